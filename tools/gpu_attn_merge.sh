#!/bin/bash
# Split-KV decode attention as one launch (csrc/attention_merge.hip.h) against the two-launch form: the parity tests, the per-layer attention time of both
# forms (tools/attn_bench.py --merged) and an alternated A/B of decode behind 1500- and 3900-token prompts.   usage: bash tools/gpu_attn_merge.sh <tag>
TAG=${1:-am}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_engine_gpu.py -m gpu -q -p no:cacheprovider -k "attention_decode_split or split_attention or decode_at_8k or long_context" > $OUT/pytest.txt 2>&1; echo "exit $?" >> $OUT/pytest.txt; tail -8 $OUT/pytest.txt
timeout 300 python tools/attn_bench.py --merged --models 8b --cases 600:8,1023:8,2047:8,3071:8,3072:32,4095:32 > $OUT/attn.txt 2>&1
timeout 200 python tools/attn_bench.py --merged --models 70b --cases 2047:8,4095:32 >> $OUT/attn.txt 2>&1
cat $OUT/attn.txt
for rep in 1 2; do
  for pl in 1500 3900; do
    for v in merged two; do
      f=""; [ $v = merged ] && f="--attention-merge"
      echo "== prompt $pl $v (rep $rep)" >> $OUT/ab.txt
      timeout 300 python bench.py --no-also --no-cpu-baseline --no-pmc-note --prompt-bench 0 --prompt-len $pl --ctx 4096 --steps 64 $f 2>/dev/null | tail -1 | cut -c1-200 >> $OUT/ab.txt
    done
  done
done
cat $OUT/ab.txt
