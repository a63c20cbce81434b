#!/bin/bash
# The prompt pass with its small launches folded (token maxima from the RMSNorm / SiLU launches, K splits summed by the consuming launch, RoPE + cache store
# as one launch): parity tests, then an alternated A/B of the engine's prompt pass.   usage: bash tools/gpu_row_max.sh <tag>
TAG=${1:-rm}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider -k "row_maxima or folded or rope_kv_store or gemm_quant_f16 or prefill or prompt or logits_match or golden" > $OUT/pytest.txt 2>&1; echo "exit $?" >> $OUT/pytest.txt; tail -6 $OUT/pytest.txt
for mix in Q8_0 Q4_K_M; do
  timeout 400 python tools/prefill_bench.py --no-kernels --ab-row-max --tokens 64,256,1024 --mix $mix 2>&1 | grep "prompt of" >> $OUT/ab.txt
done
cat $OUT/ab.txt
