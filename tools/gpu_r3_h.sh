#!/bin/bash
# round 3: column-split Q8_0 GEMV (gemv_colsplit.hip.h) -- parity, then A/B on one build (NTK_GEMV_COLSPLIT=0 = the row form)
TAG=${1:-r03l}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -p no:cacheprovider -k "gemv" > $OUT/pytest_k.log 2>&1; echo "exit $?" >> $OUT/pytest_k.log; tail -5 $OUT/pytest_k.log
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider -k "logits_match_reference_host_code or 8b_width or generate_tokens" > $OUT/pytest_e.log 2>&1; echo "exit $?" >> $OUT/pytest_e.log; tail -3 $OUT/pytest_e.log
SH="8b.q,8b.kv,8b.qkv_fused,8b.o+res,8b.gate|up+silu,lm_head"
for rep in 1 2; do
echo "== row form (NTK_GEMV_COLSPLIT=0)"; NTK_GEMV_COLSPLIT=0 timeout 300 python tools/gemv_bench.py --dtypes Q8_0 --shapes "$SH" 2>&1
echo "== column split"; timeout 300 python tools/gemv_bench.py --dtypes Q8_0 --shapes "$SH" 2>&1
done > $OUT/gemv_ab.txt 2>&1
cat $OUT/gemv_ab.txt
for rep in 1 2 3; do
NTK_GEMV_COLSPLIT=0 timeout 600 python bench.py --no-cpu-baseline --no-also --prompt-bench 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('headline row-form', b['value'], b['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --no-also --prompt-bench 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('headline colsplit', b['value'], b['ms_per_step'])"
done 2>&1 | tee $OUT/bench_ab.txt
