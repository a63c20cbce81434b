#!/bin/bash
# round 3: is the kernel-argument prefetch of the GEMV prologue still needed now that GemvParams is 3 cache lines (AttnFuse moved out)?
# libntransformer_hip_old.so here = the current tree built with -DNTK_GEMV_NO_KARG_PREFETCH
TAG=${1:-r03k}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
OLD=$GRAFT_REPO_ROOT/ntransformer_amd/libntransformer_hip_old.so
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider -k "long_context" > $OUT/pytest_e.log 2>&1; echo "exit $?" >> $OUT/pytest_e.log; tail -3 $OUT/pytest_e.log
SH="8b.qkv_fused,8b.o+res,8b.gate|up+silu,8b.down+res"
for rep in 1 2; do
echo "== no prefetch"; NTK_LIB_PATH=$OLD timeout 300 python tools/gemv_bench.py --dtypes Q8_0,Q4_K --shapes "$SH" 2>&1
echo "== prefetch"; timeout 300 python tools/gemv_bench.py --dtypes Q8_0,Q4_K --shapes "$SH" 2>&1
done > $OUT/gemv_ab.txt 2>&1
cat $OUT/gemv_ab.txt
for rep in 1 2 3; do
NTK_LIB_PATH=$OLD timeout 600 python bench.py --no-cpu-baseline --no-also --prompt-bench 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('headline no-prefetch', b['value'], b['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --no-also --prompt-bench 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('headline prefetch', b['value'], b['ms_per_step'])"
done 2>&1 | tee $OUT/bench_ab.txt
