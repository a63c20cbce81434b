#!/bin/bash
# round 3, second GPU pass: the LDS-DMA row ring of the K-quant GEMV -- parity of every GEMV test with it, then A/B on one build
# (NTK_GEMV_DMA=0 = the register-prefetch form): launches one by one and the two K-quant decode workloads
TAG=${1:-r03b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -p no:cacheprovider -k "gemv" > $OUT/pytest_gemv.log 2>&1; echo "exit $?" >> $OUT/pytest_gemv.log; tail -6 $OUT/pytest_gemv.log
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider -k "logits_match_reference_host_code or q4_k_m_mix or 70b_width" > $OUT/pytest_engine.log 2>&1; echo "exit $?" >> $OUT/pytest_engine.log; tail -4 $OUT/pytest_engine.log
SH="8b.qkv_fused,8b.o+res,8b.gate|up+silu,lm_head,70b.qkv_fused,70b.o+res,70b.gate|up+silu"
for rep in 1 2; do
echo "== register prefetch (NTK_GEMV_DMA=0)"; NTK_GEMV_DMA=0 timeout 300 python tools/gemv_bench.py --dtypes Q4_K,Q6_K,Q5_K --shapes "$SH" 2>&1
echo "== LDS-DMA ring"; timeout 300 python tools/gemv_bench.py --dtypes Q4_K,Q6_K,Q5_K --shapes "$SH" 2>&1
done > $OUT/gemv_ab.txt 2>&1
cat $OUT/gemv_ab.txt
for rep in 1 2; do
for m in "8b Q4_K_M" "70b Q4_K_M"; do set -- $m
NTK_GEMV_DMA=0 timeout 600 python bench.py --model $1 --mix $2 --no-cpu-baseline --no-also --prompt-bench 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('DMA=0', '$m', b['value'], b['ms_per_step'])"
timeout 600 python bench.py --model $1 --mix $2 --no-cpu-baseline --no-also --prompt-bench 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('DMA=1', '$m', b['value'], b['ms_per_step'])"
done; done 2>&1 | tee $OUT/bench_ab.txt
