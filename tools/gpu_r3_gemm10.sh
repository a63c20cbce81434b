#!/bin/bash
# round 3: FP16 prompt GEMM, K split until two workgroups per CU (512) against one (256), same box
TAG=${1:-r03ah}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hip_kernels.py -m gpu -q -p no:cacheprovider -k "gemm_quant_f16" > $OUT/pytest_gemm.log 2>&1; echo "exit $?" >> $OUT/pytest_gemm.log; tail -3 $OUT/pytest_gemm.log
NTK_GEMM_WGS=512 timeout 1200 python -m pytest tests/test_hip_kernels.py -m gpu -q -p no:cacheprovider -k "gemm_quant_f16" > $OUT/pytest_gemm512.log 2>&1; echo "exit $?" >> $OUT/pytest_gemm512.log; tail -3 $OUT/pytest_gemm512.log
for rep in 1 2; do for mix in Q8_0 Q4_K_M; do
echo "== split until 256 workgroups $mix"; NTK_GEMM_WGS=256 timeout 300 python tools/prefill_bench.py --no-kernels --mix $mix --tokens 64,128,256,512 --modes 2 2>&1 | grep prompt
echo "== split until 512 workgroups $mix"; NTK_GEMM_WGS=512 timeout 300 python tools/prefill_bench.py --no-kernels --mix $mix --tokens 64,128,256,512 --modes 2 2>&1 | grep prompt
done; done 2>&1 | tee $OUT/prompt_ab.txt
