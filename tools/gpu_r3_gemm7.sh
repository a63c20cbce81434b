#!/bin/bash
# round 3: FP16 prompt GEMM at three workgroups per CU (tuning build, Q8_0 without the plane prefetch) against the product build
TAG=${1:-r03ac}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
O3=$GRAFT_REPO_ROOT/ntransformer_amd/libntransformer_hip_occ3.so
NTK_GEMM_NO_PF=1 NTK_LIB_PATH=$O3 timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -p no:cacheprovider -k "gemm_quant_f16 and Q8_0 and not rejections" > $OUT/pytest_gemm.log 2>&1; echo "exit $?" >> $OUT/pytest_gemm.log; tail -3 $OUT/pytest_gemm.log
for rep in 1 2; do
echo "== 2 workgroups per CU"; NTK_GEMM_NO_PF=1 timeout 120 python tools/prefill_bench.py --bf16-only --no-engine --mixes Q8_0 2>&1 | grep -E "8b.gate/up|8b.down|70b.gate|8b.q/o" | grep "256 tok"
echo "== 3 workgroups per CU"; NTK_GEMM_NO_PF=1 NTK_LIB_PATH=$O3 timeout 120 python tools/prefill_bench.py --bf16-only --no-engine --mixes Q8_0 2>&1 | grep -E "8b.gate/up|8b.down|70b.gate|8b.q/o" | grep "256 tok"
done | tee $OUT/gemm_ab.txt
echo "== 2 workgroups per CU"; NTK_GEMM_NO_PF=1 timeout 300 python tools/prefill_bench.py --no-kernels --mix Q8_0 --tokens 1024 --modes 2 2>&1 | grep prompt | tee -a $OUT/gemm_ab.txt
echo "== 3 workgroups per CU"; NTK_GEMM_NO_PF=1 NTK_LIB_PATH=$O3 timeout 300 python tools/prefill_bench.py --no-kernels --mix Q8_0 --tokens 1024 --modes 2 2>&1 | grep prompt | tee -a $OUT/gemm_ab.txt
