#!/bin/bash
# round 3: the two-FP16-piece prompt GEMM (gemm_f16.hip) against the previous commit's three-BF16-piece form
# (ntransformer_amd/libntransformer_hip_old.so) on the same box: parity tests, per-matrix table, 1024-token prompt pass
TAG=${1:-r03p}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
OLD=$GRAFT_REPO_ROOT/ntransformer_amd/libntransformer_hip_old.so
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -p no:cacheprovider -k "gemm_quant_bf16 or gemm_quant_f16" > $OUT/pytest_gemm.log 2>&1; echo "exit $?" >> $OUT/pytest_gemm.log; tail -5 $OUT/pytest_gemm.log
NTK_GEMM_NO_PF=1 timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -p no:cacheprovider -k "gemm_quant_bf16 or gemm_quant_f16" > $OUT/pytest_gemm_nopf.log 2>&1; echo "exit $?" >> $OUT/pytest_gemm_nopf.log; tail -3 $OUT/pytest_gemm_nopf.log
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider -k "batched_prefill or logits_match_reference_host_code or golden_greedy" > $OUT/pytest_engine.log 2>&1; echo "exit $?" >> $OUT/pytest_engine.log; tail -4 $OUT/pytest_engine.log
for rep in 1 2; do
echo "== old (3 x BF16)"; NTK_LIB_PATH=$OLD timeout 300 python tools/prefill_bench.py --bf16-only --no-engine --mixes Q8_0,Q4_K,Q6_K 2>&1
echo "== new, no prefetch"; NTK_GEMM_NO_PF=1 timeout 300 python tools/prefill_bench.py --bf16-only --no-engine --mixes Q8_0,Q4_K,Q6_K 2>&1
echo "== new"; timeout 300 python tools/prefill_bench.py --bf16-only --no-engine --mixes Q8_0,Q4_K,Q6_K 2>&1
done > $OUT/gemm_ab.txt 2>&1
cat $OUT/gemm_ab.txt
for mix in Q8_0 Q4_K_M; do
echo "== old $mix"; NTK_LIB_PATH=$OLD timeout 300 python tools/prefill_bench.py --no-kernels --mix $mix --tokens 64,1024 --modes 2 2>&1
echo "== new, no prefetch $mix"; NTK_GEMM_NO_PF=1 timeout 300 python tools/prefill_bench.py --no-kernels --mix $mix --tokens 64,1024 --modes 2 2>&1
echo "== new $mix"; timeout 300 python tools/prefill_bench.py --no-kernels --mix $mix --tokens 64,1024 --modes 2 2>&1
done 2>&1 | tee $OUT/prompt_ab.txt
