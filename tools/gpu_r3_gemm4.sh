#!/bin/bash
# round 3: FP16 prompt GEMM, plane prefetch on / off on the same box
TAG=${1:-r03y}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
OLD=$GRAFT_REPO_ROOT/ntransformer_amd/libntransformer_hip_old.so
T1=$GRAFT_REPO_ROOT/ntransformer_amd/libntransformer_hip_trace.so
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -p no:cacheprovider -k "gemm_quant_bf16 or gemm_quant_f16" > $OUT/pytest_gemm.log 2>&1; echo "exit $?" >> $OUT/pytest_gemm.log; tail -3 $OUT/pytest_gemm.log
NTK_GEMM_NO_PF=1 timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -p no:cacheprovider -k "gemm_quant_bf16 or gemm_quant_f16" > $OUT/pytest_gemm_nopf.log 2>&1; echo "exit $?" >> $OUT/pytest_gemm_nopf.log; tail -3 $OUT/pytest_gemm_nopf.log
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider -k "batched_prefill or logits_match_reference_host_code or golden_greedy" > $OUT/pytest_engine.log 2>&1; echo "exit $?" >> $OUT/pytest_engine.log; tail -3 $OUT/pytest_engine.log
{ echo "== level 1"; NTK_LIB_PATH=$T1 timeout 200 python tools/gemm_f16_trace.py 2>&1; } > $OUT/gemm_trace.txt
for rep in 1 2; do
echo "== no prefetch"; NTK_GEMM_NO_PF=1 timeout 120 python tools/prefill_bench.py --bf16-only --no-engine --mixes Q8_0,Q4_K 2>&1 | grep -E "8b.gate/up|8b.down|70b.gate" | grep "256 tok"
echo "== prefetch"; timeout 120 python tools/prefill_bench.py --bf16-only --no-engine --mixes Q8_0,Q4_K,Q6_K 2>&1 | grep -E "8b.gate/up|8b.down|70b.gate" | grep "256 tok"
done | tee $OUT/gemm_ab.txt
for mix in Q8_0 Q4_K_M; do
echo "== no prefetch $mix"; NTK_GEMM_NO_PF=1 timeout 300 python tools/prefill_bench.py --no-kernels --mix $mix --tokens 1024 --modes 2 2>&1 | grep prompt
echo "== new $mix"; timeout 300 python tools/prefill_bench.py --no-kernels --mix $mix --tokens 64,256,1024 --modes 2 2>&1 | grep prompt
done 2>&1 | tee $OUT/prompt_ab.txt
