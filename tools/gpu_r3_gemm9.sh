#!/bin/bash
# round 3: FP16 prompt GEMM, K-quant minimum term on the matrix cores against the per-step FMA form (previous commit's library), same box
TAG=${1:-r03af}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
OLD=$GRAFT_REPO_ROOT/ntransformer_amd/libntransformer_hip_old2.so
timeout 1200 python -m pytest tests/test_hip_kernels.py -m gpu -q -p no:cacheprovider -k "gemm_quant_f16" > $OUT/pytest_gemm.log 2>&1; echo "exit $?" >> $OUT/pytest_gemm.log; tail -4 $OUT/pytest_gemm.log
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider -k "batched_prefill or logits_match_reference_host_code or golden_greedy" > $OUT/pytest_engine.log 2>&1; echo "exit $?" >> $OUT/pytest_engine.log; tail -3 $OUT/pytest_engine.log
for rep in 1 2; do
echo "== per-step FMAs"; NTK_LIB_PATH=$OLD timeout 120 python tools/prefill_bench.py --bf16-only --no-engine --mixes Q4_K 2>&1 | grep -E "8b.gate/up|8b.down|70b.gate|70b.down|70b.q/o" | grep "256 tok"
echo "== matrix cores"; timeout 120 python tools/prefill_bench.py --bf16-only --no-engine --mixes Q4_K 2>&1 | grep -E "8b.gate/up|8b.down|70b.gate|70b.down|70b.q/o" | grep "256 tok"
done | tee $OUT/gemm_ab.txt
for rep in 1 2; do
echo "== per-step FMAs Q4_K_M"; NTK_LIB_PATH=$OLD timeout 300 python tools/prefill_bench.py --no-kernels --mix Q4_K_M --tokens 256,1024 --modes 2 2>&1 | grep prompt
echo "== matrix cores Q4_K_M"; timeout 300 python tools/prefill_bench.py --no-kernels --mix Q4_K_M --tokens 256,1024 --modes 2 2>&1 | grep prompt
done 2>&1 | tee $OUT/prompt_ab.txt
