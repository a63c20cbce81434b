#!/bin/bash
# round 3: FP16 prompt GEMM, two 64-token chunks per workgroup (where the grid allows) against one, same box
TAG=${1:-r03ae}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hip_kernels.py -m gpu -q -p no:cacheprovider -k "gemm_quant_f16" > $OUT/pytest_gemm.log 2>&1; echo "exit $?" >> $OUT/pytest_gemm.log; tail -4 $OUT/pytest_gemm.log
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider -k "batched_prefill or logits_match_reference_host_code or golden_greedy" > $OUT/pytest_engine.log 2>&1; echo "exit $?" >> $OUT/pytest_engine.log; tail -3 $OUT/pytest_engine.log
for rep in 1 2; do
echo "== one chunk per workgroup"; NTK_GEMM_CW=1 timeout 120 python tools/prefill_bench.py --bf16-only --no-engine --mixes Q8_0,Q4_K 2>&1 | grep -E "8b.gate/up|70b.gate|70b.down|70b.q/o" | grep "256 tok"
echo "== two where the grid allows"; timeout 120 python tools/prefill_bench.py --bf16-only --no-engine --mixes Q8_0,Q4_K 2>&1 | grep -E "8b.gate/up|70b.gate|70b.down|70b.q/o" | grep "256 tok"
done | tee $OUT/gemm_ab.txt
for mix in Q8_0 Q4_K_M; do
echo "== one chunk per workgroup $mix"; NTK_GEMM_CW=1 timeout 300 python tools/prefill_bench.py --no-kernels --mix $mix --tokens 512,1024 --modes 2 2>&1 | grep prompt
echo "== two where the grid allows $mix"; timeout 300 python tools/prefill_bench.py --no-kernels --mix $mix --tokens 512,1024 --modes 2 2>&1 | grep prompt
done 2>&1 | tee $OUT/prompt_ab.txt
