#!/bin/bash
# round 3: FP16 prompt GEMM, two chunks per workgroup + K split in two for the narrow matrices, against the previous commit's library, same box
TAG=${1:-r03am}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
OLD=$GRAFT_REPO_ROOT/ntransformer_amd/libntransformer_hip_old3.so
timeout 1200 python -m pytest tests/test_hip_kernels.py -m gpu -q -p no:cacheprovider -k "gemm_quant_f16" > $OUT/pytest_gemm.log 2>&1; echo "exit $?" >> $OUT/pytest_gemm.log; tail -3 $OUT/pytest_gemm.log
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider -k "batched_prefill_fills or logits_match_reference_host_code or golden_greedy" > $OUT/pytest_engine.log 2>&1; echo "exit $?" >> $OUT/pytest_engine.log; tail -3 $OUT/pytest_engine.log
for rep in 1 2; do for mix in Q8_0 Q4_K_M; do
echo "== before $mix"; NTK_LIB_PATH=$OLD timeout 300 python tools/prefill_bench.py --no-kernels --mix $mix --tokens 512,768,1024 --modes 2 2>&1 | grep prompt
echo "== narrow matrices: two chunks + K split $mix"; timeout 300 python tools/prefill_bench.py --no-kernels --mix $mix --tokens 512,768,1024 --modes 2 2>&1 | grep prompt
done; done 2>&1 | tee $OUT/prompt_ab.txt
