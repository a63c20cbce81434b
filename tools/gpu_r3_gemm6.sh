#!/bin/bash
# round 3: FP16 prompt GEMM, (row tile, chunk) order per XCD: 8 x 8 blocks against tile-after-tile, same box
TAG=${1:-r03aa}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -p no:cacheprovider -k "gemm_quant_bf16 or gemm_quant_f16" > $OUT/pytest_gemm.log 2>&1; echo "exit $?" >> $OUT/pytest_gemm.log; tail -3 $OUT/pytest_gemm.log
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider -k "batched_prefill or logits_match_reference_host_code or golden_greedy" > $OUT/pytest_engine.log 2>&1; echo "exit $?" >> $OUT/pytest_engine.log; tail -3 $OUT/pytest_engine.log
for rep in 1 2; do for mix in Q8_0 Q4_K_M; do
echo "== tile after tile $mix"; NTK_GEMM_MAP=0 timeout 300 python tools/prefill_bench.py --no-kernels --mix $mix --tokens 512,1024 --modes 2 2>&1 | grep prompt
echo "== 8 x 8 $mix"; timeout 300 python tools/prefill_bench.py --no-kernels --mix $mix --tokens 512,1024 --modes 2 2>&1 | grep prompt
done; done 2>&1 | tee $OUT/prompt_ab.txt
