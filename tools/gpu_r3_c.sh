#!/bin/bash
# round 3, third GPU pass: the KV-head form of the long-context decode attention (parity + by-context timing against the per-head
# form, NTK_ATT_GQA=0), and the remaining full-depth parity tests (8B Q4_K_M 32 layers, 70B width 16 layers x 2)
TAG=${1:-r03c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -p no:cacheprovider -k "attention" > $OUT/pytest_att.log 2>&1; echo "exit $?" >> $OUT/pytest_att.log; tail -6 $OUT/pytest_att.log
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider -k "long_context" > $OUT/pytest_long.log 2>&1; echo "exit $?" >> $OUT/pytest_long.log; tail -4 $OUT/pytest_long.log
echo "== per-head split form (NTK_ATT_GQA=0)"; NTK_ATT_GQA=0 timeout 300 python tools/attn_bench.py 2>&1 | tee $OUT/attn_old.txt
echo "== KV-head form"; timeout 300 python tools/attn_bench.py 2>&1 | tee $OUT/attn_new.txt
rm -f gpurun_out/parity_depth.jsonl
timeout 1500 python -m pytest tests/test_parity_depth.py -m gpu -q -p no:cacheprovider -k "not 8b_q8_0" > $OUT/pytest_depth.log 2>&1; echo "exit $?" >> $OUT/pytest_depth.log; tail -15 $OUT/pytest_depth.log
cp gpurun_out/parity_depth.jsonl $OUT/ 2>/dev/null
