#!/bin/bash
# round 3, fourth GPU pass: KV-head long-context attention after the merge rewrite (parity + by-context timing), and the VALU
# reductions of the K-quant GEMV loop (deferred SiLU, integer plane recombination, hoisted chunk offsets) against the previous
# commit's library (ntransformer_amd/libntransformer_hip_old.so) on the same box
TAG=${1:-r03d}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
OLD=$GRAFT_REPO_ROOT/ntransformer_amd/libntransformer_hip_old.so
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -p no:cacheprovider -k "attention or gemv" > $OUT/pytest_k.log 2>&1; echo "exit $?" >> $OUT/pytest_k.log; tail -6 $OUT/pytest_k.log
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider -k "long_context or logits_match_reference_host_code or q4_k_m_mix or 70b_width" > $OUT/pytest_e.log 2>&1; echo "exit $?" >> $OUT/pytest_e.log; tail -4 $OUT/pytest_e.log
echo "== KV-head form"; timeout 200 python tools/attn_bench.py 2>&1 | tee $OUT/attn_new.txt
SH="8b.gate|up+silu,lm_head,8b.down+res,70b.qkv_fused,70b.gate|up+silu,70b.down+res"
for rep in 1 2; do
echo "== old"; NTK_LIB_PATH=$OLD timeout 300 python tools/gemv_bench.py --dtypes Q4_K,Q6_K,Q8_0 --shapes "$SH" 2>&1
echo "== new"; timeout 300 python tools/gemv_bench.py --dtypes Q4_K,Q6_K,Q8_0 --shapes "$SH" 2>&1
done > $OUT/gemv_ab.txt 2>&1
cat $OUT/gemv_ab.txt
for rep in 1 2; do
for m in "8b Q4_K_M" "70b Q4_K_M" "70b Q6_K"; do set -- $m
NTK_LIB_PATH=$OLD timeout 600 python bench.py --model $1 --mix $2 --no-cpu-baseline --no-also --prompt-bench 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('old', '$m', b['value'], b['ms_per_step'])"
timeout 600 python bench.py --model $1 --mix $2 --no-cpu-baseline --no-also --prompt-bench 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('new', '$m', b['value'], b['ms_per_step'])"
done; done 2>&1 | tee $OUT/bench_ab.txt
