#!/bin/bash
# round 3: what a step of the FP16 prompt GEMM spends its time on -- timing-only ablation builds (NTK_GEMM_ABLATE bits)
TAG=${1:-r03t}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for ab in 0 1 2 4 8 16 32 3 7 15; do
  L=$GRAFT_REPO_ROOT/ntransformer_amd/libntransformer_hip_ab$ab.so; [ $ab = 0 ] && L=$GRAFT_REPO_ROOT/ntransformer_amd/libntransformer_hip.so
  echo "== ablate $ab"; NTK_LIB_PATH=$L timeout 120 python tools/prefill_bench.py --bf16-only --no-engine --mixes Q8_0,Q4_K 2>&1 | grep -E "8b.gate/up|8b.down|70b.gate" | grep "256 tok"
done | tee $OUT/gemm_ablate.txt
