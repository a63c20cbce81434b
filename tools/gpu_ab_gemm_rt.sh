#!/bin/bash
# Prompt GEMM planner A/B on the tuning library: tile height (NTK_GEMM_RT) and grid target (NTK_GEMM_WGS) at 32 ... 256 prompt tokens.  Result (round 5):
# every variant within 3 % of the default plan (profiles/r05_prefill_row_max_ab.txt's neighbourhood; recorded in DESIGN 8.4).   usage: bash tools/gpu_ab_gemm_rt.sh
export NTK_LIB_PATH=$PWD/ntransformer_amd/libntransformer_hip_tune.so
for v in "" "NTK_GEMM_RT=1" "NTK_GEMM_RT=1 NTK_GEMM_WGS=512" "NTK_GEMM_WGS=512" "NTK_GEMM_RT=1 NTK_GEMM_WGS=1024"; do
  echo "== $v"
  env $v timeout 200 python tools/prefill_bench.py --no-kernels --tokens 32,64,128,192,256 --modes 2 2>&1 | grep "prompt of"
done
