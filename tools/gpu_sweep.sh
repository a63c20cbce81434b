#!/bin/bash
# GEMV tuning sweep on the GPU box: grid size / waves per workgroup / ablations, Q8_0 8B shapes.
OUT=gpurun_out/${1:-sweep}; mkdir -p $OUT
SH="8b.qkv_fused,8b.o+res,8b.gate|up+silu,8b.down+res,lm_head"
run() { echo "== $*" | tee -a $OUT/sweep.log; env "$@" python tools/gemv_bench.py --dtypes Q8_0 --shapes "$SH" 2>&1 | tee -a $OUT/sweep.log; }
run NTK_GEMV_MAX_WG=512 NTK_GEMV_WAVES=8
run NTK_GEMV_MAX_WG=256 NTK_GEMV_WAVES=8
run NTK_GEMV_MAX_WG=1024 NTK_GEMV_WAVES=4
run NTK_GEMV_MAX_WG=512 NTK_GEMV_WAVES=4
run NTK_GEMV_MAX_WG=256 NTK_GEMV_WAVES=4
run NTK_GEMV_MAX_WG=512 NTK_GEMV_WAVES=8 NTK_GEMV_ABLATE=1
run NTK_GEMV_MAX_WG=512 NTK_GEMV_WAVES=8 NTK_GEMV_ABLATE=2
run NTK_GEMV_MAX_WG=512 NTK_GEMV_WAVES=8 NTK_GEMV_ABLATE=6
run NTK_GEMV_MAX_WG=512 NTK_GEMV_WAVES=8 NTK_GEMV_ABLATE=7
