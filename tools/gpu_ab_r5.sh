#!/bin/bash
# Round-5 same-box A/B of the compile-time variants of tools/build_variants.sh, alternated REPS times against the tuning build:
#   nx (-DNTK_GEMV_NO_XWAIT, gemv.hip) on the 8B Q8_0 headline; kt / rl / ktrl (gemv_rp.hip) on 8B Q4_K_M.  Parity of a winner is checked
#   by the full GPU suite on the merged tree, not here.   usage (GPU box): bash tools/gpu_ab_r5.sh <tag>     output: gpurun_out/<tag>/ab.txt
TAG=${1:-r5ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT; D=$PWD/ntransformer_amd
SH="8b.qkv_fused,8b.o+res,8b.gate|up+silu,8b.down+res"
for rep in $(seq ${REPS:-3}); do
  for V in tune nx; do L=$D/libntransformer_hip_$V.so; echo "== $V Q8_0 (rep $rep)"
    NTK_LIB_PATH=$L timeout 120 python tools/gemv_bench.py --dtypes Q8_0 --shapes "$SH" 2>&1 | grep "^Q8_0" | cut -c1-70
    NTK_LIB_PATH=$L timeout 200 python bench.py --model 8b --mix Q8_0 --no-also --no-cpu-baseline --prompt-bench 0 2>/dev/null | cut -c1-120
  done
  for V in tune kt rl ktrl; do L=$D/libntransformer_hip_$V.so; echo "== $V Q4_K_M (rep $rep)"
    NTK_LIB_PATH=$L timeout 120 python tools/gemv_bench.py --rp --dtypes Q4_K --shapes "$SH" 2>&1 | grep "rp " | cut -c1-70
    NTK_LIB_PATH=$L timeout 200 python bench.py --model 8b --mix Q4_K_M --no-also --no-cpu-baseline --prompt-bench 0 2>/dev/null | cut -c1-120
  done
done 2>&1 | tee $OUT/ab.txt
