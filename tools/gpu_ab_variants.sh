#!/bin/bash
# Same-box A/B of library variants (tools/build_variants.sh) against the tuning build of the product, through NTK_LIB_PATH.
#   usage (GPU box): bash tools/gpu_ab_variants.sh <tag> <variant> [<variant> ...]         e.g.  ... r05ab nx kt rl ktrl
# Per library, alternated REPS (default 2) times: the kernel parity tests (once), tools/gemv_bench.py on the short launches, and bench.py on
# 8B Q8_0 (the headline), 8B Q4_K_M and 70B Q4_K_M.  Output: gpurun_out/<tag>/ab.txt
TAG=${1:-ab}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; D=$PWD/ntransformer_amd
SH="8b.qkv_fused,8b.o+res,8b.gate|up+silu,8b.down+res,70b.o+res,70b.down+res"
for V in "$@"; do
  NTK_LIB_PATH=$D/libntransformer_hip_$V.so timeout 300 python -m pytest tests/test_gemv_rp.py tests/test_hip_kernels.py -m gpu -q -x -k "gemv" -p no:cacheprovider 2>&1 | tail -1 | sed "s/^/$V parity: /"
done | tee $OUT/ab.txt
for rep in $(seq ${REPS:-2}); do
for V in tune "$@"; do L=$D/libntransformer_hip_$V.so
  echo "== $V (rep $rep)"
  NTK_LIB_PATH=$L timeout 120 python tools/gemv_bench.py --dtypes Q8_0 --shapes "$SH" 2>&1 | grep "^Q8_0" | cut -c1-70
  NTK_LIB_PATH=$L timeout 120 python tools/gemv_bench.py --rp --dtypes Q4_K --shapes "$SH" 2>&1 | grep "rp " | cut -c1-70
  for W in "--model 8b --mix Q8_0" "--model 8b --mix Q4_K_M" "--model 70b --mix Q4_K_M --steps 32"; do
    NTK_LIB_PATH=$L timeout 200 python bench.py $W --no-also --no-cpu-baseline --prompt-bench 0 2>/dev/null | cut -c1-120
  done
done; done 2>&1 | tee -a $OUT/ab.txt
