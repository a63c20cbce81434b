#!/bin/bash
# batched prompt GEMM: what does each phase cost?  NTK_GEMM_ABLATE is a compile-time switch (1 = no dequantisation,
# 2 = VALU FMAs instead of MFMA, 4 = no weight loads after the first tile): rebuild the library per variant, restore at the end.
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
for ab in ${ABLATES:-0 1 2 3 4 7}; do
  echo "== NTK_GEMM_ABLATE=$ab"
  touch ntransformer_amd/csrc/gemm_prefill.hip
  make -s -C ntransformer_amd/csrc HIPFLAGS="$FLAGS -DNTK_GEMM_ABLATE=$ab" >/dev/null 2>&1
  timeout 300 python tools/prefill_bench.py --no-engine 2>&1 | grep -E "8b.q/o|8b.gate|70b.gate|70b.down"
done
touch ntransformer_amd/csrc/gemm_prefill.hip; make -s -C ntransformer_amd/csrc >/dev/null 2>&1
