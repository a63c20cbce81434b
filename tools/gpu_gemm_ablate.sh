#!/bin/bash
# batched prompt GEMM: what does each phase cost?  (NTK_GEMM_ABLATE: 1 = no dequantisation, 2 = VALU FMAs instead of MFMA)
for ab in ${ABLATES:-0 1 2 3 4 7}; do echo "== NTK_GEMM_ABLATE=$ab"; NTK_GEMM_ABLATE=$ab timeout 300 python tools/prefill_bench.py --no-engine 2>&1 | grep -E "8b.q/o|8b.gate|70b.gate|70b.down" ; done
