#!/bin/bash
timeout 600 python -m pytest tests/test_hip_kernels.py -q -x -k "gemm_quant_bf16" 2>&1 | tail -3
for cfg in "0 512" "2 256"; do set -- $cfg
  echo "== RT=$1 WGS=$2"
  NTK_GEMM_RT=$1 NTK_GEMM_WGS=$2 timeout 300 python tools/prefill_bench.py --no-engine --bf16-only 2>&1 | grep "bf16" | awk '{print $1, $2, $4, $6, $7}' | tr '\n' ';'; echo
done
timeout 300 python tools/prefill_bench.py --no-kernels 2>&1 | grep "prompt of" | grep "=2" | tail -4
