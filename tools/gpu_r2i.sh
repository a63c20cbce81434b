#!/bin/bash
timeout 300 python -m pytest tests/test_hip_kernels.py -q -x -k "gemm_quant_bf16" 2>&1 | tail -4
timeout 300 python -m pytest tests/test_engine_gpu.py -q -x -k "batched_prefill or logits_match_reference or 8b_q4_k_m or 70b_width" 2>&1 | tail -3
timeout 300 python tools/prefill_bench.py --no-engine 2>&1 | grep "bf16" | head -18
timeout 300 python tools/prefill_bench.py --no-kernels 2>&1 | grep "prompt of" | tail -8
