#!/bin/bash
timeout 300 python -m pytest tests/test_hip_kernels.py -q -x -k "attention_prefill" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_engine_gpu.py -q -x -k "batched_prefill or long_context or logits_match_reference or full_depth" 2>&1 | tail -3
timeout 300 python tools/prefill_bench.py 2>&1 | grep -v "^Model\|^Free\|^Tokenizer" | tail -12
