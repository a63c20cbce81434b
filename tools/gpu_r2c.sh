#!/bin/bash
timeout 120 python bench.py --no-also --no-cpu-baseline --steps 64 --persistent 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['ms_per_step'], 'ms', d['config']['path'][:20])"
timeout 120 python tools/persistent_trace.py 2>&1 | grep -v "^Model\|^Free\|^Tokenizer" | tail -10
timeout 200 python -m pytest tests/test_engine_gpu.py -q -x -k "persistent_token_kernel" 2>&1 | tail -3
