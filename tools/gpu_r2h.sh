#!/bin/bash
timeout 300 python -m pytest tests/test_engine_gpu.py -q -x -k "attention_inside or logits_match_reference or generate_tokens or 8b_width or long_context" 2>&1 | tail -3
for f in 1 0; do
  echo "== NTK_FUSE_ATTENTION=$f"
  for m in "8b Q8_0 128" "8b Q4_K_M 128" "70b Q4_K_M 48"; do set -- $m
    NTK_FUSE_ATTENTION=$f timeout 200 python bench.py --no-also --no-cpu-baseline --model $1 --mix $2 --steps $3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['value'], 'tok/s', d['ms_per_step'], 'ms; launches', d['roofline']['launches_per_token'], 'gemv avg', d['roofline']['avg_launch_us'], 'us')"
  done
done
