#!/bin/bash
for w in 1 0; do
  echo "== NTK_GEMV_WARM=$w"
  for m in "8b Q8_0 128" "8b Q4_K_M 128" "70b Q4_K_M 48"; do set -- $m
    NTK_GEMV_WARM=$w timeout 200 python bench.py --no-also --no-cpu-baseline --model $1 --mix $2 --steps $3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['value'], 'tok/s', d['ms_per_step'], 'ms; gemv avg launch', d['roofline']['avg_launch_us'], 'us frac', d['roofline']['frac'])"
  done
done
timeout 300 python -m pytest tests/test_engine_gpu.py tests/test_hip_kernels.py -q -x -k "logits_match or gemv_fused or 8b_width" 2>&1 | tail -2
