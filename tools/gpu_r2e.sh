#!/bin/bash
timeout 300 python -m pytest tests/test_hip_kernels.py -q -x -k "gemv" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_engine_gpu.py -q -x -k "q4_k_m or 70b_width or logits_match_reference" 2>&1 | tail -3
for m in "8b Q4_K_M 128" "70b Q4_K_M 48"; do set -- $m
  timeout 200 python bench.py --no-also --no-cpu-baseline --model $1 --mix $2 --steps $3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['value'], 'tok/s', d['ms_per_step'], 'ms launches', d['roofline']['launches_per_token'], 'frac', d['hbm_fraction_of_8TBs_end_to_end'])"
done
