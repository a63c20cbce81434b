#!/bin/bash
# Round 5: contexts beyond 4096 on the GPU box: the kernel + engine parity tests, then the split-count sweep of tools/attn_bench.py at 8K / 32K / 128K.
TAG=${1:-lc}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider -k "beyond_4096 or 8k_and_33k or beyond_3072" > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
( timeout 300 python tools/attn_bench.py --max-seq 8192 --layers 16 --cases 8191:16,8191:32,8191:64,8191:128
  timeout 300 python tools/attn_bench.py --max-seq 32768 --layers 6 --cases 32767:32,32767:64,32767:128,32767:256
  timeout 400 python tools/attn_bench.py --max-seq 131072 --layers 3 --models 8b --cases 131071:64,131071:128,131071:256,131071:512 ) > $OUT/attn_long.txt 2>&1
cat $OUT/attn_long.txt | grep pos
