#!/bin/bash
OUT=gpurun_out/xstage; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
SH="8b.qkv_fused,8b.o+res,8b.gate|up+silu,8b.down+res,lm_head"
for rep in 1 2; do timeout 300 python tools/gemv_bench.py --dtypes Q8_0,Q4_K --shapes "$SH" 2>&1 | tee -a $OUT/gemv.txt; done
timeout 300 python tools/gemv_trace.py --shapes "8b.o+res,8b.qkv_fused" 2>&1 | tee $OUT/gemv_trace.txt
for rep in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-also --prompt-bench 0 2>/dev/null | tee $OUT/bench.json | cut -c1-140
timeout 600 python bench.py --mix Q4_K_M --no-cpu-baseline --no-also --prompt-bench 0 2>/dev/null | tee $OUT/bench_q4km.json | cut -c1-140
done
