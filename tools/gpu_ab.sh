#!/bin/bash
# A/B of two builds of the library on one box: ntransformer_amd/libntransformer_hip_old.so (NTK_LIB_PATH) against the current one
OLD=$GRAFT_REPO_ROOT/ntransformer_amd/libntransformer_hip_old.so
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
SH="8b.qkv_fused,8b.o+res,8b.gate|up+silu,8b.down+res"
for rep in 1 2; do
echo "== old"; NTK_LIB_PATH=$OLD timeout 300 python tools/gemv_bench.py --dtypes Q8_0 --shapes "$SH" 2>&1
echo "== new"; timeout 300 python tools/gemv_bench.py --dtypes Q8_0 --shapes "$SH" 2>&1
done
echo "== attn old"; NTK_LIB_PATH=$OLD timeout 300 python tools/attn_bench.py 2>&1 | grep "^8b" | head -4
echo "== attn new"; timeout 300 python tools/attn_bench.py 2>&1 | grep "^8b" | head -4
for rep in 1 2; do
NTK_LIB_PATH=$OLD timeout 600 python bench.py --no-cpu-baseline --no-also --prompt-bench 0 2>/dev/null | cut -c1-140
timeout 600 python bench.py --no-cpu-baseline --no-also --prompt-bench 0 2>/dev/null | cut -c1-140
done
