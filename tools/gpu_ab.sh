#!/bin/bash
# A/B of two builds of the library on one box: ntransformer_amd/libntransformer_hip_old.so (NTK_LIB_PATH) against the current one
# (old build: git worktree add /tmp/oldtree <rev>; make -C /tmp/oldtree/ntransformer_amd/csrc ../libntransformer_hip.so; copy it)
OLD=$GRAFT_REPO_ROOT/ntransformer_amd/libntransformer_hip_old.so
SH=${SH:-"8b.qkv_fused,8b.o+res,8b.gate|up+silu,8b.down+res,70b.gate|up+silu,70b.down+res"}
DT=${DT:-"Q8_0,Q4_K"}
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -p no:cacheprovider -k "gemv" 2>&1 | tail -2
for rep in 1 2; do
echo "== old"; NTK_LIB_PATH=$OLD timeout 300 python tools/gemv_bench.py --dtypes $DT --shapes "$SH" 2>&1
echo "== new"; timeout 300 python tools/gemv_bench.py --dtypes $DT --shapes "$SH" 2>&1
done
for rep in 1 2; do
NTK_LIB_PATH=$OLD timeout 600 python bench.py --no-cpu-baseline --no-also --prompt-bench 0 2>/dev/null | cut -c1-140
timeout 600 python bench.py --no-cpu-baseline --no-also --prompt-bench 0 2>/dev/null | cut -c1-140
done
