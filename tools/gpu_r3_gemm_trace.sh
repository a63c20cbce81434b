#!/bin/bash
# round 3: step timeline of the FP16 prompt GEMM (trace builds)
TAG=${1:-r03q}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
T1=$GRAFT_REPO_ROOT/ntransformer_amd/libntransformer_hip_trace.so
T2=$GRAFT_REPO_ROOT/ntransformer_amd/libntransformer_hip_trace2.so
{
echo "== level 1, prefetch"; NTK_LIB_PATH=$T1 timeout 200 python tools/gemm_f16_trace.py 2>&1
echo "== level 1, no prefetch"; NTK_GEMM_NO_PF=1 NTK_LIB_PATH=$T1 timeout 200 python tools/gemm_f16_trace.py 2>&1
echo "== level 2, prefetch"; NTK_LIB_PATH=$T2 timeout 200 python tools/gemm_f16_trace.py 2>&1
echo "== level 2, no prefetch"; NTK_GEMM_NO_PF=1 NTK_LIB_PATH=$T2 timeout 200 python tools/gemm_f16_trace.py 2>&1
} | tee $OUT/gemm_trace.txt
