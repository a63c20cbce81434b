export NTK_LIB_PATH=$PWD/ntransformer_amd/libntransformer_hip_tune.so
for v in "" "NTK_GEMM_RT=1" "NTK_GEMM_RT=1 NTK_GEMM_WGS=512" "NTK_GEMM_WGS=512" "NTK_GEMM_RT=1 NTK_GEMM_WGS=1024"; do
  echo "== $v"
  env $v timeout 200 python tools/prefill_bench.py --no-kernels --tokens 32,64,128,192,256 --modes 2 2>&1 | grep "prompt of"
done
