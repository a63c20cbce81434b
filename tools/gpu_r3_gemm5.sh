#!/bin/bash
# round 3: FP16 prompt GEMM -- the wider checks (every GEMM / engine / parity-depth / TP test), clock calibration, prompt passes
TAG=${1:-r03z}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
T1=$GRAFT_REPO_ROOT/ntransformer_amd/libntransformer_hip_trace.so
timeout 1200 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -p no:cacheprovider -k "gemm" > $OUT/pytest_gemm.log 2>&1; echo "exit $?" >> $OUT/pytest_gemm.log; tail -3 $OUT/pytest_gemm.log
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_parity_depth.py tests/test_tp_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_engine.log 2>&1; echo "exit $?" >> $OUT/pytest_engine.log; tail -3 $OUT/pytest_engine.log
{ NTK_LIB_PATH=$T1 timeout 200 python tools/gemm_f16_trace.py 2>&1; } > $OUT/gemm_trace.txt; grep tokens $OUT/gemm_trace.txt
for m in "8b Q8_0" "8b Q4_K_M" "70b Q4_K_M" "70b Q6_K"; do set -- $m
timeout 900 python bench.py --model $1 --mix $2 --no-cpu-baseline --no-also --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('$m', b['value'], 'tok/s decode; prompt', b['config'].get('prompt_pass'))"
done 2>&1 | tee $OUT/prompt_models.txt
