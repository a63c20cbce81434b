#!/bin/bash
# Round 5: does the merged gemv_rp.hip prologue / epilogue change (kernel-argument touch + late residual) cost the 70B models anything?  Same-box A/B, alternated twice,
# of the tuning build against a build without it (tools/build_variants.sh "old=-DNTK_RP_NO_KERNARG_TOUCH -DNTK_RP_RESID_EPILOGUE"); then the new tests.
TAG=${1:-ab70}; OUT=gpurun_out/$TAG; mkdir -p $OUT; D=$PWD/ntransformer_amd
for rep in 1 2; do for V in tune old; do
  echo "== $V (rep $rep)"
  NTK_LIB_PATH=$D/libntransformer_hip_$V.so timeout 300 python bench.py --model 70b --mix Q4_K_M --steps 48 --no-also --no-cpu-baseline --prompt-bench 0 2>/dev/null | cut -c1-130
  NTK_LIB_PATH=$D/libntransformer_hip_$V.so timeout 300 python bench.py --model 70b --mix Q6_K --steps 48 --no-also --no-cpu-baseline --prompt-bench 0 2>/dev/null | cut -c1-130
done; done 2>&1 | tee $OUT/ab.txt
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -p no:cacheprovider -k "one_resident_copy or rebases or matrix_core_gemv" > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
