#!/bin/bash
# Builds tuning variants of the library beside the product (CPU only; hipcc cross-compiles): one per compile-time experiment switch.
#   usage: bash tools/build_variants.sh [name=FLAGS ...]      default: the four experiments DESIGN.md section 8 lists for the next round
# Each variant = ntransformer_amd/libntransformer_hip_<name>.so (git-ignored; travels to the GPU box), built with -DNTK_TUNE + its flags in
# its own object directory (csrc/build_<name>/, not pushed: .gpurunignore).  A/B on one box: tools/gpu_ab_variants.sh.
set -eu
cd "$(dirname "$0")/../ntransformer_amd/csrc"
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Wno-unused-variable -DNTK_TUNE"
VARS=("$@")
[ ${#VARS[@]} -gt 0 ] || VARS=("nx=-DNTK_GEMV_NO_XWAIT" "kt=-DNTK_RP_KERNARG_TOUCH" "rl=-DNTK_RP_RESID_LATE" "ktrl=-DNTK_RP_KERNARG_TOUCH -DNTK_RP_RESID_LATE")
make -s -j8 tune
for v in "${VARS[@]}"; do
  name=${v%%=*}; flags=${v#*=}
  echo "== $name: $flags"
  make -s -j8 BUILD=build_$name LIBNAME=libntransformer_hip_$name.so HIPFLAGS="$BASE $flags" ../libntransformer_hip_$name.so
  grep -q "^ntransformer_amd/csrc/build_$name$" ../../.gpurunignore || echo "ntransformer_amd/csrc/build_$name" >> ../../.gpurunignore
done
ls -la ../libntransformer_hip_*.so
