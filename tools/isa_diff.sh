#!/bin/bash
# Are the gfx950 instructions of two revisions of the kernels the same?  Compiles every csrc/*.hip of both revisions with the Makefile's
# flags and diffs the device assembly (comments, debug labels and the per-translation-unit __hip_cuid_* symbols dropped).
# Used to tie a late host-side / comment-only / ifdef-only commit to the GPU passes of an earlier one.
# usage: bash tools/isa_diff.sh <revA> <revB> [file.hip ...]        (no GPU needed)
set -u
A=$1; B=$2; shift 2
FILES=${*:-"gemv.hip gemv_rp.hip gemm_prefill.hip attention.hip elementwise.hip sampling.hip gemm_f16.hip tp.hip attention_mfma.hip"}
T=$(mktemp -d)
for r in $A $B; do mkdir -p $T/$r; git archive $r ntransformer_amd/csrc include | tar -x -C $T/$r; done
rc=0
for f in $FILES; do
  for r in $A $B; do
    ( cd $T/$r/ntransformer_amd/csrc && [ -f $f ] && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$T/$r/include -I. -c $f -o $T/$r/${f%.hip}.o --save-temps 2>/dev/null
      grep -v "^\s*;\|^\s*\.\(file\|loc\|ident\)\|\.Ltmp\|\.Lfunc\|__hip_cuid_\|^\s*$" ${f%.hip}-hip-amdgcn-amd-amdhsa-gfx950.s 2>/dev/null | sed 's/;.*$//' > $T/$r/${f%.hip}.code.s ) &
  done; wait
  n=$(diff $T/$A/${f%.hip}.code.s $T/$B/${f%.hip}.code.s | wc -l)
  echo "$f: $n differing lines of $(wc -l < $T/$B/${f%.hip}.code.s)"
  [ "$n" = 0 ] || rc=1
done
rm -rf $T
exit $rc
