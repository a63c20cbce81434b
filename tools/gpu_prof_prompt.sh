#!/bin/bash
# rocprofv3 kernel trace of the engine's prompt pass at given lengths (per-kernel statistics, csv).   usage: bash tools/gpu_prof_prompt.sh <tag> <tokens> [mix]
TAG=${1:-pp}; TOK=${2:-64}; MIX=${3:-Q8_0}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/t$TOK -o pp --output-format csv -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --no-kernels --tokens $TOK --modes 2 --mix $MIX > $GRAFT_REPO_ROOT/$OUT/run_$TOK.txt 2>&1 ); echo "exit $?"
tail -3 $OUT/run_$TOK.txt
f=$(find $OUT/t$TOK -name "*kernel_stats.csv" | head -1); echo $f; head -30 $f | cut -c1-230
