#!/bin/bash
# round 3: per-kernel time of a 1024-token 8B Q8_0 prompt pass (rocprofv3 --kernel-trace --stats)
TAG=${1:-r03ab}; MIX=${2:-Q8_0}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p -o g -- python $R/tools/prefill_bench.py --no-kernels --mix $MIX --tokens 1024 --modes 2 > $R/$OUT/run.log 2>&1 )
f=$(find $OUT/p -name "*kernel_stats.csv" | head -1); cp $f $OUT/prompt_1024_8b_${MIX}_kernel_stats.csv; head -14 $f | cut -c1-200; rm -rf $OUT/p
