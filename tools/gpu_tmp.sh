#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=gpurun_out/prefprof; mkdir -p $OUT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/t -o p -- python $R/tools/prefill_bench.py --no-kernels --tokens 1024 --modes 2 > $R/$OUT/run.log 2>&1 ); echo "exit $?"
grep "prompt of" $OUT/run.log
F=$(find $OUT/t -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -14 $F | cut -c1-200 | tee $OUT/kernel_stats_head.txt
rm -rf $OUT/t
