#!/bin/bash
# rocprofv3 kernel trace of bench.py (decode window summary -> profiles/).  Tries the hipGraph path first; rocprofv3 1.1.0
# has been seen to segfault inside hipGraph replay, in which case the eager (--no-graph) run is profiled: same kernels.
TAG=${1:-prof}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/$name -o bench -- python $GRAFT_REPO_ROOT/bench.py "$@" > $GRAFT_REPO_ROOT/$OUT/$name.json 2> $GRAFT_REPO_ROOT/$OUT/$name.err ); echo "$name exit $?"; }
run graph --steps 32 --warmup 4 --no-cpu-baseline
if [ -f $OUT/graph/bench_results.db ]; then python tools/prof_summary.py $OUT/graph/bench_results.db > $OUT/summary_graph.txt; cat $OUT/summary_graph.txt; fi
run eager --steps 32 --warmup 4 --no-cpu-baseline --no-graph
if [ -f $OUT/eager/bench_results.db ]; then python tools/prof_summary.py $OUT/eager/bench_results.db > $OUT/summary_eager.txt; cat $OUT/summary_eager.txt; fi
cat $OUT/graph.json $OUT/eager.json
ls -la $OUT
