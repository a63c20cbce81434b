#!/bin/bash
# Round-1 evidence run: full GPU test suite, default bench (with CPU baseline), rocprof of the same command, other configs.
set -u
TAG=${1:-r01}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_8b_q8_0.json 2> $OUT/bench.err; cat $OUT/bench_8b_q8_0.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err )
python tools/prof_summary.py $OUT/prof/bench_results.db > $OUT/prof_summary_8b_q8_0.txt 2>&1; cat $OUT/prof_summary_8b_q8_0.txt
python tools/gemv_bench.py --json $OUT/gemv_bench.json > $OUT/gemv_bench.log 2>&1
timeout 600 python bench.py --mix Q4_K_M --no-cpu-baseline > $OUT/bench_8b_q4_k_m.json 2>> $OUT/bench.err; cat $OUT/bench_8b_q4_k_m.json
timeout 1200 python bench.py --model 70b --mix Q4_K_M --steps 64 --warmup 4 --no-cpu-baseline > $OUT/bench_70b_q4_k_m.json 2>> $OUT/bench.err; cat $OUT/bench_70b_q4_k_m.json
timeout 1200 python bench.py --model 70b --mix Q6_K --steps 64 --warmup 4 --no-cpu-baseline > $OUT/bench_70b_q6_k.json 2>> $OUT/bench.err; cat $OUT/bench_70b_q6_k.json
tail -5 $OUT/bench.err
rm -f $OUT/prof/*.db.bak; ls -la $OUT $OUT/prof
