#!/bin/bash
# Last call of a round: the whole GPU suite and smoke() on the final tree, then the default bench line and the driver-style line (what the driver will run).
TAG=${1:-final}; OUT=gpurun_out/$TAG; mkdir -p $OUT
git -C $GRAFT_REPO_ROOT rev-parse HEAD > $OUT/commit.txt 2>/dev/null
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench.err; wc -c $OUT/bench_default.json; cut -c1-200 $OUT/bench_default.json
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2>> $OUT/bench.err; wc -c $OUT/bench_driver_style.json; cut -c1-160 $OUT/bench_driver_style.json
