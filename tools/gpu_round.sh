#!/bin/bash
# One GPU-box visit: parity tests, smoke, micro-benchmarks, bench, rocprof.  Everything lands in gpurun_out/.
# usage: tools/gpu_round.sh <tag> [stages...]   stages: tests smoke micro bench prof bench70 (default: all but bench70)
set -u
TAG=${1:-r01}; shift || true
STAGES=${*:-tests smoke micro bench prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
has() { [[ " $STAGES " == *" $1 "* ]]; }
rocminfo 2>/dev/null | grep -m2 -E "Marketing Name|gfx" > $OUT/device.txt; nproc >> $OUT/device.txt; free -g | head -2 >> $OUT/device.txt
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log
fi
if has tests_all; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log
fi
if has smoke; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
fi
if has micro; then
  timeout 600 python tools/gemv_bench.py --json $OUT/gemv_bench.json > $OUT/gemv_bench.log 2>&1; tail -40 $OUT/gemv_bench.log
fi
if has bench; then
  timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cat $OUT/bench.json; tail -3 $OUT/bench.err
  timeout 300 python bench.py --no-graph --no-cpu-baseline --steps 64 > $OUT/bench_nograph.json 2>> $OUT/bench.err; cat $OUT/bench_nograph.json
  timeout 300 python bench.py --no-fuse --no-cpu-baseline --steps 32 > $OUT/bench_nofuse.json 2>> $OUT/bench.err; cat $OUT/bench_nofuse.json
fi
if has prof; then
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 4 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err )
  find $OUT/prof -name "*kernel_stats*" | head; f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
  find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
fi
if has bench70; then
  timeout 1500 python bench.py --model 70b --mix Q4_K_M --steps 64 --warmup 4 --no-cpu-baseline > $OUT/bench_70b_q4km.json 2> $OUT/bench70.err; cat $OUT/bench_70b_q4km.json; tail -3 $OUT/bench70.err
fi
ls -la $OUT
