#!/bin/bash
# quick GPU pass: full parity suite + default bench line (+ optional extra commands passed as arguments)
TAG=${1:-quick}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -15 $OUT/pytest_gpu.log
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_8b_q8_0.json 2> $OUT/bench.err; cat $OUT/bench_8b_q8_0.json
for c in "$@"; do echo "== $c"; eval "$c"; done
