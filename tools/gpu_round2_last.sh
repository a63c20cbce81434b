#!/bin/bash
# last pass of round 2 on the final tree: GPU parity suite, default bench line, 70B Q6_K line (BASELINE config 5)
TAG=${1:-r02last}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
cp gpurun_out/parity_observed.jsonl $OUT/parity_observed.jsonl 2>/dev/null
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench.err; cut -c1-200 $OUT/bench_default.json
timeout 900 python bench.py --model 70b --mix Q6_K --no-cpu-baseline --no-also --prompt-bench 0 > $OUT/bench_70b_q6_k.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_70b_q6_k.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
