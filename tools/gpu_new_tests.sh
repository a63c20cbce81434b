#!/bin/bash
# The tests added since the last full GPU pass, on one box.   usage: bash tools/gpu_new_tests.sh <tag> "<pytest -k expression>" [files...]
TAG=${1:-nt}; K=$2; shift 2; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest ${@:-tests} -m gpu -q -p no:cacheprovider -k "$K" > $OUT/pytest.txt 2>&1; echo "exit $?" >> $OUT/pytest.txt; tail -25 $OUT/pytest.txt
tail -4 gpurun_out/parity_depth.jsonl 2>/dev/null | cut -c1-700
