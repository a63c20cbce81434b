#!/bin/bash
# round 3, first GPU pass: the new full-depth parity test at the headline config, regression of the refactored engine paths,
# tensor-parallel tests (fine-grained comm buffer, 8 processes), a default bench line of this tree on this box
TAG=${1:-r03a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rm -f gpurun_out/parity_depth.jsonl gpurun_out/parity_observed.jsonl
timeout 900 python -m pytest tests/test_parity_depth.py -m gpu -q -x -p no:cacheprovider -k "8b_q8_0" > $OUT/pytest_depth.log 2>&1; echo "exit $?" >> $OUT/pytest_depth.log; tail -15 $OUT/pytest_depth.log
cp gpurun_out/parity_depth.jsonl $OUT/ 2>/dev/null
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -p no:cacheprovider -k "logits_match_reference_host_code or experiments or generate_tokens or long_context or synthetic_loader" > $OUT/pytest_engine.log 2>&1; echo "exit $?" >> $OUT/pytest_engine.log; tail -5 $OUT/pytest_engine.log
timeout 600 python -m pytest tests/test_tp_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest_tp.log 2>&1; echo "exit $?" >> $OUT/pytest_tp.log; tail -12 $OUT/pytest_tp.log
timeout 600 python bench.py --no-cpu-baseline --prompt-bench 0 > $OUT/bench.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench.json; python - <<'PY'
import json
try:
    b=json.load(open("gpurun_out/%s/bench.json" % "r03a"))
    print("also:", [(a.get("workload","")[:28], a.get("value")) for a in b["config"].get("also",[])])
except Exception as e: print("bench parse", e)
PY
