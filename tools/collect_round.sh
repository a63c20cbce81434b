#!/bin/bash
# Copies the summaries of a tools/gpu_round_final.sh pass (gpurun_out/<tag>/) into profiles/ under the round's names.
# usage: bash tools/collect_round.sh <tag> <rNN>
TAG=${1:-r05final}; R=${2:-r05}; S=gpurun_out/$TAG; D=profiles
set -e
cp $S/pytest_gpu.log $D/${R}_pytest_gpu.log
cp $S/smoke.log $D/${R}_smoke.log
cp $S/parity_depth.jsonl $D/${R}_parity_depth.jsonl; cp $S/parity_observed.jsonl $D/${R}_parity_observed.jsonl
cp $S/bench_default.json $D/${R}_bench_default_8b_q8_0_with_also.json
cp $S/bench_driver_style_8b_q8_0.json $D/${R}_bench_driver_style_n1_steps20.json
for K in 8b_q8_0 8b_q4_k_m 70b_q4_k_m 70b_q6_k 8b_q8_0_ctx3900 8b_q8_0_ctx32768; do
  [ -f $S/bench_$K.json ] || continue
  cp $S/bench_$K.json $D/${R}_bench_$K.json
  [ -f $S/trace_$K.json ] && cp $S/trace_$K.json $D/${R}_bench_${K}_profiled.json
  cp $S/summary_trace_$K.txt $D/${R}_rocprofv3_kernel_trace_$K.txt
  [ -f $S/pmc_summary_$K.txt ] && cp $S/pmc_summary_$K.txt $D/${R}_pmc_fetch_write_$K.txt
done
python - $S/pmc_traffic.json $D/pmc_traffic.json <<'PY'
import json, sys
new, old = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
old.update(new)
json.dump(old, open(sys.argv[2], "w"), indent=1, sort_keys=True)
PY
python - $S/trace_gemv.json $D/trace_gemv.json <<'PY'
import json, os, sys
new = json.load(open(sys.argv[1]))
old = json.load(open(sys.argv[2])) if os.path.exists(sys.argv[2]) else {}
old.update(new)
json.dump(old, open(sys.argv[2], "w"), indent=1, sort_keys=True)
PY
[ -f $S/reference_transformer_on_rp_kernels.txt ] && cp $S/reference_transformer_on_rp_kernels.txt $D/${R}_reference_transformer_on_rp_kernels.txt
cp $S/gemv_bench.txt $D/${R}_gemv_microbench.txt
cp $S/attention_by_context.txt $D/${R}_attention_by_context.txt
cp $S/prefill_bench.txt $D/${R}_prefill_bench.txt
cat $S/commit.txt 2>/dev/null
