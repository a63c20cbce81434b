#!/bin/bash
# Copies the summaries of a tools/gpu_round3_final.sh pass (gpurun_out/<tag>/) into profiles/ under their round-3 names.
TAG=${1:-r03final}; S=gpurun_out/$TAG; D=profiles
set -e
cp $S/pytest_gpu.log $D/r03_pytest_gpu.log
cp $S/smoke.log $D/r03_smoke.log
cp $S/parity_depth.jsonl $D/r03_parity_depth.jsonl; cp $S/parity_observed.jsonl $D/r03_parity_observed.jsonl
cp $S/bench_default.json $D/r03_bench_default_8b_q8_0_with_also.json
cp $S/bench_driver_style_8b_q8_0.json $D/r03_bench_driver_style_n1_steps20.json
for K in 8b_q8_0 8b_q4_k_m 70b_q4_k_m; do
  cp $S/bench_$K.json $D/r03_bench_$K.json
  cp $S/summary_trace_$K.txt $D/r03_rocprofv3_kernel_trace_$K.txt
  cp $S/pmc_summary_$K.txt $D/r03_pmc_fetch_write_$K.txt
done
python - $S/pmc_traffic.json $D/pmc_traffic.json <<'PY'
import json, sys
new, old = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
old.update(new)
json.dump(old, open(sys.argv[2], "w"), indent=1, sort_keys=True)
PY
cp $S/gemv_bench.txt $D/r03_gemv_microbench.txt
cp $S/attention_by_context.txt $D/r03_attention_by_context.txt
cp $S/prefill_bench.txt $D/r03_prefill_bench.txt
cp $S/prompt_1024_8b_q8_0_kernel_stats.csv $D/r03_rocprofv3_prompt_1024_8b_q8_0_kernel_stats.csv
cp $S/prompt_gemm_pmc_8b_q8_0.txt $D/r03_prompt_gemm_pmc_8b_q8_0.txt
cat $S/commit.txt 2>/dev/null
