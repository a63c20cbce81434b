#!/bin/bash
# Round-2 evidence pass: GPU parity suite, default bench line (with config.also and the reference-CLI CPU baseline), prompt
# benchmark, rocprofv3 kernel trace of the bench command and the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs,
# kernel-trace only), sampling benchmark.  Outputs under gpurun_out/<tag>/; the summaries are copied to profiles/ by hand.
set -u
TAG=${1:-r02final}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
cp gpurun_out/parity_observed.jsonl $OUT/parity_observed.jsonl 2>/dev/null
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench.err; cat $OUT/bench_default.json
timeout 600 python tools/prefill_bench.py > $OUT/prefill_bench.txt 2>&1; grep "prompt of" $OUT/prefill_bench.txt
timeout 600 python tools/prefill_bench.py --no-kernels --mix Q4_K_M > $OUT/prefill_bench_q4_k_m.txt 2>&1; grep "prompt of" $OUT/prefill_bench_q4_k_m.txt
prof() { name=$1; shift; ( cd /tmp && timeout 600 rocprofv3 "$@" > $R/$OUT/$name.json 2> $R/$OUT/$name.err ); echo "$name exit $?"; }
prof trace --kernel-trace --stats -d $R/$OUT/trace -o bench -- python $R/bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-also --prompt-bench 0
[ -f $OUT/trace/bench_results.db ] && python tools/prof_summary.py $OUT/trace/bench_results.db > $OUT/summary_trace.txt && cat $OUT/summary_trace.txt
prof pmc_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-also --no-graph --prompt-bench 0
prof pmc_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-also --no-graph --prompt-bench 0
prof pmc_fetch_q4km --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch_q4km -o bench -- python $R/bench.py --mix Q4_K_M --steps 8 --warmup 2 --no-cpu-baseline --no-also --no-graph --prompt-bench 0
prof pmc_write_q4km --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write_q4km -o bench -- python $R/bench.py --mix Q4_K_M --steps 8 --warmup 2 --no-cpu-baseline --no-also --no-graph --prompt-bench 0
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
F2=$(ls $OUT/pmc_fetch_q4km/*counter_collection.csv 2>/dev/null | head -1); W2=$(ls $OUT/pmc_write_q4km/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$F2" ] && python tools/pmc_summary.py $F2 $W2 --json $OUT/pmc_traffic.json --key 8b_q4_k_m > $OUT/pmc_summary_q4km.txt 2>&1; cat $OUT/pmc_summary_q4km.txt
F=$(ls $OUT/pmc_fetch/*counter_collection.csv 2>/dev/null | head -1); W=$(ls $OUT/pmc_write/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$F" ] && python tools/pmc_summary.py $F $W --json $OUT/pmc_traffic.json --key 8b_q8_0 --algorithmic-bytes-per-launch 61811624 > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
for d in pmc_fetch pmc_write pmc_fetch_q4km pmc_write_q4km trace; do rm -rf $OUT/$d; done
timeout 300 python tools/gemv_bench.py --dtypes Q8_0,Q4_K,Q6_K > $OUT/gemv_bench.txt 2>&1; grep "Q8_0" $OUT/gemv_bench.txt
timeout 300 python tools/attn_bench.py > $OUT/attention_by_context.txt 2>&1; grep "^8b" $OUT/attention_by_context.txt | head -8
[ -f ntransformer_amd/libntransformer_hip_trace.so ] && timeout 300 python tools/gemv_trace.py > $OUT/gemv_launch_timeline.txt 2>&1
timeout 600 python bench.py --prompt-len 3900 --no-cpu-baseline --no-also --prompt-bench 0 > $OUT/bench_ctx3900.json 2>/dev/null; cut -c1-160 $OUT/bench_ctx3900.json
timeout 300 python tools/sampling_bench.py > $OUT/sampling_bench.txt 2>&1; tail -8 $OUT/sampling_bench.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log; tail -3 $OUT/smoke.log
tail -5 $OUT/bench.err; du -sh $OUT; ls $OUT
