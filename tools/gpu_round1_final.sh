#!/bin/bash
# Round-1, third GPU pass: parity suite after the DPP-reduction / cursor / A16 rewrite, micro-bench, bench lines,
# rocprofv3 kernel trace and the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel-trace only).
set -u
TAG=${1:-r01d}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log
timeout 600 python tools/gemv_bench.py --json $OUT/gemv_bench.json > $OUT/gemv_bench.log 2>&1; tail -50 $OUT/gemv_bench.log
timeout 900 python bench.py > $OUT/bench_8b_q8_0.json 2> $OUT/bench.err; cat $OUT/bench_8b_q8_0.json
timeout 600 python bench.py --mix Q4_K_M --no-cpu-baseline > $OUT/bench_8b_q4_k_m.json 2>> $OUT/bench.err; cat $OUT/bench_8b_q4_k_m.json
timeout 1200 python bench.py --model 70b --mix Q4_K_M --steps 64 --warmup 4 --no-cpu-baseline > $OUT/bench_70b_q4_k_m.json 2>> $OUT/bench.err; cat $OUT/bench_70b_q4_k_m.json
timeout 1200 python bench.py --model 70b --mix Q6_K --steps 64 --warmup 4 --no-cpu-baseline > $OUT/bench_70b_q6_k.json 2>> $OUT/bench.err; cat $OUT/bench_70b_q6_k.json
prof() { name=$1; shift; ( cd /tmp && timeout 600 rocprofv3 "$@" > $R/$OUT/$name.json 2> $R/$OUT/$name.err ); echo "$name exit $?"; }
prof trace --kernel-trace --stats -d $R/$OUT/trace -o bench -- python $R/bench.py --steps 32 --warmup 4 --no-cpu-baseline
[ -f $OUT/trace/bench_results.db ] && python tools/prof_summary.py $OUT/trace/bench_results.db > $OUT/summary_trace.txt && cat $OUT/summary_trace.txt
prof pmc_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-graph
prof pmc_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-graph
prof pmc_fetch_q4km --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch_q4km -o bench -- python $R/bench.py --mix Q4_K_M --steps 8 --warmup 2 --no-cpu-baseline --no-graph
prof pmc_write_q4km --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write_q4km -o bench -- python $R/bench.py --mix Q4_K_M --steps 8 --warmup 2 --no-cpu-baseline --no-graph
F2=$(ls $OUT/pmc_fetch_q4km/*counter_collection.csv 2>/dev/null | head -1); W2=$(ls $OUT/pmc_write_q4km/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$F2" ] && python tools/pmc_summary.py $F2 $W2 --json $OUT/pmc_traffic.json --key 8b_q4_k_m > $OUT/pmc_summary_q4km.txt 2>&1; cat $OUT/pmc_summary_q4km.txt
F=$(ls $OUT/pmc_fetch/*counter_collection.csv 2>/dev/null | head -1); W=$(ls $OUT/pmc_write/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$F" ] && python tools/pmc_summary.py $F $W --json $OUT/pmc_traffic.json --key 8b_q8_0 --algorithmic-bytes-per-launch 61811624 > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
# keep the merge-back small: the per-dispatch CSVs are large
for d in pmc_fetch pmc_write pmc_fetch_q4km pmc_write_q4km; do for f in $OUT/$d/*kernel_trace.csv $OUT/$d/*agent_info.csv; do rm -f $f; done; done
[ -n "$F" ] && gzip -9 $F; [ -n "$W" ] && gzip -9 $W; [ -n "$F2" ] && gzip -9 $F2; [ -n "$W2" ] && gzip -9 $W2
tail -5 $OUT/bench.err; du -sh $OUT; ls $OUT
