#!/bin/bash
# Evidence pass of a round on ONE tree and ONE box: GPU parity suite (incl. the full-depth arbiter tests and, with NT_RUN_SLOW=1, the
# 80-layer 70B pass), the default bench line (five workloads + reference-CLI CPU baseline), the driver-style line, and for each workload
# the metric names a rocprofv3 kernel trace of the bench command (the profiled process's own bench line kept beside it: it carries the
# shader clock of the profiled pass), the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel-trace only) and the same command
# un-profiled.  Outputs under gpurun_out/<tag>/; tools/collect_round.sh copies the summaries to profiles/.
set -u
TAG=${1:-r05final}; RN=${2:-r05}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
git -C $R rev-parse HEAD > $OUT/commit.txt 2>/dev/null
env | grep -c "^ROCP" > /dev/null
rm -f gpurun_out/parity_depth.jsonl gpurun_out/parity_observed.jsonl
NT_RUN_SLOW=${NT_RUN_SLOW:-1} timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
cp gpurun_out/parity_observed.jsonl gpurun_out/parity_depth.jsonl $OUT/ 2>/dev/null
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench.err; wc -c $OUT/bench_default.json; cut -c1-300 $OUT/bench_default.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --prompt-bench 0 > $OUT/bench_driver_style_8b_q8_0.json 2>> $OUT/bench.err
prof() { name=$1; shift; ( cd /tmp && timeout 900 rocprofv3 "$@" > $R/$OUT/$name.json 2> $R/$OUT/$name.err ); echo "$name exit $?"; }
for w in "8b Q8_0 32" "8b Q4_K_M 32" "70b Q4_K_M 16" "70b Q6_K 16"; do set -- $w; M=$1; X=$2; ST=$3; K=${M}_$(echo $X | tr A-Z a-z)
  read BYTES NL <<< $(python tools/gemv_bytes.py $M $X)
  prof trace_$K --kernel-trace --stats -d $R/$OUT/trace_$K -o bench -- python $R/bench.py --model $M --mix $X --steps $ST --warmup 4 --no-cpu-baseline --no-also --prompt-bench 0
  # (rocprofv3 1.1.0 has been seen to segfault inside hipGraph replay: the eager launches are the same kernels)
  [ -f $OUT/trace_$K/bench_results.db ] || prof trace_$K --kernel-trace --stats -d $R/$OUT/trace_$K -o bench -- python $R/bench.py --model $M --mix $X --steps $ST --warmup 4 --no-cpu-baseline --no-also --no-graph --prompt-bench 0
  [ -f $OUT/trace_$K/bench_results.db ] && python tools/prof_summary.py $OUT/trace_$K/bench_results.db --gemv-bytes-per-token $BYTES --json $OUT/trace_gemv.json --key $K --file ${RN}_rocprofv3_kernel_trace_$K.txt > $OUT/summary_trace_$K.txt && head -14 $OUT/summary_trace_$K.txt
  prof pmc_fetch_$K --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch_$K -o bench -- python $R/bench.py --model $M --mix $X --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-graph --prompt-bench 0
  prof pmc_write_$K --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write_$K -o bench -- python $R/bench.py --model $M --mix $X --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-graph --prompt-bench 0
  F=$(ls $OUT/pmc_fetch_$K/*counter_collection.csv 2>/dev/null | head -1); W=$(ls $OUT/pmc_write_$K/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$F" ] && python tools/pmc_summary.py $F $W --json $OUT/pmc_traffic.json --key $K --algorithmic-bytes-per-launch $(python -c "print($BYTES / $NL)") > $OUT/pmc_summary_$K.txt 2>&1; tail -3 $OUT/pmc_summary_$K.txt
  # the same workload un-profiled, same box, same commit: the line the trace is checked against (tools/check_evidence.py)
  timeout 600 python bench.py --model $M --mix $X --steps $ST --warmup 4 --no-cpu-baseline --no-also --prompt-bench 0 > $OUT/bench_$K.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_$K.json
  rm -rf $OUT/trace_$K $OUT/pmc_fetch_$K $OUT/pmc_write_$K
done
# the long-context workload (8B Q8_0 behind a 3900-token prompt: split attention on the matrix cores + combine): kernel trace + the same un-profiled
K=8b_q8_0_ctx3900; read BYTES NL <<< $(python tools/gemv_bytes.py 8b Q8_0)
prof trace_$K --kernel-trace --stats -d $R/$OUT/trace_$K -o bench -- python $R/bench.py --prompt-len 3900 --steps 16 --warmup 4 --no-cpu-baseline --no-also --prompt-bench 0
[ -f $OUT/trace_$K/bench_results.db ] || prof trace_$K --kernel-trace --stats -d $R/$OUT/trace_$K -o bench -- python $R/bench.py --prompt-len 3900 --steps 16 --warmup 4 --no-cpu-baseline --no-also --no-graph --prompt-bench 0
[ -f $OUT/trace_$K/bench_results.db ] && python tools/prof_summary.py $OUT/trace_$K/bench_results.db --gemv-bytes-per-token $BYTES --json $OUT/trace_gemv.json --key $K --file ${RN}_rocprofv3_kernel_trace_$K.txt > $OUT/summary_trace_$K.txt && head -12 $OUT/summary_trace_$K.txt
timeout 600 python bench.py --prompt-len 3900 --steps 64 --warmup 4 --no-cpu-baseline --no-also --prompt-bench 0 > $OUT/bench_$K.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_$K.json
rm -rf $OUT/trace_$K
# decode behind a 32768-token prompt (contexts beyond 4096): kernel trace + the same un-profiled
K=8b_q8_0_ctx32768
prof trace_$K --kernel-trace --stats -d $R/$OUT/trace_$K -o bench -- python $R/bench.py --prompt-len 32768 --ctx 33024 --steps 8 --warmup 2 --no-cpu-baseline --no-also --no-graph --prompt-bench 0
[ -f $OUT/trace_$K/bench_results.db ] && python tools/prof_summary.py $OUT/trace_$K/bench_results.db --gemv-bytes-per-token $BYTES --json $OUT/trace_gemv.json --key $K --file ${RN}_rocprofv3_kernel_trace_$K.txt > $OUT/summary_trace_$K.txt && head -12 $OUT/summary_trace_$K.txt
timeout 600 python bench.py --prompt-len 32768 --ctx 33024 --steps 32 --warmup 4 --no-cpu-baseline --no-also --prompt-bench 0 > $OUT/bench_$K.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_$K.json
rm -rf $OUT/trace_$K
# the reference's unmodified nt::Transformer over the binding with the matrix-core K-quant GEMV (NT_HIP_AUTO_REPACK=1): which kernels ran
if [ -x oracle/_ref/ref_logits_hip ]; then
  ( cd /tmp && NT_HIP_AUTO_REPACK=1 NT_HIP_REPACK_STATS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace_refrp -o ref -- $R/oracle/_ref/ref_logits_hip $R/tests/golden/tiny_q4_k_m.gguf 128 /tmp/ref_rp.bin 4 0 256 5 9 17 3 4 5 6 > $R/$OUT/ref_rp.log 2>&1 )
  F=$(ls $OUT/trace_refrp/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$F" ] && ( grep nt_hip_repack $OUT/ref_rp.log; cut -d, -f1-4 $F | head -12 ) > $OUT/reference_transformer_on_rp_kernels.txt; cat $OUT/reference_transformer_on_rp_kernels.txt | cut -c1-160
  rm -rf $OUT/trace_refrp
fi
( echo "== raw GGUF blocks (csrc/gemv.hip)"; timeout 300 python tools/gemv_bench.py --dtypes Q8_0,Q4_K,Q6_K; echo "== engine repack, matrix cores (csrc/gemv_rp.hip)"; timeout 300 python tools/gemv_bench.py --rp --dtypes Q4_K,Q5_K,Q6_K ) > $OUT/gemv_bench.txt 2>&1; grep "rp " $OUT/gemv_bench.txt | head -9
timeout 300 python tools/attn_bench.py > $OUT/attention_by_context.txt 2>&1; grep "^8b" $OUT/attention_by_context.txt | head -8
timeout 300 python tools/attn_bench.py --max-seq 32768 --layers 6 --models 8b --cases 8191:32,32767:32 >> $OUT/attention_by_context.txt 2>&1
timeout 600 python tools/prefill_bench.py --no-kernels --mix Q8_0 --tokens 64,256,1024 --modes 2 > $OUT/prefill_bench.txt 2>&1; timeout 300 python tools/prefill_bench.py --no-kernels --mix Q4_K_M --tokens 64,256,1024 --modes 2 >> $OUT/prefill_bench.txt 2>&1; grep "prompt of" $OUT/prefill_bench.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log; tail -3 $OUT/smoke.log
# optional same-box A/B against an older build of the library (ntransformer_amd/libntransformer_hip_old.so, a tuning build; not committed)
if [ -f ntransformer_amd/libntransformer_hip_old.so ]; then
  SH="8b.qkv_fused,8b.o+res,8b.gate|up+silu,8b.down+res,70b.o+res,70b.down+res"
  for V in old tune tune_nx; do L=$PWD/ntransformer_amd/libntransformer_hip_$V.so; [ -f $L ] || continue
    echo "== $V"; NTK_LIB_PATH=$L timeout 200 python tools/gemv_bench.py --rp --dtypes Q4_K --shapes "$SH" 2>&1 | grep "rp " | cut -c1-70
    NTK_LIB_PATH=$L timeout 200 python bench.py --mix Q4_K_M --no-also --no-cpu-baseline --prompt-bench 0 --steps 64 2>/dev/null | cut -c1-130
    NTK_LIB_PATH=$L timeout 200 python bench.py --prompt-len 3900 --no-also --no-cpu-baseline --prompt-bench 0 --steps 64 2>/dev/null | cut -c1-130
  done > $OUT/ab_old_new.txt 2>&1
  cat $OUT/ab_old_new.txt
fi
tail -5 $OUT/bench.err; du -sh $OUT; ls $OUT
