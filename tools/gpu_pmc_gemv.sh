#!/bin/bash
# SQ counters of the GEMV launches per weight format: where do the waves of the K-quant decoders wait?  (PMC pass, kernel-trace only)
TAG=${1:-pmcgemv}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $R/$OUT/pmc -o g -- python $R/tools/gemv_bench.py --dtypes Q8_0,Q4_K,Q6_K --shapes "8b.gate|up+silu,lm_head,70b.gate|up+silu" > $R/$OUT/run.log 2> $R/$OUT/run.err ); echo "exit $?"
F=$(ls $OUT/pmc/*counter_collection.csv | head -1)
python - "$F" <<'PY' | tee $OUT/summary.txt
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if 'gemv_quant' not in k: continue
    m = re.search(r'gemv_quant_kernel<(\d+)', k)
    acc[(m.group(1) if m else k[:40], r['Grid_Size'], r.get('LDS_Block_Size', ''))][r['Counter_Name']].append(float(r['Counter_Value']))
names = {'2': 'Q8_0', '3': 'Q4_0', '4': 'Q4_K', '5': 'Q6_K', '6': 'Q5_K'}
for (k, g, lds), c in sorted(acc.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    wc = m.get('SQ_WAVE_CYCLES', 1)
    print('%-5s grid %-8s lds %-6s n %-4d' % (names.get(k, k), g, lds, len(c['SQ_WAVE_CYCLES'])),
          'wave_cyc %.3g valu_insts %.3g | wait_any %.0f%% wait_inst %.0f%% (lds %.0f%%) active %.0f%% (valu %.0f%% lds %.0f%%)' % (
              wc, m.get('SQ_INSTS_VALU', 0), 100 * m.get('SQ_WAIT_ANY', 0) / wc, 100 * m.get('SQ_WAIT_INST_ANY', 0) / wc,
              100 * m.get('SQ_WAIT_INST_LDS', 0) / wc, 100 * m.get('SQ_ACTIVE_INST_ANY', 0) / wc,
              100 * m.get('SQ_ACTIVE_INST_VALU', 0) / wc, 100 * m.get('SQ_ACTIVE_INST_LDS', 0) / wc))
PY
rm -rf $OUT/pmc
