#!/bin/bash
# SQ counters of the batched prompt GEMM (where do the waves wait?) -- PMC pass, kernel-trace only.
TAG=${1:-pmcgemm}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $R/$OUT/pmc -o g -- python $R/tools/prefill_bench.py --no-engine > $R/$OUT/run.log 2> $R/$OUT/run.err ); echo "exit $?"
F=$(ls $OUT/pmc/*counter_collection.csv | head -1)
python - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if 'gemm_quant' not in k: continue
    acc[(k, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for (k, g), c in sorted(acc.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    wc = m.get('SQ_WAVE_CYCLES', 1)
    print(k[-40:], 'grid', g, 'n', len(c['SQ_WAVE_CYCLES']), ' '.join('%s=%.3g' % (n.replace('SQ_', ''), v) for n, v in sorted(m.items())),
          '| wait_any %.0f%% wait_inst %.0f%% active %.0f%% mfma_busy/busy %.0f%%' % (100 * m.get('SQ_WAIT_ANY', 0) / wc, 100 * m.get('SQ_WAIT_INST_ANY', 0) / wc,
           100 * m.get('SQ_ACTIVE_INST_ANY', 0) / wc, 100 * m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(m.get('SQ_BUSY_CYCLES', 1), 1)))
PY
rm -rf $OUT/pmc
