#!/bin/bash
# After a late change to the prompt path only: its tests, smoke(), and one full default-style bench line (short CPU leg skipped).   usage: bash tools/gpu_last_check.sh <tag>
TAG=${1:-last}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 200 python -m pytest tests/test_hip_kernels.py tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider -k "row_maxima or folded or rope_kv_store or gemm_quant_f16 or prefill or prompt or logits_match or golden" > $OUT/pytest.txt 2>&1; echo "exit $?" >> $OUT/pytest.txt; tail -3 $OUT/pytest.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 200 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; wc -c $OUT/bench.json; cut -c1-160 $OUT/bench.json
