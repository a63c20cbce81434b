#!/bin/bash
# round 3: batched online softmax + hardware exponential in the decode attention walk, against the previous commit's library on the same box
TAG=${1:-r03i}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
OLD=$GRAFT_REPO_ROOT/ntransformer_amd/libntransformer_hip_old.so
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -p no:cacheprovider -k "attention" > $OUT/pytest_k.log 2>&1; echo "exit $?" >> $OUT/pytest_k.log; tail -3 $OUT/pytest_k.log
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider -k "long_context or logits_match_reference_host_code or generate_tokens" > $OUT/pytest_e.log 2>&1; echo "exit $?" >> $OUT/pytest_e.log; tail -3 $OUT/pytest_e.log
echo "== new"; timeout 200 python tools/attn_bench.py 2>&1 | grep -E "nsplit  1|nsplit  8" | tee $OUT/attn_new.txt
echo "== old"; NTK_LIB_PATH=$OLD timeout 200 python tools/attn_bench.py 2>&1 | grep -E "nsplit  1|nsplit  8" | tee $OUT/attn_old.txt
for rep in 1 2; do
NTK_LIB_PATH=$OLD timeout 600 python bench.py --no-cpu-baseline --no-also --prompt-bench 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('headline old', b['value'], b['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --no-also --prompt-bench 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('headline new', b['value'], b['ms_per_step'])"
NTK_LIB_PATH=$OLD timeout 600 python bench.py --prompt-len 3900 --steps 64 --no-cpu-baseline --no-also --prompt-bench 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('ctx3900 old', b['value'], b['ms_per_step'])"
timeout 600 python bench.py --prompt-len 3900 --steps 64 --no-cpu-baseline --no-also --prompt-bench 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('ctx3900 new', b['value'], b['ms_per_step'])"
done 2>&1 | tee $OUT/bench_ab.txt
