#!/bin/bash
# round-2 run B: first contact of the persistent decode kernel with the hardware (everything under `timeout`)
mkdir -p gpurun_out/r02b
export OMP_WAIT_POLICY=passive
( timeout 600 python -m pytest tests/test_engine_gpu.py -q -x -k "logits_match_reference_host_code or generate_tokens or long_context or 8b_width" ) > gpurun_out/r02b/pytest_small.log 2>&1
tail -15 gpurun_out/r02b/pytest_small.log
( timeout 300 python bench.py --no-also --no-cpu-baseline --steps 64 ) > gpurun_out/r02b/bench_persistent.json 2> gpurun_out/r02b/bench_persistent.err
cat gpurun_out/r02b/bench_persistent.json; tail -3 gpurun_out/r02b/bench_persistent.err
( timeout 300 python bench.py --no-also --no-cpu-baseline --steps 64 --no-persistent ) > gpurun_out/r02b/bench_launches.json 2> gpurun_out/r02b/bench_launches.err
cat gpurun_out/r02b/bench_launches.json; tail -3 gpurun_out/r02b/bench_launches.err
( timeout 300 python bench.py --no-also --no-cpu-baseline --steps 64 --mix Q4_K_M ) > gpurun_out/r02b/bench_persistent_q4km.json 2> gpurun_out/r02b/bench_persistent_q4km.err
cat gpurun_out/r02b/bench_persistent_q4km.json; tail -3 gpurun_out/r02b/bench_persistent_q4km.err
