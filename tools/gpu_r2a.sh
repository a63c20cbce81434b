#!/bin/bash
# round-2 run A: the whole GPU suite (new real-config parity tests included) + the default bench line
mkdir -p gpurun_out/r02a
export OMP_WAIT_POLICY=passive
( time python -m pytest tests -m gpu -q --durations=15 ) > gpurun_out/r02a/pytest.log 2>&1
tail -30 gpurun_out/r02a/pytest.log
( time python bench.py ) > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
cat gpurun_out/r02a/bench.json
tail -5 gpurun_out/r02a/bench.err
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; free -g | head -2; df -h /dev/shm | tail -1
