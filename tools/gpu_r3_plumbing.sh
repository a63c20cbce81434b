#!/bin/bash
# round 3: the N > 1 path of bench.py on the one GPU of a gpurun box -- two ranks (replicas) sharing it, launched exactly as the driver
# launches N ranks on N GPUs (torch.distributed.run, RCCL barrier + max-reduce around the timed region).  Plumbing evidence, not a scaling number.
TAG=${1:-r03an}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 16 --warmup 4 --no-cpu-baseline --no-also --prompt-bench 0 > $OUT/plumbing_2ranks.json 2> $OUT/plumbing_2ranks.err; echo "exit $?"; cut -c1-400 $OUT/plumbing_2ranks.json; tail -3 $OUT/plumbing_2ranks.err
