#!/usr/bin/env python3
"""Tensor-parallel decode benchmark.

  multi-GPU node, one process per GPU:
      python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/tp_bench.py --model 70b --mix Q4_K_M
  one GPU, ranks as threads of one process (what the single-GPU boxes of this project can run: the ranks time-share the device,
  so this measures the overhead of the sliced path and of the exchange, not a speed-up):
      python tools/tp_bench.py --share-gpu --tp 2 --model 8b --mix Q8_0

Prints one JSON line: tokens/s of the GROUP (all ranks decode the same sequence), strong scaling."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntransformer_amd import engine as E, tp  # noqa: E402


def build(rank, world, a):
    eng = E.Engine()
    if world > 1:
        eng.tp_configure(rank, world)
    eng.load_synthetic(E.synth_spec(a.model, a.mix), a.ctx)
    return eng


def prompt_tokens(a):
    r = np.random.Generator(np.random.Philox(key=[20260925, 99]))
    vocab = 128000 if a.model in ("8b", "70b") else 256
    return [128000 if a.model in ("8b", "70b") else 256] + [int(t) for t in r.integers(0, vocab, a.prompt_len - 1)]


def run_rank(eng, a, barrier, out, key):
    prompt = prompt_tokens(a)
    eng.forward(prompt, 0)
    pos = len(prompt)
    toks = eng.decode_greedy_steps(prompt[-1], pos, a.warmup)
    pos += a.warmup
    barrier()
    t0 = time.perf_counter()
    toks = eng.decode_greedy_steps(toks[-1], pos, a.steps)
    out[key] = (time.perf_counter() - t0, toks, eng.tp_error())
    barrier()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="70b", choices=["tiny", "small", "8b", "70b"])
    ap.add_argument("--mix", default="Q4_K_M")
    ap.add_argument("--ctx", type=int, default=4096)
    ap.add_argument("--prompt-len", type=int, default=16)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--tp", type=int, default=0, help="ranks (only with --share-gpu; under torch.distributed.run it is WORLD_SIZE)")
    ap.add_argument("--gpus", type=int, default=0, help="accepted for the driver's command line (bench.py's flag): the rank count is WORLD_SIZE")
    ap.add_argument("--share-gpu", action="store_true")
    a = ap.parse_args()
    out = {}
    if a.share_gpu:
        world = max(1, a.tp)
        engines = [build(r, world, a) for r in range(world)]
        if world > 1:
            raws = [e.tp_export()[1] for e in engines]
            for e in engines:
                e.tp_connect(raws=raws)
        bar = threading.Barrier(world)
        threads = [threading.Thread(target=run_rank, args=(engines[k], a, bar.wait, out, k)) for k in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        rank, elapsed = 0, max(v[0] for v in out.values())
        same = all(out[k][1] == out[0][1] for k in out)
        errs = [out[k][2] for k in sorted(out)]
    else:
        import torch
        import torch.distributed as dist
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        os.environ.setdefault("NTK_DEVICE", os.environ.get("LOCAL_RANK", "0"))
        dist.init_process_group("gloo")
        eng = build(rank, world, a)
        if world > 1:
            tp.connect_over_torch(eng, rank, world)
        run_rank(eng, a, dist.barrier, out, 0)
        t = torch.tensor([out[0][0]], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor(out[0][1][:8], dtype=torch.int64)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        same = all(bool((g == gathered[0]).all()) for g in gathered)
        errs = [out[0][2]]
    if rank == 0:   # one line in bench.py's format: `python -m torch.distributed.run --nproc-per-node N ... tools/tp_bench.py --gpus N --steps K --warmup W`
        print(json.dumps({"metric": "decode tokens/sec, ONE sequence over %d tensor-parallel ranks (%s %s, resident, greedy, batch 1)" % (world, a.model, a.mix),
                          "value": round(a.steps / elapsed, 3), "unit": "tokens/s", "n_gpus": 1 if a.share_gpu else world, "tp": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 4), "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "Llama-3.1-%s-shaped %s, %d-token prompt, greedy decode, every rank holds 1/%d of each projection "
                                                 "(csrc/tp.hip: two peer-read all-reduces of the hidden vector per layer)" % (a.model.upper(), a.mix, a.prompt_len, world),
                                     "ctx": a.ctx, "parallelism": "tp%d" % world, "ranks_share_one_gpu": bool(a.share_gpu)},
                          "token_streams_identical": same, "tp_error": errs}), flush=True)


if __name__ == "__main__":
    main()
