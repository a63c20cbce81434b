"""Replica-parallel timing for the decode path (the path shards across REQUESTS only --
one whole-model replica per GPU, no data-path collective).  `timed_steps` is the measurement protocol bench.py
uses on every rank: barrier + device synchronise on both sides of exactly K steps, MAX over ranks, and the
whole-job rate N*K/max.  torch.distributed is used only as rendezvous/barrier plumbing (gloo by default, RCCL on request)."""
from __future__ import annotations

import os
import time
from typing import Callable, Optional, Tuple


def env_ranks() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend: str, local_rank: int):
    """Returns (torch.distributed module with the process group initialised, backend used), or (None, None) for a single
    process.  The group carries nothing but a barrier and a MAX-reduce of one double per run -- replicas share no data
     -- so the default is gloo on CPU tensors: no communicator bring-up, no GPU IPC, nothing that can wedge
    one rank while the others wait.  NT_DIST_BACKEND=nccl runs the same two calls over RCCL."""
    rank, _, world = env_ranks()
    if world <= 1:
        return None, None
    import torch
    import torch.distributed as dist
    backend = os.environ.get("NT_DIST_BACKEND", backend)
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", init_method="env://", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
        return dist, "nccl"
    # gloo's transport prints "[Gloo] Rank r is connected to ..." on the C++ stdout: keep the process's stdout for the one
    # JSON line of the bench contract by pointing fd 1 at stderr while the group comes up
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        dist.init_process_group(backend="gloo", init_method="env://", rank=rank, world_size=world)
        dist.barrier()
    finally:
        os.dup2(saved, 1)
        os.close(saved)
    return dist, "gloo"


def timed_steps(run_steps: Callable[[int], object], steps: int, device_sync: Callable[[], None], dist=None,
                backend: Optional[str] = None) -> Tuple[float, float, object]:
    """Time `run_steps(steps)` on every rank.  Returns (max_elapsed_seconds, whole_job_steps_per_second, result)."""
    world = dist.get_world_size() if dist is not None else 1

    def fence():
        if dist is not None:
            dist.barrier()
            if backend == "nccl":
                import torch
                torch.cuda.synchronize()
        device_sync()

    fence()
    t0 = time.perf_counter()
    result = run_steps(steps)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, world * steps / elapsed, result


def shard_requests(n_requests: int, rank: int, world: int):
    """Independent sequences are the only unit the path shards by: request i runs on replica i % world."""
    return [i for i in range(n_requests) if i % world == rank]
