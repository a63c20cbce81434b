"""Tensor-parallel decoding: one engine per rank, the ranks' communication buffers connected through hipIpc handles.

`connect_over_files` exchanges the 64-byte handles through a directory (no torch needed: the tests use it);
`connect_over_torch` through an initialised torch.distributed group (tools/tp_bench.py under torch.distributed.run, gloo).
Every rank must afterwards drive its engine with the same calls and the same tokens (the engines exchange partial
vectors inside the forward pass and produce bit-identical logits)."""
from __future__ import annotations

import os
import time
from typing import List

from . import engine as E


def connect_over_files(eng: "E.Engine", rank: int, world: int, directory: str, timeout_s: float = 120.0, run_id: str = "") -> None:
    """run_id: a string every rank of THIS run shares (a job id, a start time): handle files are named after it, so files left in
    the directory by an earlier run are never read (a stale handle maps a buffer that no longer exists)."""
    handle, _ = eng.tp_export()
    stem = "tp_handle_%s_" % run_id if run_id else "tp_handle_"
    tmp = os.path.join(directory, "%s%d.tmp" % (stem, rank))
    with open(tmp, "wb") as f:
        f.write(handle)
    os.replace(tmp, os.path.join(directory, "%s%d" % (stem, rank)))
    handles: List[bytes] = []
    t0 = time.time()
    for r in range(world):
        path = os.path.join(directory, "%s%d" % (stem, r))
        while not os.path.exists(path):
            if time.time() - t0 > timeout_s:
                raise TimeoutError("rank %d never published its handle" % r)
            time.sleep(0.01)
        with open(path, "rb") as f:
            handles.append(f.read())
    eng.tp_connect(handles=handles)


def connect_over_torch(eng: "E.Engine", rank: int, world: int) -> None:
    import torch
    import torch.distributed as dist
    handle, _ = eng.tp_export()
    mine = torch.tensor(list(handle), dtype=torch.uint8)
    gathered = [torch.zeros(64, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(gathered, mine)
    eng.tp_connect(handles=[bytes(t.tolist()) for t in gathered])
