#!/usr/bin/env python3
"""Micro-benchmark of the dequant-fused GEMV launches at the real Llama-3.1 8B / 70B shapes: achieved HBM GB/s
per shape (algorithmic bytes = rows * row_bytes).  Successive launches walk a >1 GB pool of weight bytes so
the 256 MiB Infinity Cache cannot serve them (in decode every token streams the whole model).
usage: python tools/gemv_bench.py [--dtypes Q8_0,Q4_K,Q6_K] [--json out.json]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntransformer_amd import _lib, gguf as G, ops  # noqa: E402
from ntransformer_amd.ops import DeviceBuffer as DB  # noqa: E402

SHAPES = {  # name: (kind, rows..., in)
    "8b.q": ("plain", 4096, 4096), "8b.kv": ("plain", 1024, 4096), "8b.qkv_fused": ("qkv", (4096, 1024, 1024), 4096),
    "8b.o+res": ("resid", 4096, 4096), "8b.gate|up+silu": ("silu", 14336, 4096), "8b.down+res": ("resid", 4096, 14336),
    "lm_head": ("plain", 128256, 4096),
    "70b.qkv_fused": ("qkv", (8192, 1024, 1024), 8192), "70b.o+res": ("resid", 8192, 8192),
    "70b.gate|up+silu": ("silu", 28672, 8192), "70b.down+res": ("resid", 8192, 28672),
}
GT = {"Q8_0": G.GGML_Q8_0, "Q4_0": G.GGML_Q4_0, "Q4_K": G.GGML_Q4_K, "Q5_K": G.GGML_Q5_K, "Q6_K": G.GGML_Q6_K}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtypes", default="Q8_0,Q4_K,Q6_K")
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--pool-mb", type=int, default=1536)
    ap.add_argument("--json", default=None)
    ap.add_argument("--rp", action="store_true", help="the matrix-core GEMV over repacked tensors (ntk_gemv_rp_fused); GB/s still counts GGUF bytes")
    ap.add_argument("--nw", type=int, default=0, help="--rp with a tuning build (make tune, NTK_LIB_PATH): waves per workgroup, 0 = planner")
    ap.add_argument("--sweep", action="store_true", help="--rp with a tuning build: every (waves per workgroup, workgroups per CU) the planner could take, per shape")
    a = ap.parse_args()
    ops.init(0)
    L = _lib.lib()
    global HIP
    HIP = C.CDLL("libamdhip64.so")
    if a.nw:
        L.ntk_tune_rp_waves(a.nw)
    pool_bytes = a.pool_mb << 20
    rng = np.random.default_rng(0)
    # valid-looking blocks are irrelevant for timing; small-magnitude bytes keep fp16 scales finite
    pool = DB.from_numpy(rng.integers(0, 60, pool_bytes, dtype=np.uint8))
    results = []
    ev0, ev1 = L.ntk_event_create(), L.ntk_event_create()
    for dname in a.dtypes.split(","):
        gt = GT[dname]
        dt = G.GGML_TO_DT[gt]
        for sname in a.shapes.split(","):
            kind, rows, in_f = SHAPES[sname]
            rlist = list(rows) if isinstance(rows, tuple) else [rows]
            nmat = 2 if kind == "silu" else len(rlist)
            rb = G.row_bytes(gt, in_f)
            per_launch = rb * (sum(rlist) if kind != "silu" else 2 * rlist[0])
            per_launch_al = (per_launch + 4095) // 4096 * 4096
            if a.rp:
                if gt not in (G.GGML_Q4_K, G.GGML_Q5_K, G.GGML_Q6_K):
                    continue
                rpb = [ops.rp_bytes(dt, r, in_f) for r in (rlist if kind != "silu" else [rlist[0], rlist[0]])]
                rpb = [(b + 255) // 256 * 256 for b in rpb]
                per_launch_al = (sum(rpb) + 4095) // 4096 * 4096
            nslots = max(2, pool_bytes // per_launch_al)
            x = DB.from_numpy(rng.standard_normal(in_f).astype(np.float32))
            nw = DB.from_numpy(np.ones(in_f, np.float32))
            ys = [DB.zeros(max(r, 1) * 4) for r in (rlist if kind != "silu" else [rlist[0], rlist[0]])]

            def launch(slot):
                base = pool.ptr + slot * per_launch_al
                if a.rp:
                    segs, off = [], 0
                    for i, b in enumerate(rpb):
                        segs.append((base + off, ys[i], rlist[0] if kind == "silu" else rlist[i], dt))
                        off += b
                    ops.gemv_rp_fused(segs, x, in_f, norm_w=nw if kind in ("qkv", "silu") else None, eps=1e-5,
                                      resid=ys[0] if kind == "resid" else None, silu_pair=kind == "silu")
                    return
                if kind == "plain":
                    ops.launch_gemv(ys[0], base, x, rlist[0], in_f, dt)
                elif kind == "resid":
                    ops.gemv_fused([(base, ys[0], rlist[0], dt)], x, in_f, resid=ys[0])
                elif kind == "qkv":
                    segs, off = [], 0
                    for i, r in enumerate(rlist):
                        segs.append((base + off, ys[i], r, dt))
                        off += r * rb
                    ops.gemv_fused(segs, x, in_f, norm_w=nw, eps=1e-5)
                else:
                    ops.gemv_fused([(base, ys[0], rlist[0], dt), (base + rlist[0] * rb, ys[1], rlist[0], dt)], x, in_f,
                                   norm_w=nw, eps=1e-5, silu_pair=True)
            def measure():
                for s in range(min(nslots, 4)):
                    launch(s)
                ops.synchronize()
                n = max(nslots, 200 if per_launch < (64 << 20) else 20)
                # capture the n launches into one hipGraph: back-to-back on the device like a decode token, no host
                # launch cost in the measurement (eager launches from Python are host-bound below ~10 us per kernel)
                stream = L.ntk_stream(0)
                graph, gexec = C.c_void_p(), C.c_void_p()
                assert HIP.hipStreamBeginCapture(C.c_void_p(stream), 1) == 0
                for i in range(n):
                    launch(i % nslots)
                assert HIP.hipStreamEndCapture(C.c_void_p(stream), C.byref(graph)) == 0
                assert HIP.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
                HIP.hipGraphLaunch(gexec, C.c_void_p(stream))
                ops.synchronize()
                L.ntk_event_record(ev0, None)
                HIP.hipGraphLaunch(gexec, C.c_void_p(stream))
                L.ntk_event_record(ev1, None)
                L.ntk_event_synchronize(ev1)
                ms = C.c_float()
                L.ntk_event_elapsed_ms(ev0, ev1, C.byref(ms))
                HIP.hipGraphExecDestroy(gexec)
                HIP.hipGraphDestroy(graph)
                return ms.value * 1e3 / n
            if a.rp and a.sweep:   # every geometry the planner could take for this launch (tuning build)
                rows_out = []
                for nwv in (16, 14, 12, 10, 8, 7, 6, 5, 4):
                    for pc in (1, 2, 3, 4):
                        L.ntk_tune_rp_plan(nwv, pc)
                        try:
                            us = measure()
                        except _lib.NtkError:
                            continue
                        plan = (C.c_int * 3)()
                        L.ntk_tune_rp_last_plan(plan)
                        rows_out.append((us, nwv, pc, plan[1], plan[2]))
                L.ntk_tune_rp_plan(0, 0)
                us0 = measure()
                plan = (C.c_int * 3)()
                L.ntk_tune_rp_last_plan(plan)
                rows_out.sort()
                print("%-8s %-18s planner: nw %d grid %d lds %d -> %.2f us; best: %s" % (dname + ".rp", sname, plan[0], plan[1], plan[2], us0,
                      "  ".join("nw%d/cu%d g%d %.2f" % (r[1], r[2], r[3], r[0]) for r in rows_out[:6])), flush=True)
                continue
            us = measure()
            gbs = per_launch / (us * 1e-6) / 1e9
            results.append({"dtype": dname + (".rp" if a.rp else ""), "shape": sname, "bytes": per_launch, "us": round(us, 2), "GBs": round(gbs, 1),
                            "frac_8TBs": round(gbs / 8000, 4)})
            print("%-8s %-18s %9.2f MB %8.2f us %8.1f GB/s  %5.1f%% of 8 TB/s" % (dname + (".rp" if a.rp else ""), sname, per_launch / 1e6, us, gbs, gbs / 80), flush=True)
    if a.json:
        json.dump(results, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
