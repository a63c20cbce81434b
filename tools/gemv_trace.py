#!/usr/bin/env python3
"""Timeline of a GEMV launch inside a dependent chain (tuning aid; needs the trace build: make -C ntransformer_amd/csrc trace).
Thread 0 of every workgroup stamps the constant 100 MHz clock at: entry, x landed, activations in registers, first row landed,
first row decoded, last row landed, end.  For each shape a hipGraph of back-to-back launches over rotating weight slots is
replayed; for the launches in the middle of the chain the tool prints the gap to the previous launch's last workgroup and the
spread (min / median / max over workgroups) of every stamp relative to the launch's first entry.
usage: python tools/gemv_trace.py [--dtypes Q8_0] [--shapes ...]
       python tools/gemv_trace.py --rp --dtypes Q4_K,Q6_K     (the matrix-core GEMV over the engine's repack, csrc/gemv_rp.hip: its own stamps)"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntransformer_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libntransformer_hip_trace.so")
from ntransformer_amd import gguf as G, ops  # noqa: E402
from ntransformer_amd.ops import DeviceBuffer as DB  # noqa: E402

SHAPES = {
    "8b.qkv_fused": ("qkv", (4096, 1024, 1024), 4096), "8b.o+res": ("resid", 4096, 4096),
    "8b.gate|up+silu": ("silu", 14336, 4096), "8b.down+res": ("resid", 4096, 14336),
}
GT = {"Q8_0": G.GGML_Q8_0, "Q4_K": G.GGML_Q4_K, "Q5_K": G.GGML_Q5_K, "Q6_K": G.GGML_Q6_K}
RP_SLOTS, RP_WG, RP_EV = 64, 512, 12
RP_NAMES = ["entry", "x requested", "items requested", "image written", "loop done", "last item done", "shares in LDS", "stored"]
SLOTS, WG, EV = 64, 512, 14
NAMES = ["entry", "x landed", "x in regs", "row0 landed", "row0 done", "last landed", "end", "x requested", "row0 requested", "image stored", "image barrier", "own row read"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtypes", default="Q8_0")
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--pool-mb", type=int, default=1536, help="weight pool the launches rotate over (small = Infinity-Cache resident)")
    ap.add_argument("--rp", action="store_true", help="ntk_gemv_rp_fused over repacked copies of the pool's matrices (Q4_K / Q5_K / Q6_K)")
    a = ap.parse_args()
    ops.init(0)
    L = _lib.lib()
    L.ntk_debug_gemv_trace.argtypes = [C.POINTER(C.c_ulonglong), C.c_size_t]
    L.ntk_debug_gemv_trace.restype = C.c_int
    if a.rp:
        L.ntk_debug_rp_trace.argtypes = [C.POINTER(C.c_ulonglong), C.c_size_t]
        L.ntk_debug_rp_trace.restype = C.c_int
        return main_rp(a, L)
    HIP = C.CDLL("libamdhip64.so")
    rng = np.random.default_rng(0)
    pool_bytes = a.pool_mb << 20
    pool = DB.from_numpy(rng.integers(0, 60, pool_bytes, dtype=np.uint8))
    stream = L.ntk_stream(0)
    for dname in a.dtypes.split(","):
        launched = 0   # launches issued so far in this format = the library's trace slot counter (one per format)
        gt = GT[dname]
        dt = G.GGML_TO_DT[gt]
        for sname in a.shapes.split(","):
            kind, rows, in_f = SHAPES[sname]
            rlist = list(rows) if isinstance(rows, tuple) else [rows]
            rb = G.row_bytes(gt, in_f)
            per_launch = rb * (sum(rlist) if kind != "silu" else 2 * rlist[0])
            per_al = (per_launch + 4095) // 4096 * 4096
            nslots = max(2, pool_bytes // per_al)
            x = DB.from_numpy(rng.standard_normal(in_f).astype(np.float32))
            nw = DB.from_numpy(np.ones(in_f, np.float32))
            ys = [DB.zeros(max(r, 1) * 4) for r in (rlist if kind != "silu" else [rlist[0], rlist[0]])]

            def launch(slot):
                base = pool.ptr + slot * per_al
                if kind == "resid":
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

            n = 24
            graph, gexec = C.c_void_p(), C.c_void_p()
            assert HIP.hipStreamBeginCapture(C.c_void_p(stream), 1) == 0
            first = launched
            for i in range(n):
                launch(i % nslots)
                launched += 1
            assert HIP.hipStreamEndCapture(C.c_void_p(stream), C.byref(graph)) == 0
            assert HIP.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
            for _ in range(3):
                HIP.hipGraphLaunch(gexec, C.c_void_p(stream))
            ops.synchronize()
            buf = (C.c_ulonglong * (SLOTS * WG * EV))()
            assert L.ntk_debug_gemv_trace(buf, SLOTS * WG * EV) == 0
            t = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(SLOTS, WG, EV)
            HIP.hipGraphExecDestroy(gexec); HIP.hipGraphDestroy(graph)
            print("== %s %s: %.2f MB per launch, chain of %d (10 ns clock; us relative to the launch's first workgroup entry)" % (dname, sname, per_launch / 1e6, n))
            gaps, totals, rows_ = [], [], []
            agg = {k: [] for k in range(1, 12)}
            for j in range(8, n - 2):   # middle of the chain
                cur, prev = t[(first + j) % SLOTS], t[(first + j - 1) % SLOTS]
                live = cur[:, 0] > 0
                nwg = int(live.sum())
                t0 = cur[live, 0].min()
                gaps.append((t0 - prev[prev[:, 6] > 0, 6].max()) / 100.0)
                totals.append((cur[live, 6].max() - t0) / 100.0)
                for k in range(1, 12):
                    v = cur[live, k]
                    v = v[v > 0]
                    if len(v): agg[k].append(((v.min() - t0) / 100.0, (np.median(v) - t0) / 100.0, (v.max() - t0) / 100.0))
                rows_.append((nwg, (cur[live, 0].max() - t0) / 100.0, cur[live, 13].min(), cur[live, 13].max()))
            print("   workgroups %d, rows per wave %d..%d; gap from the previous launch's last end to the first entry: %.2f us (median); entries spread over %.2f us"
                  % (rows_[0][0], rows_[0][2], rows_[0][3], np.median(gaps), np.median([r[1] for r in rows_])))
            for k in (7, 8, 1, 9, 10, 11, 2, 3, 4, 5, 6):
                if agg[k]:
                    m = np.median(np.array(agg[k]), axis=0)
                    print("   %-14s min %6.2f  median %6.2f  max %6.2f us" % (NAMES[k], m[0], m[1], m[2]))
            print("   launch (first entry -> last end) %.2f us; per launch in the chain %.2f us" % (np.median(totals), np.median(totals) + np.median(gaps)))


def main_rp(a, L):
    """Same chain, the matrix-core GEMV: stamps of rp_body (thread 0 of every workgroup = wave 0)."""
    HIP = C.CDLL("libamdhip64.so")
    rng = np.random.default_rng(0)
    stream = L.ntk_stream(0)
    launched = 0   # ntk_gemv_rp_fused counts its launches (one counter for all formats)
    for dname in a.dtypes.split(","):
        gt = GT[dname]
        dt = G.GGML_TO_DT[gt]
        for sname in a.shapes.split(","):
            kind, rows, in_f = SHAPES[sname]
            rlist = list(rows) if isinstance(rows, tuple) else ([rows, rows] if kind == "silu" else [rows])
            rb = G.row_bytes(gt, in_f)
            per_launch = rb * sum(rlist)
            nslots = max(2, min(48, (a.pool_mb << 20) // per_launch))
            slots = []
            for _ in range(nslots):   # distinct repacked copies (random bytes: timing only)
                mats = []
                for r in rlist:
                    raw = DB.from_numpy(rng.integers(0, 60, r * rb, dtype=np.uint8))
                    mats.append(ops.rp_pack(raw, r, in_f, dt))
                slots.append(mats)
            x = DB.from_numpy(rng.standard_normal(in_f).astype(np.float32))
            nw = DB.from_numpy(np.ones(in_f, np.float32))
            ys = [DB.zeros(max(r, 1) * 4) for r in rlist]

            def launch(slot):
                m = slots[slot]
                if kind == "resid":
                    ops.gemv_rp_fused([(m[0], ys[0], rlist[0], dt)], x, in_f, resid=ys[0])
                elif kind == "qkv":
                    ops.gemv_rp_fused([(m[i], ys[i], r, dt) for i, r in enumerate(rlist)], x, in_f, norm_w=nw, eps=1e-5)
                else:
                    ops.gemv_rp_fused([(m[0], ys[0], rlist[0], dt), (m[1], ys[1], rlist[0], dt)], x, in_f, norm_w=nw, eps=1e-5, silu_pair=True)

            n = 24
            graph, gexec = C.c_void_p(), C.c_void_p()
            assert HIP.hipStreamBeginCapture(C.c_void_p(stream), 1) == 0
            first = launched
            for i in range(n):
                launch(i % nslots)
                launched += 1
            assert HIP.hipStreamEndCapture(C.c_void_p(stream), C.byref(graph)) == 0
            assert HIP.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
            for _ in range(3):
                HIP.hipGraphLaunch(gexec, C.c_void_p(stream))
            ops.synchronize()
            buf = (C.c_ulonglong * (RP_SLOTS * RP_WG * RP_EV))()
            assert L.ntk_debug_rp_trace(buf, RP_SLOTS * RP_WG * RP_EV) == 0
            t = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(RP_SLOTS, RP_WG, RP_EV)
            HIP.hipGraphExecDestroy(gexec); HIP.hipGraphDestroy(graph)
            print("== %s.rp %s: %.2f MB per launch, chain of %d (10 ns clock; us relative to the launch's first workgroup entry)" % (dname, sname, per_launch / 1e6, n))
            gaps, totals, geo = [], [], None
            agg = {k: [] for k in range(1, 8)}
            for j in range(8, n - 2):   # middle of the chain
                cur, prev = t[(first + j) % RP_SLOTS], t[(first + j - 1) % RP_SLOTS]
                live = cur[:, 0] > 0
                t0 = cur[live, 0].min()
                gaps.append((t0 - prev[prev[:, 7] > 0, 7].max()) / 100.0)
                totals.append((cur[live, 7].max() - t0) / 100.0)
                for k in range(1, 8):
                    v = cur[live, k]
                    v = v[v > 0]
                    if len(v): agg[k].append(((v.min() - t0) / 100.0, (np.median(v) - t0) / 100.0, (v.max() - t0) / 100.0))
                geo = (int(live.sum()), (cur[live, 0].max() - t0) / 100.0, int(cur[live, 8].min()), int(cur[live, 8].max()), int(cur[live, 9].max()))
            print("   workgroups %d x %d waves, items per workgroup %d..%d; gap from the previous launch's last store to the first entry: %.2f us (median); entries spread over %.2f us"
                  % (geo[0], geo[4], geo[2], geo[3], np.median(gaps), geo[1]))
            for k in range(1, 8):
                if agg[k]:
                    m = np.median(np.array(agg[k]), axis=0)
                    print("   %-16s min %6.2f  median %6.2f  max %6.2f us" % (RP_NAMES[k], m[0], m[1], m[2]))
            print("   launch (first entry -> last store) %.2f us; per launch in the chain %.2f us" % (np.median(totals), np.median(totals) + np.median(gaps)))
            del slots


if __name__ == "__main__":
    main()
