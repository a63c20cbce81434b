#!/usr/bin/env python3
"""Decode attention (ntk_attention_decode_fused: RoPE + KV store + GQA attention, one launch) by context length, at the
Llama-3.1 8B / 70B head geometries.  KV bytes per launch = 2 * (pos + 1) * n_kv_heads * head_dim * 2 (each KV head read
once in the algorithmic count).  hipGraph-timed, 64 launches over rotating layer caches (--layers, default 40: 671 MB of cache at 4096
positions, past the 256 MB Infinity Cache like the 32 / 80 layers of a model; rounds 1-3 rotated over 8 = 134 MB, which the
Infinity Cache held).
usage: python tools/attn_bench.py [--json out.json] [--layers N] [--long]   (--long: only the split regime, 600 .. 4095 positions)"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntransformer_amd import _lib, ops  # noqa: E402
from ntransformer_amd.ops import DeviceBuffer as DB  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--layers", type=int, default=40)
    ap.add_argument("--long", action="store_true")
    ap.add_argument("--cases", default=None, help="pos:nsplit,... instead of the built-in table")
    ap.add_argument("--models", default="8b,70b")
    ap.add_argument("--merged", action="store_true", help="time every split case in both forms: with the combine launch and as one launch (ntk_attention_decode_split_merged)")
    ap.add_argument("--max-seq", type=int, default=4096, help="context the caches are allocated for (round 5: 8192 ... 131072; use fewer --layers)")
    a = ap.parse_args()
    ops.init(0)
    L = _lib.lib()
    HIP = C.CDLL("libamdhip64.so")
    rng = np.random.default_rng(0)
    ev0, ev1 = L.ntk_event_create(), L.ntk_event_create()
    stream = L.ntk_stream(0)
    res = []
    for name, nh, nkv, hd in (("8b", 32, 8, 128), ("70b", 64, 8, 128)):
        if name not in a.models.split(","):
            continue
        max_seq, nl = a.max_seq, a.layers
        per = nkv * hd
        if max_seq <= 4096:
            kc = [DB.from_numpy(rng.standard_normal(max_seq * per).astype(np.float16)) for _ in range(nl)]
            vc = [DB.from_numpy(rng.standard_normal(max_seq * per).astype(np.float16)) for _ in range(nl)]
        else:   # long contexts: one host array, nl device copies (distinct addresses are what defeats the Infinity Cache, not distinct values)
            hk = rng.standard_normal(max_seq * per, dtype=np.float32).astype(np.float16)
            kc = [DB.from_numpy(hk) for _ in range(nl)]
            vc = [DB.from_numpy(hk[::-1].copy()) for _ in range(nl)]
        q = DB.from_numpy(rng.standard_normal(nh * hd).astype(np.float32))
        k = DB.from_numpy(rng.standard_normal(per).astype(np.float32))
        v = DB.from_numpy(rng.standard_normal(per).astype(np.float32))
        out = DB.zeros(nh * hd * 4)
        scratch = DB.zeros(int(L.ntk_attention_split_scratch_bytes(nh, hd, 256)))
        cases = ((16, 1), (128, 1), (128, 2), (255, 1), (255, 2), (255, 4), (320, 1), (320, 2), (320, 4), (320, 8), (320, 16), (320, 32), (512, 1), (512, 2), (512, 4), (512, 8), (512, 32), (1024, 1), (1024, 4), (1024, 8), (1024, 16), (1024, 32), (2048, 8), (2048, 16), (2048, 32), (4095, 1), (4095, 4), (4095, 8), (4095, 16), (4095, 32), (4095, 64))
        if a.long:
            cases = tuple((p_, n_) for p_ in (600, 1023, 2047, 4095) for n_ in (1, 4, 8, 16, 32) if not (n_ == 1 and p_ > 1023))
        if a.cases:
            cases = tuple(tuple(int(t) for t in c.split(":")) for c in a.cases.split(","))
        if a.merged:
            cases = tuple((p_, n_, m_) for p_, n_ in cases for m_ in ((0, 1) if 1 < n_ <= 64 else (0,)))
        else:
            cases = tuple((p_, n_, 0) for p_, n_ in cases)
        for pos, nsplit, merged in cases:
            split_fn = L.ntk_attention_decode_split_merged if merged else L.ntk_attention_decode_split
            dpos = DB.from_numpy(np.array([pos], np.int32))
            n = 64 if max_seq <= 4096 else 16
            def launch(i):
                if nsplit == 1:
                    ops.attention_decode_fused(out, q, k, v, kc[i % nl], vc[i % nl], dpos, nh, nkv, hd, max_seq, 1.0 / np.sqrt(hd), 500000.0)
                else:
                    _lib.check(split_fn(out.ptr, q.ptr, k.ptr, v.ptr, kc[i % nl].ptr, vc[i % nl].ptr, dpos.ptr, None, nh, nkv, hd,
                                                            max_seq, 1.0 / np.sqrt(hd), 500000.0, 1.0, nsplit, scratch.ptr, None), "split")
            launch(0); ops.synchronize()
            graph, gexec = C.c_void_p(), C.c_void_p()
            assert HIP.hipStreamBeginCapture(C.c_void_p(stream), 1) == 0
            for i in range(n): launch(i)
            assert HIP.hipStreamEndCapture(C.c_void_p(stream), C.byref(graph)) == 0
            assert HIP.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
            HIP.hipGraphLaunch(gexec, C.c_void_p(stream)); ops.synchronize()
            L.ntk_event_record(ev0, None); HIP.hipGraphLaunch(gexec, C.c_void_p(stream)); L.ntk_event_record(ev1, None)
            L.ntk_event_synchronize(ev1)
            ms = C.c_float(); L.ntk_event_elapsed_ms(ev0, ev1, C.byref(ms))
            HIP.hipGraphExecDestroy(gexec); HIP.hipGraphDestroy(graph)
            us = ms.value * 1e3 / n
            kvb = 2 * (pos + 1) * per * 2
            res.append({"model": name, "pos": pos, "nsplit": nsplit, "merged": merged, "us": round(us, 2), "kv_MB": round(kvb / 1e6, 3), "GBs": round(kvb / us / 1e3, 1)})
            print("%-4s pos %5d nsplit %2d: %8.2f us per layer (%s), KV %7.3f MB -> %7.1f GB/s"
                  % (name, pos, nsplit, us, "single pass" if nsplit == 1 else ("split, one launch" if merged else "split + combine launch"), kvb / 1e6, kvb / us / 1e3), flush=True)
    if a.json: json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
