#!/usr/bin/env python3
"""Randomised parity sweep of the FP16 prompt GEMM (ntk_gemm_quant_ws / _multi) against the oracle's per-token GEMV: random formats,
token counts (ragged chunks, odd chunk counts), row counts (multiples of 16, ragged row tiles), column counts (whole units), residual,
several matrices per launch.  usage: python tools/gemm_fuzz.py [--cases 60] [--seed 1]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntransformer_amd import gguf as G, ops
from ntransformer_amd.ops import DeviceBuffer as DB
from oracle import oracle as O

ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=60); ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
ops.init(0)
r = np.random.Generator(np.random.Philox(key=[a.seed, 77]))
FMT = {"Q8_0": G.GGML_Q8_0, "Q4_0": G.GGML_Q4_0, "Q4_K": G.GGML_Q4_K, "Q5_K": G.GGML_Q5_K, "Q6_K": G.GGML_Q6_K}
worst, t0 = 0.0, time.time()
for case in range(a.cases):
    name = list(FMT)[int(r.integers(0, 5))]
    gt = FMT[name]; dt = G.GGML_TO_DT[gt]
    unit = 128 if name == "Q8_0" else 256
    in_f = unit * int(r.integers(1, 4096 // unit + 1))
    T = int(r.choice([1, 17, 63, 64, 65, 127, 129, 200, 257, 300, 449]))
    nseg = int(r.choice([1, 1, 2, 3]))
    outs = [16 * int(r.integers(1, 40)) for _ in range(nseg)]
    if r.random() < 0.25: outs[0] = 16 * int(r.integers(128, 200))     # >= 2048 rows: two row tiles per wave
    X = (r.standard_normal((T, in_f)) * np.exp(r.uniform(-4, 4, (T, 1)))).astype(np.float32)
    Ws = [np.frombuffer(G.synth_tensor(r, gt, o, in_f), np.uint8) for o in outs]
    Xd = DB.from_numpy(X); Wd = [DB.from_numpy(w) for w in Ws]
    Yd = [DB.from_numpy(np.full((T, o), np.nan, np.float32)) for o in outs]
    if nseg == 1:
        R = r.standard_normal((T, outs[0])).astype(np.float32) if r.random() < 0.5 else None
        if R is not None: Yd[0] = DB.from_numpy(R)
        st = ops.gemm_quant_ws(Yd[0], Wd[0], Xd, T, outs[0], in_f, dt, resid=Yd[0] if R is not None else None)
    else:
        R = None
        st = ops.gemm_quant_ws_multi([(Wd[i], Yd[i], outs[i], dt) for i in range(nseg)], Xd, T, in_f)
    assert st == 0, (case, name, T, outs, in_f, st)
    for i, o in enumerate(outs):
        got = Yd[i].numpy(np.float32).reshape(T, o)
        assert np.isfinite(got).all(), (case, name, T, outs, in_f)
        for t in sorted(set([0, T // 2, T - 1, int(r.integers(0, T))])):
            ref = O.gemv(Ws[i], X[t], o, in_f, dt)
            if R is not None: ref = ref + R[t]
            tol = 4e-6 * np.sqrt(in_f) * max(1.0, float(np.abs(ref).max()))
            err = float(np.abs(got[t] - ref).max())
            worst = max(worst, err / tol)
            assert err <= tol, (case, name, T, outs, in_f, i, t, err, tol)
print("gemm fuzz ok: %d cases, worst error / tolerance %.3f, %.1f s" % (a.cases, worst, time.time() - t0))
