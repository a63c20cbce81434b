#!/usr/bin/env python3
"""Phase timeline of the batched prompt GEMM (needs a trace build: make -C ntransformer_amd/csrc clean all HIPFLAGS+=-DNTK_GEMM_TRACE).
Prints, for wave 0 of workgroup 0, the clock ticks between the phase boundaries of each tile round."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntransformer_amd import _lib, gguf as G, ops
from ntransformer_amd.ops import DeviceBuffer as DB

ops.init(0)
L = _lib.lib()
rng = np.random.default_rng(0)
for dname, gt, out_f, in_f in (("Q8_0", G.GGML_Q8_0, 14336, 4096), ("Q4_K", G.GGML_Q4_K, 28672, 8192)):
    T = 16
    rb = G.row_bytes(gt, in_f)
    W = DB.from_numpy(rng.integers(0, 60, out_f * rb, dtype=np.uint8)); X = DB.from_numpy(rng.standard_normal((T, in_f)).astype(np.float32)); Y = DB.zeros(T * out_f * 4)
    for _ in range(3): ops.gemm_quant(Y, W, X, T, out_f, in_f, G.GGML_TO_DT[gt])
    ops.synchronize()
    buf = (C.c_ulonglong * 256)()
    assert L.ntk_debug_gemm_trace(buf) == 0
    t = np.array(buf[:], dtype=np.uint64).astype(np.int64)
    names = ["staged", "B1", "mfma", "partials", "B2", "reduced"]
    print("%s %dx%d: entry -> chunk prologue %d cycles" % (dname, out_f, in_f, t[1] - t[0]))
    i, rnd = 2, 0
    while i + 6 <= 256 and t[i + 5] > 0 and rnd < 8:
        d = np.diff(t[i - 1:i + 6])
        print("  round %d: " % rnd + "  ".join("%s %6d" % (n, x) for n, x in zip(names, d)) + "   | total %d" % (t[i + 5] - t[i - 1]))
        i += 6; rnd += 1
