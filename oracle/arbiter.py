"""oracle/arbiter.py -- TEST INFRASTRUCTURE ONLY.

ctypes view of oracle/libarbiter_f64.so (arbiter_f64.cpp: the decode path with every accumulation in float64) and
`ArbiterModel`, the same host orchestration as oracle.OracleModel (reference src/model/transformer.cpp:604-669,
attention.cpp:120-211, ffn.cpp:85-134) over it.  The arbiter is the third party of the full-depth parity tests
(tests/test_parity_depth.py): the F32 restatement (liboracle) and the HIP engine are both compared with it.

Two ways to run it:
  * free:   it rounds its OWN K / V to half when it stores them (reference attention.cu:338) -- the exact function;
  * forced: after computing its own K / V rows it COMPARES their half roundings with the rows an implementation X put into its
            cache, records every mismatch (size in half ulps, and how far the arbiter's pre-rounding value sat from the rounding
            boundary), then CONTINUES WITH X's ROWS.  The rounding decisions -- the only discontinuity on the path -- are then X's,
            and what is left between X's logits and the arbiter's is X's accumulated F32 error alone.

Only tests/ may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

from . import oracle as O

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libarbiter_f64.so")
        src = os.path.join(_HERE, "arbiter_f64.cpp")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE, "libarbiter_f64.so"])
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        _LIB = C.CDLL(so)
        _LIB.arb_set_threads(C.c_int(O.host_cpu_budget()[0]))
        _LIB.arb_h2d.restype = C.c_double
        _LIB.arb_h2d.argtypes = [C.c_uint16]
        _LIB.arb_d2h.restype = C.c_uint16
        _LIB.arb_d2h.argtypes = [C.c_double]
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def dequant(W: np.ndarray, rows: int, in_f: int, dtype: int) -> np.ndarray:
    out = np.empty((rows, in_f), np.float64)
    lib().arb_dequant(_p(out), _p(np.ascontiguousarray(W)), C.c_int(rows), C.c_int(in_f), C.c_int(dtype))
    return out


def gemm(W: np.ndarray, X: np.ndarray, out_f: int, in_f: int, dtype: int) -> np.ndarray:
    X = _f64(X).reshape(-1, in_f)
    Y = np.empty((X.shape[0], out_f), np.float64)
    rc = lib().arb_gemm(_p(Y), _p(np.ascontiguousarray(W)), _p(X), C.c_int(X.shape[0]), C.c_int(out_f), C.c_int(in_f), C.c_int(dtype))
    assert rc == 0, "arbiter: unsupported dtype %d" % dtype
    return Y


def rmsnorm(x: np.ndarray, w: np.ndarray, eps: float) -> np.ndarray:
    x = _f64(x)
    x2 = x.reshape(-1, x.shape[-1])
    o = np.empty_like(x2)
    lib().arb_rmsnorm(_p(o), _p(x2), _p(np.ascontiguousarray(w, dtype=np.float32)), C.c_int(x2.shape[0]), C.c_int(x2.shape[1]), C.c_float(eps))
    return o.reshape(x.shape)


def rope(q: np.ndarray, k: np.ndarray, positions: Sequence[int], nh: int, nkv: int, hd: int, theta: float, fscale: float = 1.0):
    q, k = _f64(q).copy(), _f64(k).copy()
    pos = np.ascontiguousarray(positions, dtype=np.int32)
    lib().arb_rope(_p(q), _p(k), _p(pos), C.c_int(len(pos)), C.c_int(nh), C.c_int(nkv), C.c_int(hd), C.c_float(theta), C.c_float(fscale))
    return q, k


def kv_store(kc: np.ndarray, vc: np.ndarray, k: np.ndarray, v: np.ndarray, T: int, per: int, start: int, max_seq: int):
    """stores the half roundings; returns (mid_k, mid_v): distance of every value to its rounding boundary"""
    k, v = _f64(k).reshape(T, per), _f64(v).reshape(T, per)
    mk, mv = np.empty_like(k), np.empty_like(v)
    lib().arb_kv_store(_p(kc), _p(vc), _p(k), _p(v), C.c_int(T), C.c_int(per), C.c_int(start), C.c_int(max_seq), _p(mk), _p(mv))
    return mk, mv


def attention(Q: np.ndarray, kc: np.ndarray, vc: np.ndarray, T: int, start: int, nh: int, nkv: int, hd: int, scale: float) -> np.ndarray:
    Q = _f64(Q).reshape(T, nh * hd)
    o = np.empty_like(Q)
    lib().arb_attention(_p(o), _p(Q), _p(kc), _p(vc), C.c_int(T), C.c_int(start), C.c_int(nh), C.c_int(nkv), C.c_int(hd), C.c_float(scale))
    return o


def silu_mul(g: np.ndarray, u: np.ndarray) -> np.ndarray:
    g, u = _f64(g), _f64(u)
    o = np.empty_like(g)
    lib().arb_silu_mul(_p(o), _p(g), _p(u), C.c_long(g.size))
    return o


def half_spacing(x: np.ndarray) -> np.ndarray:
    """spacing of the IEEE halves around |x| (float64 array)"""
    ax = np.maximum(np.abs(x), 2.0 ** -14)
    return 2.0 ** (np.floor(np.log2(ax)) - 10)


def kv_excess(rows_half: np.ndarray, exact: np.ndarray) -> float:
    """max over elements of (|stored half - exact| - half an ulp) / RMS of the row.  A correct implementation's stored half is the
    rounding of ITS pre-rounding value, so this is bounded by its F32 error before the rounding (in units of the row's RMS)."""
    stored = np.ascontiguousarray(rows_half, dtype=np.uint16).view(np.float16).astype(np.float64).reshape(exact.shape)
    rms = np.sqrt((exact ** 2).mean(axis=1, keepdims=True))
    return float(((np.abs(stored - exact) - 0.5 * half_spacing(exact)) / rms).max())


def half_ulp_distance(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """distance between two arrays of IEEE halves in units of representable values (sign-magnitude -> ordered integers)"""
    def ordered(h):
        h = h.astype(np.int32)
        return np.where(h & 0x8000, -(h & 0x7FFF), h & 0x7FFF)
    return np.abs(ordered(a) - ordered(b))


class ArbiterModel:
    """The resident forward in float64.  Shares the GGUF view, the configuration and the embedding rows (F32 data, reference
    transformer.cpp:419-599) with an oracle.OracleModel."""

    def __init__(self, om: "O.OracleModel"):
        self.om = om
        L, S, per = om.n_layers, om.max_seq, om.nkv * om.hd
        self.k_cache = np.zeros((L, S * per), np.uint16)
        self.v_cache = np.zeros((L, S * per), np.uint16)
        self.kv_report = []   # forced runs: one record per (layer, call)

    def _gemm(self, name: str, X: np.ndarray, out_f: int, in_f: int) -> np.ndarray:
        return gemm(self.om.f.raw(name), X, out_f, in_f, self.om.f.dtype(name))

    def layer(self, i: int, hidden: np.ndarray, start_pos: int, forced=None) -> np.ndarray:
        """one layer on hidden [T, H] (float64).  forced = (k_rows, v_rows) uint16 [T, per] of an implementation X for THESE positions."""
        om = self.om
        T, H = hidden.shape
        q_dim, kv_dim = om.nh * om.hd, om.nkv * om.hd
        p = "blk.%d." % i
        positions = [start_pos + t for t in range(T)]
        x = rmsnorm(hidden, om.f.f32(p + "attn_norm.weight"), om.eps)
        q = self._gemm(p + "attn_q.weight", x, q_dim, H)
        k = self._gemm(p + "attn_k.weight", x, kv_dim, H)
        v = self._gemm(p + "attn_v.weight", x, kv_dim, H)
        q, k = rope(q.reshape(-1), k.reshape(-1), positions, om.nh, om.nkv, om.hd, om.theta)
        mk, mv = kv_store(self.k_cache[i], self.v_cache[i], k, v.reshape(-1), T, kv_dim, start_pos, om.max_seq)
        if forced is not None:
            lo, hi = start_pos * kv_dim, (start_pos + T) * kv_dim
            for name, cache, rows, mid, val in (("k", self.k_cache[i], forced[0], mk, k.reshape(T, kv_dim)),
                                                ("v", self.v_cache[i], forced[1], mv, v.reshape(T, kv_dim))):
                mine = cache[lo:hi].reshape(T, kv_dim)
                rows = np.ascontiguousarray(rows, dtype=np.uint16).reshape(T, kv_dim)
                diff = mine != rows
                n = int(diff.sum())
                rec = {"layer": i, "start_pos": start_pos, "which": name, "elements": int(diff.size), "mismatches": n,
                       "max_half_ulps": 0, "max_boundary_distance_over_row_rms": 0.0, "max_excess_over_row_rms": kv_excess(rows, val)}
                if n:
                    rms = np.sqrt((val ** 2).mean(axis=1, keepdims=True))            # of the pre-rounding row
                    rec["max_half_ulps"] = int(half_ulp_distance(mine, rows)[diff].max())
                    rec["max_boundary_distance_over_row_rms"] = float((mid / rms)[diff].max())
                self.kv_report.append(rec)
                cache[lo:hi] = rows.reshape(-1)                                       # continue with X's decisions
        att = attention(q, self.k_cache[i], self.v_cache[i], T, start_pos, om.nh, om.nkv, om.hd, float(om.scale))
        hidden = hidden + self._gemm(p + "attn_output.weight", att, H, q_dim)
        x = rmsnorm(hidden, om.f.f32(p + "ffn_norm.weight"), om.eps)
        act = silu_mul(self._gemm(p + "ffn_gate.weight", x, om.inter, H), self._gemm(p + "ffn_up.weight", x, om.inter, H))
        return hidden + self._gemm(p + "ffn_down.weight", act, H, om.inter)

    def forward(self, tokens: Sequence[int], start_pos: int, forced_cache=None) -> np.ndarray:
        """logits (float64) of the last token.  forced_cache = (K, V) uint16 [L, max_seq * per]: X's whole cache."""
        om = self.om
        T = len(tokens)
        per = om.nkv * om.hd
        hidden = om.embed(tokens).astype(np.float64)
        for i in range(om.n_layers):
            forced = None
            if forced_cache is not None:
                lo, hi = start_pos * per, (start_pos + T) * per
                forced = (forced_cache[0][i][lo:hi], forced_cache[1][i][lo:hi])
            hidden = self.layer(i, hidden, start_pos, forced)
        last = rmsnorm(hidden[T - 1], om.f.f32("output_norm.weight"), om.eps)
        return self._gemm(om.out_name, last[None, :], om.vocab, om.hidden)[0]
