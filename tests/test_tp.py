"""Tensor-parallel slices (csrc/tp.hip, engine/model.cpp): host-side checks that need no GPU.
The column slice of a GGUF matrix is a re-packing of whole quantisation blocks; with it W . x = sum over ranks of
W_r . x_r, which is what the exchange step adds up on the device."""
import ctypes as C

import numpy as np
import pytest

from ntransformer_amd import engine as E
from ntransformer_amd import gguf as G
from oracle import oracle as O

QUANT = {"Q8_0": G.GGML_Q8_0, "Q4_K": G.GGML_Q4_K, "Q6_K": G.GGML_Q6_K, "Q5_K": G.GGML_Q5_K, "Q4_0": G.GGML_Q4_0}


def slice_columns(W, dt, out_f, in_f, rank, world):
    L = E._bind()
    loc = np.zeros(len(W) // world, np.uint8)
    st = L.nt_tp_slice_columns(loc.ctypes.data_as(C.c_void_p), W.ctypes.data_as(C.c_void_p), int(dt), out_f, in_f, rank, world)
    return st, loc


@pytest.mark.parametrize("qname", sorted(QUANT))
@pytest.mark.parametrize("world", [2, 4])
def test_column_slices_partition_the_gemv(qname, world):
    gt = QUANT[qname]
    dt = G.GGML_TO_DT[gt]
    out_f, in_f = 48, 2048
    r = np.random.default_rng(world * 100 + gt)
    W = np.frombuffer(G.synth_tensor(r, gt, out_f, in_f), np.uint8).copy()
    x = r.standard_normal(in_f).astype(np.float32)
    full = O.gemv(W, x, out_f, in_f, dt)
    rb_full, rb_loc = G.row_bytes(gt, in_f), G.row_bytes(gt, in_f // world)
    assert rb_loc * world == rb_full
    acc = np.zeros(out_f, np.float64)
    for rank in range(world):
        st, loc = slice_columns(W, dt, out_f, in_f, rank, world)
        assert st == 0
        # byte for byte: row r of the slice = bytes [rank * rb_loc, (rank + 1) * rb_loc) of row r
        want = W.reshape(out_f, rb_full)[:, rank * rb_loc:(rank + 1) * rb_loc]
        assert np.array_equal(loc.reshape(out_f, rb_loc), want)
        xs = x[rank * (in_f // world):(rank + 1) * (in_f // world)]
        acc += O.gemv(loc, np.ascontiguousarray(xs), out_f, in_f // world, dt)
    assert np.abs(acc - full).max() <= 1e-5 * max(1.0, np.abs(full).max())


@pytest.mark.parametrize("qname", sorted(QUANT))
@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("in_f", [8192, 28672])
def test_column_slices_at_70b_shapes_are_byte_ranges_of_whole_blocks(qname, world, in_f):
    """Wo (8192 columns) and ffn_down (28672 columns) of the Llama-3.1-70B shape over 2 / 4 / 8 ranks, every GGUF format: the slice a
    rank uploads is, row by row, the byte range of its whole quantisation blocks -- re-packed, never re-quantised -- and the ranks'
    partial products add up to the full row product (checked on a few rows: the packing is per row)."""
    gt = QUANT[qname]
    dt = G.GGML_TO_DT[gt]
    out_f = 6
    r = np.random.default_rng(world * 1000 + gt + in_f)
    W = np.frombuffer(G.synth_tensor(r, gt, out_f, in_f), np.uint8).copy()
    rb_full = G.row_bytes(gt, in_f)
    blk_w, blk_b = (32, rb_full * 32 // in_f) if qname in ("Q8_0", "Q4_0") else (256, rb_full * 256 // in_f)
    assert (in_f // world) % blk_w == 0      # 70B shapes divide into whole blocks for every format up to 8 ranks
    rb_loc = (in_f // world) // blk_w * blk_b
    x = r.standard_normal(in_f).astype(np.float32)
    full = O.gemv(W, x, out_f, in_f, dt).astype(np.float64)
    acc = np.zeros(out_f, np.float64)
    for rank in range(world):
        st, loc = slice_columns(W, dt, out_f, in_f, rank, world)
        assert st == 0
        assert np.array_equal(loc.reshape(out_f, rb_loc), W.reshape(out_f, rb_full)[:, rank * rb_loc:(rank + 1) * rb_loc])
        xs = np.ascontiguousarray(x[rank * (in_f // world):(rank + 1) * (in_f // world)])
        acc += O.gemv(loc, xs, out_f, in_f // world, dt)
    assert np.abs(acc - full).max() <= 2e-5 * max(1.0, np.abs(full).max())


def test_column_slices_must_be_whole_blocks():
    L = E._bind()
    W = np.zeros(G.row_bytes(G.GGML_Q4_K, 256) * 16, np.uint8)
    dst = np.zeros(len(W), np.uint8)
    dt = G.GGML_TO_DT[G.GGML_Q4_K]
    assert L.nt_tp_slice_columns(dst.ctypes.data_as(C.c_void_p), W.ctypes.data_as(C.c_void_p), int(dt), 16, 256, 0, 2) == -2   # half a super-block
    assert L.nt_tp_slice_columns(dst.ctypes.data_as(C.c_void_p), W.ctypes.data_as(C.c_void_p), int(dt), 16, 256, 2, 2) == -2   # rank out of range
    assert L.nt_tp_slice_columns(None, W.ctypes.data_as(C.c_void_p), int(dt), 16, 256, 0, 1) == -5


def test_tp_configure_rejects_bad_worlds():
    eng = E.Engine()
    try:
        L = eng.L
        assert L.nt_engine_tp_configure(eng.h, 0, 0) == -2
        assert L.nt_engine_tp_configure(eng.h, 2, 2) == -2
        assert L.nt_engine_tp_configure(eng.h, 0, 9) == -2
        assert L.nt_engine_tp_configure(eng.h, 1, 2) == 0
        assert L.nt_engine_tp_export(eng.h, None, None) == -5      # nothing loaded yet
    finally:
        eng.close()
