"""GPU parity tests of the engine-owned repack and the matrix-core decode GEMV that reads it (csrc/gemv_rp.hip), through the C ABI:
the repacked weights ARE the GGUF weights (bit for bit), the kernel's lane maps and its integer image of x are what it assumes,
and every launch form the engine uses matches the oracle's restatement of the reference kernels (gemm.cu:158-470, rmsnorm.cu:16-70,
elementwise.cu:23-32, gemm.cu:713-725) at the GEMV tolerance of tests/test_hip_kernels.py.

Run on the MI355X box:  python -m pytest tests/test_gemv_rp.py -m gpu -x -q
"""
import ctypes as C

import numpy as np
import pytest

from ntransformer_amd import _lib
from ntransformer_amd import gguf as G
from ntransformer_amd import ops
from ntransformer_amd.ops import DeviceBuffer as DB
from oracle import int_activation as IA
from oracle import oracle as O

pytestmark = pytest.mark.gpu

KQ = {"Q4_K": G.GGML_Q4_K, "Q5_K": G.GGML_Q5_K, "Q6_K": G.GGML_Q6_K}


@pytest.fixture(scope="module", autouse=True)
def _device():
    ops.init(0)
    yield
    ops.synchronize()


def rng(seed):
    return np.random.Generator(np.random.Philox(key=[20260926, seed]))


def tol_for(y, in_f):
    return 4e-6 * np.sqrt(in_f) * max(1.0, float(np.abs(y).max()))


def packed(Wraw, out_f, in_f, gt):
    """the repacked form of a raw GGUF matrix, on the device"""
    dt = G.GGML_TO_DT[gt]
    return ops.rp_pack(DB.from_numpy(np.frombuffer(Wraw, np.uint8)), out_f, in_f, dt)


# ------------------------------------------------------------------------------- what the kernel assumes about the hardware
def test_mfma_i8_lane_maps():
    r = rng(1)
    A = r.integers(-128, 128, (16, 64), dtype=np.int8)
    B = r.integers(-128, 128, (64, 16), dtype=np.int8)      # asymmetric: a swapped row / column map cannot pass
    Dd, Ad, Bd = DB.zeros(16 * 16 * 4), DB.from_numpy(A), DB.from_numpy(B)
    _lib.check(_lib.lib().ntk_debug_mfma_i8_probe(Dd.ptr, Ad.ptr, Bd.ptr, None), "probe")
    D = Dd.numpy(np.int32).reshape(16, 16)
    assert np.array_equal(D, A.astype(np.int32) @ B.astype(np.int32))


def expected_image(x, nsub):
    """numpy restatement of rp_convert_quad (oracle/int_activation.py; its own CPU tests: tests/test_int_activation_cpu.py)"""
    return IA.digit_image(x, nsub)


@pytest.mark.parametrize("nsub", [8, 16])
@pytest.mark.parametrize("in_f,nwaves", [(256, 4), (1024, 16), (4096, 8), (14336, 14), (28672, 16)])
def test_prologue_image_is_the_digit_decomposition(nsub, in_f, nwaves):
    r = rng(in_f + nsub)
    x = r.standard_normal(in_f).astype(np.float32)
    x[5] *= 300.0          # an outlier channel
    x[in_f - 3] = 0.0
    if in_f >= 1024:
        x[512:768] = 0.0   # an all-zero super-block
        x[256:512] *= 1e-30
    out, xd = DB.zeros(4 * in_f + 68 * (in_f // 256)), DB.from_numpy(x)
    _lib.check(_lib.lib().ntk_debug_rp_prologue(out.ptr, xd.ptr, None, C.c_float(0.0), in_f, nsub, nwaves, None), "prologue")
    got = out.numpy(np.uint8)
    want = expected_image(x, nsub)
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:8]
    # ... and the digits reconstruct x to half a unit of 2^(e-22) <= 2^-22 of the super-block's largest magnitude
    d = got[:3 * in_f].view(np.int8).reshape(3, in_f).astype(np.float64)
    inv = got[4 * in_f + 64 * (in_f // 256):].view(np.float32).astype(np.float64)
    rec = (d[0] + 256.0 * d[1] + 65536.0 * d[2]) * np.repeat(inv, 256)
    blockmax = np.repeat(np.abs(x.reshape(-1, 256)).max(1), 256).astype(np.float64)
    assert (np.abs(rec - x) <= blockmax * 2.0 ** -22 + 1e-300).all()


def test_prologue_image_with_rmsnorm():
    in_f = 4096
    r = rng(77)
    x = r.standard_normal(in_f).astype(np.float32)
    w = (1.0 + 0.1 * r.standard_normal(in_f)).astype(np.float32)
    out, xd, wd = DB.zeros(4 * in_f + 68 * (in_f // 256)), DB.from_numpy(x), DB.from_numpy(w)
    _lib.check(_lib.lib().ntk_debug_rp_prologue(out.ptr, xd.ptr, wd.ptr, C.c_float(1e-5), in_f, 8, 16, None), "prologue")
    got = out.numpy(np.uint8)
    xn = O.rmsnorm(x, w, 1e-5).reshape(-1).astype(np.float64)
    d = got[:3 * in_f].view(np.int8).reshape(3, in_f).astype(np.float64)
    inv = got[4 * in_f + 64 * (in_f // 256):].view(np.float32).astype(np.float64)
    rec = (d[0] + 256.0 * d[1] + 65536.0 * d[2]) * np.repeat(inv, 256)
    assert np.abs(rec - xn).max() <= 4e-6 * np.abs(xn).max()


# ------------------------------------------------------------------------------- the repacked weights are the GGUF weights
@pytest.mark.parametrize("qname", sorted(KQ))
@pytest.mark.parametrize("out_f,in_f", [(1, 256), (16, 512), (37, 1024), (130, 4096), (48, 14336)])
def test_repacked_weights_dequantise_to_the_gguf_weights_bit_for_bit(qname, out_f, in_f):
    gt = KQ[qname]
    r = rng(out_f * 3 + in_f + gt)
    W = G.synth_tensor(r, gt, out_f, in_f)
    dt = G.GGML_TO_DT[gt]
    assert ops.rp_bytes(dt, out_f, in_f) == ((out_f + 15) // 16) * (in_f // 256) * {G.GGML_Q4_K: 2368, G.GGML_Q5_K: 2880, G.GGML_Q6_K: 3360}[gt]
    rp = packed(W, out_f, in_f, gt)
    out = DB.from_numpy(np.full(out_f * in_f, np.nan, np.float32))
    _lib.check(_lib.lib().ntk_rp_dequant(out.ptr, rp.ptr, out_f, in_f, dt, None), "rp_dequant")
    got = out.numpy(np.float32)
    want = G.dequantize(np.frombuffer(W, np.uint8), gt, out_f * in_f)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("qname", sorted(KQ))
@pytest.mark.parametrize("out_f,in_f", [(1, 256), (16, 512), (37, 1024), (130, 4096), (48, 14336)])
def test_unpack_restores_the_gguf_bytes(qname, out_f, in_f):
    """Round 5: one resident copy of a K-quant matrix -- the engine frees the uploaded GGUF bytes after the repack and unpacks a tensor into a
    scratch in front of the launches that read raw blocks (prompt GEMM, 1:1 ntk_gemv).  ntk_rp_unpack(ntk_rp_pack(W)) must be W, byte for byte
    (same integers, same 6-bit / int8 scales, same FP16 d / dmin; block layouts of reference src/core/types.h:112-137), ragged tiles included."""
    gt = KQ[qname]
    r = rng(out_f * 5 + in_f + gt)
    W = G.synth_tensor(r, gt, out_f, in_f)
    dt = G.GGML_TO_DT[gt]
    rp = packed(W, out_f, in_f, gt)
    raw = np.frombuffer(W, np.uint8)
    back = ops.rp_unpack(rp, out_f, in_f, dt, raw.size).numpy(np.uint8)[:raw.size]
    assert np.array_equal(back, raw), int((back != raw).sum())


def test_repack_rejects_what_it_does_not_take():
    L = _lib.lib()
    assert ops.rp_bytes(G.GGML_TO_DT[G.GGML_Q8_0], 16, 256) == 0
    assert ops.rp_bytes(G.GGML_TO_DT[G.GGML_Q4_K], 16, 128) == 0
    d = DB.zeros(4096)
    assert L.ntk_rp_pack(d.ptr, d.ptr, 16, 256, G.GGML_TO_DT[G.GGML_Q8_0], None) == -1
    assert L.ntk_rp_pack(d.ptr, d.ptr, 16, 100, G.GGML_TO_DT[G.GGML_Q4_K], None) == -2
    assert L.ntk_rp_pack(None, d.ptr, 16, 256, G.GGML_TO_DT[G.GGML_Q4_K], None) == -5
    y = DB.zeros(64)
    x = DB.zeros(1024)
    assert L.ntk_gemv_rp(y.ptr, d.ptr, x.ptr, 16, 256, G.GGML_TO_DT[G.GGML_Q8_0], None) == -1
    assert L.ntk_gemv_rp(y.ptr, d.ptr, x.ptr, 16, 255, G.GGML_TO_DT[G.GGML_Q4_K], None) == -2


# ------------------------------------------------------------------------------- GEMV vs oracle
# (the last: in_features = 32768, the stated limit of ntk_gemv_rp_fused -- 128 super-blocks per row, a 140 KB activation image)
SHAPES = [(3, 256), (64, 512), (257, 1024), (129, 2048), (512, 4096), (1024, 4096), (96, 8192), (515, 14336), (130, 28672), (4096, 4096), (70, 32768)]


@pytest.mark.parametrize("qname", sorted(KQ))
@pytest.mark.parametrize("out_f,in_f", SHAPES)
def test_gemv_rp_matches_oracle(qname, out_f, in_f):
    gt = KQ[qname]
    r = rng(out_f * 7 + in_f + gt)
    W = np.frombuffer(G.synth_tensor(r, gt, out_f, in_f), np.uint8)
    x = r.standard_normal(in_f).astype(np.float32)
    dt = G.GGML_TO_DT[gt]
    ref = O.gemv(W, x, out_f, in_f, dt)
    rp = packed(W, out_f, in_f, gt)
    yd = DB.from_numpy(np.full(out_f, np.nan, np.float32))
    ops.gemv_rp_fused([(rp, yd, out_f, dt)], DB.from_numpy(x), in_f)
    ops.synchronize()
    y = yd.numpy(np.float32)
    assert np.isfinite(y).all()
    assert np.abs(y - ref).max() <= tol_for(ref, in_f), np.abs(y - ref).max()


@pytest.mark.parametrize("qname", sorted(KQ))
def test_gemv_rp_outlier_channels_and_scaled_inputs(qname):
    """activations spanning e^+-3 inside every super-block, a few channels 1000 x the rest, and whole-vector scales far from 1"""
    gt = KQ[qname]
    out_f, in_f = 256, 4096
    r = rng(900 + gt)
    W = np.frombuffer(G.synth_tensor(r, gt, out_f, in_f), np.uint8)
    dt = G.GGML_TO_DT[gt]
    rp = packed(W, out_f, in_f, gt)
    for scale in (1.0, 1e-20, 1e15):
        x = (r.standard_normal(in_f) * np.exp(3.0 * r.uniform(-1, 1, in_f))).astype(np.float32)
        x[r.integers(0, in_f, 6)] *= 1000.0
        x = (x * np.float32(scale)).astype(np.float32)
        ref = O.gemv(W, x, out_f, in_f, dt)
        yd = DB.from_numpy(np.full(out_f, np.nan, np.float32))
        ops.gemv_rp_fused([(rp, yd, out_f, dt)], DB.from_numpy(x), in_f)
        ops.synchronize()
        y = yd.numpy(np.float32)
        assert np.abs(y - ref).max() <= 4e-6 * np.sqrt(in_f) * float(np.abs(ref).max()), (scale, np.abs(y - ref).max(), np.abs(ref).max())


@pytest.mark.parametrize("qname", sorted(KQ))
def test_gemv_rp_fused_at_the_in_features_limit(qname):
    """in_features = 32768 (the limit include/ntk.h states for ntk_gemv_rp_fused; the raw path sees it through the reference KAT
    tests/test_gemm.cpp:339-368): RMSNorm prologue + residual epilogue + plain rows, against the oracle; 32769+ is refused."""
    gt = KQ[qname]
    dt = G.GGML_TO_DT[gt]
    out_f, in_f = 48, 32768
    r = rng(32768 + gt)
    W = np.frombuffer(G.synth_tensor(r, gt, out_f, in_f), np.uint8)
    x = r.standard_normal(in_f).astype(np.float32)
    nw = (1.0 + 0.1 * r.standard_normal(in_f)).astype(np.float32)
    res = r.standard_normal(out_f).astype(np.float32)
    rp = packed(W, out_f, in_f, gt)
    xn = O.rmsnorm(x, nw, 1e-5).reshape(-1)
    ref = O.gemv(W, xn, out_f, in_f, dt)
    yd = DB.from_numpy(np.full(out_f, np.nan, np.float32))
    ops.gemv_rp_fused([(rp, yd, out_f, dt)], DB.from_numpy(x), in_f, norm_w=DB.from_numpy(nw), eps=1e-5)
    y = yd.numpy(np.float32)
    assert np.isfinite(y).all() and np.abs(y - ref).max() <= tol_for(ref, in_f), np.abs(y - ref).max()
    ref2 = res + O.gemv(W, x, out_f, in_f, dt)
    yd = DB.from_numpy(res.copy())
    ops.gemv_rp_fused([(rp, yd, out_f, dt)], DB.from_numpy(x), in_f, resid=yd)
    y = yd.numpy(np.float32)
    assert np.isfinite(y).all() and np.abs(y - ref2).max() <= tol_for(ref2, in_f), np.abs(y - ref2).max()
    arr = (ops.GemvSeg * 1)()
    arr[0].W, arr[0].y, arr[0].rows, arr[0].dtype = rp.ptr, yd.ptr, out_f, int(dt)
    assert _lib.lib().ntk_gemv_rp_fused(arr, 1, DB.zeros(4 * 33024).ptr, 33024, None, 0.0, None, 0, None) == -2   # NTK_E_SHAPE


def test_gemv_rp_zero_and_tiny_inputs():
    gt = G.GGML_Q4_K
    out_f, in_f = 32, 1024
    r = rng(5)
    W = np.frombuffer(G.synth_tensor(r, gt, out_f, in_f), np.uint8)
    dt = G.GGML_TO_DT[gt]
    rp = packed(W, out_f, in_f, gt)
    for x in (np.zeros(in_f, np.float32), np.full(in_f, 1e-38, np.float32)):
        yd = DB.from_numpy(np.full(out_f, np.nan, np.float32))
        ops.gemv_rp_fused([(rp, yd, out_f, dt)], DB.from_numpy(x), in_f)
        ops.synchronize()
        y = yd.numpy(np.float32)
        ref = O.gemv(W, x, out_f, in_f, dt)
        assert np.isfinite(y).all() and np.abs(y - ref).max() <= 1e-6 * max(float(np.abs(ref).max()), 1e-30) + 1e-37


@pytest.mark.parametrize("qname", sorted(KQ))
@pytest.mark.parametrize("in_f,rows", [(4096, (4096, 1024, 1024)), (8192, (1024, 128, 128)), (1024, (20, 3, 50))])
def test_gemv_rp_fused_norm_qkv(qname, in_f, rows):
    """norm + Q | K | V as one launch: launch_rmsnorm + 3 x launch_gemv"""
    gt = KQ[qname]
    dt = G.GGML_TO_DT[gt]
    r = rng(in_f + sum(rows) + gt)
    Ws = [np.frombuffer(G.synth_tensor(r, gt, o, in_f), np.uint8) for o in rows]
    x = r.standard_normal(in_f).astype(np.float32)
    nw = (1.0 + 0.1 * r.standard_normal(in_f)).astype(np.float32)
    xn = O.rmsnorm(x, nw, 1e-5).reshape(-1)
    ys = [DB.from_numpy(np.full(o, np.nan, np.float32)) for o in rows]
    rps = [packed(W, o, in_f, gt) for W, o in zip(Ws, rows)]
    ops.gemv_rp_fused([(rp, y, o, dt) for rp, y, o in zip(rps, ys, rows)], DB.from_numpy(x), in_f, norm_w=DB.from_numpy(nw), eps=1e-5)
    ops.synchronize()
    for W, y, o in zip(Ws, ys, rows):
        ref = O.gemv(W, xn, o, in_f, dt)
        assert np.abs(y.numpy() - ref).max() <= tol_for(ref, in_f)


@pytest.mark.parametrize("other", ["Q6_K", "Q5_K"])
@pytest.mark.parametrize("norm", [True, False])
@pytest.mark.parametrize("in_f,rows", [(4096, (4096, 1024, 1024)), (8192, (2048, 256, 256)), (512, (33, 7, 18))])
def test_gemv_rp_two_formats_one_launch(other, norm, in_f, rows):
    """llama.cpp's Q4_K_M: attn_q / attn_k in Q4_K, attn_v in Q6_K or Q5_K -- one launch, also with V first"""
    r = rng(in_f + sum(rows) + len(other))
    for order in ((G.GGML_Q4_K, G.GGML_Q4_K, KQ[other]), (KQ[other], G.GGML_Q4_K, G.GGML_Q4_K)):
        Ws = [np.frombuffer(G.synth_tensor(r, gt, o, in_f), np.uint8) for gt, o in zip(order, rows)]
        x = r.standard_normal(in_f).astype(np.float32)
        nw = (1.0 + 0.1 * r.standard_normal(in_f)).astype(np.float32)
        xin = O.rmsnorm(x, nw, 1e-5).reshape(-1) if norm else x
        ys = [DB.from_numpy(np.full(o, np.nan, np.float32)) for o in rows]
        rps = [packed(W, o, in_f, gt) for W, o, gt in zip(Ws, rows, order)]
        ops.gemv_rp_fused([(rp, y, o, G.GGML_TO_DT[gt]) for rp, y, o, gt in zip(rps, ys, rows, order)], DB.from_numpy(x), in_f,
                          norm_w=DB.from_numpy(nw) if norm else None, eps=1e-5)
        ops.synchronize()
        for W, y, o, gt in zip(Ws, ys, rows, order):
            ref = O.gemv(W, xin, o, in_f, G.GGML_TO_DT[gt])
            assert np.abs(y.numpy() - ref).max() <= tol_for(ref, in_f)


@pytest.mark.parametrize("qname", sorted(KQ))
@pytest.mark.parametrize("in_f,out_f", [(4096, 4096), (14336, 4096), (28672, 8192), (1024, 77)])
def test_gemv_rp_residual_in_place(qname, in_f, out_f):
    """hidden += W . x (launch_gemv + launch_add_inplace), resid == y"""
    gt = KQ[qname]
    dt = G.GGML_TO_DT[gt]
    r = rng(in_f + out_f + gt)
    rows = min(out_f, 600)   # (the oracle's time; the full launch geometries are covered by test_gemv_rp_full_size_launches_are_linear)
    W = np.frombuffer(G.synth_tensor(r, gt, rows, in_f), np.uint8)
    x = r.standard_normal(in_f).astype(np.float32)
    h = r.standard_normal(rows).astype(np.float32)
    ref = h + O.gemv(W, x, rows, in_f, dt)
    hd = DB.from_numpy(h)
    ops.gemv_rp_fused([(packed(W, rows, in_f, gt), hd, rows, dt)], DB.from_numpy(x), in_f, resid=hd)
    ops.synchronize()
    assert np.abs(hd.numpy() - ref).max() <= tol_for(ref, in_f)


@pytest.mark.parametrize("qname", sorted(KQ))
@pytest.mark.parametrize("in_f,inter", [(4096, 14336), (8192, 1024), (512, 40)])
def test_gemv_rp_norm_gate_up_silu(qname, in_f, inter):
    """norm + gate | up + SiLU(gate) * up as one launch (launch_rmsnorm, 2 x launch_gemv, launch_silu_mul)"""
    gt = KQ[qname]
    dt = G.GGML_TO_DT[gt]
    r = rng(in_f + inter + gt)
    rows = min(inter, 700)
    Wg = np.frombuffer(G.synth_tensor(r, gt, rows, in_f), np.uint8)
    Wu = np.frombuffer(G.synth_tensor(r, gt, rows, in_f), np.uint8)
    x = r.standard_normal(in_f).astype(np.float32)
    nw = (1.0 + 0.1 * r.standard_normal(in_f)).astype(np.float32)
    xn = O.rmsnorm(x, nw, 1e-5).reshape(-1)
    g, u = O.gemv(Wg, xn, rows, in_f, dt), O.gemv(Wu, xn, rows, in_f, dt)
    ref = O.silu_mul(g, u)
    yg, yu = DB.from_numpy(np.full(rows, np.nan, np.float32)), DB.zeros(rows * 4)
    ops.gemv_rp_fused([(packed(Wg, rows, in_f, gt), yg, rows, dt), (packed(Wu, rows, in_f, gt), yu, rows, dt)], DB.from_numpy(x), in_f,
                      norm_w=DB.from_numpy(nw), eps=1e-5, silu_pair=True)
    ops.synchronize()
    tol = tol_for(g, in_f) * (1.0 + float(np.abs(u).max())) + tol_for(u, in_f) * (1.0 + float(np.abs(g).max()))
    assert np.abs(yg.numpy() - ref).max() <= tol


def test_gemv_rp_full_size_launches_are_linear():
    """the real 8B / 70B launch geometries (every workgroup / wave split the planner produces at those sizes): y(a x1 + x2) against
    a y(x1) + y(x2) from the same kernel -- a size-independent property -- and a sampled set of rows against the oracle"""
    r = rng(4242)
    for gt, out_f, in_f in ((G.GGML_Q4_K, 28672, 4096), (G.GGML_Q6_K, 128256, 4096), (G.GGML_Q4_K, 8192, 28672), (G.GGML_Q6_K, 4096, 14336)):
        dt = G.GGML_TO_DT[gt]
        rb = G.row_bytes(gt, in_f)
        base = np.frombuffer(G.synth_tensor(r, gt, 256, in_f), np.uint8)
        reps = (out_f + 255) // 256
        W = np.tile(base, reps)[: out_f * rb]
        rp = packed(W, out_f, in_f, gt)
        x1 = r.standard_normal(in_f).astype(np.float32)
        x2 = r.standard_normal(in_f).astype(np.float32)
        outs = []
        for x in (x1, x2, (np.float32(2.0) * x1 + x2).astype(np.float32)):
            yd = DB.from_numpy(np.full(out_f, np.nan, np.float32))
            ops.gemv_rp_fused([(rp, yd, out_f, dt)], DB.from_numpy(x), in_f)
            ops.synchronize()
            outs.append(yd.numpy(np.float32))
        ref = O.gemv(base, x1, 256, in_f, dt)
        t = tol_for(ref, in_f)
        assert np.isfinite(outs[0]).all()
        assert np.abs(outs[0][:256] - ref).max() <= t
        # the tiled rows repeat (up to the summation order: a tile may be split between waves at another super-block)
        assert np.abs(outs[0].reshape(-1)[: (out_f // 256) * 256].reshape(-1, 256) - outs[0][:256]).max() <= t
        assert np.abs(outs[2] - (2.0 * outs[0] + outs[1])).max() <= 4 * t
