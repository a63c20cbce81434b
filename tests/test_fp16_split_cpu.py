"""CPU tests of the ACTIVATION ARITHMETIC of the prompt GEMM (csrc/gemm_f16.hip) through its numpy restatement (oracle/fp16_split.py): the per-token
power of two, the two FP16 pieces and their error bound, the row maxima taken from bit patterns (what ntk_rmsnorm_rowmax / ntk_silu_mul_rowmax leave for
ntk_gemm_quant_ws_rm), and the GEMM evaluated on the pieces against the restatement of the reference's per-token float GEMV (oracle.gemv:
gemm.cu:95-470) at the tolerance of the GPU tests.  (The kernels are held to the same contracts on the GPU: tests/test_hip_kernels.py.)"""
import numpy as np
import pytest

from ntransformer_amd import gguf as G
from oracle import fp16_split as FS
from oracle import oracle as O


def rng(seed):
    return np.random.Generator(np.random.Philox(key=[20260930, seed]))


def tokens(r, T, in_f):
    X = r.standard_normal((T, in_f)).astype(np.float32)
    if T >= 4:
        X[1] *= np.float32(1e-20)            # a token of tiny values
        X[2] = 0.0                           # an all-zero token
        X[3, r.integers(0, in_f, 4)] *= 1000.0   # outlier channels (massive activations)
    return X


@pytest.mark.parametrize("T,in_f", [(1, 256), (7, 1024), (64, 4096)])
def test_token_scale_puts_the_largest_magnitude_in_fp16s_top_binade_but_one(T, in_f):
    X = tokens(rng(T + in_f), T, in_f)
    s, inv = FS.token_scales(X)
    assert np.array_equal(s * inv, np.ones(T, np.float32))                                    # both normal powers of two
    m = np.abs(X).max(axis=1)
    live = m >= np.float32(2.0 ** -100)
    assert ((m[live] * s[live] >= 2.0 ** 14) & (m[live] * s[live] < 2.0 ** 15)).all()
    s2, inv2 = FS.token_scales(X, from_bits=False)                                            # the pre-pass's own pass over X gives the same scales
    assert np.array_equal(s, s2) and np.array_equal(inv, inv2)


def test_row_maxima_from_bit_patterns():
    X = tokens(rng(5), 8, 512)
    assert np.array_equal(FS.row_max_bits(X).view(np.float32), np.abs(X).max(axis=1))
    X[4, 7] = -np.inf
    X[5, 9] = np.nan
    mb = FS.row_max_bits(X)
    assert mb[4] == 0x7F800000 and mb[5] == 0x7F800000                                        # inf, and a NaN counted as the largest binade
    es = FS.scale_exponent(mb)
    assert es[4] == 13 and es[5] == 13 and FS.scale_exponent(np.uint32(0)) == 253             # clamps: an all-zero token, the largest binade


@pytest.mark.parametrize("T,in_f", [(5, 256), (64, 4096)])
def test_two_pieces_reconstruct_x_to_one_f32_ulp(T, in_f):
    X = tokens(rng(T * 3 + in_f), T, in_f)
    s, inv = FS.token_scales(X)
    h1, h2 = FS.split(X, s)
    assert np.isfinite(h1.astype(np.float32)).all() and np.abs(h1.astype(np.float32)).max() < 2.0 ** 15 + 16
    xs = X.astype(np.float64) * s.astype(np.float64)[:, None]
    err = np.abs(xs - h1.astype(np.float64) - h2.astype(np.float64))
    normal_h2 = np.abs(xs - h1.astype(np.float64)) >= 2.0 ** -14                              # h2 a normal FP16 number: relative 2^-11 of a 2^-11 remainder
    assert (err[normal_h2] <= np.abs(xs[normal_h2]) * 2.0 ** -23).all()                       # the documented bound: one F32 ulp of the scaled activation
    assert (err[~normal_h2] <= 2.0 ** -25).all()                                              # subnormal h2: half of FP16's smallest step
    rec = FS.reconstruct(h1, h2, inv)
    big = np.abs(X) >= np.abs(X).max(axis=1, keepdims=True) * 2.0 ** -16                      # within 2^-16 of the token's largest: one F32 ulp (CPU arithmetic
                                                                                              # found gemm_f16.hip's header saying 2^-17: h2 is subnormal from 2^-16 down)
    assert (np.abs(rec - X)[big] <= np.abs(X[big]).astype(np.float64) * 2.0 ** -23).all()


@pytest.mark.parametrize("qname,gt", [("Q8_0", G.GGML_Q8_0), ("Q4_K", G.GGML_Q4_K), ("Q6_K", G.GGML_Q6_K)])
def test_gemm_on_the_pieces_matches_the_reference_gemv(qname, gt):
    T, out_f, in_f = 9, 48, 1024
    r = rng(gt)
    raw = np.frombuffer(G.synth_tensor(r, gt, out_f, in_f), np.uint8)
    X = tokens(r, T, in_f)
    dt = G.GGML_TO_DT[gt]
    ref = np.stack([O.gemv(raw, X[t], out_f, in_f, dt) for t in range(T)])
    # the dequantised matrix, column by column through the same restatement: W[:, k] = gemv(e_k)
    W = np.zeros((out_f, in_f), np.float64)
    eye = np.zeros(in_f, np.float32)
    for k in range(in_f):
        eye[k] = 1.0
        W[:, k] = O.gemv(raw, eye, out_f, in_f, dt)
        eye[k] = 0.0
    Y = FS.gemm_two_piece(W, X)
    scale = np.abs(ref).max(axis=1, keepdims=True) + 1e-30
    assert (np.abs(Y - ref) <= 2e-5 * scale + 1e-6).all(), float((np.abs(Y - ref) / scale).max())
