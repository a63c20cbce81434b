"""TEST INFRASTRUCTURE (not product code; only tests/ may import this).

numpy restatement of the ARITHMETIC of the matrix-core decode GEMV (ntransformer_amd/csrc/gemv_rp.hip, the `int24-block` activation
form of bench.py), so that what the kernel is meant to compute can be checked on the CPU against the restatement of the reference
kernels (oracle.gemv = reference src/cuda/gemm.cu:158-255 Q4_K, :265-365 Q5_K, :387-470 Q6_K):

  * per 256-column super-block of x one exponent e (frexp of the largest magnitude) and the integers X = rint(x 2^(22 - e)),
    |X| <= 2^22, i.e. |x - X 2^(e - 22)| <= 2^(e - 23) <= 2^-22 of the super-block's largest |x| -- `digit_image` writes them as the kernel's LDS
    image: three signed base-256 digit planes, the digits of the sub-block sums of X, and 2^(e - 22) per super-block;
  * the weights' own integers (4 / 5 / 6-bit q, 6-bit or int8 sub-block scales, 6-bit minima), taken from the GGUF block exactly as
    the reference kernels read them;
  * y[row] = sum over super-blocks 2^(e - 22) ( d sum_j sc_j (sum_k q_jk X_k) - dmin sum_j m_j (sum_k X_k) )      (Q4_K, Q5_K)
           = sum over super-blocks 2^(e - 22)   d sum_j sc_j (sum_k (q_jk - 32) X_k)                               (Q6_K)
    with every inner sum an exact integer (the kernel: v_mfma_i32_16x16x64_i8 per digit plane, v_mad_i32_i24 for the scales).
`gemv_int24` evaluates that in int64 / float64: the value the kernel's F32 epilogue rounds.
"""
import numpy as np

from ntransformer_amd import gguf as G


def block_integers(x):
    """x [in] float32 -> (X int64 [in], inv float64 [in / 256]): x ~ X * inv[super-block]"""
    x = np.asarray(x, np.float32)
    nsb = x.size // 256
    X = np.zeros(x.size, np.int64)
    inv = np.zeros(nsb, np.float64)
    for sb in range(nsb):
        v = x[256 * sb:256 * sb + 256]
        am = np.float32(np.abs(v).max())
        e = int(np.frexp(am)[1]) if am > 0 else 0
        e = max(e, -100)
        X[256 * sb:256 * sb + 256] = np.rint(v.astype(np.float64) * 2.0 ** (22 - e)).astype(np.int64)
        inv[sb] = 2.0 ** (e - 22)
    return X, inv


def digit_image(x, nsub):
    """The kernel's LDS image of x (rp_convert_quad): [plane 0 | plane 1 | plane 2 | zero plane] of `in` bytes each, then per
    super-block 64 bytes = 4 digits x 16 slots of the sub-block sums (nsub = 8: Q4_K / Q5_K, 16: Q6_K), then 2^(e - 22) as float32."""
    x = np.asarray(x, np.float32)
    in_f = x.size
    nsb = in_f // 256
    X, inv = block_integers(x)
    assert np.abs(X).max() <= 2 ** 22
    img = np.zeros(4 * in_f + 68 * nsb, np.uint8)
    Y = (X + 0x808080) ^ 0x808080            # signed base-256 digits: X = d0 + 256 d1 + 65536 d2, each digit an int8
    for p in range(3):
        img[p * in_f:(p + 1) * in_f] = ((Y >> (8 * p)) & 0xFF).astype(np.uint8)
    S = X.reshape(nsb, nsub, 256 // nsub).sum(2)
    YS = (S + 0x80808080) ^ 0x80808080
    for sb in range(nsb):
        for dg in range(4):
            base = 4 * in_f + 64 * sb + 16 * dg
            img[base:base + nsub] = ((YS[sb] >> (8 * dg)) & 0xFF).astype(np.uint8)
    img[4 * in_f + 64 * nsb:] = inv.astype(np.float32).view(np.uint8)
    return img


def kquant_fields(raw, ggml_type, out_f, in_f):
    """Integer fields of a K-quant matrix: q [rows, nb, nsub, w] (Q6_K: already minus 32), sc [rows, nb, nsub], m or None, d, dmin."""
    be, bb = G.BLOCK[ggml_type]
    nb = in_f // 256
    b = np.frombuffer(raw, np.uint8)[:out_f * nb * bb].reshape(out_f, nb, bb)
    if ggml_type in (G.GGML_Q4_K, G.GGML_Q5_K):
        d = b[..., 0:2].copy().view("<f2").astype(np.float64)[..., 0]
        dm = b[..., 2:4].copy().view("<f2").astype(np.float64)[..., 0]
        s, m = G._kq_scales(b[..., 4:16])
        if ggml_type == G.GGML_Q4_K:
            qs = b[..., 16:144].reshape(out_f, nb, 4, 32).astype(np.int64)
            lo, hi = qs & 0xF, qs >> 4
        else:
            qh = b[..., 16:48].astype(np.int64)[:, :, None, :]
            ql = b[..., 48:176].reshape(out_f, nb, 4, 32).astype(np.int64)
            c = np.arange(4)[None, None, :, None]
            lo = (ql & 0xF) + (((qh >> (2 * c)) & 1) << 4)
            hi = (ql >> 4) + (((qh >> (2 * c + 1)) & 1) << 4)
        q = np.stack([lo, hi], 3).reshape(out_f, nb, 8, 32)          # sub-block order 0..7
        return q, s.astype(np.int64), m.astype(np.int64), d, dm
    if ggml_type == G.GGML_Q6_K:
        ql = b[..., 0:128].reshape(out_f, nb, 2, 64).astype(np.int64)
        qh = b[..., 128:192].reshape(out_f, nb, 2, 32).astype(np.int64)
        sc = b[..., 192:208].copy().view(np.int8).reshape(out_f, nb, 16).astype(np.int64)
        d = b[..., 208:210].copy().view("<f2").astype(np.float64)[..., 0]
        q1 = ((ql[..., :32] & 0xF) | (((qh >> 0) & 3) << 4)) - 32
        q2 = ((ql[..., 32:] & 0xF) | (((qh >> 2) & 3) << 4)) - 32
        q3 = ((ql[..., :32] >> 4) | (((qh >> 4) & 3) << 4)) - 32
        q4 = ((ql[..., 32:] >> 4) | (((qh >> 6) & 3) << 4)) - 32
        # element order of a half (128 columns): q1 (32), q2 (32), q3 (32), q4 (32); sub-block = 16 columns, scale index 8 half + 2 g + (l >= 16)
        q = np.stack([q1, q2, q3, q4], 3).reshape(out_f, nb, 16, 16)
        return q, sc, None, d, None
    raise ValueError(ggml_type)


def gemv_int24(raw, ggml_type, out_f, in_f, x):
    """y [out_f] float64: the integer-activation GEMV, exact integer dot products, scales / minima / sums in float64."""
    X, inv = block_integers(x)
    q, sc, m, d, dm = kquant_fields(raw, ggml_type, out_f, in_f)
    nb, nsub = q.shape[1], q.shape[2]
    Xb = X.reshape(nb, nsub, 256 // nsub)
    dots = np.einsum("rbjk,bjk->rbj", q, Xb)                          # exact: |.| < 2^5 * 2^22 * 32 < 2^63
    main = (sc * dots).sum(2)                                          # [rows, nb] integers
    y = d * main.astype(np.float64)
    if m is not None:
        y = y - dm * (m * Xb.sum(2)[None]).sum(2).astype(np.float64)
    return (y * inv[None]).sum(1)
