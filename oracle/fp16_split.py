"""TEST INFRASTRUCTURE (not product code; only tests/ may import this).

numpy restatement of the ACTIVATION ARITHMETIC of the prompt GEMM (ntransformer_amd/csrc/gemm_f16.hip), so that what its operand
pre-pass is meant to produce can be checked on the CPU against the restatement of the reference kernels (oracle.gemv applied token by token =
the reference's prefill loop, attention.cpp:144-162, ffn.cpp:96-133):

  * per TOKEN one power of two s = 2^(es - 127), es = clamp(268 - em, 1, 253), em = the exponent field of the token's largest |x| (row_scale_kernel /
    scale_exp_of_max; round 5: the largest |x| may come from the launch that produced x -- rmsnorm_rowmax_kernel, silu_mul_rowmax_kernel -- as the maximum
    of the BIT PATTERNS of |x|, a NaN pattern counting as the largest binade): the largest |x| s lies in [2^14, 2^15);
  * every scaled activation as TWO FP16 pieces h1 = rn16(x s), h2 = rn16(x s - h1) (split_x_kernel): x s - h1 is exact in F32, so
    |x s - h1 - h2| <= 2^-23 |x s| wherever h2 is a normal FP16 number, and <= 2^-25 (one half of FP16's smallest subnormal step) below that;
  * the weights' integers are exact in FP16, the products exact in the F32 accumulator of v_mfma_f32_16x16x32_f16; block scales multiply F32 block sums
    and 1 / s the finished sum -- `gemm_two_piece` evaluates that chain in float64 on the two pieces: the value the kernel's F32 sums round.
"""
import numpy as np


def scale_exponent(max_bits):
    """uint32 bit pattern of the token's largest |x| -> exponent field of s (gemm_f16.hip: scale_exp_of_max)."""
    em = (np.asarray(max_bits, np.uint32) >> np.uint32(23)) & np.uint32(0xFF)
    return np.clip(268 - em.astype(np.int64), 1, 253)


def row_max_bits(X):
    """[T, in] float32 -> uint32 [T]: what the producing launches leave -- the maximum of the bit patterns of |x| (non-negative floats order like their
    bits; anything above +inf's pattern, i.e. a NaN, is clamped to +inf's)."""
    b = np.ascontiguousarray(X, np.float32).view(np.uint32) & np.uint32(0x7FFFFFFF)
    return np.minimum(b.max(axis=1), np.uint32(0x7F800000))


def token_scales(X, from_bits=True):
    """-> (s [T] float32, 1 / s [T] float32).  from_bits False: row_scale_kernel's own pass (fmaxf over |x|, which drops NaNs)."""
    X = np.ascontiguousarray(X, np.float32)
    if from_bits:
        mb = row_max_bits(X)
    else:
        m = np.fmax.reduce(np.abs(X), axis=1, initial=np.float32(0))      # fmaxf: a NaN operand is ignored
        mb = m.astype(np.float32).view(np.uint32)
    es = scale_exponent(mb)
    s = (es.astype(np.uint32) << np.uint32(23)).view(np.float32)
    inv = ((254 - es).astype(np.uint32) << np.uint32(23)).view(np.float32)
    return s, inv


def split(X, s):
    """-> (h1, h2) float16 [T, in]: the two operand planes (split_x_kernel; the conversions round to nearest even like v_cvt_f16_f32)."""
    xs = (np.ascontiguousarray(X, np.float32) * s[:, None]).astype(np.float32)          # exact: s is a power of two (no overflow: |x s| < 2^15)
    h1 = xs.astype(np.float16)
    rem = (xs - h1.astype(np.float32)).astype(np.float32)                                # exact in F32 (13 significant bits)
    h2 = rem.astype(np.float16)
    return h1, h2


def reconstruct(h1, h2, inv):
    """what the two planes stand for: (h1 + h2) / s, in float64"""
    return (h1.astype(np.float64) + h2.astype(np.float64)) * inv.astype(np.float64)[:, None]


def gemm_two_piece(dequant_rows, X):
    """Y[t, r] = sum_k W[r, k] ((h1 + h2)[t, k] / s_t) in float64, W = the dequantised weights (float64 [out, in]): the exact value of the kernel's chain
    up to its F32 accumulation."""
    s, inv = token_scales(X)
    h1, h2 = split(X, s)
    return reconstruct(h1, h2, inv) @ np.asarray(dequant_rows, np.float64).T
