"""CPU tests of the ARITHMETIC of the matrix-core decode GEMV (csrc/gemv_rp.hip), through its numpy restatement
(oracle/int_activation.py): the integer image of x loses at most half a unit of 2^(e - 22), i.e. at most 2^-22 of a super-block's largest magnitude, per element, and the
GEMV evaluated on it agrees with the restatement of the reference's float kernels (oracle.gemv: gemm.cu:158-470) at the GEMV
tolerance of the GPU tests -- with outlier channels, all-zero and denormal-scale super-blocks.  (The kernel itself is held to the same
restatements on the GPU: tests/test_gemv_rp.py compares its LDS image with `digit_image` byte for byte and its results with oracle.gemv.)
"""
import numpy as np
import pytest

from ntransformer_amd import gguf as G
from oracle import int_activation as IA
from oracle import oracle as O

KQ = {"Q4_K": G.GGML_Q4_K, "Q5_K": G.GGML_Q5_K, "Q6_K": G.GGML_Q6_K}


def rng(seed):
    return np.random.Generator(np.random.Philox(key=[20260926, seed]))


def activations(r, in_f, outliers):
    x = r.standard_normal(in_f).astype(np.float32)
    if outliers:
        x[r.integers(0, in_f, 8)] *= 60.0       # eight outlier channels (what one exponent per 256 columns has to survive)
        if in_f >= 1024:
            x[512:768] = 0.0                     # an all-zero super-block
            x[256:512] *= np.float32(1e-30)      # a super-block of tiny values
    return x


@pytest.mark.parametrize("outliers", [False, True])
@pytest.mark.parametrize("in_f", [256, 1024, 4096])
def test_integer_image_reconstructs_x(in_f, outliers):
    x = activations(rng(in_f + outliers), in_f, outliers)
    X, inv = IA.block_integers(x)
    assert np.abs(X).max() <= 2 ** 22
    blockmax = np.repeat(np.abs(x.reshape(-1, 256)).max(1), 256).astype(np.float64)
    assert (np.abs(X * np.repeat(inv, 256) - x) <= blockmax * 2.0 ** -22 + 1e-300).all()
    for nsub in (8, 16):
        img = IA.digit_image(x, nsub)
        d = img[:3 * in_f].view(np.int8).reshape(3, in_f).astype(np.int64)
        assert np.array_equal(d[0] + 256 * d[1] + 65536 * d[2], X)            # the three digit planes ARE X
        assert not img[3 * in_f:4 * in_f].any()                                # the zero plane
        nsb = in_f // 256
        sums = X.reshape(nsb, nsub, 256 // nsub).sum(2)
        dig = img[4 * in_f:4 * in_f + 64 * nsb].view(np.int8).reshape(nsb, 4, 16).astype(np.int64)
        got = dig[:, 0] + 256 * dig[:, 1] + 65536 * dig[:, 2] + 16777216 * dig[:, 3]
        assert np.array_equal(got[:, :nsub], sums) and not got[:, nsub:].any()  # the digits of the sub-block sums
        assert np.array_equal(img[4 * in_f + 64 * nsb:].view(np.float32).astype(np.float64), inv)


@pytest.mark.parametrize("outliers", [False, True])
@pytest.mark.parametrize("fmt", list(KQ))
@pytest.mark.parametrize("out_f,in_f", [(16, 256), (48, 1024), (32, 4096)])
def test_integer_activation_gemv_matches_the_reference_restatement(fmt, out_f, in_f, outliers):
    gt = KQ[fmt]
    r = rng(out_f * 7 + in_f + gt + outliers)
    raw = G.synth_tensor(r, gt, out_f, in_f)
    x = activations(r, in_f, outliers)
    y = IA.gemv_int24(raw, gt, out_f, in_f, x)
    # (a) against the exact evaluation of the same weights on the same x (float64): what the integer image costs
    Wd = G.dequantize(raw, gt, out_f * in_f).astype(np.float64).reshape(out_f, in_f)
    exact = Wd @ x.astype(np.float64)
    blockmax = np.repeat(np.abs(x.reshape(-1, 256)).max(1), 256).astype(np.float64)
    bound = np.abs(Wd) @ (blockmax * 2.0 ** -22) + 1e-12 * np.abs(exact).max()
    assert (np.abs(y - exact) <= bound).all(), float(np.abs(y - exact).max())
    # (b) against the restatement of the reference's F32 kernels, at the tolerance the GPU tests hold the HIP kernels to
    ref = O.gemv(np.frombuffer(raw, np.uint8), x, out_f, in_f, G.GGML_TO_DT[gt]).astype(np.float64)
    tol = 4e-6 * np.sqrt(in_f) * max(1.0, float(np.abs(ref).max()))
    assert np.abs(y - ref).max() <= tol, (float(np.abs(y - ref).max()), tol)
