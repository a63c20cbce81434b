"""The oracle against every known-answer vector the reference's own tests hold for this path
(reference tests/test_gemm.cpp, tests/test_tensor.cpp) and against bit-level fp16 facts.
The same vectors are run through the HIP library in tests/test_hip_kernels.py."""
import numpy as np
import pytest

from ntransformer_amd import gguf as G
from oracle import oracle as O

from kat_vectors import KATS, q4_0_block, q6_k_block


@pytest.mark.parametrize("name", sorted(KATS))
def test_reference_gemv_kats(name):
    k = KATS[name]
    y = O.gemv(k["W"], k["x"], k["out"], k["in"], k["dtype"])
    assert np.allclose(y, k["expect"], atol=k["tol"], rtol=0), (name, y)


def test_silu_mul_kat():
    # reference tests/test_gemm.cpp:173-192
    g = np.array([0.0, 1.0, -1.0, 2.0], np.float32)
    u = np.ones(4, np.float32)
    exp = np.array([0.0, 0.731, -0.269, 1.762], np.float32)
    assert np.allclose(O.silu_mul(g, u), exp, atol=0.01)


def test_rmsnorm_kat():
    # reference tests/test_gemm.cpp:212-235: x=[1,2,3,4], w=1, eps=1e-5 -> x / sqrt(7.5)
    x = np.array([1, 2, 3, 4], np.float32)
    y = O.rmsnorm(x, np.ones(4, np.float32), 1e-5)
    assert np.allclose(y, x / np.sqrt(7.5 + 1e-5), atol=1e-6)


def test_block_sizes_match_reference_types_h():
    # reference src/core/types.h:37-88, tests/test_tensor.cpp:125-135 (the :128 assert there is stale: 36 vs 34)
    assert G.BLOCK[G.GGML_Q4_0] == (32, 18)
    assert G.BLOCK[G.GGML_Q8_0] == (32, 34)
    assert G.BLOCK[G.GGML_Q4_K] == (256, 144)
    assert G.BLOCK[G.GGML_Q5_K] == (256, 176)
    assert G.BLOCK[G.GGML_Q6_K] == (256, 210)
    assert G.row_bytes(G.GGML_Q4_0, 1024) == 576


def test_fp16_roundtrip_all_bit_patterns():
    bits = np.arange(65536, dtype=np.uint16)
    ours = np.array([O.h2f(int(b)) for b in bits[::7]], np.float32)
    ref = bits[::7].view(np.float16).astype(np.float32)
    ok = (ours == ref) | (np.isnan(ours) & np.isnan(ref))
    assert ok.all()


def test_f2h_round_to_nearest_even():
    rng = np.random.default_rng(3)
    vals = np.concatenate([
        rng.standard_normal(4000).astype(np.float32) * np.float32(10.0) ** rng.integers(-9, 6, 4000).astype(np.float32),
        np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e6, -1e6, 6.1e-5, 5.96e-8, 2.98e-8, 2.9802322e-8, 1e-9,
                  np.inf, -np.inf], np.float32),
        # exact ties in normal and subnormal range
        np.array([1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, 2.0 ** -24 * 1.5, 2.0 ** -24 * 2.5], np.float32)])
    ours = np.array([O.f2h(float(v)) for v in vals], np.uint16)
    with np.errstate(over="ignore"):
        ref = vals.astype(np.float16).view(np.uint16)
    assert (ours == ref).all(), np.flatnonzero(ours != ref)[:5]
