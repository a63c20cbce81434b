"""Pins of the float64 arbiter (oracle/arbiter_f64.cpp, oracle/arbiter.py) -- the third party of tests/test_parity_depth.py.
CPU only.  The arbiter is not a restatement of the reference's summation order; it must (1) decode the GGUF blocks to exactly the
values the reference's own numpy dequantisers give (tests/golden/dequant_*.npz, made from reference tools/decompose_gguf.py),
(2) round to half exactly like __float2half (reference attention.cu:338) wherever a float would, and (3) agree with the F32
restatement -- which carries the reference's known-answer vectors -- to F32 rounding level on every operator and on the golden
models' logits once the half roundings are the restatement's."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from ntransformer_amd import gguf as G
from oracle import arbiter as A
from oracle import oracle as O
from test_oracle_golden import CASES, golden_model

DEQ = {"q8_0": G.GGML_Q8_0, "q4_k": G.GGML_Q4_K, "q5_k": G.GGML_Q5_K, "q6_k": G.GGML_Q6_K}


@pytest.mark.parametrize("name", sorted(DEQ))
def test_arbiter_block_decoding_matches_reference_python_dequant(name):
    z = np.load(os.path.join(GOLDEN, "dequant_%s.npz" % name))
    raw, out_f, in_f, ref = z["raw"], int(z["out_f"]), int(z["in_f"]), z["ref"]
    w = A.dequant(raw, out_f, in_f, G.GGML_TO_DT[DEQ[name]])
    # the reference dequantiser works in F32: the exact value differs from it by at most one F32 rounding
    assert np.allclose(w, ref.astype(np.float64), rtol=1.2e-7, atol=0), np.abs(w - ref).max()
    x = np.random.default_rng(1).standard_normal((3, in_f))
    y = A.gemm(raw, x, out_f, in_f, G.GGML_TO_DT[DEQ[name]])
    assert np.allclose(y, x @ w.T, rtol=1e-12, atol=1e-12)


def test_arbiter_half_rounding_is_float2half_on_floats():
    """double -> half in ONE rounding equals the restatement's F32 -> half (checked against the reference's KV-store semantics:
    subnormals, +-0, the overflow threshold 65520, ties to even) for every value a float can hold near the interesting places."""
    r = np.random.default_rng(0)
    vals = np.concatenate([r.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** r.integers(-9, 5, 20000).astype(np.float32),
                           np.array([0.0, -0.0, 65504.0, 65519.996, 65520.0, 70000.0, -65520.0, 5.9604645e-08, 2.9802322e-08,
                                     2.98023e-08, 6.1035156e-05, 6.0975552e-05, 1.0009766, 1.0004883, 1.0014648], np.float32)])
    for v in vals:
        assert A.lib().arb_d2h(float(v)) == O.f2h(float(v)), float(v)
    for h in list(range(0, 0x7C00, 37)) + [0x7BFF, 0x0001, 0x03FF, 0x0400]:
        assert A.lib().arb_h2d(h) == O.h2f(h)
        assert A.lib().arb_d2h(A.lib().arb_h2d(h)) == h
    # a double strictly between a float and the rounding boundary must not be rounded twice
    mid = (1.0 + 2.0 ** -11)                       # exactly between the halves 1.0 and 1.0009766
    assert A.lib().arb_d2h(mid) == 0x3C00          # tie -> even
    assert A.lib().arb_d2h(mid * (1 + 2.0 ** -40)) == 0x3C01
    assert A.lib().arb_d2h(mid * (1 - 2.0 ** -40)) == 0x3C00


def test_arbiter_operators_agree_with_the_restatement():
    r = np.random.default_rng(3)
    H, nh, nkv, hd, T, start, S = 512, 8, 2, 64, 5, 7, 32
    x = r.standard_normal((T, H)).astype(np.float32)
    w = (1 + 0.05 * r.standard_normal(H)).astype(np.float32)
    assert np.abs(A.rmsnorm(x, w, 1e-5) - O.rmsnorm(x, w, 1e-5)).max() <= 2e-6
    q = r.standard_normal(T * nh * hd).astype(np.float32)
    k = r.standard_normal(T * nkv * hd).astype(np.float32)
    pos = list(range(start, start + T))
    qa, ka = A.rope(q, k, pos, nh, nkv, hd, 500000.0)
    qo, ko = O.rope(q, k, pos, nh, nkv, hd, 500000.0)
    assert np.abs(qa - qo).max() <= 2e-6 and np.abs(ka - ko).max() <= 2e-6
    kc = r.standard_normal(S * nkv * hd).astype(np.float16).view(np.uint16)
    vc = r.standard_normal(S * nkv * hd).astype(np.float16).view(np.uint16)
    out_a = A.attention(qo, kc, vc, T, start, nh, nkv, hd, 0.125)
    out_o = O.attention_prefill(qo, kc, vc, T, start, nh, nkv, hd, S, 0.125).reshape(T, nh * hd)
    assert np.abs(out_a - out_o).max() <= 5e-6
    out_d = O.attention_decode(qo[:nh * hd], kc, vc, start + 1, nh, nkv, hd, S, 0.125)
    assert np.abs(out_a[0] - out_d).max() <= 5e-6
    g, u = r.standard_normal(1000).astype(np.float32), r.standard_normal(1000).astype(np.float32)
    assert np.abs(A.silu_mul(g, u) - O.silu_mul(g, u)).max() <= 1e-6
    kc2, vc2 = np.zeros(S * nkv * hd, np.uint16), np.zeros(S * nkv * hd, np.uint16)
    kc3, vc3 = kc2.copy(), vc2.copy()
    A.kv_store(kc2, vc2, ko, k, T, nkv * hd, start, S)
    O.copy_to_kv_cache(kc3, vc3, ko, k, T, nkv, hd, start, S)
    assert np.array_equal(kc2, kc3) and np.array_equal(vc2, vc3)      # F32 inputs: the two roundings coincide


@pytest.mark.parametrize("name,shape,mix", CASES)
def test_arbiter_logits_on_the_golden_models(name, shape, mix, tmp_path):
    """Forced to the restatement's half roundings the arbiter reproduces the golden logits (reference host code + restatement)
    to accumulated-F32-error level; running free it differs by the flips alone -- the decomposition the depth tests rely on."""
    path, z = golden_model(name, shape, mix, tmp_path)
    m = O.OracleModel(path, int(z["ctx"]))
    prompt = [int(t) for t in z["prompt"]]
    want = m.forward(prompt, 0)
    assert np.array_equal(want, z["logits"][0])
    forced = A.ArbiterModel(m)
    got = forced.forward(prompt, 0, (m.k_cache, m.v_cache))
    assert np.abs(got - want).max() <= 5e-5, np.abs(got - want).max()
    for rec in forced.kv_report:   # every stored half is the rounding of a value within F32 error of the exact one
        assert rec["max_excess_over_row_rms"] <= 1e-5, rec
    free = A.ArbiterModel(m).forward(prompt, 0)
    assert np.abs(free - want).max() <= 1e-3
