"""Pin the oracle (and the numpy block decoder in ntransformer_amd/gguf.py) to outputs of the reference:
  * tests/golden/dequant_*.npz   -- reference tools/decompose_gguf.py dequantisers (imported in the build
                                    container by tools/make_golden.py)
  * tests/golden/*_logits.npz    -- the reference's unmodified host code + CPU kernels (oracle/_ref/ref_logits)
"""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN
from ntransformer_amd import gguf as G
from oracle import oracle as O

DEQ = {"q8_0": G.GGML_Q8_0, "q4_k": G.GGML_Q4_K, "q5_k": G.GGML_Q5_K, "q6_k": G.GGML_Q6_K}


@pytest.mark.parametrize("name", sorted(DEQ))
def test_block_decoding_matches_reference_python_dequant(name):
    z = np.load(os.path.join(GOLDEN, "dequant_%s.npz" % name))
    raw, out_f, in_f, ref = z["raw"], int(z["out_f"]), int(z["in_f"]), z["ref"]
    gt = DEQ[name]
    ours = G.dequantize(raw, gt, out_f * in_f).reshape(out_f, in_f)
    assert np.allclose(ours, ref, rtol=2e-7, atol=0), np.abs(ours - ref).max()
    # oracle GEMV == dequantised matrix . x (fp64) to fp32 accumulation accuracy
    rng = np.random.default_rng(1)
    x = rng.standard_normal(in_f).astype(np.float32)
    y = O.gemv(raw, x, out_f, in_f, G.GGML_TO_DT[gt])
    y64 = ref.astype(np.float64) @ x.astype(np.float64)
    assert np.allclose(y, y64, rtol=0, atol=3e-6 * np.abs(ref).max() * np.abs(x).sum()), np.abs(y - y64).max()
    # one-hot x picks single weights out exactly: catches any wrong scale index / nibble order
    for j in (0, 1, 15, 16, 31, 32, 63, 64, 95, 96, 127, 128, 255, 256, 300, 511):
        e = np.zeros(in_f, np.float32)
        e[j] = 1.0
        col = O.gemv(raw, e, out_f, in_f, G.GGML_TO_DT[gt])
        assert np.allclose(col, ref[:, j], rtol=3e-7, atol=1e-9), (j, col, ref[:, j])


@pytest.mark.parametrize("name", ["q8_0", "q4_k", "q6_k"])
def test_embed_row_matches_reference_python_dequant(name):
    z = np.load(os.path.join(GOLDEN, "dequant_%s.npz" % name))
    raw, out_f, in_f, ref = z["raw"], int(z["out_f"]), int(z["in_f"]), z["ref"]
    for r in range(out_f):
        row = O.embed_row(raw, r, in_f, G.GGML_TO_DT[DEQ[name]])
        assert np.allclose(row, ref[r], rtol=2e-7, atol=0)


def test_embed_row_q5_k_is_zero_like_reference():
    # reference src/model/transformer.cpp:595-598 has no Q5_K branch: prints an error and zero-fills
    z = np.load(os.path.join(GOLDEN, "dequant_q5_k.npz"))
    row = O.embed_row(z["raw"], 0, int(z["in_f"]), G.DT_Q5_K)
    assert not row.any()


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for c in iter(lambda: f.read(1 << 20), b""):
            h.update(c)
    return h.hexdigest()


CASES = [("tiny_q8_0", G.TINY, "Q8_0"), ("tiny_q4_k_m", G.TINY, "Q4_K_M"), ("tiny_mixed", G.TINY, "MIXED"),
         ("small_q8_0", G.SMALL, "Q8_0"), ("small_q4_k_m", G.SMALL, "Q4_K_M"), ("small_q6_k", G.SMALL, "Q6_K")]


def golden_model(name, shape, mix, tmp_path):
    """Path of the GGUF a golden logits file was generated from (committed, or regenerated from its seed)."""
    z = np.load(os.path.join(GOLDEN, name + "_logits.npz"))
    path = os.path.join(GOLDEN, name + ".gguf")
    if not os.path.exists(path):
        path = str(tmp_path / (name + ".gguf"))
        G.make_synthetic_llama(path, shape, mix, seed=20260925)
    assert _sha(path) == str(z["gguf_sha256"]), "synthetic generator drifted from the golden run"
    return path, z


@pytest.mark.parametrize("name,shape,mix", CASES)
def test_oracle_model_reproduces_reference_host_logits(name, shape, mix, tmp_path):
    path, z = golden_model(name, shape, mix, tmp_path)
    m = O.OracleModel(path, max_context=int(z["ctx"]))
    prompt, forced = list(z["prompt"]), list(z["forced"])
    outs = [m.forward(prompt, 0)]
    pos = len(prompt)
    for t in forced:
        outs.append(m.forward([int(t)], pos))
        pos += 1
    nxt = m.argmax(outs[-1])
    for _ in range(int(z["n_greedy"])):
        outs.append(m.forward([nxt], pos))
        pos += 1
        nxt = m.argmax(outs[-1])
    outs = np.stack(outs)
    assert outs.shape == z["logits"].shape
    # same kernels, same call sequence -> identical bits; anything else means the orchestration drifted
    assert np.array_equal(outs, z["logits"]), np.abs(outs - z["logits"]).max()
    assert np.array_equal(outs.argmax(1), z["argmax"])


@pytest.mark.skipif(not os.path.exists("/root/reference/src/main.cpp"), reason="reference tree not present (GPU box)")
def test_reference_own_kernel_tests_pass_against_oracle():
    """reference tests/test_gemm.cpp, unmodified, linked with the CPU restatement."""
    import subprocess
    assert O.build_ref()
    exe = os.path.join(os.path.dirname(O.__file__), "_ref", "test_gemm_cpu")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0
    assert "FAIL" not in r.stderr and r.stderr.count("PASS") == 6, r.stderr
