"""GGUF writer/reader round trip and the algorithmic-bytes table of SURVEY.md section 8(d)."""
import numpy as np

from ntransformer_amd import gguf as G


def test_roundtrip(tmp_path):
    p = str(tmp_path / "t.gguf")
    types = G.make_synthetic_llama(p, G.TINY, "MIXED", seed=5)
    f = G.read_gguf(p)
    assert f.version == 3
    assert f.meta["llama.embedding_length"] == 256
    assert len(f.meta["tokenizer.ggml.tokens"]) == 512
    assert f.data_offset % 32 == 0
    for name, gt in types.items():
        ti = f.tensors[name]
        assert ti.ggml_type == gt and ti.offset % 32 == 0
        assert f.raw(name).size == ti.nbytes
    # ggml dim order: in_features first
    assert f.tensors["blk.0.ffn_down.weight"].dims == (512, 256)
    assert f.tensors["blk.0.attn_k.weight"].dims == (256, 128)


def test_synthetic_weights_have_unit_scale():
    rng = np.random.default_rng(0)
    for gt in (G.GGML_Q8_0, G.GGML_Q4_0, G.GGML_Q4_K, G.GGML_Q5_K, G.GGML_Q6_K, G.GGML_F16, G.GGML_F32):
        w = G.dequantize(G.synth_tensor(rng, gt, 8, 1024), gt, 8 * 1024)
        assert np.isfinite(w).all()
        assert 0.5 < w.std() * np.sqrt(1024) < 2.0, (gt, w.std() * np.sqrt(1024))


def test_algorithmic_bytes_per_token_match_survey():
    # SURVEY.md section 8(d) / BASELINE.md section 2 (weights + norms term only)
    def weights_norms(shape, mix):
        t = G.tensor_types(shape, mix)
        hd = shape.hidden // shape.heads
        kv = 2 * shape.layers * shape.kv_heads * hd * 2
        emb = G.row_bytes(t["token_embd.weight"], shape.hidden)
        return G.algorithmic_bytes_per_token(shape, t, pos=0) - 2 * kv - emb
    assert weights_norms(G.LLAMA_8B, "Q8_0") == 7_974_764_544
    assert weights_norms(G.LLAMA_8B, "Q4_K_M") == 4_617_396_224
    assert weights_norms(G.LLAMA_70B, "Q4_K_M") == 41_921_527_808
    assert weights_norms(G.LLAMA_70B, "Q6_K") == 57_018_400_768
