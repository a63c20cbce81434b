"""Checks of the EXPERIMENTS=1 library (ntransformer_amd/libntransformer_hip_exp.so), run by
tests/test_engine_gpu.py::test_experiments_library_matches_the_launch_path in a subprocess with NTK_LIB_PATH pointing at it.
Both structures lost to the launch path; these checks keep the negative results reproducible."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ntransformer_amd import engine as E   # noqa: E402
from test_oracle_golden import CASES, golden_model   # noqa: E402
from pathlib import Path   # noqa: E402


def persistent(name, shape, mix, tmp_path):
    """One decode token as ONE persistent launch (csrc/decode_persistent.hip: weights prefetched by LDS-DMA across operators,
    activations handed between workgroups through the in-launch grid barrier) against the 5-launches-per-layer path on the
    same KV cache: same operators and per-row arithmetic, so the logits agree far inside the tolerance (only the RMSNorm and
    attention reduction orders differ); eager and hipGraph replay; the bounded-wait error word must stay clear."""
    path, z = golden_model(name, shape, mix, tmp_path)
    prompt = [int(t) for t in z["prompt"]]
    fed = [int(t) for t in z["fed"][1:]][:6] + [5, 9, 300 % 256, 17]
    outs = {}
    for mode in ("launches", "persistent", "persistent_graph"):
        eng = E.Engine()
        eng.load(path, int(z["ctx"]))
        eng.set_option("persistent", mode != "launches")
        if mode != "launches" and "persistent" not in eng.decode_path():
            eng.close()
            return "skipped (model does not qualify: dense or mixed gate/up tensors)"
        lg = [eng.forward(prompt, 0)]
        pos = len(prompt)
        for t in fed:
            lg.append(eng.decode_fused(t, pos, mode == "persistent_graph"))
            pos += 1
        toks = eng.decode_greedy_steps(fed[-1], pos, 8)       # device argmax loop through the same kernel
        outs[mode] = (np.stack(lg), toks)
        eng.close()
    for mode in ("persistent", "persistent_graph"):
        err = np.abs(outs[mode][0] - outs["launches"][0]).max()
        assert np.isfinite(outs[mode][0]).all() and err <= 5e-4, (name, mode, err)   # summation order differs; the logits bar is 1e-3
    assert outs["persistent"][1] == outs["persistent_graph"][1]
    return "ok"


def attention_in_wo(name, shape, mix, tmp_path):
    """Short contexts: RoPE + KV store + attention run as extra workgroups IN FRONT of the Wo projection's grid
    (ntk_attention_gemv_fused: the GEMV workgroups request their first weight rows, then wait for the heads), one launch
    less per layer.  Same arithmetic as ntk_attention_decode_fused + ntk_gemv_fused; compared on the same KV cache, eager and
    hipGraph replay, across many positions (the sync words must return to zero after every launch)."""
    path, z = golden_model(name, shape, mix, tmp_path)
    prompt = [int(t) for t in z["prompt"]]
    r = np.random.Generator(np.random.Philox(key=[20260925, 31]))
    fed = [int(t) for t in r.integers(0, 256, 24)]
    outs = {}
    for mode in ("separate", "fused", "fused_graph"):
        eng = E.Engine()
        eng.load(path, int(z["ctx"]))
        eng.set_option("fuse_attention", mode != "separate")
        lg = [eng.forward(prompt, 0)]
        pos = len(prompt)
        for t in fed:
            lg.append(eng.decode_fused(t, pos, mode == "fused_graph"))
            pos += 1
        toks = eng.decode_greedy_steps(fed[-1], pos, 16)
        outs[mode] = (np.stack(lg), toks)
        eng.close()
    for mode in ("fused", "fused_graph"):
        err = np.abs(outs[mode][0] - outs["separate"][0]).max()
        assert np.isfinite(outs[mode][0]).all() and err <= 5e-4, (name, mode, err)
    assert outs["fused"][1] == outs["fused_graph"][1]
    return "ok"


def main():
    lib = os.environ.get("NTK_LIB_PATH", "")
    assert lib.endswith("libntransformer_hip_exp.so"), "run with NTK_LIB_PATH=<...>/libntransformer_hip_exp.so"
    with tempfile.TemporaryDirectory() as d:
        for name, shape, mix in CASES:
            print("persistent", name, persistent(name, shape, mix, Path(d)), flush=True)
        for name, shape, mix in [c for c in CASES if c[0] in ("tiny_q8_0", "tiny_q4_k_m", "small_q8_0", "small_q6_k")]:
            print("attention_in_wo", name, attention_in_wo(name, shape, mix, Path(d)), flush=True)
    print("experiments ok")


if __name__ == "__main__":
    main()
