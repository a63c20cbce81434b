#!/usr/bin/env python3
"""Generate tests/golden/* from the reference itself.  Runs ONLY in the build container, where
/root/reference exists; the outputs are committed because the reference cannot travel to the GPU box.

Two sources of truth are tapped:

1. The reference's own numpy dequantisers (reference tools/decompose_gguf.py:219-388: dequant_q6_k,
   dequant_q8_0, dequant_q4_k, dequant_q5_k), imported from where they lie.  They are an independent
   statement of the GGUF block formats by the reference's author  ->  tests/golden/dequant_<type>.npz
   {raw: uint8 block bytes, out_f, in_f, ref: float32[out_f, in_f]}.

2. The reference's unmodified host code (GGUF loader, Transformer::forward, Attention/FFN/RMSNorm
   orchestration, CPU embedding dequant) compiled from /root/reference by oracle/Makefile and linked
   with the CPU restatement of its kernels (oracle/_ref/ref_logits)  ->  tests/golden/<model>_logits.npz
   {prompt, forced, fed, argmax, logits[steps, vocab], gguf_sha256}.  The tiny GGUF inputs are
   committed next to them; the `small` (head_dim 128) model is regenerated from its seed at test time
   and checked against gguf_sha256.

usage: python tools/make_golden.py
"""
import hashlib
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ntransformer_amd import gguf as G  # noqa: E402
from oracle import oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SEED = 20260925


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def golden_dequant():
    spec = importlib.util.spec_from_file_location("decompose_gguf", "/root/reference/tools/decompose_gguf.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    fns = {"q8_0": (G.GGML_Q8_0, ref.dequant_q8_0), "q4_k": (G.GGML_Q4_K, ref.dequant_q4_k),
           "q5_k": (G.GGML_Q5_K, ref.dequant_q5_k), "q6_k": (G.GGML_Q6_K, ref.dequant_q6_k)}
    out_f, in_f = 6, 512
    for name, (gt, fn) in fns.items():
        rng = np.random.Generator(np.random.Philox(key=[SEED, gt]))
        raw = np.frombuffer(G.synth_tensor(rng, gt, out_f, in_f), np.uint8).copy()
        # exercise the full scale range incl. high 6-bit scale bits and negative int8 sub-scales
        deq = np.asarray(fn(raw.tobytes(), out_f, in_f), np.float32)
        np.savez_compressed(os.path.join(GOLD, "dequant_%s.npz" % name), raw=raw, out_f=out_f, in_f=in_f, ref=deq)
        print("dequant", name, deq.shape, "rms %.4f" % deq.std())


def golden_logits():
    O.build_ref()
    cases = [("tiny_q8_0", G.TINY, "Q8_0", True), ("tiny_q4_k_m", G.TINY, "Q4_K_M", True),
             ("tiny_mixed", G.TINY, "MIXED", True), ("small_q8_0", G.SMALL, "Q8_0", False),
             ("small_q4_k_m", G.SMALL, "Q4_K_M", False), ("small_q6_k", G.SMALL, "Q6_K", False)]
    for name, shape, mix, keep in cases:
        path = os.path.join(GOLD, name + ".gguf") if keep else os.path.join("/tmp", name + ".gguf")
        G.make_synthetic_llama(path, shape, mix, seed=SEED)
        rng = np.random.Generator(np.random.Philox(key=[SEED, 99]))
        prompt = [shape.bos] + [int(t) for t in rng.integers(0, shape.vocab, 7)]
        forced = [int(t) for t in rng.integers(0, shape.vocab, 4)]
        fed, am, lg = O.run_ref_logits(path, prompt, forced, n_greedy=4, ctx=128)
        np.savez_compressed(os.path.join(GOLD, name + "_logits.npz"), prompt=np.array(prompt, np.int32),
                            forced=np.array(forced, np.int32), fed=fed, argmax=am, logits=lg,
                            gguf_sha256=sha256(path), ctx=128, n_greedy=4)
        print(name, "logits", lg.shape, "rms %.3f" % lg.std(), "gguf %.1f KB" % (os.path.getsize(path) / 1024))
        if not keep:
            os.remove(path)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    golden_dequant()
    golden_logits()
