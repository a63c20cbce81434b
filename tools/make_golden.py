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


def spm_vocab():
    """SentencePiece-flavoured vocabulary with scores and <0xXX> byte tokens (Llama-1/2 style)."""
    toks = [b"<unk>", b"<s>", b"</s>"] + [b"<0x%02X>" % b for b in range(256)]
    types = [2, 3, 3] + [6] * 256
    sp = "\u2581".encode()
    words = [sp, b"t", b"h", b"e", b"l", b"o", b"w", b"r", b"d", b"a", b"n", b"s", b"i", b"c", b"u", b"m", b"p", b".", b",",
             b"he", b"th", b"the", sp + b"the", sp + b"t", b"ll", b"lo", b"llo", b"ello", b"hello", sp + b"hello", sp + b"h",
             b"wor", b"ld", b"world", sp + b"world", sp + b"w", b"an", b"and", sp + b"and", sp + b"a", b"in", b"ing",
             sp + sp, b"er", b"es", b"on", sp + b"s", b"st", b"ca", b"cat", sp + b"cat", b"\xc3\xa9", b"caf", sp + b"caf\xc3\xa9"]
    toks += words
    types += [1] * len(words)
    scores = [0.0] * 259 + [-float(i) * 0.5 - (3.0 if len(w) == 1 else 0.0) for i, w in enumerate(words)]
    return toks, scores, types


TEXTS = ["hello world", "the cat and the hat", " leading space", "trailing space ", "two  spaces", "caf\u00e9 au lait",
         "tabs\tand\nnewlines are bytes", "MiXeD CaSe 12345 !?", "\u65e5\u672c\u8a9e", "", "a", "the the the the"]


def golden_host_logic():
    import json
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_host")
    out = {"tokenizer": {}, "sampler": []}
    spm = os.path.join(GOLD, "vocab_spm.gguf")
    toks, scores, types = spm_vocab()
    G.write_vocab_gguf(spm, toks, scores, types, bos=1, eos=2)
    txt = "/tmp/_tok_lines.txt"
    with open(txt, "w", encoding="utf-8") as f:
        for t in TEXTS:
            f.write(t.replace("\n", " ") + "\n")
    for name, path in (("vocab_spm.gguf", spm), ("tiny_q8_0.gguf", os.path.join(GOLD, "tiny_q8_0.gguf"))):
        r = subprocess.run([exe, "tok", path, txt], capture_output=True, check=True)
        lines = r.stdout.decode().split("\n")[:len(TEXTS)]
        cases = []
        for t, l in zip(TEXTS, lines):
            ids = [int(x) for x in l.split()]
            d = subprocess.run([exe, "detok", path] + [str(i) for i in ids], capture_output=True, check=True).stdout
            cases.append({"text": t.replace("\n", " "), "ids": ids, "detok_hex": d.hex()})
        out["tokenizer"][name] = cases
    rng = np.random.Generator(np.random.Philox(key=[SEED, 4242]))
    logits = (rng.standard_normal(512) * 2.5).astype(np.float32)
    logits[[7, 99]] = logits.max() + 0.5          # an exact tie for the maximum: first index must win
    lp = os.path.join(GOLD, "sampler_logits.f32")
    logits.tofile(lp)
    for cfg in ({"temperature": 0.7, "top_k": 40, "top_p": 0.9, "repeat_penalty": 1.1, "repeat_window": 64, "seed": 42},
                {"temperature": 1.0, "top_k": 0, "top_p": 1.0, "repeat_penalty": 1.0, "repeat_window": 64, "seed": 7},
                {"temperature": 0.0, "top_k": 40, "top_p": 0.9, "repeat_penalty": 1.0, "repeat_window": 64, "seed": 1},
                {"temperature": 0.5, "top_k": 5, "top_p": 0.5, "repeat_penalty": 1.3, "repeat_window": 4, "seed": 123456789012345}):
        recent = [7, 99, 7, 3]
        args = [exe, "sample", lp, "512", repr(cfg["temperature"]), str(cfg["top_k"]), repr(cfg["top_p"]),
                repr(cfg["repeat_penalty"]), str(cfg["repeat_window"]), str(cfg["seed"]), "24"] + [str(t) for t in recent]
        r = subprocess.run(args, capture_output=True, check=True)
        out["sampler"].append({"cfg": cfg, "recent": recent, "draws": [int(x) for x in r.stdout.split()]})
    with open(os.path.join(GOLD, "host_logic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("host logic:", {k: len(v) for k, v in out["tokenizer"].items()}, len(out["sampler"]), "sampler configs")


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    golden_dequant()
    golden_logits()
    golden_host_logic()
