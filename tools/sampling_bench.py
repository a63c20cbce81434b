#!/usr/bin/env python3
"""Decode rate with the reference CLI's DEFAULT sampling settings (-t 0.7 --top-k 40 --top-p 0.9 --repeat-penalty 1.1,
reference src/main.cpp:31-37): sampler on the device (ntk_sample_top_k) against the host sampler (513 KB logits download +
partial_sort of 128 256 candidates per token), and greedy for scale.  Engine::Stats::decode_tok_s of Engine::generate."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ntransformer_amd import engine as E

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="8b"); ap.add_argument("--mix", default="Q8_0"); ap.add_argument("-n", type=int, default=128)
a = ap.parse_args()
spec = E.synth_spec(a.model, a.mix)
rng = np.random.Generator(np.random.Philox(key=[20260925, 99]))
prompt = [spec.bos] + [int(t) for t in rng.integers(0, spec.vocab, 15)]
eng = E.Engine()
eng.load_synthetic(spec, 4096)
for name, dev, kw in (("greedy, device argmax", 1, dict(temperature=0.0, repeat_penalty=1.0)),
                      ("default sampling, device sampler", 1, dict(temperature=0.7, top_k=40, top_p=0.9, repeat_penalty=1.1)),
                      ("default sampling, host sampler", 0, dict(temperature=0.7, top_k=40, top_p=0.9, repeat_penalty=1.1)),
                      ("greedy + repeat penalty 1.1, device", 1, dict(temperature=0.0, repeat_penalty=1.1)),
                      ("greedy + repeat penalty 1.1, host", 0, dict(temperature=0.0, repeat_penalty=1.1))):
    eng.set_option("device_sampling", dev)
    best = 0.0
    for rep in range(2):
        toks = eng.generate_tokens(prompt, a.n, seed=42, repeat_window=64, stop_at_eos=False, **kw)
        st = eng.stats()
        best = max(best, st.decode_tok_s)
    print("%-42s %8.1f tok/s  (%d tokens, %.3f ms/token)" % (name, best, st.gen_tokens, 1e3 / best))
eng.close()
