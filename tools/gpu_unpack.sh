#!/bin/bash
# round 5: the unpack kernel (ntk_rp_unpack): byte-exactness tests + what one resident copy costs a prompt pass (tools/gpu_suite.sh's second part)
TAG=${1:-up}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gemv_rp.py -m gpu -q -p no:cacheprovider -k "unpack or in_features_limit or matches_oracle" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
python - > $OUT/repack_modes.txt 2>&1 <<'PY'
import time, numpy as np
from ntransformer_amd import engine as E
for model, mix in (("8b", "Q4_K_M"), ("70b", "Q6_K"), ("70b", "Q4_K_M")):
    spec = E.synth_spec(model, mix, layers=None if model == "8b" else 16)
    for level in (1, 2):
        eng = E.Engine()
        eng.set_option("repack", level)
        eng.load_synthetic(spec, 4096)
        rng = np.random.Generator(np.random.Philox(key=[1, 2]))
        prompt = [spec.bos] + [int(t) for t in rng.integers(0, spec.vocab, 1023)]
        eng.forward(prompt, 0)
        t0 = time.perf_counter(); lg = eng.forward(prompt, 0); dt = time.perf_counter() - t0
        tok = int(np.argmax(lg))
        out = eng.decode_greedy_steps(tok, 1024, 32)
        t0 = time.perf_counter(); out = eng.decode_greedy_steps(out[-1], 1056, 64); dd = time.perf_counter() - t0
        print("%s %s%s repack=%d: weights %.2f GB, resident %.2f GB | prompt 1024 tokens %.1f ms = %.0f tok/s | decode %.1f tok/s | logits checksum %.6f"
              % (model, mix, "" if model == "8b" else " (16 layers)", level, eng.weight_bytes() / 1e9, eng.resident_weight_bytes() / 1e9, dt * 1e3, 1024 / dt, 64 / dd, float(np.abs(lg).sum())), flush=True)
        eng.close()
PY
grep repack= $OUT/repack_modes.txt
