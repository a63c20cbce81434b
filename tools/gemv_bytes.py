#!/usr/bin/env python3
"""Algorithmic bytes of the matrices the GEMV launches of one decode token stream (weights minus the embedding table), and the
number of GEMV launches per token of the fused path -- the two constants the trace / PMC summaries divide by.
usage: python tools/gemv_bytes.py 8b Q8_0  ->  "<bytes_per_token> <launches_per_token>" """
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ntransformer_amd import engine as E  # noqa: E402
from ntransformer_amd import gguf as G  # noqa: E402


def main():
    model, mix = sys.argv[1], sys.argv[2]
    spec = E.synth_spec(model, mix)
    b = bench._gemv_bytes_per_token(spec, mix)
    # fused path: norm+Q|K|V, Wo+res, norm+gate|up+SiLU, down+res per layer, + the LM head
    print(b, 4 * spec.layers + 1)


if __name__ == "__main__":
    main()
