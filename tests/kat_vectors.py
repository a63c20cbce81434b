"""Known-answer vectors restated from the reference's own kernel tests (reference tests/test_gemm.cpp).
Shared by the oracle tests (CPU) and the HIP parity tests (GPU) so both sides are pinned to the same
numbers.  Block byte layouts per reference src/core/types.h:96-137."""
import numpy as np

from ntransformer_amd import gguf as G


def _h(x: float) -> bytes:
    return np.float16(x).tobytes()


def q4_0_block(d: float, nib: int) -> bytes:
    return _h(d) + bytes([(nib << 4) | nib] * 16)


def q6_k_block(ql: int, qh: int, sc: int, d: float) -> bytes:
    return bytes([ql] * 128) + bytes([qh] * 64) + np.full(16, sc, np.int8).tobytes() + _h(d)


KATS = {
    # test_gemm.cpp:18-64: W 4x3 rows 1..12, x = ones -> {6, 15, 24, 33}
    "f32_4x3": dict(W=np.arange(1, 13, dtype=np.float32), x=np.ones(3, np.float32), out=4, **{"in": 3},
                    dtype=G.DT_F32, expect=[6, 15, 24, 33], tol=1e-3),
    # test_gemm.cpp:66-162: 2x32 Q4_0, d=0.5, nibbles 10 / 7, x = ones -> {32, -16}
    "q4_0_2x32": dict(W=np.frombuffer(q4_0_block(0.5, 10) + q4_0_block(0.5, 7), np.uint8), x=np.ones(32, np.float32),
                      out=2, **{"in": 32}, dtype=G.DT_Q4_0, expect=[32, -16], tol=0.1),
    # test_gemm.cpp:258-327: 2x256 Q6_K, row0 ql=0x11 qh=0xAA (q=33-32=+1), row1 ql=0xFF qh=0x55 (q=31-32=-1)
    "q6_k_2x256": dict(W=np.frombuffer(q6_k_block(0x11, 0xAA, 1, 1.0) + q6_k_block(0xFF, 0x55, 1, 1.0), np.uint8),
                       x=np.ones(256, np.float32), out=2, **{"in": 256}, dtype=G.DT_Q6_K, expect=[256, -256], tol=0.5),
    # test_gemm.cpp:330-397: in=32768 forces the reference's no-shared-memory variant -> +-32768
    "q6_k_2x32768": dict(W=np.frombuffer(q6_k_block(0x11, 0xAA, 1, 1.0) * 128 + q6_k_block(0xFF, 0x55, 1, 1.0) * 128, np.uint8),
                         x=np.ones(32768, np.float32), out=2, **{"in": 32768}, dtype=G.DT_Q6_K,
                         expect=[32768, -32768], tol=1.0),
}
