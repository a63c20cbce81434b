"""The weight ring of csrc/gemm_f16.hip is written by inline-asm loads and read by inline-asm LDS stores with explicit waits in
between; the compiler does not know the loads are asynchronous.  The build is correct only if no compiler-generated instruction
touches a ring register while its load is in flight -- tools/check_gemm_isa.py compiles the kernels to ISA and walks every
instantiation (prologue, the loop, once more around the back edge).  Runs wherever hipcc does (no GPU needed)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_ring_registers_are_untouched_while_their_loads_are_in_flight():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_gemm_isa.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if "gemm_quant_f16_kernel" in l]
    assert len(lines) >= 16 and all(l.rstrip().endswith("ok") for l in lines), r.stdout[-3000:]
