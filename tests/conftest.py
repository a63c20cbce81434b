"""pytest configuration: `gpu` marker + import path.  `-m "not gpu"` runs here (no GPU);
`-m gpu` runs on the MI355X box and goes through the C ABI of the HIP library."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: builder-run (NT_RUN_SLOW=1): minutes of CPU oracle on a full-size model")


def pytest_sessionstart(session):
    """A clean checkout has no built library (binaries are git-ignored): build the product once (hipcc cross-compiles gfx950
    without a GPU) so that the host-logic / ABI tests can load it.  Compiling is not a fallback: nothing runs on the CPU."""
    lib = os.path.join(ROOT, "ntransformer_amd", "libntransformer_hip.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "ntransformer_amd", "csrc"), "all"])
