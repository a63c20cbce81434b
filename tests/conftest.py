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
