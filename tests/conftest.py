import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mpas-ocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


def pytest_sessionstart(session):
    """Bring the two shared libraries up to date with their sources (a no-op when they are): the HIP library
    cross-compiles without a GPU, and a stale .so would test yesterday's code."""
    import subprocess
    for d in (os.path.join(ROOT, "mpas-ocean.jl_amd"), os.path.join(ROOT, "oracle")):
        subprocess.run(["make", "-C", d, "--no-print-directory", "-j4"], check=True, stdout=subprocess.DEVNULL)
