import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pythonic-disort_amd"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The CPU suite checks the C-ABI library's exported symbols: build it (hipcc cross-compiles without a GPU)
    when the in-tree librtd.so is missing, e.g. on a fresh checkout."""
    lib = os.path.join(ROOT, "pythonic-disort_amd", "pydisort_amd", "librtd.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess
        subprocess.run([sys.executable, os.path.join(ROOT, "pythonic-disort_amd", "build.py")], check=True)
