import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "u-dales_amd"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch ships its own copies of the ROCm runtime libraries.  A process that loads libudcore.so (which links the system's
    # libamdhip64 / librccl / librocfft) and uses it BEFORE it imports torch aborts at exit ("double free or corruption" /
    # "free(): invalid pointer" from the two runtimes' tear-down; found when a subset of the suite ran a udcore test before the
    # first test that imports torch) -- the other order is fine, and it is bench.py's.  So: torch first, whatever the test order.
    try:
        import torch      # noqa: F401
    except ImportError:
        pass
