import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# torch ships its own copy of the HIP runtime: a process that first initialises the system's (through libpaml_amd.so) and then
# imports torch ends up with "No HIP GPUs are available" in torch.  The tests that use torch for device memory therefore need it
# loaded before the engine library, whatever subset of tests is selected — as bench.py does.
try:
    import torch  # noqa: F401
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")
