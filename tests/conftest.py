import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# The oracle runs on the host: oneDNN convolutions at batch 1-4 are fastest with ~16 threads on the
# MI355X box (256 hardware threads; the default of 128 is 4-10x slower, measured in tests/probes).
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on CPU")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=False)


@pytest.fixture(scope="session", autouse=True)
def _build_lib():
    """Make sure the HIP library exists (hipcc cross-compiles without a GPU)."""
    from diffpure_amd import build

    try:
        build.build()
    except Exception as e:  # no hipcc on this machine: tests that need the .so will fail loudly
        print("WARNING: could not build libdiffpure_hip.so:", e)
