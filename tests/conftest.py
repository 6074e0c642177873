import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle runs on torch's CPU kernels, which crawl when every one of a GPU box's 256 host threads joins a small problem (the tiny-config
    # oracle took 8 minutes instead of 1 in a run that happened to start with the default): cap the pool for the whole session
    import torch
    torch.set_num_threads(min(int(os.environ.get("EW_ORACLE_THREADS", "32")), os.cpu_count() or 1))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
