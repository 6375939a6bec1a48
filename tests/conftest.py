import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle (float64 torch on small per-frame tensors) oversubscribes badly on many-core hosts — the same step took 306 s with 256
    # threads and 3.9 s with 8 (bench.py: cpu_baseline) — and the GPU boxes share their host cores between jobs: a fixed, modest thread count
    # makes the oracle-heavy parity tests both faster and repeatable (HARP_TEST_THREADS overrides)
    import torch
    torch.set_num_threads(max(1, min(int(os.environ.get("HARP_TEST_THREADS", "16")), os.cpu_count() or 1)))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
