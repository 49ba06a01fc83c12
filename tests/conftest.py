import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box with -m gpu)')
    # The CPU oracle runs on PyTorch-CPU.  On the 256-thread GPU hosts torch's default (one thread per logical CPU) makes its small
    # convolutions ~250x slower than 16 threads do (bench.py's cpu_baseline.thread_sweep_seconds: 1.5 s against 6 ms for one 3x3
    # conv) -- the round-5 GPU suite took 30 minutes that way, 25 of them in the oracle.
    import torch
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'))
    return load
