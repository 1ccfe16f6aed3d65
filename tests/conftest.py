import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    # the precision verdicts the GPU tests measure are remembered for the SESSION only (plan.py's disk cache, shared with the bench.py /
    # tools subprocesses the tests start): nothing is read from, or left in, the home directory
    if "EGOEGO_HIP_CACHE" not in os.environ:
        import tempfile
        os.environ["EGOEGO_HIP_CACHE"] = tempfile.mkdtemp(prefix="egoego_hip_cache_")
    # the fp32 CPU oracle runs chains of 1-8 windows: on the GPU box's 128+ hardware threads torch's default pool makes each small
    # matmul SLOWER (a 1000-step chain of 4 windows: 195 s on 128 threads; the whole GPU suite 468 -> 236 s with 16); the oracle's results are compared within
    # tolerances, never bit for bit with a thread count in between
    import torch
    if torch.get_num_threads() > 16:
        torch.set_num_threads(16)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "stage2_golden.npz"))
