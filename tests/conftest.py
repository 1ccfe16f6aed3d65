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


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "stage2_golden.npz"))
