import os
import sys
from pathlib import Path

import pytest

# GPU tests: fill the sample buffer with NaN bit patterns before every trace launch, so that a (pixel, sample)
# the kernel fails to write cannot hide behind a previous render's value or a zero-initialised allocation.
# (RBRT_POISON_SAMPLES and the scheduling knobs some tests set are lab knobs: the library reads them only with
# RBRT_HIP_LAB=1, include/rbrt_hip_debug.h)
os.environ.setdefault("RBRT_HIP_LAB", "1")
os.environ.setdefault("RBRT_POISON_SAMPLES", "1")

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): oracle/pyoracle.py over oracle/librbrt_oracle.so."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def hip():
    """The product library. No fallback: a missing .so or a missing GPU is an error, not a skip."""
    import rbrt_amd
    rbrt_amd.load_hip()
    n = rbrt_amd.device_count()
    assert n >= 1, "gpu-marked test running without a HIP device"
    return rbrt_amd
