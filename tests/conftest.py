import os
import sys

import numpy as np
import pytest

os.environ.setdefault("VCMI_TEST_HOOKS", "1")       # enables the library's vcmi_debug_force test hook for this process
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def fixture_model():
    """Trained joint GMM of the reference's test/models/clb_to_slt_gmm32_order40_diff.jld (M=32, Dj=80).
    Arrays are the raw Julia memory images: weights (M), means [m][d], covars [m][col][row]."""
    z = load_golden("model_clb_to_slt_gmm32_order40_diff.npz")
    return z["weights"], z["means"], z["covars"]


@pytest.fixture(scope="session")
def joint_model():
    """The reference's other trained model, test/models/clb_and_slt_gmm32_order40.jld (the joint, non-differential model
    test/vc.jl:40-51 converts with; M=32, Dj=80), extracted by oracle/gen_golden_joint.py.  Same array conventions."""
    z = load_golden("model_clb_and_slt_gmm32_order40.npz")
    return z["weights"], z["means"], z["covars"]


def julia_model(w, mu, sig):
    """numpy [m][d] / [m][col][row] buffers -> Julia-shaped (Dj,M) / (Dj,Dj,M) Fortran arrays (no data change)."""
    M, Dj = mu.shape
    return w, np.asfortranarray(mu.T), np.asfortranarray(np.transpose(sig, (2, 1, 0)))


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))


def frame_relerr(Y, Yref):
    """max over frames of |y - yref| / |yref| for (D,T)-shaped arrays."""
    num = np.linalg.norm(np.asarray(Y) - np.asarray(Yref), axis=0)
    den = np.linalg.norm(np.asarray(Yref), axis=0)
    return float(np.max(num / np.maximum(den, 1e-300)))
