"""CPU: the committed golden vectors re-verified against THIRD-PARTY implementations (oracle/crosscheck.py): sklearn
GaussianMixture, scipy multivariate_normal / logsumexp / solveh_banded, numpy.linalg.solve and 50-digit mpmath.
These are the pins of the floating-point oracle (the reference's own tests assert only isfinite); reference lines covered:
src/gmm.jl:24-30,44-47, src/gmmmap.jl:33-36,109-117, src/trajectory_gmmmap.jl:82-105, the E-step of bin/train_gmm.jl:103."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import crosscheck as cc


@pytest.mark.parametrize("k,swap", [("fwd", False), ("swap", True)])
def test_fixture_model_conversion_vs_sklearn_scipy_lapack(fixture_model, k, swap):
    w, mu, sig = fixture_model
    z = load_golden("gmmmap_fixture_model.npz")
    out = cc.check_conversion(w, mu, sig, z[f"X_{k}"], z[f"Y_{k}"], z[f"P_{k}"], z[f"idx_{k}"], swap=swap, tol=1e-9)
    assert out["fvconvert_vs_numpy_solve"] < 1e-9


def test_config1_conversion_vs_third_party():
    z = load_golden("gmmmap_cfg1_D24_M8_T1000.npz")
    from oracle import np_oracle as npo
    P = npo.GMMMap(z["weights"], z["means"], z["covars"]).predict_proba(z["X"])
    cc.check_conversion(z["weights"], z["means"], z["covars"], z["X"], z["Y"], P, z["idx"], tol=1e-9)


def test_fixture_model_conversion_vs_mpmath_50_digits(fixture_model):
    w, mu, sig = fixture_model
    z = load_golden("gmmmap_fixture_model.npz")
    out = cc.check_conversion_mpmath(w, mu, sig, z["X_fwd"], z["Y_fwd"], z["P_fwd"], frames=16, dps=50, tol=1e-9)
    assert out["frames"] == 16 and out["fvconvert_vs_mpmath"] < 1e-9     # north_star bar: 1e-5


def test_trajectory_solve_vs_lapack_banded(fixture_model):
    w, mu, sig = fixture_model
    z = load_golden("trajectory_fixture_model.npz")
    out = cc.check_trajectory(w, mu, sig, z["X"], z["Y"], z["mhat"], z["Ey"], tol=1e-6)
    assert out["trajectory_vs_solveh_banded"] < 1e-6


def test_estep_statistics_vs_sklearn():
    z = load_golden("estep_diag_N2000_D80_M16.npz")
    cc.check_estep_diag(z["X"], z["w"], z["mu"], z["var"], z["S0"], z["S1"], z["S2"], z["loglik"])
    f = load_golden("estep_full_N1000_D80_M8.npz")
    cc.check_estep_full(f["X"], f["w"], f["mu"], f["sigma"], f["S0"], f["S1"], f["S2"], f["loglik"])
