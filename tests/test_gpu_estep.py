"""GPU parity: diagonal E-step sufficient statistics vs the golden vectors and the C oracle.
Tolerance 1e-9 relative to the largest statistic (sums of ~N terms in different association orders)."""
import ctypes

import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-9


@pytest.fixture(scope="module")
def vc():
    import voiceconversion_jl_amd as m
    assert m.device_count() >= 1
    return m


def _force_generic(vc, on):
    from voiceconversion_jl_amd import _lib
    _lib.debug_force(_lib.DBG_ESTEP_GENERIC if on else 0)


@pytest.mark.parametrize("generic", [False, True])
def test_golden(vc, generic):
    z = load_golden("estep_diag_N2000_D80_M16.npz")
    _force_generic(vc, generic)
    try:
        S0, S1, S2, ll = vc.estep_diag(z["X"].T, z["w"], z["mu"].T, z["var"].T)
    finally:
        _force_generic(vc, False)
    assert relerr(S0, z["S0"]) < TOL and relerr(S1, z["S1"].T) < TOL and relerr(S2, z["S2"].T) < TOL
    assert abs(ll - float(z["loglik"])) < TOL * abs(float(z["loglik"]))
    assert abs(S0.sum() - 2000) < 1e-6          # responsibilities sum to one per frame


@pytest.mark.parametrize("N,Dj,M", [(5000, 80, 128), (777, 80, 100), (1, 80, 3), (300, 48, 8), (1000, 6, 2), (70000, 10, 4)])
def test_vs_oracle(vc, N, Dj, M):
    from oracle import c_oracle as co, np_oracle as npo
    w, mu, _ = npo.synth_model(3000 + N, Dj, M)
    rg = np.random.default_rng(N)
    var = np.exp(rg.uniform(np.log(1e-3), 0.0, (M, Dj)))
    comp = rg.choice(M, size=N, p=w)
    X = mu[comp] + rg.standard_normal((N, Dj)) * np.sqrt(var[comp])
    r0, r1, r2, rl = co.estep_diag(X, w, mu, var)
    S0, S1, S2, ll = vc.estep_diag(X.T, w, mu.T, var.T)
    assert relerr(S0, r0) < TOL and relerr(S1, r1.T) < TOL and relerr(S2, r2.T) < TOL
    assert abs(ll - rl) < TOL * abs(rl)


def test_device_resident_deterministic_and_additive(vc):
    """Run-to-run bit-identical (fixed-order reductions, no atomics) and additive over frame shards -- the
    property the multi-GPU all-reduce relies on."""
    import torch
    from oracle import np_oracle as npo
    Dj, M, N = 80, 128, 40000
    w, mu, _ = npo.synth_model(77, Dj, M)
    rg = np.random.default_rng(7)
    var = np.exp(rg.uniform(np.log(1e-3), 0.0, (M, Dj)))
    comp = rg.choice(M, size=N, p=w)
    X = torch.from_numpy(mu[comp] + rg.standard_normal((N, Dj)) * np.sqrt(var[comp])).cuda()
    a = vc.estep_diag_dev(X.t(), w, mu.T, var.T)
    b = vc.estep_diag_dev(X.t(), w, mu.T, var.T)
    assert torch.equal(a, b)
    h = vc.estep_diag_dev(X[:N // 2].t(), w, mu.T, var.T) + vc.estep_diag_dev(X[N // 2:].t(), w, mu.T, var.T)
    assert float((a - h).abs().max() / a.abs().max()) < 1e-12
    S0, S1, S2, ll = vc.unpack_stats(a.cpu().numpy(), Dj, M)
    assert abs(S0.sum() - N) < 1e-6 * N
    # M-step sanity: re-estimated means stay close to the generating means for well-populated components
    w2, mu2, var2 = vc.mstep_diag(S0, S1, S2)
    big = S0 > 200
    assert np.max(np.abs(mu2[:, big] - mu.T[:, big])) < 0.2


def test_errors(vc):
    with pytest.raises(vc.PosDefException):
        vc.estep_diag(np.zeros((4, 10)), np.ones(2) / 2, np.zeros((4, 2)), np.zeros((4, 2)))
    with pytest.raises(vc.DimensionMismatch):
        vc.estep_diag(np.zeros((5, 10)), np.ones(2) / 2, np.zeros((4, 2)), np.ones((4, 2)))
