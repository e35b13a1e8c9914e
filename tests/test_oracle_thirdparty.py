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


@pytest.mark.parametrize("k,swap", [("fwd", False), ("swap", True)])
def test_joint_model_conversion_vs_sklearn_scipy_lapack(joint_model, k, swap):
    """The reference's second trained model (test/vc.jl:40-51), same third-party pins."""
    w, mu, sig = joint_model
    z = load_golden("gmmmap_joint_model.npz")
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


def test_mc2e_vs_frequency_domain_evaluation():
    """mc2e (src/align.jl:48, third-party MelGeneralizedCepstrums): the golden energies (SPTK freqt + c2ir recursions in
    both oracle restatements) against a frequency-domain evaluation that uses neither recursion -- the all-pass-warped
    log spectrum exponentiated on a 65536-point grid, numpy.fft.ifft, energy of the first fftlen samples.  Truncating the
    linear cepstrum to order fftlen-1 cannot change h[0..fftlen-1], so the two agree to rounding; the energy beyond fftlen
    samples (Parseval minus truncated) is what the fixed response length leaves out."""
    z = load_golden("align_mcep_case.npz")
    out = cc.check_mc2e(z["src"], 0.41, 256, z["energy"], tol=1e-10)
    assert out["mc2e_vs_frequency_domain"] < 1e-12 and out["energy_beyond_fftlen_relative"] < 1e-9
    # a long, slowly decaying response: here the truncation term is visible and has the right sign
    c = np.zeros((1, 3)); c[0] = [0.2, 0.9, 0.3]
    from oracle import c_oracle as co
    e_t, e_f = cc.mc2e_frequency_domain(c, 0.41, 16)
    assert abs(co.mc2e(c, 0.41, 16)[0] - e_t[0]) < 1e-12 * e_t[0] and e_f[0] > e_t[0] * (1 + 1e-6)
    # and the C and numpy restatements on other (alpha, fftlen)
    from oracle import np_oracle as npo
    rng = np.random.default_rng(3)
    mc = rng.standard_normal((5, 13)) * np.exp(-0.4 * np.arange(13)) * 0.4
    for alpha, n in ((0.35, 128), (0.0, 64), (0.55, 512)):
        e_t, _ = cc.mc2e_frequency_domain(mc, alpha, n)
        assert np.max(np.abs(co.mc2e(mc, alpha, n) - e_t) / e_t) < 1e-11
        assert np.max(np.abs(npo.mc2e(mc, alpha, n) - e_t) / e_t) < 1e-11


def test_gv_ascent_vs_dense_numpy(fixture_model):
    """TrajectoryGVGMMMap fvconvert (src/trajectory_gmmmap.jl:139-168): the golden GV trajectory (100 epochs) against a
    dense evaluation -- W materialised with scipy.sparse, dense normal equations solved by numpy.linalg.solve,
    numpy.var(ddof=1), numpy.linalg.inv -- that shares no code with either oracle restatement."""
    w, mu, sig = fixture_model
    t = load_golden("trajectory_fixture_model.npz")
    g = load_golden("gv_fixture_model.npz")
    out = cc.check_gv(w, mu, sig, t["X"], g["muv"], g["sigmavv"], g["Y_gv"], epochs=100, alpha=1.0e-5, tol=1e-6)
    assert out["gv_ascent_vs_dense_numpy"] < 1e-9


def test_gvgrad_and_one_step_vs_mpmath_50_digits():
    """`gvgrad` (src/trajectory_gmmmap.jl:171-189) of both restatements equals (T-1)/T times the gradient of the GV
    log-density, taken by central differences in 50-digit arithmetic; and one whole ascent step (:163-166) of the C oracle
    agrees with a 50-digit dense evaluation."""
    import synthdata as sd
    from oracle import c_oracle as co
    from oracle import np_oracle as npo
    rng = np.random.default_rng(5)
    y = rng.standard_normal((7, 3)) * np.array([1.0, 0.5, 2.0]) + 3
    muv = np.array([1.2, 0.3, 3.0])
    A = rng.standard_normal((3, 3))
    Sv = A @ A.T + np.eye(3)
    out = cc.check_gvgrad_is_the_gv_likelihood_gradient(y, muv, Sv, npo.gvgrad)
    assert out["gvgrad_vs_mpmath_gradient_times_(T-1)/T"] < 1e-12
    w2, mu2, sig2 = sd.synth_model(77, 12, 2, lam_lo=1e-2)
    st = np.cumsum(sd.sample_frames(78, w2, mu2, sig2, 6, 0, 3), axis=0)
    X = npo.push_delta(st)
    muv2, Sv2 = np.array([0.8, 1.1, 0.6]), np.diag([0.2, 0.1, 0.3]) + 0.02

    class Step:
        alpha = 1e-3

        def __call__(self, y0):
            return co.TrajectoryGMMMap(co.GMMMap(w2, mu2, sig2)).fvconvert_gv(X, muv2, Sv2, 1, self.alpha)

    assert cc.check_gv_step_mpmath(w2, mu2, sig2, X, muv2, Sv2, Step())["gv_step_vs_mpmath"] < 1e-12

    class StepNp(Step):
        def __call__(self, y0):
            tj = npo.TrajectoryGMMMap(npo.GMMMap(w2, mu2, sig2))
            return npo.trajgv_fvconvert(tj, X, muv2, Sv2, epochs=1, alpha=self.alpha)

    assert cc.check_gv_step_mpmath(w2, mu2, sig2, X, muv2, Sv2, StepNp())["gv_step_vs_mpmath"] < 1e-12
