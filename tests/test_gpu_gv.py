"""GPU parity for the SURVEY 8(f) rank-2 / rank-4 rows: TrajectoryGVGMMMap (src/trajectory_gmmmap.jl:114-189),
VarianceScaling (src/gv.jl) and diffgmm (src/diffgmm.jl) vs the C oracle.
Tolerance 1e-6 relative: the GV ascent starts from the trajectory solve, which the oracle and the GPU agree on to
~1e-9 (different direct solvers, cond(P) ~ 1e6); north_star: 1e-5."""
import numpy as np
import pytest

from conftest import julia_model, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture(scope="module")
def vc():
    import voiceconversion_jl_amd as m
    assert m.device_count() >= 1
    return m


def _gv_stats(rng, Y):
    """GV mean slightly above the converted track's variance, and a dense SPD GV covariance."""
    D = Y.shape[1]
    muv = Y.var(axis=0, ddof=1) * 1.3
    A = rng.standard_normal((D, D))
    return muv, A @ A.T / D * np.mean(muv) ** 2 * 0.1 + np.diag(muv ** 2 * 0.05)


def _utterances(npo, rng, w, mu, sig, D, Ts):
    Xs = []
    for T in Ts:
        static = npo.sample_frames(int(rng.integers(1 << 30)), w, mu, sig, T, 0, D)
        static = np.cumsum(static, axis=0) / np.sqrt(np.arange(1, T + 1))[:, None]
        Xs.append(npo.push_delta(static))
    return Xs


def test_fixture_model_gv(vc, fixture_model):
    """the reference's own model (test/models, read as static 20 + delta 20), 100 frames, default epochs / alpha"""
    from conftest import load_golden
    from oracle import c_oracle as co
    z = load_golden("trajectory_fixture_model.npz")
    w, mu, sig = fixture_model
    ref = co.TrajectoryGMMMap(co.GMMMap(w, mu, sig))
    muv, Sv = _gv_stats(np.random.default_rng(0), z["Y"])
    want = ref.fvconvert_gv(z["X"], muv, Sv, 100, 1.0e-5)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    t = vc.TrajectoryGMMMap(g, 100)
    tgv = vc.TrajectoryGVGMMMap(t, muv, Sv)
    assert len(tgv) == 100 and vc.dim(tgv) == 40 and vc.ncomponents(tgv) == 32
    got = vc.fvconvert(tgv, z["X"].T)
    assert got.shape == (20, 100) and relerr(got, want.T) < TOL
    # the ascent really moved the trajectory away from both the plain solve and its rescaled start
    assert relerr(got, z["Y"].T) > 1e-3
    # epochs = 0 is the eq. (58) initialisation alone = VarianceScaling(mu^v) of the plain trajectory
    init = vc.fvconvert(tgv, z["X"].T, epochs=0)
    assert relerr(init, vc.fvpostf(vc.VarianceScaling(muv), vc.fvconvert(t, z["X"].T))) < 1e-12
    with pytest.raises(AssertionError):
        vc.TrajectoryGVGMMMap(t, -muv, Sv)
    # ... and the HIP path against the INDEPENDENT dense evaluation (oracle/crosscheck.py: scipy.sparse W, dense LU,
    # numpy.var -- shares no code with either oracle restatement), golden GV statistics, and one ascent step against
    # 50-digit mpmath on the library's own output
    from oracle import crosscheck as cc
    gz = load_golden("gv_fixture_model.npz")
    tg = vc.TrajectoryGVGMMMap(t, gz["muv"], gz["sigmavv"])
    got2 = vc.fvconvert(tg, z["X"].T)
    out = cc.check_gv(w, mu, sig, z["X"], gz["muv"], gz["sigmavv"], got2.T, epochs=100, alpha=1.0e-5, tol=TOL)
    assert out["gv_ascent_vs_dense_numpy"] < TOL


def test_one_gv_step_vs_mpmath_50_digits(vc):
    """One ascent step (src/trajectory_gmmmap.jl:163-166) of the HIP path against a dense 50-digit mpmath evaluation
    (tiny case: static D = 3, M = 2, T = 6)."""
    import synthdata as sd
    from oracle import crosscheck as cc
    w2, mu2, sig2 = sd.synth_model(77, 12, 2, lam_lo=1e-2)
    st = np.cumsum(sd.sample_frames(78, w2, mu2, sig2, 6, 0, 3), axis=0)
    X = np.ascontiguousarray(vc.push_delta(np.asfortranarray(st.T)).T)
    muv2, Sv2 = np.array([0.8, 1.1, 0.6]), np.diag([0.2, 0.1, 0.3]) + 0.02
    g = vc.GMMMap(*julia_model(w2, mu2, sig2))
    tgv = vc.TrajectoryGVGMMMap(vc.TrajectoryGMMMap(g, 6), muv2, Sv2)

    class Step:
        alpha = 1e-3

        def __call__(self, y0):
            return vc.fvconvert(tgv, X.T, epochs=1, alpha=self.alpha).T

    assert cc.check_gv_step_mpmath(w2, mu2, sig2, X, muv2, Sv2, Step())["gv_step_vs_mpmath"] < 1e-10


@pytest.mark.parametrize("D,M,Ts,epochs", [(40, 8, [300, 37], 20), (12, 4, [2, 3, 17, 50], 100), (25, 5, [64, 129], 30),
                                           (13, 3, [40, 9], 25), (35, 4, [70], 10)])
def test_vs_oracle_batch(vc, D, M, Ts, epochs):
    """config-5 shape (static D=40) and ragged / tiny utterances (T=2: both stencil neighbours missing somewhere;
    tiles of 16 frames with several mixtures; D odd -> padded k-steps)"""
    from oracle import c_oracle as co, np_oracle as npo
    w, mu, sig = npo.synth_model(600 + D, 4 * D, M, lam_lo=1e-3)
    ref = co.TrajectoryGMMMap(co.GMMMap(w, mu, sig))
    g = vc.GMMMap(*julia_model(w, mu, sig))
    t = vc.TrajectoryGMMMap(g, max(Ts))
    rng = np.random.default_rng(D)
    Xs = _utterances(npo, rng, w, mu, sig, D, Ts)
    y_long = ref.fvconvert(Xs[0])[0]
    muv, Sv = _gv_stats(rng, y_long if y_long.shape[0] > 2 else np.vstack([y_long, y_long + 1.0]))
    tgv = vc.TrajectoryGVGMMMap(t, muv, Sv)
    got = tgv.fvconvert_batch([x.T for x in Xs], epochs=epochs, alpha=1.0e-5)
    for x, y in zip(Xs, got):
        want = ref.fvconvert_gv(x, muv, Sv, epochs, 1.0e-5)
        assert relerr(y, want.T) < TOL, x.shape
    with pytest.raises(vc.DimensionMismatch):
        tgv.fvconvert_batch([Xs[0][:1].T])            # a one-frame trajectory has no variance


def test_vc_chunks_with_gv(vc):
    """vc(c::TrajectoryConverter, fm) dispatches on the GV converter too (src/common.jl:31-63): chunks of length(c)"""
    from oracle import c_oracle as co, np_oracle as npo
    D, M, T, L = 12, 4, 70, 30
    w, mu, sig = npo.synth_model(77, 4 * D, M, lam_lo=1e-3)
    ref = co.TrajectoryGMMMap(co.GMMMap(w, mu, sig))
    rng = np.random.default_rng(3)
    X = _utterances(npo, rng, w, mu, sig, D, [T])[0]
    muv, Sv = _gv_stats(rng, ref.fvconvert(X)[0])
    fm = np.concatenate([np.linspace(0, 1, T)[:, None], X], axis=1)
    tgv = vc.TrajectoryGVGMMMap(vc.TrajectoryGMMMap(vc.GMMMap(*julia_model(w, mu, sig)), L), muv, Sv)
    out = vc.vc(tgv, fm.T)
    assert out.shape == (D + 1, T) and np.array_equal(out[0], fm[:, 0])
    for b in range(0, T, L):
        want = ref.fvconvert_gv(X[b:b + L], muv, Sv, 100, 1.0e-5)
        assert relerr(out[1:, b:b + L], want.T) < TOL


def test_variance_scaling_and_diffgmm(vc, fixture_model):
    from oracle import c_oracle as co, np_oracle as npo
    rng = np.random.default_rng(5)
    src = rng.standard_normal((700, 25)) * rng.uniform(0.1, 3.0, 25) + rng.standard_normal(25)
    s2 = rng.uniform(0.5, 2.0, 25)
    got = vc.fvpostf(vc.VarianceScaling(s2), src.T)
    assert relerr(got, co.variance_scaling(src, s2).T) < 1e-12
    assert np.allclose(got.var(axis=1, ddof=1), s2, rtol=1e-12) and np.allclose(got.mean(axis=1), src.mean(axis=0), atol=1e-12)
    big = rng.standard_normal((300000, 40)) * rng.uniform(0.1, 3.0, 40) + 5.0     # many workgroup chunks
    s40 = rng.uniform(0.5, 2.0, 40)
    gb = vc.fvpostf(vc.VarianceScaling(s40), big.T)
    assert relerr(gb, npo.variance_scaling(big, s40).T) < 1e-12
    buf = np.asfortranarray(src.T.copy())
    assert vc.fvpostf_(vc.VarianceScaling(s2), buf) is buf and np.array_equal(buf, got)
    # diffgmm: parameters, then the converter built from them against the oracle built from the oracle's transform
    w, mu, sig = fixture_model
    mu_j, sig_j = julia_model(w, mu, sig)[1:]
    md, sd = vc.diffgmm(mu_j, sig_j)
    mo, so = co.diffgmm(mu, sig)
    assert np.array_equal(md, mo.T) and relerr(sd, np.transpose(so, (2, 1, 0))) < 1e-15
    X = npo.sample_frames(9, w, mu, sig, 200, 0, 40)
    y = vc.fvconvert(vc.GMMMap(w, md, sd), X.T)
    assert relerr(y, co.GMMMap(w, mo, so).fvconvert(X).T) < 1e-9
    # the differential converter predicts y - x: same posterior, E[y|x] - x
    full = vc.fvconvert(vc.GMMMap(*julia_model(w, mu, sig)), X.T)
    assert relerr(y, full - X.T) < 1e-6


def test_gv_more_utterances_than_compute_units(vc):
    """Ragged batch larger than the CU count: every workgroup runs several utterances one after the other; each result
    equals the single-utterance call bit for bit (the frame order inside a mixture group comes from atomics, but each
    frame's product is computed independently)."""
    from oracle import c_oracle as co, np_oracle as npo
    D, M = 12, 4
    w, mu, sig = npo.synth_model(911, 4 * D, M, lam_lo=1e-3)
    ref = co.TrajectoryGMMMap(co.GMMMap(w, mu, sig))
    t = vc.TrajectoryGMMMap(vc.GMMMap(*julia_model(w, mu, sig)), 40)
    rng = np.random.default_rng(8)
    Ts = [int(v) for v in rng.integers(2, 41, size=600)]
    Xs = _utterances(npo, rng, w, mu, sig, D, Ts)
    muv, Sv = _gv_stats(rng, ref.fvconvert(Xs[int(np.argmax(Ts))])[0])
    tgv = vc.TrajectoryGVGMMMap(t, muv, Sv)
    got = tgv.fvconvert_batch([x.T for x in Xs], epochs=7, alpha=1.0e-5)
    for i in list(range(0, 600, 41)) + [599]:
        assert np.array_equal(vc.fvconvert(tgv, Xs[i].T, epochs=7, alpha=1.0e-5), got[i])
        assert relerr(got[i], ref.fvconvert_gv(Xs[i], muv, Sv, 7, 1.0e-5).T) < TOL


def test_one_team_and_two_team_kernels_agree(vc):
    """traj_gv2_kernel (gather waves beside the MFMA waves; the default when the frame permutation fits in LDS) against
    traj_gv_kernel (forced through the debug hook; the path of very long utterances): same ascent, different thread counts in the
    moment reductions -> agreement to rounding."""
    from oracle import c_oracle as co, np_oracle as npo
    D, M = 20, 6
    w, mu, sig = npo.synth_model(620, 4 * D, M, lam_lo=1e-3)
    ref = co.TrajectoryGMMMap(co.GMMMap(w, mu, sig))
    t = vc.TrajectoryGMMMap(vc.GMMMap(*julia_model(w, mu, sig)), 200)
    rng = np.random.default_rng(21)
    Xs = _utterances(npo, rng, w, mu, sig, D, [200, 33, 2])
    muv, Sv = _gv_stats(rng, ref.fvconvert(Xs[0])[0])
    tgv = vc.TrajectoryGVGMMMap(t, muv, Sv)
    two = tgv.fvconvert_batch([x.T for x in Xs], epochs=25, alpha=1.0e-5)
    from voiceconversion_jl_amd import _lib
    _lib.debug_force(_lib.DBG_GV_ONE_TEAM)
    try:
        one = tgv.fvconvert_batch([x.T for x in Xs], epochs=25, alpha=1.0e-5)
    finally:
        _lib.debug_force(0)
    for a, b in zip(two, one):
        assert relerr(a, b) < 1e-10
