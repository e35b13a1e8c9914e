"""GPU parity: full-covariance E-step (bin/train_gmm.jl:84-103, covariance_type="full") vs the golden vectors
and the C oracle.  Tolerance 1e-9 relative to the largest statistic (sums of ~N terms in other orders)."""
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


def _check(got, ref, N):
    S0, S1, S2, ll = got
    r0, r1, r2, rl = ref
    assert relerr(S0, r0) < TOL and relerr(S1, r1.T) < TOL
    assert relerr(S2, np.transpose(r2, (2, 1, 0))) < TOL
    assert abs(ll - rl) < TOL * abs(rl)
    assert abs(S0.sum() - N) < 1e-6 * max(N, 1)
    assert np.array_equal(S2, np.transpose(S2, (1, 0, 2)))      # exactly symmetric


def test_golden(vc):
    z = load_golden("estep_full_N1000_D80_M8.npz")
    got = vc.estep_full(z["X"].T, z["w"], z["mu"].T, np.transpose(z["sigma"], (2, 1, 0)))
    _check(got, (z["S0"], z["S1"], z["S2"], float(z["loglik"])), 1000)


# Dj in {32,48,64,80} take the MFMA statistics kernel, the others the generic one; M not a multiple of 8,
# a single frame, and a frame count that is not a multiple of the 64-frame block are the ragged cases.
@pytest.mark.parametrize("N,Dj,M", [(3000, 80, 32), (777, 80, 13), (1, 80, 3), (1500, 48, 8), (900, 64, 5),
                                    (2000, 32, 16), (1000, 6, 2), (500, 50, 4), (40000, 10, 4),
                                    (2500, 160, 6), (130, 160, 3), (700, 128, 4), (600, 96, 9), (300, 100, 3), (800, 66, 5), (450, 25, 3),
                                    (350, 154, 2)])
def test_vs_oracle(vc, N, Dj, M):
    from oracle import c_oracle as co, np_oracle as npo
    w, mu, sig = npo.synth_model(5000 + N + Dj, Dj, M, lam_lo=1e-3)
    X = npo.sample_frames(N, w, mu, sig, N, 0, Dj)
    ref = co.estep_full(X, w, mu, sig)
    got = vc.estep_full(X.T, w, mu.T, np.transpose(sig, (2, 1, 0)))
    _check(got, ref, N)


@pytest.mark.parametrize("N,Dj,M,lam_lo", [(20000, 80, 64, 1e-5), (9000, 80, 16, 1e-1), (5000, 160, 6, 1e-3), (4096, 48, 40, 1e-4),
                                          (12000, 64, 200, 1e-5), (7000, 32, 300, 1e-3)])
def test_frame_lists_of_the_statistics_kernel(vc, N, Dj, M, lam_lo):
    """From 4096 frames on (and up to 32 mixture groups) the statistics kernel walks, per mixture group, the list of the
    frames that have a responsibility != 0 there (built in frame order from the softmax kernel's group mask) instead of
    staging every frame once per group.  Against the oracle, against the all-frames path (test hook) and run to run; peaked
    models (short lists), broad ones (every frame in every list) and mixture counts beyond 32 groups (lists off)."""
    import torch
    from oracle import c_oracle as co, np_oracle as npo
    from voiceconversion_jl_amd import _lib
    w, mu, sig = npo.synth_model(6100 + N, Dj, M, lam_lo=lam_lo)
    X = npo.sample_frames(N, w, mu, sig, N, 0, Dj)
    sg = np.transpose(sig, (2, 1, 0))
    Xd = torch.from_numpy(X).cuda()
    a = vc.estep_full_dev(Xd.t(), w, mu.T, sg)
    assert torch.equal(a, vc.estep_full_dev(Xd.t(), w, mu.T, sg))
    _lib.debug_force(_lib.DBG_ESTEP_FULL_NO_LISTS)
    try:
        b = vc.estep_full_dev(Xd.t(), w, mu.T, sg)
    finally:
        _lib.debug_force(0)
    assert float((a - b).abs().max() / b.abs().max()) < 1e-12
    if N * M <= 600_000:
        _check(vc.estep_full(X.T, w, mu.T, sg), co.estep_full(X, w, mu, sig), N)


def test_zero_weight_and_not_pd(vc):
    from oracle import c_oracle as co, np_oracle as npo
    w, mu, sig = npo.synth_model(9, 80, 6, lam_lo=1e-3)
    X = npo.sample_frames(9, w, mu, sig, 400, 0, 80)
    w[2] = 0.0
    w /= w.sum()
    ref = co.estep_full(X, w, mu, sig)
    got = vc.estep_full(X.T, w, mu.T, np.transpose(sig, (2, 1, 0)))
    _check(got, ref, 400)
    assert got[0][2] == 0.0 and not got[2][:, :, 2].any()
    bad = sig.copy()
    bad[1] = -bad[1]
    with pytest.raises(vc.PosDefException):
        vc.estep_full(X.T, w, mu.T, np.transpose(bad, (2, 1, 0)))


def test_device_resident_deterministic_and_additive(vc):
    """Run-to-run bit-identical and additive over frame shards (what the all-reduce relies on); the diagonal
    special case reproduces the diagonal E-step."""
    import torch
    from oracle import np_oracle as npo
    Dj, M, N = 80, 16, 20000
    w, mu, sig = npo.synth_model(78, Dj, M, lam_lo=1e-3)
    X = npo.sample_frames(78, w, mu, sig, N, 0, Dj)
    Xd = torch.from_numpy(X).cuda()                  # (N,Dj) row-major == (Dj,N) Julia image
    sg = np.transpose(sig, (2, 1, 0))
    a = vc.estep_full_dev(Xd.t(), w, mu.T, sg)
    b = vc.estep_full_dev(Xd.t(), w, mu.T, sg)
    assert torch.equal(a, b)
    h = N // 2 + 37
    s = vc.estep_full_dev(Xd[:h].t(), w, mu.T, sg) + vc.estep_full_dev(Xd[h:].t(), w, mu.T, sg)
    assert float((s - a).abs().max() / a.abs().max()) < 1e-12
    var = np.exp(np.random.default_rng(1).uniform(np.log(1e-2), 0.0, (M, Dj)))
    dsig = np.zeros((M, Dj, Dj))
    dsig[:, np.arange(Dj), np.arange(Dj)] = var
    f0, f1, f2, fl = vc.estep_full(Xd.t(), w, mu.T, np.transpose(dsig, (2, 1, 0)))
    d0, d1, d2, dl = vc.estep_diag(Xd.t(), w, mu.T, var.T)
    assert relerr(f0, d0) < TOL and relerr(f1, d1) < TOL and abs(fl - dl) < TOL * abs(dl)
    assert relerr(f2[np.arange(Dj), np.arange(Dj), :], d2) < TOL


def test_em_iterations_match_oracle_em(vc):
    """fit_full: EM on the device follows the same trajectory as EM driven by the oracle's E-step, and never
    decreases the mean log-likelihood."""
    import torch
    from oracle import np_oracle as npo
    Dj, M, N = 32, 4, 6000
    w, mu, sig = npo.synth_model(5, Dj, M, lam_lo=1e-2)
    X = npo.sample_frames(5, w, mu, sig, N, 0, Dj)
    rg = np.random.default_rng(0)
    mu0 = X[rg.choice(N, M, replace=False)].T
    sig0 = np.repeat(np.cov(X.T)[:, :, None], M, axis=2)
    w0 = np.full(M, 1.0 / M)
    Xd = torch.from_numpy(X).cuda().t()
    w1, mu1, sig1, hist = vc.fit_full(Xd, w0, mu0, sig0, n_iter=8, tol=0.0)
    assert len(hist) == 8 and all(b >= a - 1e-9 for a, b in zip(hist, hist[1:])) and hist[-1] > hist[0]
    wr, mur, sigr, ref = w0, mu0, sig0, []
    for _ in range(8):
        S0, S1, S2, ll = npo.estep_full(X, wr, mur.T, np.transpose(sigr, (2, 1, 0)))
        ref.append(ll / N)
        wr, mur, sigr = vc.mstep_full(S0, S1.T, np.transpose(S2, (2, 1, 0)))
    assert np.allclose(hist, ref, rtol=1e-8, atol=0)
    assert relerr(mu1, mur) < 1e-7 and relerr(sig1, sigr) < 1e-7


def test_device_resident_em_matches_host_em(vc):
    """EMState (parameters, M-step and Cholesky whitening on the device) follows the same trajectory as the EM loop
    whose M-step runs on the host (fit_full) and as the oracle-driven EM."""
    import torch
    from oracle import np_oracle as npo
    Dj, M, N = 32, 4, 6000
    w, mu, sig = npo.synth_model(5, Dj, M, lam_lo=1e-2)
    X = npo.sample_frames(5, w, mu, sig, N, 0, Dj)
    rg = np.random.default_rng(0)
    mu0 = X[rg.choice(N, M, replace=False)].T
    sig0 = np.repeat(np.cov(X.T)[:, :, None], M, axis=2)
    w0 = np.full(M, 1.0 / M)
    Xd = torch.from_numpy(X).cuda().t()
    em = vc.EMState(w0, mu0, sig0, min_covar=1e-7)
    hist = []
    for _ in range(8):
        st = em.estep(Xd)
        hist.append(em.mstep(st) / N)
    w1, mu1, sig1 = em.get()
    wh, muh, sigh, hh = vc.fit_full(Xd, w0, mu0, sig0, n_iter=8, tol=0.0)
    assert np.allclose(hist, hh, rtol=1e-10, atol=0)
    assert relerr(w1, wh) < 1e-9 and relerr(mu1, muh) < 1e-9 and relerr(sig1, sigh) < 1e-9
    assert np.array_equal(sig1, np.transpose(sig1, (1, 0, 2)))
    # generic (non-MFMA) dimension takes the same path through px_prep_kernel's row-major outputs
    Dg = 10
    wg, mug, sgg = npo.synth_model(6, Dg, 3, lam_lo=1e-2)
    Xg = npo.sample_frames(6, wg, mug, sgg, 2000, 0, Dg)
    e2 = vc.EMState(wg, mug.T, np.transpose(sgg, (2, 1, 0)))
    got = vc.unpack_full_stats(e2.estep(torch.from_numpy(Xg).cuda().t()).cpu().numpy(), Dg, 3)
    ref = npo.estep_full(Xg, wg, mug, sgg)
    assert relerr(got[0], ref[0]) < TOL and relerr(got[1], ref[1].T) < TOL and abs(got[3] - ref[3]) < TOL * abs(ref[3])


@pytest.mark.parametrize("Dj,M,N", [(160, 3, 3000), (112, 2, 1500)])
def test_device_resident_em_beyond_98_dimensions(vc, Dj, M, N):
    """Joint dimensions whose covariance does not fit px_prep_kernel's two LDS images (e.g. the 160-dimensional model of
    delta features) are prepared by px_prep_packed_kernel (packed lower triangle, in-place inverse): the device EM
    follows the host-M-step EM, and a not-PD covariance is reported."""
    import torch
    from oracle import np_oracle as npo
    w, mu, sig = npo.synth_model(50 + Dj, Dj, M, lam_lo=1e-2)
    X = npo.sample_frames(51 + Dj, w, mu, sig, N, 0, Dj)
    rg = np.random.default_rng(1)
    mu0 = X[rg.choice(N, M, replace=False)].T
    sig0 = np.repeat(np.cov(X.T)[:, :, None], M, axis=2)
    w0 = np.full(M, 1.0 / M)
    Xd = torch.from_numpy(X).cuda().t()
    em = vc.EMState(w0, mu0, sig0, min_covar=1e-7)
    hist = []
    for _ in range(4):
        hist.append(em.mstep(em.estep(Xd)) / N)
    w1, mu1, sig1 = em.get()
    wh, muh, sigh, hh = vc.fit_full(Xd, w0, mu0, sig0, n_iter=4, tol=0.0)
    assert np.allclose(hist, hh, rtol=1e-9, atol=0)
    assert relerr(w1, wh) < 1e-8 and relerr(mu1, muh) < 1e-8 and relerr(sig1, sigh) < 1e-8
    # first E-step against the oracle
    e2 = vc.EMState(w, mu.T, np.transpose(sig, (2, 1, 0)))
    got = vc.unpack_full_stats(e2.estep(Xd).cpu().numpy(), Dj, M)
    ref = npo.estep_full(X, w, mu, sig)
    assert relerr(got[0], ref[0]) < TOL and relerr(got[1], ref[1].T) < TOL and abs(got[3] - ref[3]) < TOL * abs(ref[3])
    bad = np.transpose(sig, (2, 1, 0)).copy()
    bad[:, :, 1] = -bad[:, :, 1]
    with pytest.raises(vc.PosDefException):           # the E-step is asynchronous: the flag is read by the M-step
        eb = vc.EMState(w, mu.T, bad)
        eb.mstep(eb.estep(Xd))


def test_train_gmm_refine_and_init(vc):
    """train_gmm (bin/train_gmm.jl:84-103): refine keeps improving a pretrained model; a k-means-initialised fit is
    monotone, beats a single Gaussian and is reproducible for a fixed seed; a non-PD start raises PosDefException."""
    import torch
    from oracle import np_oracle as npo
    Dj, M, N = 16, 4, 12000
    w, mu, sig = npo.synth_model(11, Dj, M, lam_lo=1e-1)
    mu = mu * 4.0                                            # well separated mixtures
    X = npo.sample_frames(11, w, mu, sig, N, 0, Dj)
    Xd = torch.from_numpy(X).cuda().t()
    true_ll = npo.estep_full(X, w, mu, sig)[3] / N
    r = vc.train_gmm(Xd, n_components=M, n_iter=5, refine=(w, mu.T, np.transpose(sig, (2, 1, 0))))
    assert all(b >= a - 1e-9 for a, b in zip(r["loglik"], r["loglik"][1:])) and r["loglik"][0] > true_ll - 1e-9
    a = vc.train_gmm(Xd, n_components=M, n_iter=60, n_init=2, seed=3)
    b = vc.train_gmm(Xd, n_components=M, n_iter=60, n_init=2, seed=3)
    assert np.array_equal(a["means"], b["means"]) and a["loglik"] == b["loglik"]
    assert all(y >= x - 1e-9 for x, y in zip(a["loglik"], a["loglik"][1:]))
    one = npo.estep_full(X, np.ones(1), X.mean(0)[None, :], np.cov(X.T)[None, :, :])[3] / N
    assert a["loglik"][-1] > one + 1.0 and a["loglik"][-1] > true_ll - 0.2
    assert abs(a["weights"].sum() - 1.0) < 1e-9 and a["covars"].shape == (Dj, Dj, M)
    bad = np.transpose(sig, (2, 1, 0)).copy()
    bad[:, :, 1] = -bad[:, :, 1]
    with pytest.raises(vc.PosDefException):
        vc.train_gmm(Xd, n_components=M, n_iter=2, refine=(w, mu.T, bad))


def test_empty_mixture_survives_the_mstep(vc):
    """A mixture that receives no responsibility (S0 = 0) must not poison the model: with the old sklearn.mixture.GMM
    guards (w = S0/(sum + 10 eps) + eps, mu = S1/(S0 + 10 eps), Sigma = S2/(S0 + 10 eps) - mu mu' + min_covar I) it comes
    out as weight eps, mean 0, covariance min_covar I -- finite and positive definite -- and the next E-step runs."""
    import torch
    from oracle import np_oracle as npo
    Dj, M, N = 12, 3, 4000
    w, mu, sig = npo.synth_model(91, Dj, M, lam_lo=1e-1)
    X = npo.sample_frames(92, w[:2] / w[:2].sum(), mu[:2], sig[:2], N, 0, Dj)      # frames from mixtures 0 and 1 only
    mu = mu.copy()
    mu[2] = 1.0e3                                                                  # mixture 2 is nowhere near any frame
    em = vc.EMState(w, mu.T, np.transpose(sig, (2, 1, 0)), min_covar=1e-7)
    Xd = torch.from_numpy(X).cuda().t()
    st = em.estep(Xd)
    S0 = st.cpu().numpy()[:M]
    assert S0[2] == 0.0 and abs(S0[:2].sum() - N) < 1e-6
    ll = em.mstep(st)                                                              # would raise PosDefException on NaN
    w2, mu2, sg2 = em.get()
    assert np.isfinite(ll) and np.all(np.isfinite(w2)) and np.all(np.isfinite(mu2)) and np.all(np.isfinite(sg2))
    assert w2[2] == np.finfo(np.float64).eps and np.all(mu2[:, 2] == 0.0)
    assert np.allclose(sg2[:, :, 2], 1e-7 * np.eye(Dj), rtol=0, atol=0)
    ll2 = em.mstep(em.estep(Xd))                                                   # and EM goes on
    assert np.isfinite(ll2) and ll2 >= ll - 1e-6 * abs(ll)


def test_device_em_state_rejects_dimensions_beyond_its_staging(vc):
    """vcmi_gmm_em_create stages the means of one mixture in a 256-double LDS array (M-step): larger joint dimensions are
    refused with DimensionMismatch at construction instead of overrunning it."""
    from voiceconversion_jl_amd.train import EMState
    Dj, M = 260, 2
    w = np.full(M, 1.0 / M)
    mu = np.zeros((Dj, M), order="F")
    sigma = np.asfortranarray(np.stack([np.eye(Dj)] * M, axis=2))
    with pytest.raises(vc.DimensionMismatch):
        EMState(w, mu, sigma)
