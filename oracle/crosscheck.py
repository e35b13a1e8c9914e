"""Third-party cross-checks of the floating-point oracle -- TEST INFRASTRUCTURE ONLY.

The reference pins nothing numeric about fvconvert / predict_proba / the trajectory solve / the E-step (its tests only
assert `isfinite`, test/vc.jl:26,50,72) and no Julia exists in this image, so the oracle's two restatements
(vc_oracle.c, np_oracle.py) were, after round 1, checked only against each other -- both written by the same builder.
This module checks the GOLDEN VECTORS (tests/golden/*.npz) against code the builder did not write:

  posterior / predict   src/gmm.jl:24-30,44-47     sklearn.mixture.GaussianMixture with injected parameters
                                                   (predict_proba / predict) AND scipy.stats.multivariate_normal.logpdf
                                                   + scipy.special.logsumexp
  E[y|x]                src/gmmmap.jl:33-36,109-117 numpy.linalg.solve on the RAW covariance blocks (LAPACK gesv; the
                                                   reference uses the LU inverse `^-1` of the raw block too)
  the same, 50 digits   (16 frames)                mpmath: Cholesky / inverse / log / exp in 50-digit arithmetic -- bounds
                                                   the rounding error of the golden vectors themselves
  trajectory solve      src/trajectory_gmmmap.jl:95-105  scipy.linalg.solveh_banded (LAPACK pbsv) on the band of
                                                   W'D^-1W assembled from scipy.sparse products, D^-1 from numpy.linalg.inv
  diag / full E-step    bin/train_gmm.jl:84-103    sklearn GaussianMixture('diag' / 'full') responsibilities and
                                                   score_samples (the successor of the sklearn.mixture.GMM the
                                                   reference calls; same log-density mathematics)

  mc2e                  src/align.jl:48            frequency-domain evaluation (round 3): H(w) = exp(sum_m c(m) z~^-m) on a
                                                   65536-point grid, numpy.fft.ifft -> impulse response, energy of its first
                                                   `fftlen` samples; Parseval for the whole response.  No freqt / c2ir
                                                   recursion (what the oracle restates from SPTK): check_mc2e
  GV ascent             src/trajectory_gmmmap.jl:139-189  (round 3) dense numpy evaluation with W materialised through
                                                   scipy.sparse, numpy.linalg.solve / inv, numpy.var(ddof=1): check_gv;
                                                   `gvgrad` against central differences of log N(v(y); mu_v, Sigma_vv) in
                                                   50-digit mpmath (it is (T-1)/T times that gradient: the reference writes
                                                   2/T where d var/dy gives 2/(T-1)); one ascent step in 50-digit mpmath

Used by oracle/gen_golden.py (asserted before a fixture is written) and by tests/test_oracle_thirdparty.py (re-verifies
the committed fixtures on every CPU run).  This is the ceiling of what can be pinned without Julia: third-party
implementations of the same published formulas, not the reference's own run-time output."""
import numpy as np


def _jl_cov(sig_m):
    """[col][row] buffer of one (D,D) Julia matrix -> ordinary numpy matrix [row][col]."""
    return np.asarray(sig_m).T


def split_blocks(mu, sig, swap=False):
    """src/gmmmap.jl:41-52 (+ swap :74-78) on numpy [m][d] / [m][col][row] buffers -> per-mixture row-major blocks."""
    M, Dj = mu.shape
    D = Dj // 2
    S = np.stack([_jl_cov(sig[m]) for m in range(M)])
    xs, ys = (slice(D, Dj), slice(0, D)) if swap else (slice(0, D), slice(D, Dj))
    return mu[:, xs], mu[:, ys], S[:, xs, xs], S[:, xs, ys], S[:, ys, xs], S[:, ys, ys]


def hermitian_upper(S):
    """Array(Hermitian(S)): the upper triangle mirrored (src/gmm.jl:16)."""
    U = np.triu(S)
    return U + np.triu(S, 1).T


def sklearn_gmm(w, means, covs, covariance_type="full"):
    from sklearn.mixture import GaussianMixture
    from sklearn.mixture._gaussian_mixture import _compute_precision_cholesky

    gm = GaussianMixture(n_components=len(w), covariance_type=covariance_type)
    gm.weights_ = np.asarray(w, dtype=np.float64)
    gm.means_ = np.asarray(means, dtype=np.float64)
    gm.covariances_ = np.asarray(covs, dtype=np.float64)
    gm.precisions_cholesky_ = _compute_precision_cholesky(gm.covariances_, covariance_type)
    return gm


def posterior_scipy(w, mux, Sxx_h, X):
    from scipy.special import logsumexp
    from scipy.stats import multivariate_normal

    lpr = np.stack([multivariate_normal.logpdf(X, mean=mux[m], cov=Sxx_h[m], allow_singular=False) + np.log(w[m])
                    for m in range(len(w))], axis=1)
    lpr = lpr.reshape(len(X), len(w))
    return np.exp(lpr - logsumexp(lpr, axis=1, keepdims=True)), lpr


def check_conversion(w, mu, sig, X, Y, P, idx, swap=False, tol=1e-9):
    """Golden fvconvert / predict_proba / predict outputs against sklearn, scipy and numpy.linalg.solve.
    Returns the observed maxima."""
    mux, muy, Sxx, Sxy, Syx, Syy = split_blocks(mu, sig, swap)
    Sh = np.stack([hermitian_upper(s) for s in Sxx])
    gm = sklearn_gmm(w, mux, Sh)
    P_sk = gm.predict_proba(X)
    P_sp, lpr = posterior_scipy(w, mux, Sh, X)
    out = {"posterior_vs_sklearn": float(np.max(np.abs(P - P_sk))), "posterior_vs_scipy": float(np.max(np.abs(P - P_sp)))}
    assert out["posterior_vs_sklearn"] < tol and out["posterior_vs_scipy"] < tol, out
    assert np.array_equal(idx, gm.predict(X) + 1) and np.array_equal(idx, np.argmax(lpr, axis=1) + 1)
    # E[y|x] = sum_m p_m (mu^y_m + Syx_m Sxx_m^-1 (x - mu^x_m)) with the RAW (unsymmetrised) Sxx block, src/gmmmap.jl:35
    Yc = np.zeros_like(Y)
    for m in range(len(w)):
        Z = np.linalg.solve(Sxx[m], (X - mux[m]).T)            # (D,T)
        Yc += P_sp[:, m:m + 1] * (muy[m] + (Syx[m] @ Z).T)
    err = np.linalg.norm(Y - Yc, axis=1) / np.linalg.norm(Yc, axis=1)
    out["fvconvert_vs_numpy_solve"] = float(err.max())
    assert out["fvconvert_vs_numpy_solve"] < tol, out
    return out


def check_conversion_mpmath(w, mu, sig, X, Y, P, swap=False, frames=16, dps=50, tol=1e-9):
    """The same quantities for a few frames in 50-digit arithmetic: what the golden vectors are off by in absolute
    terms (rounding of an FP64 evaluation with cond(Sxx) ~ 1e6-1e7)."""
    import mpmath as mp

    mux, muy, Sxx, Sxy, Syx, Syy = split_blocks(mu, sig, swap)
    M, D = mux.shape
    frames = min(frames, len(X))
    with mp.workdps(dps):
        tomp = lambda a: mp.matrix(np.asarray(a).tolist())  # noqa: E731
        log2pi = mp.log(2 * mp.pi)
        lpr = [[None] * M for _ in range(frames)]
        E = [[None] * M for _ in range(frames)]
        for m in range(M):
            Sh = tomp(hermitian_upper(Sxx[m]))
            L = mp.cholesky(Sh)
            Li = mp.inverse(L)
            logdet = 2 * sum(mp.log(L[i, i]) for i in range(D))
            A = tomp(Syx[m]) * mp.inverse(tomp(Sxx[m]))          # raw block, src/gmmmap.jl:35
            for t in range(frames):
                d = tomp((X[t] - 0.0).tolist()) - tomp(mux[m].tolist())
                z = Li * d
                q = sum(z[i] * z[i] for i in range(D))
                lpr[t][m] = mp.log(mp.mpf(float(w[m]))) - (D * log2pi + logdet + q) / 2
                E[t][m] = tomp(muy[m].tolist()) + A * d
        errP = errY = mp.mpf(0)
        for t in range(frames):
            u = max(lpr[t])
            s = sum(mp.exp(l - u) for l in lpr[t])
            p = [mp.exp(l - u) / s for l in lpr[t]]
            y = [sum(p[m] * E[t][m][i] for m in range(M)) for i in range(D)]
            errP = max(errP, max(abs(p[m] - mp.mpf(float(P[t, m]))) for m in range(M)))
            num = mp.sqrt(sum((y[i] - mp.mpf(float(Y[t, i]))) ** 2 for i in range(D)))
            errY = max(errY, num / mp.sqrt(sum(v * v for v in y)))
        out = {"posterior_vs_mpmath": float(errP), "fvconvert_vs_mpmath": float(errY), "frames": frames, "digits": dps}
    assert out["posterior_vs_mpmath"] < tol and out["fvconvert_vs_mpmath"] < tol, out
    return out


def check_trajectory(w, mu, sig, X, Y, mhat, Ey, tol=1e-6):
    """Golden trajectory conversion (src/trajectory_gmmmap.jl:65-110) against scipy.linalg.solveh_banded: P = W'D^-1W
    assembled with scipy.sparse from the explicit W and numpy.linalg.inv blocks, its band extracted, LAPACK pbsv."""
    import scipy.sparse as sp
    from scipy.linalg import solveh_banded

    T, D2 = X.shape
    D = D2 // 2
    mux, muy, Sxx, Sxy, Syx, Syy = split_blocks(mu, sig, False)
    Sh = np.stack([hermitian_upper(s) for s in Sxx])
    gm = sklearn_gmm(w, mux, Sh)
    mh = gm.predict(X) + 1                                        # src/trajectory_gmmmap.jl:82
    assert np.array_equal(mh, mhat)
    E = np.empty((T, D2))
    Dinv = []
    for t in range(T):
        m = mh[t] - 1
        A = Syx[m] @ np.linalg.inv(Sxx[m])                        # src/gmmmap.jl:35
        E[t] = muy[m] + A @ (X[t] - mux[m])                       # :85-89
        Dinv.append(np.linalg.inv(Syy[m] - A @ Sxy[m]))           # :24-28
    assert np.max(np.abs(E - Ey)) < 1e-9 * np.max(np.abs(Ey))
    # W (2DT x DT), src/trajectory_gmmmap.jl:39-61
    rows, cols, vals = [], [], []
    for t in range(T):
        for d in range(D):
            rows.append(2 * D * t + d); cols.append(D * t + d); vals.append(1.0)
            if t > 0:
                rows.append(2 * D * t + D + d); cols.append(D * (t - 1) + d); vals.append(-0.5)
            if t < T - 1:
                rows.append(2 * D * t + D + d); cols.append(D * (t + 1) + d); vals.append(0.5)
    W = sp.csc_matrix((vals, (rows, cols)), shape=(2 * D * T, D * T))
    Dblk = sp.block_diag(Dinv, format="csc")
    Pm = (W.T @ Dblk @ W).toarray()
    r = W.T @ (Dblk @ E.reshape(-1))
    Pm = 0.5 * (Pm + Pm.T)
    bw = 3 * D - 1                                                # scalar half-bandwidth of the block-pentadiagonal matrix
    assert np.max(np.abs(np.triu(Pm, bw + 1))) == 0.0
    ab = np.zeros((bw + 1, D * T))
    for k in range(bw + 1):                                       # upper form: ab[bw - k, j] = P[j - k, j]
        ab[bw - k, k:] = np.diagonal(Pm, k)
    y = solveh_banded(ab, r, lower=False).reshape(T, D)
    err = float(np.max(np.abs(y - Y)) / np.max(np.abs(y)))
    assert err < tol, err
    return {"trajectory_vs_solveh_banded": err}


def check_estep_diag(X, w, mu, var, S0, S1, S2, loglik, tol=1e-9):
    gm = sklearn_gmm(w, mu, var, "diag")
    R = gm.predict_proba(X)
    ll = float(gm.score_samples(X).sum())
    out = {"S0": float(np.max(np.abs(R.sum(0) - S0)) / np.max(np.abs(S0))),
           "S1": float(np.max(np.abs(R.T @ X - S1)) / np.max(np.abs(S1))),
           "S2": float(np.max(np.abs(R.T @ (X * X) - S2)) / np.max(np.abs(S2))),
           "loglik": abs(ll - float(loglik)) / abs(ll)}
    assert max(out.values()) < tol, out
    return out


def check_estep_full(X, w, mu, sigma, S0, S1, S2, loglik, tol=1e-9):
    """sigma, S2: [m][col][row] buffers (symmetric up to rounding; sklearn reads them as ordinary matrices)."""
    covs = np.stack([hermitian_upper(_jl_cov(s)) for s in sigma])
    gm = sklearn_gmm(w, mu, covs, "full")
    R = gm.predict_proba(X)
    ll = float(gm.score_samples(X).sum())
    S2c = np.einsum("nm,ni,nj->mij", R, X, X)
    out = {"S0": float(np.max(np.abs(R.sum(0) - S0)) / np.max(np.abs(S0))),
           "S1": float(np.max(np.abs(R.T @ X - S1)) / np.max(np.abs(S1))),
           "S2": float(np.max(np.abs(S2c - np.transpose(S2, (0, 2, 1)))) / np.max(np.abs(S2))),
           "loglik": abs(ll - float(loglik)) / abs(ll)}
    assert max(out.values()) < tol, out
    return out


# ---------------------------------------------------------------------------------------------------------------
# mc2e (src/align.jl:48; MelGeneralizedCepstrums, third party) -- frequency-domain evaluation, no freqt / c2ir recursion
# ---------------------------------------------------------------------------------------------------------------
def mc2e_frequency_domain(mc, alpha, fftlen, nfft=1 << 16):
    """Energy of the first `fftlen` samples of the impulse response of the spectrum a mel-cepstrum describes, evaluated
    WITHOUT the SPTK recursions the oracle restates (freqt: mel -> linear cepstrum; c2ir: cepstrum -> impulse response).

    The mel-cepstral model (Tokuda et al.; mgcep) is  H(z) = exp sum_m c~(m) z~^-m  with the first-order all-pass
    z~^-1 = (z^-1 - alpha) / (1 - alpha z^-1).  On the unit circle, on a dense grid of `nfft` frequencies:
        H(w_k) = exp( sum_m c~(m) * ((e^-jw_k - alpha) / (1 - alpha e^-jw_k))^m )
    H is minimum phase (exp of a causal series), so its impulse response is causal and numpy.fft.ifft(H) returns it (time
    aliasing: samples beyond nfft = 65536, far below 1e-16 here).  freqt(mc, fftlen-1, -alpha) truncates the linear
    cepstrum to order fftlen-1, which cannot change h[0..fftlen-1] (h[n] depends on c[1..n] only), so
        mc2e(mc, alpha, fftlen) = sum_{n < fftlen} h[n]^2      -- up to rounding, no truncation term.
    Also returns the whole-response energy by Parseval, mean_k |H(w_k)|^2: the difference is the energy of the response
    beyond `fftlen` samples, i.e. what the reference's fixed length leaves out (reported, not asserted as equal).
    mc (T,D) -> (e_truncated (T,), e_parseval (T,))"""
    mc = np.atleast_2d(np.asarray(mc, dtype=np.float64))
    w = 2.0 * np.pi * np.arange(nfft) / nfft
    zi = np.exp(-1j * w)
    zt = (zi - alpha) / (1.0 - alpha * zi)                       # z~^-1 on the unit circle
    powers = np.ones((mc.shape[1], nfft), dtype=np.complex128)
    for m in range(1, mc.shape[1]):
        powers[m] = powers[m - 1] * zt
    H = np.exp(mc.astype(np.complex128) @ powers)                # (T, nfft)
    h = np.fft.ifft(H, axis=1)
    assert np.max(np.abs(h.imag)) < 1e-9 * np.max(np.abs(h.real))                       # real response
    assert np.max(np.abs(h.real[:, nfft // 2:])) < 1e-12 * np.max(np.abs(h.real))       # causal: nothing at negative times
    e_trunc = np.sum(h.real[:, :fftlen] ** 2, axis=1)
    e_full = np.mean(np.abs(H) ** 2, axis=1)
    return e_trunc, e_full


def check_mc2e(mc, alpha, fftlen, energy, tol=1e-10):
    """Golden energies (oracle: SPTK freqt + c2ir recursions) against the frequency-domain evaluation above."""
    e_trunc, e_full = mc2e_frequency_domain(mc, alpha, fftlen)
    err = float(np.max(np.abs(e_trunc - energy) / energy))
    tail = float(np.max(np.abs(e_full - e_trunc) / e_full))
    assert err < tol, err
    assert np.all(e_full >= e_trunc * (1 - 1e-12))               # the truncated response cannot hold more energy
    return {"mc2e_vs_frequency_domain": err, "energy_beyond_fftlen_relative": tail}


# ---------------------------------------------------------------------------------------------------------------
# GV ascent (src/trajectory_gmmmap.jl:139-189): dense numpy / scipy.sparse evaluation and 50-digit mpmath
# ---------------------------------------------------------------------------------------------------------------
def _trajectory_system(w, mu, sig, X):
    """(W, D^-1 blocks, E, mhat) of src/trajectory_gmmmap.jl:82-96 from third-party pieces only: sklearn predict,
    numpy.linalg.inv, an explicit scipy.sparse W (src/trajectory_gmmmap.jl:39-61)."""
    import scipy.sparse as sp

    T, D2 = X.shape
    D = D2 // 2
    mux, muy, Sxx, Sxy, Syx, Syy = split_blocks(mu, sig, False)
    gm = sklearn_gmm(w, mux, np.stack([hermitian_upper(s) for s in Sxx]))
    mh = gm.predict(X)
    E = np.empty((T, D2))
    Dinv = []
    for t in range(T):
        m = mh[t]
        A = Syx[m] @ np.linalg.inv(Sxx[m])
        E[t] = muy[m] + A @ (X[t] - mux[m])
        Dinv.append(np.linalg.inv(Syy[m] - A @ Sxy[m]))
    W = sp.lil_matrix((2 * D * T, D * T))
    for t in range(T):
        for d in range(D):
            W[2 * D * t + d, D * t + d] = 1.0
            if t > 0:
                W[2 * D * t + D + d, D * (t - 1) + d] = -0.5
            if t < T - 1:
                W[2 * D * t + D + d, D * (t + 1) + d] = 0.5
    return W.tocsc(), Dinv, E, mh


def gv_ascent_dense(w, mu, sig, X, muv, Sigvv, epochs, alpha):
    """fvconvert(tgv, X) of src/trajectory_gmmmap.jl:139-168 with every operator materialised as a DENSE numpy matrix
    (W from scipy.sparse, .toarray()), the initial trajectory from numpy.linalg.solve on the dense normal equations,
    numpy.var(ddof=1) for Julia's `var`, numpy.linalg.inv for p_v.  Shares no code with oracle/np_oracle.py or
    oracle/vc_oracle.c (stencil / banded Cholesky there; dense LU here).  X (T,2D) -> y (T,D), and the first gradient."""
    import scipy.sparse as sp

    W, Dinv, E, _ = _trajectory_system(w, mu, sig, X)
    T, D2 = X.shape
    D = D2 // 2
    Wd = W.toarray()
    Dd = sp.block_diag(Dinv).toarray()
    WtD = Wd.T @ Dd
    P = WtD @ Wd
    r = WtD @ E.reshape(-1)
    y = np.linalg.solve(P, r).reshape(T, D)                                   # :146 (dense LU instead of sparse `\`)
    m0 = y.mean(axis=0)
    y = np.sqrt(muv / np.var(y, axis=0, ddof=1)) * (y - m0) + m0              # :152, eq. (58)
    pv = np.linalg.inv(Sigvv)                                                 # :127
    omega = 1.0 / (2 * T)                                                     # :154
    first = None
    for _ in range(epochs):
        gv = np.var(y, axis=0, ddof=1)                                        # :174
        g = (-2.0 / T) * (pv.T @ (gv - muv)) * (y - y.mean(axis=0))           # :181-183
        dy = omega * (-(P @ y.reshape(-1)) + r) + g.reshape(-1)               # :163
        if first is None:
            first = dy.reshape(T, D).copy()
        y = y + alpha * dy.reshape(T, D)                                      # :166, eq. (52)
    return y, first


def check_gv(w, mu, sig, X, muv, Sigvv, Y_gv, epochs=100, alpha=1.0e-5, tol=1e-6):
    """Golden GV trajectory against the dense evaluation.  Tolerance: the ascent starts from the trajectory solve, whose
    two independent solutions differ by ~cond(P) eps (1e-11 here); the ascent itself is a contraction at this step."""
    y, _ = gv_ascent_dense(w, mu, sig, X, muv, Sigvv, epochs, alpha)
    err = float(np.max(np.abs(y - Y_gv)) / np.max(np.abs(Y_gv)))
    assert err < tol, err
    return {"gv_ascent_vs_dense_numpy": err, "epochs": epochs}


def check_gvgrad_is_the_gv_likelihood_gradient(y, muv, Sigvv, gvgrad_fn, rel=1e-6):
    """What `gvgrad` is, independently of any restatement: the reference's expression (src/trajectory_gmmmap.jl:181-183)
    equals (T-1)/T times the gradient of  log N(v(y); mu_v, Sigma_vv)  with v = Julia's corrected variance -- the
    reference writes 2/T where differentiating var(y,2) gives 2/(T-1).  Verified by central differences of the
    log-density at 50 digits (mpmath), on a small y: pins the formula AND documents the quirk.
    gvgrad_fn(pv, muv, y) -> (T,D): the implementation under test (an oracle restatement, or the HIP path's output)."""
    import mpmath as mp

    T, D = y.shape
    pv = np.linalg.inv(Sigvv)
    got = gvgrad_fn(pv, muv, y)
    with mp.workdps(50):
        Pv = mp.matrix(Sigvv.tolist()) ** -1

        def logdens(Y):
            v = []
            for d in range(D):
                col = [Y[t][d] for t in range(T)]
                mean = sum(col) / T
                v.append(sum((c - mean) ** 2 for c in col) / (T - 1))
            dv = mp.matrix([v[d] - mp.mpf(float(muv[d])) for d in range(D)])
            return -(dv.T * Pv * dv)[0] / 2

        Y0 = [[mp.mpf(float(y[t, d])) for d in range(D)] for t in range(T)]
        h = mp.mpf(10) ** -20
        worst = 0.0
        for t in range(T):
            for d in range(D):
                Yp = [row[:] for row in Y0]
                Ym = [row[:] for row in Y0]
                Yp[t][d] += h
                Ym[t][d] -= h
                num = (logdens(Yp) - logdens(Ym)) / (2 * h)
                want = num * (T - 1) / T
                worst = max(worst, float(abs(mp.mpf(float(got[t, d])) - want) / (abs(want) + mp.mpf(10) ** -30)))
    assert worst < rel, worst
    return {"gvgrad_vs_mpmath_gradient_times_(T-1)/T": worst}


def check_gv_step_mpmath(w, mu, sig, X, muv, Sigvv, step_fn, dps=50, tol=1e-9):
    """One ascent step (src/trajectory_gmmmap.jl:163-166) from the eq. (58) initial trajectory, evaluated in 50-digit
    mpmath with dense W'D^-1W, against `step_fn(y0) -> y1` (the implementation under test started from the same y0).
    Small cases only (dense mpmath products)."""
    import mpmath as mp

    W, Dinv, E, _ = _trajectory_system(w, mu, sig, X)
    T, D2 = X.shape
    D = D2 // 2
    y_dense, _ = gv_ascent_dense(w, mu, sig, X, muv, Sigvv, 0, 0.0)            # eq. (58) initial value (double)
    y1 = step_fn(y_dense)
    with mp.workdps(dps):
        import scipy.sparse as sp
        Wm = mp.matrix(W.toarray().tolist())
        Dm = mp.matrix(sp.block_diag(Dinv).toarray().tolist())
        WtD = Wm.T * Dm
        yv = mp.matrix([[mp.mpf(float(v))] for v in y_dense.reshape(-1)])
        Ev = mp.matrix([[mp.mpf(float(v))] for v in E.reshape(-1)])
        lin = -(WtD * (Wm * yv)) + WtD * Ev
        Pv = mp.matrix(Sigvv.tolist()) ** -1
        means = [sum(yv[D * t + d] for t in range(T)) / T for d in range(D)]
        gv = [sum((yv[D * t + d] - means[d]) ** 2 for t in range(T)) / (T - 1) for d in range(D)]
        dv = mp.matrix([gv[d] - mp.mpf(float(muv[d])) for d in range(D)])
        pg = Pv.T * dv
        omega = mp.mpf(1) / (2 * T)
        alpha = mp.mpf(float(step_fn.alpha))
        err = mp.mpf(0)
        scale = max(abs(v) for v in yv)
        for t in range(T):
            for d in range(D):
                g = mp.mpf(-2) / T * pg[d] * (yv[D * t + d] - means[d])
                want = yv[D * t + d] + alpha * (omega * lin[D * t + d] + g)
                err = max(err, abs(mp.mpf(float(y1[t, d])) - want))
        out = float(err / scale)
    assert out < tol, out
    return {"gv_step_vs_mpmath": out, "digits": dps}
