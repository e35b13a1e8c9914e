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

Used by oracle/gen_golden.py (asserted before a fixture is written) and by tests/test_oracle_thirdparty.py (re-verifies
the committed fixtures on every CPU run).  This is the ceiling of what can be pinned without Julia: third-party
implementations of the same published formulas, not the reference's own run-time output.  GV ascent and mc2e have no
third-party implementation in this image and stay unpinned."""
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
