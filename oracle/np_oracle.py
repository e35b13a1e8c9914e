"""numpy/scipy restatement of the VoiceConversion.jl hot path -- TEST INFRASTRUCTURE ONLY.

Second, independent restatement of the reference algorithm (the first is oracle/vc_oracle.c).  It is
written the way the Julia source is written (explicit sparse W, scipy Cholesky / LU, explicit
posterior vector), so that agreement between the two restatements is evidence for both.  Used by
oracle/gen_golden.py to produce tests/golden/*.npz and by the CPU tests.  The product never imports it.

Arrays follow numpy convention [frame, feature] = the Julia (feature, frame) column-major memory image,
so `X.ravel()` is byte-for-byte what Julia would hand to ccall.  Indices returned are 1-based (Julia).

Pinning: DTW by test/dtw.jl:7-31, constructW by test/trajectory_gmmmap.jl:1-34 of the reference.
fvconvert / predict_proba / trajectory / E-step numerics are PARITY UNPINNED BY THE REFERENCE (its tests
assert isfinite only, and no Julia exists here to run it); they are pinned instead against third-party
implementations of the same published formulas -- sklearn GaussianMixture, scipy multivariate_normal /
logsumexp / solveh_banded, LAPACK gesv, 50-digit mpmath -- in oracle/crosscheck.py, re-run on the
committed fixtures by tests/test_oracle_thirdparty.py.  Since round 3 the GV ascent (a dense numpy /
scipy.sparse evaluation that shares no code with this file; gvgrad and one step in 50-digit mpmath) and mc2e (a
frequency-domain evaluation with numpy.fft instead of the SPTK recursions) are checked there too.
"""
import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
import scipy.sparse.linalg as spla

LOG2PI = float(np.log(2.0 * np.pi))


class GMMMap:
    """src/gmmmap.jl:57-96.  w (M,), mu (M,Dj) [= Julia (Dj,M)], sigma (M,Dj,Dj) indexed [m][col][row]."""

    def __init__(self, w, mu, sigma, swap=False):
        w = np.asarray(w, dtype=np.float64)
        mu = np.asarray(mu, dtype=np.float64)
        sigma = np.asarray(sigma, dtype=np.float64)
        M, Dj = mu.shape
        D = Dj >> 1                                               # src/gmmmap.jl:70
        S = np.transpose(sigma, (0, 2, 1))                        # S[m][row][col]
        mux, muy = mu[:, :D], mu[:, D:]                           # split_joint_gmm, src/gmmmap.jl:41-52
        Sxx, Sxy, Syx, Syy = S[:, :D, :D], S[:, :D, D:], S[:, D:, :D], S[:, D:, D:]
        if swap:                                                  # src/gmmmap.jl:74-78
            mux, muy = muy, mux
            Sxx, Syy = Syy, Sxx
            Sxy, Syx = Syx, Sxy
        self.D, self.M, self.w = D, M, w
        self.mux, self.muy = mux.copy(), muy.copy()
        self.Sxx, self.Sxy, self.Syx, self.Syy = (a.copy() for a in (Sxx, Sxy, Syx, Syy))
        # src/gmmmap.jl:33-36: Syx * Sxx^-1 (general inverse of the raw block)
        self.A = np.stack([self.Syx[m] @ np.linalg.inv(self.Sxx[m]) for m in range(M)])
        # src/gmm.jl:16-17: Hermitian() mirrors the upper triangle; MvNormal Cholesky-factorises it
        self.chol = []
        for m in range(M):
            U = np.triu(self.Sxx[m])
            Ssym = U + np.triu(self.Sxx[m], 1).T
            self.chol.append(np.linalg.cholesky(Ssym))            # raises LinAlgError if not PD
        self.logdet = np.array([2.0 * np.sum(np.log(np.diag(L))) for L in self.chol])

    def log_weighted(self, x):
        """lpr_m = logpdf(N(mux_m, Sxx_m), x) + log w_m  (src/gmm.jl:25-27)."""
        lpr = np.full(self.M, -np.inf)
        for m in range(self.M):
            if not self.w[m] > 0.0:
                continue
            z = sla.solve_triangular(self.chol[m], x - self.mux[m], lower=True)
            lpr[m] = -(self.D * LOG2PI + self.logdet[m]) / 2.0 - (z @ z) / 2.0 + np.log(self.w[m])
        return lpr

    def predict_proba1(self, x):
        lpr = self.log_weighted(x)
        u = np.max(lpr)
        lse = u + np.log(np.sum(np.exp(lpr - u)))                 # StatsFuns.logsumexp, src/gmm.jl:28
        return np.exp(lpr - lse)                                  # src/gmm.jl:29

    def predict_proba(self, X):
        return np.stack([self.predict_proba1(x) for x in X])     # (T,M) = Julia (M,T)

    def predict(self, X):
        return np.array([int(np.argmax(self.predict_proba1(x))) + 1 for x in X], dtype=np.int64)

    def fvconvert1(self, x):
        """src/gmmmap.jl:101-118."""
        E = np.stack([self.muy[m] + self.A[m] @ (x - self.mux[m]) for m in range(self.M)], axis=1)  # (D,M)
        return E @ self.predict_proba1(x)

    def fvconvert(self, X):
        return np.stack([self.fvconvert1(x) for x in X])


def fvconvert_batched_gemm(g, X, chunk=8192):
    """Same math as GMMMap.fvconvert restructured as ONE dense GEMM per frame chunk (numpy -> multithreaded BLAS): the
    'strong CPU baseline' of SURVEY 8d(ii).  All mixtures' whitening and regression rows are stacked into
    W = [U_1; A_1; U_2; A_2; ...] (2DM x D): G = X W' gives z_m = U_m x - U_m mux_m and A_m x for every m at once;
    l_m = c_m - |z_m|^2/2, softmax over m, y = sum_m p_m (A_m x + b_m)."""
    D, M = g.D, g.M
    U = [sla.solve_triangular(L, np.eye(D), lower=True) for L in g.chol]
    W = np.concatenate([np.concatenate([U[m], g.A[m]], axis=0) for m in range(M)], axis=0)          # (2DM, D)
    off = np.concatenate([np.concatenate([-U[m] @ g.mux[m], g.muy[m] - g.A[m] @ g.mux[m]]) for m in range(M)])
    cst = np.array([np.log(g.w[m]) - (D * LOG2PI + g.logdet[m]) / 2.0 if g.w[m] > 0 else -np.inf for m in range(M)])
    Wt = np.ascontiguousarray(W.T)
    Y = np.empty_like(X)
    for lo in range(0, X.shape[0], chunk):
        G = (X[lo:lo + chunk] @ Wt + off).reshape(-1, M, 2 * D)
        lpr = cst - 0.5 * np.einsum("tmd,tmd->tm", G[:, :, :D], G[:, :, :D])
        lpr -= lpr.max(axis=1, keepdims=True)
        P = np.exp(lpr)
        P /= P.sum(axis=1, keepdims=True)
        Y[lo:lo + chunk] = np.einsum("tm,tmd->td", P, G[:, :, D:])
    return Y


def vc_frames(g, fm):
    """vc(c::FrameByFrameConverter, fm), src/common.jl:7-26; fm is (T, D+1) here."""
    out = np.empty_like(fm)
    out[:, 1:] = g.fvconvert(fm[:, 1:])
    out[:, 0] = fm[:, 0]
    return out


# ---------------------------------------------------------------------------------------------- DTW
def dtw_fit(tmpl, seq, fstep=0, bstep=1):
    """fit!(d, template, sequence), src/dtw.jl:93-145.  tmpl (S,D), seq (T,D).
    Returns path (T,), costtable (T+1,S) [= Julia (S,T+1)], backpointer (T+1,S) int64; all 1-based."""
    S, T = tmpl.shape[0], seq.shape[0]
    D = tmpl.shape[1]
    cost = np.zeros((T + 1, S))
    bp = np.ones((T + 1, S), dtype=np.int64)
    cost[0] = np.arange(1, S + 1)
    bp[0] = np.arange(1, S + 1)
    for t in range(1, T + 1):
        v = seq[t - 1]
        for i in range(1, S + 1):
            o = 0.0
            for d in range(D):                                    # sequential, unfused (oracle's association)
                df = v[d] - tmpl[i - 1, d]
                o = o + df * df
            minindex = i
            mincost = cost[t - 1, i - 1] + o + 1.0
            for j in range(i - bstep, i + fstep + 1):
                if j < 1 or j > S:
                    continue
                tr = 0.0 if i == j + 1 else (1.0 if i == j else 2.0)
                c = cost[t - 1, j - 1] + o + tr
                if c < mincost:
                    mincost, minindex = c, j
            cost[t, i - 1] = mincost
            bp[t, i - 1] = minindex
    path = np.zeros(T, dtype=np.int64)
    if T > 0:
        path[T - 1] = int(np.argmin(cost[T])) + 1
        for i in range(T, 1, -1):
            path[i - 2] = bp[i, path[i - 1] - 1]
    return path, cost, bp


def align(src, tgt):
    """align(src, tgt), src/align.jl:8-35.  src (S,D), tgt (T,D) -> newtgt (S,D), path."""
    path, _, _ = dtw_fit(src, tgt, fstep=0, bstep=2)
    newtgt = np.zeros_like(src)
    for k in range(len(path)):
        newtgt[path[k] - 1] = tgt[k]
    S = src.shape[0]
    if len(path):
        have = set(path.tolist())
        for i in range(path[0], path[-1] + 1):
            if i in have:
                continue
            if 1 < i < S:
                newtgt[i - 1] = (newtgt[i - 2] + newtgt[i]) / 2.0
    return newtgt, path


# ----------------------------------------------------------- align_mcep / ParallelDataset
def freqt(c, order, alpha):
    """Frequency transformation of a cepstrum (SPTK `freqt`; MelGeneralizedCepstrums.freqt, third party -- restated
    from the published recursion, Tokuda et al.): c (m1+1,) -> (order+1,)."""
    m1 = len(c) - 1
    b = 1.0 - alpha * alpha
    g = np.zeros(order + 1)
    for i in range(m1, -1, -1):
        d = g.copy()
        g[0] = c[i] + alpha * d[0]
        if order >= 1:
            g[1] = b * d[0] + alpha * d[1]
        for j in range(2, order + 1):
            g[j] = d[j - 1] + alpha * (d[j] - g[j - 1])
    return g


def c2ir(c, length):
    """Cepstrum -> minimum-phase impulse response (SPTK `c2ir`): h[0] = exp(c[0]), h[n] = sum_k (k/n) c[k] h[n-k]."""
    h = np.zeros(length)
    h[0] = np.exp(c[0])
    for n in range(1, length):
        up = min(n, len(c) - 1)
        k = np.arange(1, up + 1)
        h[n] = np.dot(k * c[1:up + 1], h[n - k]) / n
    return h


def mc2e(mc, alpha, fftlen):
    """mc2e(mc, alpha, len), call site src/align.jl:48 (MelGeneralizedCepstrums, third party): energy of the impulse
    response of the spectrum a mel-cepstrum describes.  mc (T,D) -> (T,)."""
    return np.array([np.sum(c2ir(freqt(row, fftlen - 1, -alpha), fftlen) ** 2) for row in mc])


def align_mcep(src, tgt, alpha, fftlen, threshold=-14.0, remove_silence=True):
    """align_mcep, src/align.jl:38-55.  src (S,D), tgt (T,D) -> (src', newtgt') with silent source frames dropped."""
    newtgt, _ = align(src, tgt)
    if remove_silence:
        keep = np.log(mc2e(src, alpha, fftlen)) > threshold
        return src[keep], newtgt[keep]
    return src, newtgt


def parallel_dataset(pairs, diff=False, ignore0th=True, add_delta=False):
    """ParallelDataset(path; joint=true, ...).X, src/datasets.jl:52-98: per utterance drop row 1, push_delta, tgt - src,
    vcat, then hcat over utterances.  pairs: [(src (n,D), tgt (n,D))] -> (N, Dj)."""
    out = []
    for sx, tx in pairs:
        if ignore0th:
            sx, tx = sx[:, 1:], tx[:, 1:]
        if add_delta:
            sx, tx = push_delta(sx), push_delta(tx)
        if diff:
            tx = tx - sx
        out.append(np.concatenate([sx, tx], axis=1))
    return np.concatenate(out, axis=0)


# ---------------------------------------------------------------------------------------- trajectory
def constructW(D, T):
    """src/trajectory_gmmmap.jl:39-61, as a scipy CSC matrix (2DT x DT)."""
    W = sp.lil_matrix((2 * D * T, D * T))
    I = np.arange(D)
    for t in range(T):
        W[2 * D * t + I, t * D + I] = 1.0
        if t >= 1:
            W[2 * D * t + D + I, (t - 1) * D + I] = -0.5
        if t < T - 1:
            W[2 * D * t + D + I, (t + 1) * D + I] = 0.5
    return W.tocsc()


def gv_dataset(feature_matrices, ignore0th=True, add_delta=False):
    """GVDataset, src/datasets.jl:134-183 (the file loop replaced by in-memory matrices [T, D+1]): per utterance drop
    column 0 (ignore0th, :157-159), push_delta (:161-163), gv = var(tgt, 2) -- Julia's default corrected variance (:165);
    utterances whose variance has a NaN are skipped (:166-172).  Returns [nkept, Dout] (= Julia (Dout, nkept))."""
    cols = []
    for fm in feature_matrices:
        tgt = np.asarray(fm, dtype=np.float64)
        if ignore0th:
            tgt = tgt[:, 1:]
        if add_delta:
            tgt = push_delta(tgt)
        with np.errstate(invalid="ignore", divide="ignore"):
            gv = np.var(tgt, axis=0, ddof=1) if tgt.shape[0] > 1 else np.full(tgt.shape[1], np.nan)
        if not np.isnan(gv).any():
            cols.append(gv)
    return np.array(cols).reshape(len(cols), -1)


def push_delta(src):
    """src/datasets.jl:6-13; src (T,D) -> (T,2D); first/last frames keep the static copy in the delta rows."""
    T, D = src.shape
    out = np.concatenate([src, src], axis=1)
    for t in range(1, T - 1):
        out[t, D:] = -0.5 * src[t - 1] + 0.5 * src[t + 1]
    return out


class TrajectoryGMMMap:
    """src/trajectory_gmmmap.jl:3-37."""

    def __init__(self, g):
        self.g = g
        self.Dy = np.stack([np.linalg.inv(g.Syy[m] - g.A[m] @ g.Sxy[m]) for m in range(g.M)])   # :24-28

    def fvconvert(self, X):
        """src/trajectory_gmmmap.jl:65-110.  X (T,2D) -> (T,D); also returns mhat, Ey."""
        g = self.g
        T, D2 = X.shape
        D = D2 >> 1
        W = constructW(D, T)
        mhat = g.predict(X)                                               # :82
        Ey = np.stack([g.muy[m - 1] + g.A[m - 1] @ (X[t] - g.mux[m - 1]) for t, m in enumerate(mhat)])  # :85-89
        Dinv = sp.block_diag([sp.csc_matrix(self.Dy[m - 1]) for m in mhat], format="csc")             # :95
        WtD = W.T @ Dinv                                                  # :103
        y = spla.spsolve((WtD @ W).tocsc(), WtD @ Ey.ravel())             # :105
        return y.reshape(T, D), mhat, Ey


def variance_scaling(src, sigma2):
    """fvpostf(vs::VarianceScaling, src), src/gv.jl:10-15 (also eq. (58) initialisation, src/trajectory_gmmmap.jl:152).
    src (T,D): per-dimension  sqrt(sigma2 / var) * (x - mean) + mean  with Julia's corrected variance (1/(T-1))."""
    mu = src.mean(axis=0)
    return np.sqrt(sigma2 / src.var(axis=0, ddof=1)) * (src - mu) + mu


def gvgrad(pv, muv, y):
    """gvgrad(tgv, y), src/trajectory_gmmmap.jl:171-189.  y (T,D)."""
    T = y.shape[0]
    gv = y.var(axis=0, ddof=1)
    return -2.0 / T * (pv.T @ (gv - muv)) * (y - y.mean(axis=0))


def trajgv_fvconvert(tj, X, muv, Sigvv, epochs=100, alpha=1.0e-5):
    """fvconvert(tgv::TrajectoryGVGMMMap, X), src/trajectory_gmmmap.jl:139-168, literally: explicit sparse W and
    block-diagonal D^-1.  X (T,2D) -> (T,D)."""
    y0, mhat, Ey = tj.fvconvert(X)
    T, D = y0.shape
    pv = np.linalg.inv(Sigvv)                                             # :127
    y = variance_scaling(y0, muv)                                         # :152
    omega = 1.0 / (2 * T)
    W = constructW(D, T)
    Dinv = sp.block_diag([sp.csc_matrix(tj.Dy[m - 1]) for m in mhat], format="csc")
    WtD = (W.T @ Dinv).tocsc()
    WtDW = (WtD @ W).tocsc()
    rhs = WtD @ Ey.ravel()
    for _ in range(epochs):
        dy = omega * (-(WtDW @ y.ravel()) + rhs) + gvgrad(pv, muv, y).ravel()   # :163
        y = y + alpha * dy.reshape(T, D)                                  # :166, eq. (52)
    return y


def diffgmm(mu, sigma):
    """diffgmm(params), src/diffgmm.jl:9-25, on joint parameters: mu (M,2D), sigma (M,2D,2D) -> the joint parameters
    whose split gives (mu^x, mu^y - mu^x, Sxx, Sxy - Sxx, (Sxy - Sxx)', Sxx + Syy - Sxy - Syx)."""
    D = mu.shape[1] >> 1
    mu2, s2 = mu.copy(), sigma.copy()
    mu2[:, D:] = mu[:, D:] - mu[:, :D]
    Sxx, Sxy, Syx, Syy = sigma[:, :D, :D], sigma[:, :D, D:], sigma[:, D:, :D], sigma[:, D:, D:]
    s2[:, :D, D:] = Sxy - Sxx
    s2[:, D:, :D] = np.transpose(Sxy - Sxx, (0, 2, 1))
    s2[:, D:, D:] = Sxx + Syy - Sxy - Syx
    return mu2, s2


def vc_traj(tj, fm, L):
    """vc(c::TrajectoryConverter, fm), src/common.jl:31-63; fm (T, 2D+1) -> (T, D+1); chunks of L frames."""
    T = fm.shape[0]
    D = (fm.shape[1] - 1) >> 1
    out = np.empty((T, D + 1))
    b = 0
    while b < T:
        e = min(b + L, T)
        out[b:e, 1:] = tj.fvconvert(fm[b:e, 1:])[0]
        b = e
    out[:, 0] = fm[:, 0]
    return out


# ------------------------------------------------------------------------------------------ E-step
def estep_diag(X, w, mu, var):
    """Diagonal E-step.  X (N,Dj); w (M,); mu,var (M,Dj).  Returns S0,S1,S2,loglik."""
    Dj = X.shape[1]
    cst = np.log(w) - 0.5 * (Dj * LOG2PI + np.sum(np.log(var), axis=1))
    q = np.sum((X[:, None, :] - mu[None, :, :]) ** 2 / var[None, :, :], axis=2)       # (N,M)
    lpr = cst[None, :] - 0.5 * q
    u = lpr.max(axis=1, keepdims=True)
    lse = u[:, 0] + np.log(np.sum(np.exp(lpr - u), axis=1))
    gam = np.exp(lpr - lse[:, None])
    return gam.sum(0), gam.T @ X, gam.T @ (X * X), float(lse.sum())


def estep_full(X, w, mu, sigma):
    """Full-covariance E-step (bin/train_gmm.jl:84-103 via sklearn).  X (N,Dj); w (M,); mu (M,Dj); sigma (M,Dj,Dj)
    (symmetric).  Returns S0 (M,), S1 (M,Dj), S2 (M,Dj,Dj), loglik."""
    N, Dj = X.shape
    M = len(w)
    lpr = np.empty((N, M))
    for m in range(M):
        L = np.linalg.cholesky((sigma[m] + sigma[m].T) / 2.0)
        Z = sla.solve_triangular(L, (X - mu[m]).T, lower=True)
        lpr[:, m] = np.log(w[m]) - 0.5 * (Dj * LOG2PI + 2.0 * np.sum(np.log(np.diag(L)))) - 0.5 * np.sum(Z * Z, axis=0)
    u = lpr.max(axis=1, keepdims=True)
    lse = u[:, 0] + np.log(np.sum(np.exp(lpr - u), axis=1))
    gam = np.exp(lpr - lse[:, None])
    S2 = np.einsum("nm,ni,nj->mij", gam, X, X)
    return gam.sum(0), gam.T @ X, S2, float(lse.sum())


# ------------------------------------------------------------------- synthetic generators
# The input generators live in the neutral module synthdata.py at the repo root (bench.py and smoke() take their inputs
# from there, not from the checker); re-exported here for the golden-vector scripts and the tests.
from synthdata import sample_frames, synth_model  # noqa: E402,F401
