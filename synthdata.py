"""synthdata.py -- synthetic workloads of SURVEY 8(d): the model and frame generators bench.py, smoke() and the tests share.

Neutral ground: it belongs neither to the product package nor to the checker (oracle/), imports only numpy, and is the
same code on this container and on the GPU box, so nothing large has to be committed.
"""
import numpy as np


def synth_model(seed, Dj, M, lam_lo=1e-5, lam_hi=1.0):
    """Synthetic joint GMM: Dirichlet(2) weights, N(0,1) means, covariances Q diag(lam) Q' with lam
    log-uniform in [lam_lo, lam_hi] (cond ~ fixture's 1e6-1e7), exactly symmetrised."""
    rng = np.random.default_rng(seed)
    w = rng.dirichlet(2.0 * np.ones(M))
    mu = rng.standard_normal((M, Dj))
    sig = np.empty((M, Dj, Dj))
    for m in range(M):
        Q, _ = np.linalg.qr(rng.standard_normal((Dj, Dj)))
        lam = np.exp(rng.uniform(np.log(lam_lo), np.log(lam_hi), Dj))
        S = (Q * lam) @ Q.T
        sig[m] = (S + S.T) / 2.0
    return w, mu, sig


def sample_frames(seed, w, mu, sig, T, lo, hi):
    """Frames drawn from the model's own marginal over dims [lo,hi): component ~ w, then mu + L z."""
    rng = np.random.default_rng(seed)
    M = len(w)
    comp = rng.choice(M, size=T, p=w / w.sum())
    Ls = [np.linalg.cholesky((sig[m][lo:hi, lo:hi] + sig[m][lo:hi, lo:hi].T) / 2.0) for m in range(M)]
    Z = rng.standard_normal((T, hi - lo))
    X = np.empty((T, hi - lo))
    for m in range(M):
        sel = comp == m
        X[sel] = mu[m, lo:hi] + Z[sel] @ Ls[m].T
    return X
