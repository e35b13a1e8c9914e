"""GMM training with the EM state resident on the GPU -- what `gmm[:fit](dataset.X')` does in the reference's
bin/train_gmm.jl:84-103 (sklearn.mixture.GMM(n_components, covariance_type="full", n_iter, n_init, min_covar),
optionally refined from a saved model, :92-99).

The hot loop (E-step statistics, M-step, Cholesky whitening of every mixture) never leaves HBM; with a
torch.distributed process group every rank holds a shard of the frames and the only exchange per iteration is ONE
all-reduce of the packed statistics.  Initialisation (k-means on a subsample, as the old sklearn GMM's
init_params="wmc" does on the CPU) is host-side numpy: it is not part of the hot path and, being random, is not
something the reference pins either.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._arrays import current_stream_ptr, dev_matrix, jl_matrix, jl_vector
from .estep import full_stats_len, unpack_full_stats


class EMState:
    """Device-resident (w, mu, Sigma) + whitening blocks of a full-covariance GMM (vcmi_gmm_em_*)."""

    def __init__(self, w, mu, sigma, min_covar=1e-7):
        w = jl_vector(w)
        mu = jl_matrix(mu, "mu")
        sigma = np.asfortranarray(np.asarray(sigma, dtype=np.float64))
        Dj, M = mu.shape
        if sigma.ndim != 3 or sigma.shape != (Dj, Dj, M) or w.shape != (M,):
            raise _lib.DimensionMismatch(f"w {w.shape}, mu {mu.shape}, sigma {sigma.shape} are inconsistent")
        self.Dj, self.M = Dj, M
        h = C.c_void_p()
        _lib.check(_lib.lib.vcmi_gmm_em_create(Dj, M, _lib.dptr(w), _lib.dptr(mu), _lib.dptr(sigma), float(min_covar), C.byref(h)))
        self._h = h

    def __del__(self, _destroy=_lib.lib.vcmi_gmm_em_destroy):     # bound at definition: module globals may be gone at exit
        h, self._h = getattr(self, "_h", None), None
        if h:
            _destroy(h)

    def estep(self, X, out=None):
        """Local statistics of the (Dj,N) device block X -> packed device tensor [S0 | S1 | S2 | loglik]."""
        import torch

        ptr, D, N, ld = dev_matrix(X, "X")
        if D != self.Dj or (N > 1 and ld != self.Dj):
            raise _lib.DimensionMismatch("X must be a dense (Dj,N) matrix matching the model dimension")
        if out is None:
            out = torch.empty(full_stats_len(self.Dj, self.M), dtype=torch.float64, device=X.device)
        _lib.check(_lib.lib.vcmi_gmm_em_estep_dev(self._h, ptr, N, out.data_ptr(), current_stream_ptr()))
        return out

    def mstep(self, stats):
        """Parameters <- statistics (already summed over ranks); returns the log-likelihood they carry."""
        ll = np.zeros(1)
        _lib.check(_lib.lib.vcmi_gmm_em_mstep(self._h, stats.data_ptr(), current_stream_ptr(), _lib.dptr(ll)))
        return float(ll[0])

    def get(self):
        w = np.empty(self.M)
        mu = np.empty((self.Dj, self.M), order="F")
        sigma = np.empty((self.Dj, self.Dj, self.M), order="F")
        _lib.check(_lib.lib.vcmi_gmm_em_get(self._h, _lib.dptr(w), _lib.dptr(mu), _lib.dptr(sigma)))
        return w, mu, sigma


def kmeans_init(Xs, M, rng, n_iter=10):
    """k-means++ seeding and a few Lloyd iterations on a host subsample Xs (n,Dj): means for init_params='wmc'."""
    n = Xs.shape[0]
    centers = [Xs[rng.integers(n)]]
    d2 = np.sum((Xs - centers[0]) ** 2, axis=1)
    for _ in range(1, M):
        p = d2 / d2.sum() if d2.sum() > 0 else np.full(n, 1.0 / n)
        centers.append(Xs[rng.choice(n, p=p)])
        d2 = np.minimum(d2, np.sum((Xs - centers[-1]) ** 2, axis=1))
    Cn = np.asarray(centers)
    for _ in range(n_iter):
        dist = (Xs * Xs).sum(1)[:, None] - 2.0 * Xs @ Cn.T + (Cn * Cn).sum(1)[None, :]
        lab = dist.argmin(1)
        for m in range(M):
            sel = lab == m
            if sel.any():
                Cn[m] = Xs[sel].mean(0)
    return Cn


def train_gmm(X, n_components=16, n_iter=200, n_init=2, min_covar=1e-7, tol=1e-3, refine=None, seed=0, group=None,
              init_sample=50000):
    """train_gmm.jl's `gmm[:fit]`: X is this rank's (Dj,N) device-resident shard of the joint features.

    n_init random initialisations (k-means means, uniform weights, the data covariance + min_covar*I for every
    mixture -- the old sklearn GMM's init_params='wmc'), each run for at most n_iter EM iterations or until the mean
    log-likelihood per frame changes by less than tol; the best final log-likelihood wins.  refine=(w, mu, Sigma)
    starts from a pretrained model instead (bin/train_gmm.jl:92-99, init_params='').
    Returns {"weights", "means" (Dj,M), "covars" (Dj,Dj,M), "n_components", "loglik" (per-frame history), "converged"}.
    """
    import torch
    import torch.distributed as dist

    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if distributed else 0
    Dj, N = X.shape
    M = int(n_components)
    ntot = torch.tensor([float(N)], dtype=torch.float64, device=X.device)
    if distributed:
        dist.all_reduce(ntot, group=group)
    ntot = float(ntot.item())
    rng = np.random.default_rng(seed)
    best = None
    for init in range(1 if refine is not None else max(1, int(n_init))):
        if refine is not None:
            w0, mu0, sig0 = refine
        else:
            # rank 0 draws the initial model from its shard and every rank receives the same one
            pk = torch.empty(M * (1 + Dj + Dj * Dj), dtype=torch.float64, device=X.device)
            if rank == 0:
                idx = rng.choice(N, size=min(N, int(init_sample)), replace=False)
                Xs = X[:, torch.from_numpy(np.sort(idx)).to(X.device)].t().contiguous().cpu().numpy()
                mu0 = kmeans_init(Xs, M, rng).T
                cv = np.cov(Xs.T) + min_covar * np.eye(Dj)
                sig0 = np.repeat(cv[:, :, None], M, axis=2)
                w0 = np.full(M, 1.0 / M)
                pk.copy_(torch.from_numpy(np.concatenate([w0, mu0.T.ravel(), np.transpose(sig0, (2, 1, 0)).ravel()])))
            if distributed:
                dist.broadcast(pk, src=0, group=group)
            h = pk.cpu().numpy()
            w0 = h[:M].copy()
            mu0 = h[M:M + M * Dj].reshape(M, Dj).T
            sig0 = np.transpose(h[M + M * Dj:].reshape(M, Dj, Dj), (2, 1, 0))
        em = EMState(w0, mu0, sig0, min_covar)
        stats = torch.empty(full_stats_len(Dj, M), dtype=torch.float64, device=X.device)
        hist, converged = [], False
        for _ in range(int(n_iter)):
            em.estep(X, out=stats)
            if distributed:
                dist.all_reduce(stats, group=group)
            hist.append(em.mstep(stats) / ntot)
            if len(hist) > 1 and abs(hist[-1] - hist[-2]) < tol:
                converged = True
                break
        if best is None or hist[-1] > best[0]:
            best = (hist[-1], em.get(), hist, converged)
    (w, mu, sigma), hist, converged = best[1], best[2], best[3]
    return {"weights": w, "means": mu, "covars": sigma, "n_components": M, "loglik": hist, "converged": converged}


__all__ = ["EMState", "train_gmm", "kmeans_init", "unpack_full_stats"]
