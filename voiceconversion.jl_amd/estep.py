"""GMM training E-step -- the arithmetic behind `gmm[:fit](dataset.X')` in the reference's
bin/train_gmm.jl:103 (scikit-learn through PyCall; joint features from src/datasets.jl:52-77), for the
diagonal configuration BASELINE.json names and for the full-covariance one the reference script actually
builds (bin/train_gmm.jl:84-89, covariance_type="full").

estep_diag(X (Dj,N), w (M,), mu (Dj,M), var (Dj,M)) -> S0 (M,), S1 (Dj,M), S2 (Dj,M), loglik.
With a torch.distributed process group (one process per GPU) every rank passes its own shard of frames and
the sufficient statistics are summed with ONE all-reduce over RCCL (`estep_diag_allreduce`)."""
import numpy as np

from . import _lib
from ._arrays import current_stream_ptr, dev_matrix, is_torch, jl_matrix, jl_vector


def stats_len(Dj, M):
    return int(_lib.lib.vcmi_estep_stats_len(int(Dj), int(M)))


ESTEP_AUTO, ESTEP_HARD, ESTEP_SOFT = 0, 1, 2


def estep_set_path(path):
    """Which path the diagonal E-step of the calling thread takes from 65536 frames on (include/vcmi.h): ESTEP_AUTO decides per
    call from a sample of the call's own frames, ESTEP_HARD / ESTEP_SOFT pin the hard-assignment / the one-kernel path."""
    _lib.check(_lib.lib.vcmi_estep_set_path(int(path)))


def estep_get_path():
    import ctypes as C
    v = C.c_int(0)
    _lib.check(_lib.lib.vcmi_estep_get_path(C.byref(v)))
    return int(v.value)


def _params(w, mu, var):
    w = jl_vector(w)
    mu = jl_matrix(mu, "mu")
    var = jl_matrix(var, "var")
    Dj, M = mu.shape
    if var.shape != (Dj, M) or w.shape != (M,):
        raise _lib.DimensionMismatch(f"w {w.shape}, mu {mu.shape}, var {var.shape} are inconsistent")
    return w, mu, var, Dj, M


def unpack_stats(stats, Dj, M):
    """[S0 | S1 | S2 | loglik] -> (S0 (M,), S1 (Dj,M), S2 (Dj,M), loglik); works for numpy and torch buffers."""
    S0 = stats[:M]
    S1 = stats[M:M + M * Dj].reshape(M, Dj).T
    S2 = stats[M + M * Dj:M + 2 * M * Dj].reshape(M, Dj).T
    return S0, S1, S2, stats[M + 2 * M * Dj]


def estep_diag_dev(X, w, mu, var, out=None):
    """Device-resident E-step: X is a (Dj,N) torch tensor (unit stride along Dj, dense: ld == Dj).
    Returns the packed statistics as a device tensor of stats_len(Dj,M) doubles."""
    import torch

    w, mu, var, Dj, M = _params(w, mu, var)
    ptr, D, N, ld = dev_matrix(X, "X")
    if D != Dj or (N > 1 and ld != Dj):
        raise _lib.DimensionMismatch("X must be a dense (Dj,N) matrix matching the model dimension")
    if out is None:
        out = torch.empty(stats_len(Dj, M), dtype=torch.float64, device=X.device)
    _lib.check(_lib.lib.vcmi_estep_diag_dev(ptr, N, Dj, M, _lib.dptr(w), _lib.dptr(mu), _lib.dptr(var), out.data_ptr(),
                                            current_stream_ptr()))
    return out


def estep_diag(X, w, mu, var):
    if is_torch(X):
        w_, mu_, var_, Dj, M = _params(w, mu, var)
        st = estep_diag_dev(X, w_, mu_, var_).cpu().numpy()
        S0, S1, S2, ll = unpack_stats(st, Dj, M)
        return S0.copy(), np.asfortranarray(S1), np.asfortranarray(S2), float(ll)
    w, mu, var, Dj, M = _params(w, mu, var)
    X = jl_matrix(X, "X")
    if X.shape[0] != Dj:
        raise _lib.DimensionMismatch("X must be (Dj,N)")
    N = X.shape[1]
    S0 = np.empty(M)
    S1 = np.empty((Dj, M), order="F")
    S2 = np.empty((Dj, M), order="F")
    ll = np.zeros(1)
    _lib.check(_lib.lib.vcmi_estep_diag(_lib.dptr(X), N, Dj, M, _lib.dptr(w), _lib.dptr(mu), _lib.dptr(var),
                                        _lib.dptr(S0), _lib.dptr(S1), _lib.dptr(S2), _lib.dptr(ll)))
    return S0, S1, S2, float(ll[0])


def estep_diag_allreduce(X_shard, w, mu, var, group=None):
    """Multi-GPU E-step: local statistics of this rank's frame shard, then one all-reduce(sum) of the packed
    M(1+2Dj)+1 doubles (RCCL when the group's backend is "nccl").  Returns the packed global statistics."""
    from .dist import allreduce_sum_

    return allreduce_sum_(estep_diag_dev(X_shard, w, mu, var), group)


def mstep_diag(S0, S1, S2, min_covar=1e-7):
    """M-step of the old sklearn.mixture.GMM the reference configures (bin/train_gmm.jl:84-89, min_covar :18),
    from the E-step statistics (informative -- the E-step is what the parity tests pin)."""
    eps = np.finfo(np.float64).eps
    S0 = np.asarray(S0)
    w = S0 / (S0.sum() + 10 * eps) + eps
    inv = 1.0 / (S0 + 10 * eps)
    mu = S1 * inv[None, :]
    var = S2 * inv[None, :] - 2 * mu * S1 * inv[None, :] + mu * mu + min_covar
    return w, mu, var


# ----------------------------------------------------------------------------------------- full covariance
def full_stats_len(Dj, M):
    return int(_lib.lib.vcmi_estep_full_stats_len(int(Dj), int(M)))


def _params_full(w, mu, sigma):
    w = jl_vector(w)
    mu = jl_matrix(mu, "mu")
    sigma = np.asfortranarray(np.asarray(sigma, dtype=np.float64))
    Dj, M = mu.shape
    if sigma.ndim != 3 or sigma.shape != (Dj, Dj, M) or w.shape != (M,):
        raise _lib.DimensionMismatch(f"w {w.shape}, mu {mu.shape}, sigma {sigma.shape} are inconsistent")
    return w, mu, sigma, Dj, M


def unpack_full_stats(stats, Dj, M):
    """[S0 | S1 | S2 | loglik] -> (S0 (M,), S1 (Dj,M), S2 (Dj,Dj,M), loglik); numpy or torch buffers."""
    S0 = stats[:M]
    S1 = stats[M:M + M * Dj].reshape(M, Dj).T
    S2 = stats[M + M * Dj:M + M * Dj + M * Dj * Dj].reshape(M, Dj, Dj).permute(2, 1, 0) if is_torch(stats) else \
        stats[M + M * Dj:M + M * Dj + M * Dj * Dj].reshape(M, Dj, Dj).transpose(2, 1, 0)
    return S0, S1, S2, stats[M + M * Dj + M * Dj * Dj]


def estep_full_dev(X, w, mu, sigma, out=None):
    """Device-resident full-covariance E-step: X is a dense (Dj,N) torch tensor.  Returns the packed statistics
    as a device tensor of full_stats_len(Dj,M) doubles."""
    import torch

    w, mu, sigma, Dj, M = _params_full(w, mu, sigma)
    ptr, D, N, ld = dev_matrix(X, "X")
    if D != Dj or (N > 1 and ld != Dj):
        raise _lib.DimensionMismatch("X must be a dense (Dj,N) matrix matching the model dimension")
    if out is None:
        out = torch.empty(full_stats_len(Dj, M), dtype=torch.float64, device=X.device)
    _lib.check(_lib.lib.vcmi_estep_full_dev(ptr, N, Dj, M, _lib.dptr(w), _lib.dptr(mu), _lib.dptr(sigma),
                                            out.data_ptr(), current_stream_ptr()))
    return out


def estep_full(X, w, mu, sigma):
    """estep_full(X (Dj,N), w (M,), mu (Dj,M), sigma (Dj,Dj,M)) -> S0 (M,), S1 (Dj,M), S2 (Dj,Dj,M), loglik."""
    w, mu, sigma, Dj, M = _params_full(w, mu, sigma)
    if is_torch(X):
        st = estep_full_dev(X, w, mu, sigma).cpu().numpy()
        S0, S1, S2, ll = unpack_full_stats(st, Dj, M)
        return S0.copy(), np.asfortranarray(S1), np.asfortranarray(S2), float(ll)
    X = jl_matrix(X, "X")
    if X.shape[0] != Dj:
        raise _lib.DimensionMismatch("X must be (Dj,N)")
    S0 = np.empty(M)
    S1 = np.empty((Dj, M), order="F")
    S2 = np.empty((Dj, Dj, M), order="F")
    ll = np.zeros(1)
    _lib.check(_lib.lib.vcmi_estep_full(_lib.dptr(X), X.shape[1], Dj, M, _lib.dptr(w), _lib.dptr(mu), _lib.dptr(sigma),
                                        _lib.dptr(S0), _lib.dptr(S1), _lib.dptr(S2), _lib.dptr(ll)))
    return S0, S1, S2, float(ll[0])


def estep_full_allreduce(X_shard, w, mu, sigma, group=None):
    """Multi-GPU full-covariance E-step: local statistics of this rank's frames, then ONE all-reduce(sum) of
    the packed M(1+Dj+Dj^2)+1 doubles."""
    from .dist import allreduce_sum_

    return allreduce_sum_(estep_full_dev(X_shard, w, mu, sigma), group)


def mstep_full(S0, S1, S2, min_covar=1e-7):
    """M-step of sklearn.mixture.GMM(covariance_type="full") as the reference configures it
    (bin/train_gmm.jl:84-89; min_covar :18) from the E-step statistics: w = S0/sum, mu = S1/S0,
    sigma = S2/S0 - mu mu' + min_covar I."""
    eps = np.finfo(np.float64).eps
    S0 = np.asarray(S0)
    Dj = S1.shape[0]
    w = S0 / (S0.sum() + 10 * eps) + eps
    inv = 1.0 / (S0 + 10 * eps)
    mu = S1 * inv[None, :]
    sigma = S2 * inv[None, None, :] - mu[:, None, :] * mu[None, :, :] + min_covar * np.eye(Dj)[:, :, None]
    return w, mu, np.asfortranarray(sigma)


def fit_full(X, w, mu, sigma, n_iter=200, tol=1e-3, min_covar=1e-7, group=None):
    """EM driver in the shape of bin/train_gmm.jl:84-103 (n_iter :16 default 200, min_covar :18); stops when
    the mean per-frame log-likelihood changes by less than tol (the old sklearn.mixture.GMM rule).  X is this
    rank's (Dj,N) device shard; with a process group the statistics are all-reduced each iteration, so every
    rank holds the same model.  Returns (w, mu, sigma, mean_loglik_history)."""
    Dj, M = np.asarray(mu).shape
    hist = []
    for _ in range(n_iter):
        st = estep_full_allreduce(X, w, mu, sigma, group) if group is not None or _dist_ready() else \
            estep_full_dev(X, w, mu, sigma)
        S0, S1, S2, ll = unpack_full_stats(st.cpu().numpy(), Dj, M)
        hist.append(float(ll) / float(S0.sum()))
        if len(hist) > 1 and abs(hist[-1] - hist[-2]) < tol:
            break
        w, mu, sigma = mstep_full(S0, S1, S2, min_covar)
    return w, mu, sigma, hist


def _dist_ready():
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized()
