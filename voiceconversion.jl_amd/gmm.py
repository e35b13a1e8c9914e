"""Posterior helpers of the source-side GMM p(x) -- reference src/gmm.jl:24-58.

In the reference `g.px` is a Distributions.MixtureModel built by GaussianMixtureModel (src/gmm.jl:8-20);
here it is a thin view onto the GMMMap handle whose device-resident whitening blocks represent p(x)."""
import numpy as np

from . import _lib
from ._arrays import current_stream_ptr, dev_matrix, is_torch, jl_matrix, jl_vector


class GMM:
    def __init__(self, owner):
        self._owner = owner       # GMMMap keeping the libvcmi handle alive

    @property
    def _h(self):
        return self._owner._h

    def __len__(self):
        return self._owner._ncomponents()


def predict_proba(gmm, X):
    """predict_proba(gmm, x) -> (M,) / predict_proba(gmm, X (D,T)) -> (M,T); src/gmm.jl:24-41."""
    M = len(gmm)
    if is_torch(X):
        import torch

        ptr, D, T, ld = dev_matrix(X, "X")
        P = torch.empty((T, M), dtype=torch.float64, device=X.device)
        _lib.check(_lib.lib.vcmi_gmmmap_posterior_dev(gmm._h, ptr, ld, T, P.data_ptr(), current_stream_ptr()))
        return P.t()
    X = np.asarray(X)
    if X.ndim == 1:
        x = jl_vector(X)
        P = np.empty(M)
        _lib.check(_lib.lib.vcmi_gmmmap_posterior(gmm._h, _lib.dptr(x), max(len(x), 1), 1, _lib.dptr(P)) if len(x) == gmm._owner._dim()
                   else _dim_error(gmm, len(x)))
        return P
    X = jl_matrix(X, "X")
    D, T = X.shape
    if D != gmm._owner._dim():
        _dim_error(gmm, D)
    P = np.empty((M, T), order="F")
    _lib.check(_lib.lib.vcmi_gmmmap_posterior(gmm._h, _lib.dptr(X), D, T, _lib.dptr(P)))
    return P


def predict(gmm, X):
    """predict(gmm, x) -> Int / predict(gmm, X (D,T)) -> Vector{Int} (1-based, first maximum); src/gmm.jl:44-58."""
    if is_torch(X):
        import torch

        ptr, D, T, ld = dev_matrix(X, "X")
        idx = torch.empty(T, dtype=torch.int64, device=X.device)
        _lib.check(_lib.lib.vcmi_gmmmap_predict_dev(gmm._h, ptr, ld, T, idx.data_ptr(), current_stream_ptr()))
        return idx
    X = np.asarray(X)
    single = X.ndim == 1
    Xm = jl_matrix(X.reshape(-1, 1) if single else X, "X")
    D, T = Xm.shape
    if D != gmm._owner._dim():
        _dim_error(gmm, D)
    idx = np.empty(T, dtype=np.int64)
    _lib.check(_lib.lib.vcmi_gmmmap_predict(gmm._h, _lib.dptr(Xm), D, T, _lib.iptr(idx)))
    return int(idx[0]) if single else idx


def _dim_error(gmm, got):
    raise _lib.DimensionMismatch(f"Inconsistent dimentions: model dim {gmm._owner._dim()}, input {got}")
