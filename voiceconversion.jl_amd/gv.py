"""GV post filter and the differential-GMM parameter transform -- reference src/gv.jl:1-21, src/diffgmm.jl:9-25."""
import numpy as np

from . import _lib
from ._arrays import current_stream_ptr, dev_matrix, is_torch, jl_matrix


class VarianceScaling:
    """VarianceScaling(sigma2) -- src/gv.jl:6-8"""

    def __init__(self, sigma2):
        self.sigma2 = np.ascontiguousarray(np.asarray(sigma2, dtype=np.float64).reshape(-1))


def fvpostf(vs, src):
    """fvpostf(vs, src (D,T)): per row sqrt(sigma2 / var) * (x - mean) + mean with Julia's corrected variance;
    src/gv.jl:10-21.  A torch tensor on the device is filtered where it is (a new device tensor is returned)."""
    if is_torch(src):
        out = src.clone(memory_format=__import__("torch").preserve_format)
        return fvpostf_(vs, out)
    src = jl_matrix(src, "src")
    D, T = src.shape
    if vs.sigma2.shape != (D,):
        raise _lib.DimensionMismatch("sigma2 must have one entry per feature row")
    out = np.empty((D, T), order="F")
    _lib.check(_lib.lib.vcmi_variance_scaling(_lib.dptr(src), D, T, _lib.dptr(vs.sigma2), _lib.dptr(out)))
    return out


def fvpostf_(vs, src):
    """fvpostf!(vs, src): in place; src/gv.jl:10-15.  Device tensors ((D,T), unit stride along D: e.g. the rows 2..D+1 of a
    converted (D+1,T) matrix, `out[1:]`) are filtered in HBM on the current stream (vcmi_variance_scaling_dev); host arrays
    must be Fortran-ordered float64."""
    if is_torch(src):
        ptr, D, T, ld = dev_matrix(src, "src")
        if vs.sigma2.shape != (D,):
            raise _lib.DimensionMismatch("sigma2 must have one entry per feature row")
        _lib.check(_lib.lib.vcmi_variance_scaling_dev(ptr, ld, D, T, _lib.dptr(vs.sigma2), ptr, ld, current_stream_ptr()))
        return src
    src[...] = fvpostf(vs, src)
    return src


def diffgmm(mu, sigma):
    """diffgmm(params) on joint parameters mu (2D,M), sigma (2D,2D,M) -> (mu', sigma') such that
    GMMMap(w, mu', sigma') is the differential converter of [Kobayashi 2014]; src/diffgmm.jl:9-25"""
    mu = jl_matrix(mu, "mu")
    sigma = np.asfortranarray(np.asarray(sigma, dtype=np.float64))
    Dj, M = mu.shape
    if sigma.shape != (Dj, Dj, M):
        raise _lib.DimensionMismatch("sigma must be (2D,2D,M)")
    mo, so = np.empty_like(mu, order="F"), np.empty_like(sigma, order="F")
    _lib.check(_lib.lib.vcmi_diffgmm(_lib.dptr(mu), _lib.dptr(sigma), Dj, M, _lib.dptr(mo), _lib.dptr(so)))
    return mo, so
