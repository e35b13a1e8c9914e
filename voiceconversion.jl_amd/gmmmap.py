"""GMMMap: GMM-based frame-by-frame conversion -- reference src/gmmmap.jl:1-118."""
import ctypes as C

import numpy as np

from . import _lib
from ._arrays import current_stream_ptr, dev_matrix, is_torch, jl_matrix, jl_vector
from .common import FrameByFrameConverter
from .gmm import GMM


class GMMMap(FrameByFrameConverter):
    """GMMMap(weights (M,), mu (Dj,M), Sigma (Dj,Dj,M); swap=False) -- src/gmmmap.jl:57-91.

    The constructor work of the reference (split_joint_gmm :41-52, the optional source/target swap :74-78,
    A_m = Sigma^yx inv(Sigma^xx) :33-36 and the Cholesky of Hermitian(Sigma^xx) inside MvNormal,
    src/gmm.jl:16-17) runs once inside vcmi_gmmmap_create; the packed blocks stay resident in HBM."""

    def __init__(self, weights, mu, sigma, swap=False):
        w = jl_vector(weights)
        mu = jl_matrix(mu, "μ")
        sigma = np.asfortranarray(np.asarray(sigma, dtype=np.float64))
        if sigma.ndim != 3:
            raise ValueError("Σ must have shape (Dj, Dj, M)")
        Dj, M = mu.shape
        if sigma.shape != (Dj, Dj, M) or w.shape != (M,):
            raise _lib.DimensionMismatch(f"weights {w.shape}, μ {mu.shape}, Σ {sigma.shape} are inconsistent")
        h = C.c_void_p()
        _lib.check(_lib.lib.vcmi_gmmmap_create(_lib.dptr(w), _lib.dptr(mu), _lib.dptr(sigma), Dj, M, int(bool(swap)), C.byref(h)))
        self._h = h
        self._D, self._M = Dj >> 1, M
        self.px = GMM(self)                      # src/gmmmap.jl:87

    def __del__(self, _destroy=_lib.lib.vcmi_gmmmap_destroy):     # bound at definition: module globals may be gone at exit
        h = getattr(self, "_h", None)
        if h:
            _destroy(h)
            self._h = None

    def __len__(self):                           # Base.length(g) = 1, src/gmmmap.jl:93
        return 1

    def _dim(self):                              # src/gmmmap.jl:94
        return _lib.lib.vcmi_gmmmap_dim(self._h)

    def _ncomponents(self):                      # src/gmmmap.jl:95
        return _lib.lib.vcmi_gmmmap_ncomponents(self._h)

    @property
    def SyxSxxinv(self):                         # g.params.ΣʸˣΣˣˣ⁻¹ (D,D,M), src/gmmmap.jl:21
        A = np.empty((self._D, self._D, self._M), order="F")
        _lib.check(_lib.lib.vcmi_gmmmap_get_A(self._h, _lib.dptr(A)))
        return A

    def set_kernel(self, which):
        """0 auto, 1 generic VALU kernel, 2 MFMA tile kernel (used by the parity tests to cover both)."""
        _lib.check(_lib.lib.vcmi_gmmmap_set_kernel(self._h, int(which)))

    def set_prune(self, nats):
        """Posterior pruning threshold of fvconvert in nats (default 46: terms below 1e-20 of the sum are not evaluated);
        float('inf') evaluates every mixture for every frame (include/vcmi.h: vcmi_gmmmap_set_prune)."""
        _lib.check(_lib.lib.vcmi_gmmmap_set_prune(self._h, float(nats)))

    def prune_stats(self, enable=True):
        """(tile, mixture) regressions evaluated since the counter was last enabled; then restart (enable) or switch it off."""
        n = C.c_int64(0)
        _lib.check(_lib.lib.vcmi_gmmmap_prune_stats(self._h, 1 if enable else 0, C.byref(n)))
        return int(n.value)

    def convert_plan(self):
        """(MFMA instructions issued since prune_stats(True), loop shape 0 dense / 1 broad / 2 peaked / -1 no tile kernel,
        model_active_frac, model_undecided_frac) -- include/vcmi.h: vcmi_gmmmap_convert_plan."""
        n, shape, frac, und = C.c_int64(0), C.c_int(0), C.c_double(0.0), C.c_double(0.0)
        _lib.check(_lib.lib.vcmi_gmmmap_convert_plan(self._h, C.byref(n), C.byref(shape), C.byref(frac), C.byref(und)))
        return int(n.value), int(shape.value), float(frac.value), float(und.value)

    def _fvconvert(self, x, out=None):
        if is_torch(x):
            import torch

            ptr, D, T, ld = dev_matrix(x, "X")
            if D != self._D:
                raise _lib.DimensionMismatch("Inconsistent dimentions.")
            if out is None:
                out = torch.empty((T, D), dtype=torch.float64, device=x.device).t()
            optr, oD, oT, old = dev_matrix(out, "out")
            if (oD, oT) != (D, T):
                raise _lib.DimensionMismatch("output shape does not match input")
            _lib.check(_lib.lib.vcmi_gmmmap_convert_dev(self._h, ptr, ld, T, optr, old, current_stream_ptr()))
            return out
        x = np.asarray(x)
        if x.ndim == 1:                          # the reference's signature: one frame, src/gmmmap.jl:101
            v = jl_vector(x)
            if len(v) != self._D:                # src/gmmmap.jl:102
                raise _lib.DimensionMismatch("Inconsistent dimentions.")
            y = np.empty(self._D)
            _lib.check(_lib.lib.vcmi_gmmmap_convert(self._h, _lib.dptr(v), self._D, 1, _lib.dptr(y), self._D))
            return y
        X = jl_matrix(x, "X")
        D, T = X.shape
        if D != self._D:
            raise _lib.DimensionMismatch("Inconsistent dimentions.")
        if out is None:
            Y = np.empty((D, T), order="F")
        else:
            Y = out
            if Y.shape != (D, T) or Y.dtype != np.float64 or not Y.flags.f_contiguous:
                raise _lib.DimensionMismatch("out must be a Fortran-ordered float64 (D,T) array")
        _lib.check(_lib.lib.vcmi_gmmmap_convert(self._h, _lib.dptr(X), D, T, _lib.dptr(Y), D))
        return Y

    def _vc(self, fm, postfilter=None):          # src/common.jl:7-26 (+ fvpostf! of src/gv.jl:10-15 before the download)
        fm = jl_matrix(fm, "fm")
        if fm.shape[0] != self._D + 1:
            raise _lib.DimensionMismatch("Inconsistent dimentions.")
        out = np.empty_like(fm, order="F")
        if postfilter is None:
            _lib.check(_lib.lib.vcmi_vc_frames(self._h, _lib.dptr(fm), fm.shape[1], _lib.dptr(out)))
        else:
            if postfilter.sigma2.shape != (self._D,):
                raise _lib.DimensionMismatch("sigma2 must have one entry per converted feature row")
            _lib.check(_lib.lib.vcmi_vc_frames_postf(self._h, _lib.dptr(fm), fm.shape[1], _lib.dptr(postfilter.sigma2), _lib.dptr(out)))
        return out
