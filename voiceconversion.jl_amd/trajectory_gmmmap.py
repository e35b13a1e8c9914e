"""Trajectory-based (MLPG) conversion -- reference src/trajectory_gmmmap.jl:1-110, push_delta src/datasets.jl:6-13."""
import ctypes as C

import numpy as np

from . import _lib
from ._arrays import current_stream_ptr, dev_matrix, is_torch, jl_matrix
from .common import TrajectoryConverter


def constructW(D, T):
    """constructW(D, T): the (2DT x DT) sparse window matrix of src/trajectory_gmmmap.jl:39-61 -- identity for the
    static rows, -1/2 / +1/2 on the neighbouring frames for the delta rows (missing neighbours dropped).
    Host-side helper for API parity (test/trajectory_gmmmap.jl:1-34); the GPU solver applies W as a stencil and
    never materialises it."""
    import scipy.sparse as sp

    t = np.arange(T)
    d = np.arange(D)
    rows = [(2 * D * t[:, None] + d[None, :]).ravel()]
    cols = [(D * t[:, None] + d[None, :]).ravel()]
    vals = [np.ones(D * T)]
    if T >= 2:
        tt = t[1:]
        rows.append((2 * D * tt[:, None] + D + d[None, :]).ravel())
        cols.append((D * (tt[:, None] - 1) + d[None, :]).ravel())
        vals.append(np.full(D * (T - 1), -0.5))
        tt = t[:-1]
        rows.append((2 * D * tt[:, None] + D + d[None, :]).ravel())
        cols.append((D * (tt[:, None] + 1) + d[None, :]).ravel())
        vals.append(np.full(D * (T - 1), 0.5))
    return sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(2 * D * T, D * T))


def push_delta(src):
    """push_delta(src (D,T)) -> (2D,T), src/datasets.jl:6-13: delta_t = (x_{t+1} - x_{t-1})/2 for 2 <= t <= T-1;
    the first and last frame keep a copy of the static features in the delta rows (repmat artefact).
    A torch tensor on the device gives a device tensor (vcmi_push_delta_dev on the current stream): the trajectory
    converter's input is built where the features are."""
    if is_torch(src):
        import torch

        ptr, D, T, ld = dev_matrix(src, "src")
        buf = torch.empty((T, 2 * D), dtype=torch.float64, device=src.device)
        _lib.check(_lib.lib.vcmi_push_delta_dev(ptr, ld, D, T, buf.data_ptr(), 2 * D, current_stream_ptr()))
        return buf.t()
    src = jl_matrix(src, "src")
    D, T = src.shape
    out = np.empty((2 * D, T), order="F")
    _lib.check(_lib.lib.vcmi_push_delta(_lib.dptr(src), D, T, _lib.dptr(out)))
    return out


class TrajectoryGMMMap(TrajectoryConverter):
    """TrajectoryGMMMap(g::GMMMap, T) -- src/trajectory_gmmmap.jl:3-37.  `g` is a GMMMap over static+delta
    features (dim(g) = 2D).  The constructor precomputes Dy_m = inv(Sigma^yy_m - A_m Sigma^xy_m) (:24-28)."""

    def __init__(self, g, T):
        self.gmmmap = g
        h = C.c_void_p()
        _lib.check(_lib.lib.vcmi_traj_create(g._h, int(T), C.byref(h)))
        self._h = h

    def __del__(self, _destroy=_lib.lib.vcmi_traj_destroy):       # bound at definition: module globals may be gone at exit
        h = getattr(self, "_h", None)
        if h:
            _destroy(h)
            self._h = None

    def __len__(self):                      # Base.length(t) = size(W,2) / (dim/2) = T, src/trajectory_gmmmap.jl:34
        return int(_lib.lib.vcmi_traj_length(self._h))

    def _dim(self):                         # src/trajectory_gmmmap.jl:35
        return self.gmmmap._dim()

    def _ncomponents(self):                 # src/trajectory_gmmmap.jl:36
        return self.gmmmap._ncomponents()

    def _fvconvert(self, X):
        """fvconvert(tgmm, X (2D,T)) -> (D,T); src/trajectory_gmmmap.jl:65-110"""
        X = jl_matrix(X, "X")
        D2, T = X.shape
        if D2 != self._dim():               # src/trajectory_gmmmap.jl:68
            raise _lib.DimensionMismatch("Inconsistent dimentions.")
        Y = np.empty((D2 // 2, T), order="F")
        _lib.check(_lib.lib.vcmi_traj_convert(self._h, _lib.dptr(X), T, _lib.dptr(Y)))
        return Y

    def fvconvert_batch(self, Xs):
        """Batch extension: independent utterances in one launch (one workgroup per utterance)."""
        n = len(Xs)
        if n == 0:
            return []
        Xs = [jl_matrix(x, "X") for x in Xs]
        D2 = self._dim()
        for x in Xs:
            if x.shape[0] != D2:
                raise _lib.DimensionMismatch("Inconsistent dimentions.")
        T = np.array([x.shape[1] for x in Xs], dtype=np.int64)
        Ys = [np.empty((D2 // 2, int(t)), order="F") for t in T]
        dpp = C.POINTER(C.c_double) * n
        _lib.check(_lib.lib.vcmi_traj_convert_batch(self._h, n, dpp(*[_lib.dptr(x) for x in Xs]), _lib.iptr(T),
                                                    dpp(*[_lib.dptr(y) for y in Ys])))
        return Ys

    def _vc(self, fm, postfilter=None):
        """vc(c::TrajectoryConverter, fm (2D+1,T)) -> (D+1,T) in chunks of length(c) frames; src/common.jl:31-63"""
        fm = jl_matrix(fm, "fm")
        D2 = self._dim()
        if fm.shape[0] != D2 + 1:
            raise _lib.DimensionMismatch("Inconsistent dimentions.")
        T = fm.shape[1]
        out = np.empty((D2 // 2 + 1, T), order="F")
        if postfilter is None:
            _lib.check(_lib.lib.vcmi_vc_traj(self._h, _lib.dptr(fm), T, _lib.dptr(out)))
        else:       # fvpostf! (src/gv.jl:10-15) on the converted rows before the download
            if postfilter.sigma2.shape != (D2 // 2,):
                raise _lib.DimensionMismatch("sigma2 must have one entry per converted feature row")
            _lib.check(_lib.lib.vcmi_vc_traj_postf(self._h, _lib.dptr(fm), T, _lib.dptr(postfilter.sigma2), _lib.dptr(out)))
        return out


class TrajectoryGVGMMMap(TrajectoryConverter):
    """TrajectoryGVGMMMap(tgmm, mu^v, Sigma^vv) -- src/trajectory_gmmmap.jl:114-137: trajectory conversion followed
    by gradient ascent on the likelihood that includes the global variance (Toda et al. 2007, eqs. (52), (58))."""

    def __init__(self, tgmm, muv, sigmavv):
        muv = np.ascontiguousarray(np.asarray(muv, dtype=np.float64).reshape(-1))
        sigmavv = jl_matrix(sigmavv, "sigmavv")
        D = tgmm._dim() // 2
        if muv.shape != (D,) or sigmavv.shape != (D, D):
            raise _lib.DimensionMismatch("the GV statistics must have the static feature dimension")
        if np.any(muv < 0):                 # @assert sum(mu^v .< 0) == 0, src/trajectory_gmmmap.jl:124
            raise AssertionError("the GV mean must be non-negative")
        self.tgmm = tgmm
        h = C.c_void_p()
        _lib.check(_lib.lib.vcmi_trajgv_create(tgmm._h, _lib.dptr(muv), _lib.dptr(sigmavv), C.byref(h)))
        self._h = h

    def __del__(self, _destroy=_lib.lib.vcmi_trajgv_destroy):     # bound at definition: module globals may be gone at exit
        h = getattr(self, "_h", None)
        if h:
            _destroy(h)
            self._h = None

    def __len__(self):                      # src/trajectory_gmmmap.jl:132
        return len(self.tgmm)

    def _dim(self):                         # :133
        return self.tgmm._dim()

    def _ncomponents(self):                 # :134
        return self.tgmm._ncomponents()

    def _fvconvert(self, X, epochs=100, alpha=1.0e-5, verbose=False):
        """fvconvert(tgv, X (2D,T); epochs=100, alpha=1.0e-5) -> (D,T); src/trajectory_gmmmap.jl:139-168"""
        return self.fvconvert_batch([X], epochs=epochs, alpha=alpha)[0]

    def fvconvert_batch(self, Xs, epochs=100, alpha=1.0e-5):
        n = len(Xs)
        if n == 0:
            return []
        Xs = [jl_matrix(x, "X") for x in Xs]
        D2 = self._dim()
        for x in Xs:
            if x.shape[0] != D2:
                raise _lib.DimensionMismatch("Inconsistent dimentions.")
        T = np.array([x.shape[1] for x in Xs], dtype=np.int64)
        Ys = [np.empty((D2 // 2, int(t)), order="F") for t in T]
        dpp = C.POINTER(C.c_double) * n
        _lib.check(_lib.lib.vcmi_trajgv_convert_batch(self._h, n, dpp(*[_lib.dptr(x) for x in Xs]), _lib.iptr(T), int(epochs),
                                                      float(alpha), dpp(*[_lib.dptr(y) for y in Ys])))
        return Ys

    def _vc(self, fm):
        """vc(c::TrajectoryConverter, fm): chunks of length(c) frames, each converted with the default epochs / alpha;
        src/common.jl:31-63"""
        fm = jl_matrix(fm, "fm")
        D2 = self._dim()
        if fm.shape[0] != D2 + 1:
            raise _lib.DimensionMismatch("Inconsistent dimentions.")
        T, L = fm.shape[1], len(self)
        out = np.empty((D2 // 2 + 1, T), order="F")
        chunks = [np.asfortranarray(fm[1:, b:min(b + L, T)]) for b in range(0, T, L)]
        for k, y in enumerate(self.fvconvert_batch(chunks)):
            out[1:, k * L:k * L + y.shape[1]] = y
        out[0, :] = fm[0, :]
        return out
