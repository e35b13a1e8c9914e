"""ctypes loader for the C oracle (oracle/vc_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Array convention as in np_oracle.py: numpy [frame, feature] C-contiguous == Julia (feature, frame).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VCORACLE_LIB: an alternative build of the same sources (the ASan/UBSan library of tests/test_sanitizers.py)
_SO = os.environ.get("VCORACLE_LIB") or os.path.join(_HERE, "_build", "libvcoracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("vc_oracle.c", "vc_oracle_gemm.c", "vc_oracle.h", "vc_oracle_internal.h", "Makefile")]
    if os.environ.get("VCORACLE_LIB"):
        return _SO
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        # idle OpenMP workers sleep instead of spinning: the multi-threaded helpers are called from Python loops, with long
        # gaps between parallel regions, on hosts whose cores are shared (must be set before libgomp initialises)
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        L = C.CDLL(_SO)
        L.vco_gmmmap_new.restype = C.c_void_p
        L.vco_gmmmap_new.argtypes = [_dp, _dp, _dp, C.c_int, C.c_int, C.c_int]
        L.vco_gmmmap_free.argtypes = [C.c_void_p]
        L.vco_gmmmap_dim.argtypes = [C.c_void_p]
        L.vco_gmmmap_ncomponents.argtypes = [C.c_void_p]
        L.vco_gmmmap_get_A.argtypes = [C.c_void_p, _dp]
        L.vco_fvconvert.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.vco_fvconvert_batch.argtypes = [C.c_void_p, _dp, C.c_int64, _dp]
        L.vco_fvconvert_batch_mt.argtypes = [C.c_void_p, _dp, C.c_int64, _dp]
        L.vco_fvconvert_batch_gemm.argtypes = [C.c_void_p, _dp, C.c_int64, _dp]
        L.vco_logdens.argtypes = [C.c_void_p, _dp, C.c_int64, _dp]
        L.vco_predict_proba.argtypes = [C.c_void_p, _dp, C.c_int64, _dp]
        L.vco_predict.argtypes = [C.c_void_p, _dp, C.c_int64, _ip]
        L.vco_vc_frames.argtypes = [C.c_void_p, _dp, C.c_int64, _dp]
        L.vco_dtw_fit.argtypes = [_dp, C.c_int64, _dp, C.c_int64, C.c_int, C.c_int, C.c_int, _dp, _ip, _ip]
        L.vco_align.argtypes = [_dp, C.c_int64, _dp, C.c_int64, C.c_int, _dp, _ip]
        L.vco_constructW.restype = C.c_int64
        L.vco_constructW.argtypes = [C.c_int, C.c_int64, _ip, _ip, _dp]
        L.vco_push_delta.argtypes = [_dp, C.c_int, C.c_int64, _dp]
        L.vco_traj_new.restype = C.c_void_p
        L.vco_traj_new.argtypes = [C.c_void_p]
        L.vco_traj_free.argtypes = [C.c_void_p]
        L.vco_traj_fvconvert.argtypes = [C.c_void_p, _dp, C.c_int64, _dp, _ip, _dp]
        L.vco_vc_traj.argtypes = [C.c_void_p, _dp, C.c_int64, C.c_int64, _dp]
        L.vco_estep_diag.argtypes = [_dp, C.c_int64, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.vco_variance_scaling.argtypes = [_dp, C.c_int, C.c_int64, _dp, _dp]
        L.vco_trajgv_fvconvert.argtypes = [C.c_void_p, _dp, C.c_int64, _dp, _dp, C.c_int, C.c_double, _dp]
        L.vco_diffgmm.argtypes = [_dp, _dp, C.c_int, C.c_int, _dp, _dp]
        L.vco_mc2e.argtypes = [_dp, C.c_int, C.c_int64, C.c_double, C.c_int, _dp]
        L.vco_align_mcep.restype = C.c_int64
        L.vco_align_mcep.argtypes = [_dp, C.c_int64, _dp, C.c_int64, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, _dp, _dp]
        L.vco_joint_features.argtypes = [_dp, _dp, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, _dp]
        L.vco_estep_full.argtypes = [_dp, C.c_int64, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class GMMMap:
    def __init__(self, w, mu, sigma, swap=False):
        w, mu, sigma = _f64(w), _f64(mu), _f64(sigma)
        M, Dj = mu.shape
        self._h = lib().vco_gmmmap_new(_d(w), _d(mu), _d(sigma), Dj, M, int(swap))
        if not self._h:
            raise np.linalg.LinAlgError("covariance block not positive definite")
        self.D, self.M = Dj >> 1, M

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:      # (module globals may already be gone at interpreter exit)
            lib().vco_gmmmap_free(self._h)
            self._h = None

    @property
    def A(self):
        A = np.empty((self.M, self.D, self.D))
        lib().vco_gmmmap_get_A(self._h, _d(A))
        return np.transpose(A, (0, 2, 1))      # -> [m][row][col]

    def fvconvert(self, X):
        X = _f64(X)
        Y = np.empty_like(X)
        lib().vco_fvconvert_batch(self._h, _d(X), X.shape[0], _d(Y))
        return Y

    def fvconvert_mt(self, X):
        """all host cores (OpenMP); returns (Y, threads)"""
        X = _f64(X)
        Y = np.empty_like(X)
        n = lib().vco_fvconvert_batch_mt(self._h, _d(X), X.shape[0], _d(Y))
        return Y, int(n)

    def fvconvert_gemm(self, X):
        """GEMM-structured, all host cores (oracle/vc_oracle_gemm.c); returns (Y, threads)"""
        X = _f64(X)
        Y = np.empty_like(X)
        n = lib().vco_fvconvert_batch_gemm(self._h, _d(X), X.shape[0], _d(Y))
        if n <= 0:
            raise MemoryError("vco_fvconvert_batch_gemm")
        return Y, int(n)

    def logdens(self, X):
        """lpr of src/gmm.jl:25-27: (T, M) log w_m + logpdf_m(x)"""
        X = _f64(X)
        L = np.empty((X.shape[0], self.M))
        lib().vco_logdens(self._h, _d(X), X.shape[0], _d(L))
        return L

    def predict_proba(self, X):
        X = _f64(X)
        P = np.empty((X.shape[0], self.M))
        lib().vco_predict_proba(self._h, _d(X), X.shape[0], _d(P))
        return P

    def predict(self, X):
        X = _f64(X)
        idx = np.empty(X.shape[0], dtype=np.int64)
        lib().vco_predict(self._h, _d(X), X.shape[0], _i(idx))
        return idx

    def vc(self, fm):
        fm = _f64(fm)
        out = np.empty_like(fm)
        lib().vco_vc_frames(self._h, _d(fm), fm.shape[0], _d(out))
        return out


def dtw_fit(tmpl, seq, fstep=0, bstep=1, tables=True):
    tmpl, seq = _f64(tmpl), _f64(seq)
    S, D = tmpl.shape
    T = seq.shape[0]
    path = np.empty(T, dtype=np.int64)
    if tables:
        cost = np.empty((T + 1, S))
        bp = np.empty((T + 1, S), dtype=np.int64)
        lib().vco_dtw_fit(_d(tmpl), S, _d(seq), T, D, fstep, bstep, _d(cost), _i(bp), _i(path))
        return path, cost, bp
    lib().vco_dtw_fit(_d(tmpl), S, _d(seq), T, D, fstep, bstep, None, None, _i(path))
    return path


def align(src, tgt):
    src, tgt = _f64(src), _f64(tgt)
    S, D = src.shape
    T = tgt.shape[0]
    newtgt = np.empty_like(src)
    path = np.empty(T, dtype=np.int64)
    lib().vco_align(_d(src), S, _d(tgt), T, D, _d(newtgt), _i(path))
    return newtgt, path


def constructW(D, T):
    n = lib().vco_constructW(D, T, None, None, None)
    r = np.empty(n, dtype=np.int64)
    c = np.empty(n, dtype=np.int64)
    v = np.empty(n)
    lib().vco_constructW(D, T, _i(r), _i(c), _d(v))
    return r, c, v


def push_delta(src):
    src = _f64(src)
    T, D = src.shape
    out = np.empty((T, 2 * D))
    lib().vco_push_delta(_d(src), D, T, _d(out))
    return out


class TrajectoryGMMMap:
    def __init__(self, g):
        self.g = g
        self._h = lib().vco_traj_new(g._h)
        if not self._h:
            raise np.linalg.LinAlgError("singular conditional covariance")

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:      # (module globals may already be gone at interpreter exit)
            lib().vco_traj_free(self._h)
            self._h = None

    def fvconvert(self, X):
        X = _f64(X)
        T, D2 = X.shape
        Y = np.empty((T, D2 >> 1))
        mhat = np.empty(T, dtype=np.int64)
        Ey = np.empty((T, D2))
        rc = lib().vco_traj_fvconvert(self._h, _d(X), T, _d(Y), _i(mhat), _d(Ey))
        if rc:
            raise np.linalg.LinAlgError("normal matrix not positive definite")
        return Y, mhat, Ey

    def fvconvert_gv(self, X, muv, Sigvv, epochs=100, alpha=1.0e-5):
        X, muv, Sigvv = _f64(X), _f64(muv), _f64(Sigvv)
        T, D2 = X.shape
        Y = np.empty((T, D2 >> 1))
        rc = lib().vco_trajgv_fvconvert(self._h, _d(X), T, _d(muv), _d(Sigvv), int(epochs), float(alpha), _d(Y))
        if rc:
            raise np.linalg.LinAlgError("trajectory GV conversion failed")
        return Y

    def vc(self, fm, L):
        fm = _f64(fm)
        T = fm.shape[0]
        D = (fm.shape[1] - 1) >> 1
        out = np.empty((T, D + 1))
        rc = lib().vco_vc_traj(self._h, _d(fm), T, L, _d(out))
        if rc:
            raise np.linalg.LinAlgError("normal matrix not positive definite")
        return out


def estep_diag(X, w, mu, var):
    X, w, mu, var = _f64(X), _f64(w), _f64(mu), _f64(var)
    N, Dj = X.shape
    M = len(w)
    S0 = np.empty(M)
    S1 = np.empty((M, Dj))
    S2 = np.empty((M, Dj))
    ll = C.c_double(0.0)
    lib().vco_estep_diag(_d(X), N, Dj, M, _d(w), _d(mu), _d(var), _d(S0), _d(S1), _d(S2), C.byref(ll))
    return S0, S1, S2, ll.value


def estep_full(X, w, mu, sigma):
    """X (N,Dj); w (M,); mu (M,Dj); sigma (M,Dj,Dj) [m][col][row].  Returns S0 (M,), S1 (M,Dj), S2 (M,Dj,Dj) [m][col][row], loglik."""
    X, w, mu, sigma = _f64(X), _f64(w), _f64(mu), _f64(sigma)
    N, Dj = X.shape
    M = len(w)
    S0 = np.empty(M)
    S1 = np.empty((M, Dj))
    S2 = np.empty((M, Dj, Dj))
    ll = C.c_double(0.0)
    rc = lib().vco_estep_full(_d(X), N, Dj, M, _d(w), _d(mu), _d(sigma), _d(S0), _d(S1), _d(S2), C.byref(ll))
    if rc:
        raise np.linalg.LinAlgError("covariance not positive definite")
    return S0, S1, S2, ll.value


def variance_scaling(src, sigma2):
    src, sigma2 = _f64(src), _f64(sigma2)
    T, D = src.shape
    out = np.empty_like(src)
    lib().vco_variance_scaling(_d(src), D, T, _d(sigma2), _d(out))
    return out


def diffgmm(mu, sigma):
    """mu (M,2D), sigma (M,2D,2D) [m][col][row] -> transformed joint parameters in the same layout."""
    mu, sigma = _f64(mu), _f64(sigma)
    M, Dj = mu.shape
    mo, so = np.empty_like(mu), np.empty_like(sigma)
    lib().vco_diffgmm(_d(mu), _d(sigma), Dj, M, _d(mo), _d(so))
    return mo, so


def mc2e(mc, alpha, fftlen):
    mc = _f64(mc)
    T, D = mc.shape
    e = np.empty(T)
    lib().vco_mc2e(_d(mc), D, T, float(alpha), int(fftlen), _d(e))
    return e


def align_mcep(src, tgt, alpha, fftlen, threshold=-14.0, remove_silence=True):
    src, tgt = _f64(src), _f64(tgt)
    S, D = src.shape
    so, to = np.empty_like(src), np.empty_like(src)
    k = lib().vco_align_mcep(_d(src), S, _d(tgt), tgt.shape[0], D, float(alpha), int(fftlen), float(threshold),
                             int(bool(remove_silence)), _d(so), _d(to))
    return so[:k].copy(), to[:k].copy()


def joint_features(src, tgt, ignore0th=True, add_delta=False, diff=False):
    src, tgt = _f64(src), _f64(tgt)
    n, D = src.shape
    Dj = 2 * (D - int(ignore0th)) * (2 if add_delta else 1)
    out = np.empty((n, Dj))
    lib().vco_joint_features(_d(src), _d(tgt), D, n, int(ignore0th), int(add_delta), int(diff), _d(out))
    return out
