"""align_mcep and the joint training matrix -- reference src/align.jl:38-55, src/datasets.jl:52-98."""
import ctypes as C

import numpy as np

from . import _lib
from ._arrays import jl_matrix


def mc2e(mc, alpha, fftlen):
    """mc2e(mc (D,T), alpha, len) -> (T,): frame energy of a mel-cepstrum (MelGeneralizedCepstrums; src/align.jl:48)"""
    mc = jl_matrix(mc, "mc")
    D, T = mc.shape
    e = np.empty(T)
    _lib.check(_lib.lib.vcmi_mc2e(_lib.dptr(mc), D, T, float(alpha), int(fftlen), _lib.dptr(e)))
    return e


def align_mcep(src, tgt, alpha, fftlen, threshold=-14.0, remove_silence=True):
    """align_mcep(src (D,S), tgt (D,T), alpha, fftlen; threshold, remove_silence) -> (src', newtgt'); src/align.jl:38-55"""
    src, tgt = jl_matrix(src, "src"), jl_matrix(tgt, "tgt")
    if src.shape[0] != tgt.shape[0]:
        raise _lib.DimensionMismatch("order of feature vector between source and target must be equal")
    D, S = src.shape
    so, to = np.empty((D, S), order="F"), np.empty((D, S), order="F")
    k = np.zeros(1, dtype=np.int64)
    _lib.check(_lib.lib.vcmi_align_mcep(_lib.dptr(src), S, _lib.dptr(tgt), tgt.shape[1], D, float(alpha), int(fftlen),
                                        float(threshold), int(bool(remove_silence)), _lib.dptr(so), _lib.dptr(to), _lib.iptr(k)))
    n = int(k[0])
    return np.asfortranarray(so[:, :n]), np.asfortranarray(to[:, :n])


class ParallelDataset:
    """ParallelDataset(...; joint=true) built on the device from in-memory (src, tgt) mel-cepstrum pairs instead of
    *_parallel.jld files (src/datasets.jl:52-98): `X` is a (Dj, N) float64 torch tensor on the HIP device, ready for
    the E-step (train_gmm).  align=True runs align_mcep on every pair first (src/align.jl:38-55), so raw parallel
    utterances go in and the training matrix comes out without a host round trip."""

    def __init__(self, pairs, diff=False, joint=True, ignore0th=True, add_delta=False, align=True, alpha=0.41, fftlen=512,
                 threshold=-14.0, remove_silence=True, nmax=100, standarize=False):
        if standarize:
            # src/datasets.jl:105-108 writes `(X - Xmean) / Xstd` with a (D,N) and a (D,1) operand: in the reference's
            # Julia 0.5 that is a DimensionMismatch, not a standardisation -- the branch cannot have been used
            raise _lib.DimensionMismatch("standarize=true throws in the reference (src/datasets.jl:105-108); not provided")
        import torch

        pairs = list(pairs)[:nmax]                    # count >= nmax && break, src/datasets.jl:95-96
        srcs = [jl_matrix(s, "src") for s, _ in pairs]
        tgts = [jl_matrix(t, "tgt") for _, t in pairs]
        n = len(srcs)
        if n == 0:
            raise ValueError("no training pairs")
        D = srcs[0].shape[0]
        for s, t in zip(srcs, tgts):
            if s.shape[0] != D or t.shape[0] != D:
                raise _lib.DimensionMismatch("all feature matrices must share the feature dimension")
        S = np.array([s.shape[1] for s in srcs], dtype=np.int64)
        T = np.array([t.shape[1] for t in tgts], dtype=np.int64)
        Dj = 2 * (D - int(bool(ignore0th))) * (2 if add_delta else 1)
        cap = int(S.sum())
        buf = torch.empty(cap * Dj, dtype=torch.float64, device="cuda")
        nfr = np.zeros(1, dtype=np.int64)
        counts = np.zeros(n, dtype=np.int64)
        dpp = C.POINTER(C.c_double) * n
        _lib.check(_lib.lib.vcmi_parallel_dataset_dev(n, dpp(*[_lib.dptr(s) for s in srcs]), _lib.iptr(S),
                                                      dpp(*[_lib.dptr(t) for t in tgts]), _lib.iptr(T), D, int(bool(align)),
                                                      float(alpha), int(fftlen), float(threshold), int(bool(remove_silence)),
                                                      int(bool(ignore0th)), int(bool(add_delta)), int(bool(diff)),
                                                      buf.data_ptr(), cap, _lib.iptr(nfr), _lib.iptr(counts)))
        N = int(nfr[0])
        self.X = buf[:N * Dj].view(N, Dj).t()         # (Dj, N), unit stride along Dj
        self.Y = None
        if not joint:                                 # X = XY[1:D,:], Y = XY[D+1:end,:], src/datasets.jl:92-96 (views)
            self.X, self.Y = self.X[:Dj // 2], self.X[Dj // 2:]
        self.counts = counts
        self.diff = bool(diff)
        self.totalframes, self.totalphrases = N, n

    def __len__(self):
        return self.totalframes


class GVDataset:
    """GVDataset(path; ignore0th, add_delta, nmax) from in-memory feature matrices -- src/datasets.jl:134-183 (the loop
    over `*.jld` files is the caller's): `X` is the (Dout, n) matrix of per-utterance variances `var(tgt, 2)` that
    bin/train_gv.jl fits the GV model on; utterances whose variance is NaN (a single frame) are skipped as in the
    reference."""

    def __init__(self, feature_matrices, ignore0th=True, add_delta=False, nmax=100):
        fms = [jl_matrix(f, "feature_matrix") for f in list(feature_matrices)[:nmax]]   # totalphrases >= nmax && break
        n = len(fms)
        if n == 0:
            self.X = np.zeros((0, 0))                 # X = zeros(0, 0), src/datasets.jl:146
            self.totalphrases = 0
            return
        D = fms[0].shape[0]
        for f in fms:
            if f.shape[0] != D:
                raise _lib.DimensionMismatch("all feature matrices must share the feature dimension")
        T = np.array([f.shape[1] for f in fms], dtype=np.int64)
        Dout = (D - int(bool(ignore0th))) * (2 if add_delta else 1)
        out = np.empty((Dout, n), order="F")
        nk = np.zeros(1, dtype=np.int64)
        dpp = C.POINTER(C.c_double) * n
        _lib.check(_lib.lib.vcmi_gv_dataset(n, dpp(*[_lib.dptr(f) for f in fms]), _lib.iptr(T), D, int(bool(ignore0th)),
                                            int(bool(add_delta)), _lib.dptr(out), _lib.iptr(nk)))
        self.X = np.asfortranarray(out[:, :int(nk[0])])
        self.totalphrases = n
        if not np.all(np.isfinite(self.X)):            # @assert all(isfinite.(X)), src/datasets.jl:179
            raise AssertionError("non-finite global variance")
