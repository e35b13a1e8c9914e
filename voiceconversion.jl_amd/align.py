"""Alignment of parallel data -- reference src/align.jl:1-55."""
import ctypes as C

import numpy as np

from . import _lib
from ._arrays import jl_matrix


def align(src, tgt, return_path=False):
    """align(src (D,S), tgt (D,T)) -> (src, newtgt (D,S)); src/align.jl:8-35.  DTW(fstep=0, bstep=2) with the
    source as template, scatter of the target frames onto the path (later frames win) and neighbour-average
    interpolation of the skipped template frames -- all in one kernel launch."""
    src = jl_matrix(src, "src")
    tgt = jl_matrix(tgt, "tgt")
    if src.shape[0] != tgt.shape[0]:          # src/align.jl:11-13 (the reference misspells DimensionMismatch)
        raise _lib.DimensionMismatch("order of feature vector must be equal")
    D, S = src.shape
    T = tgt.shape[1]
    newtgt = np.empty((D, S), order="F")
    path = np.empty(T, dtype=np.int64)
    _lib.check(_lib.lib.vcmi_align(_lib.dptr(src), S, _lib.dptr(tgt), T, D, _lib.dptr(newtgt), _lib.iptr(path)))
    return (src, newtgt, path) if return_path else (src, newtgt)


def align_batch(srcs, tgts):
    """Batch extension: [align(s, t) for (s, t) in zip(srcs, tgts)] in one launch (one workgroup per pair)."""
    n = len(srcs)
    if n != len(tgts):
        raise ValueError("srcs and tgts must have the same length")
    if n == 0:
        return []
    ss = [jl_matrix(s, "src") for s in srcs]
    tt = [jl_matrix(t, "tgt") for t in tgts]
    D = ss[0].shape[0]
    for s, t in zip(ss, tt):
        if s.shape[0] != D or t.shape[0] != D:
            raise _lib.DimensionMismatch("order of feature vector must be equal")
    S = np.array([s.shape[1] for s in ss], dtype=np.int64)
    T = np.array([t.shape[1] for t in tt], dtype=np.int64)
    outs = [np.empty((D, int(s)), order="F") for s in S]
    dpp = C.POINTER(C.c_double) * n
    _lib.check(_lib.lib.vcmi_align_batch(n, dpp(*[_lib.dptr(s) for s in ss]), _lib.iptr(S), dpp(*[_lib.dptr(t) for t in tt]),
                                         _lib.iptr(T), D, dpp(*[_lib.dptr(o) for o in outs])))
    return [(s, o) for s, o in zip(ss, outs)]
