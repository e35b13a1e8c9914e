"""Dynamic time warping -- reference src/dtw.jl:1-147 (sub-module DTWs).

`fit!`, `update!`, `set_template!` are spelled `fit_`, `update_`, `set_template_` (Python has no `!`).
Matrices have the Julia shape (D, frames)."""
import ctypes as C

import numpy as np

from . import _lib
from ._arrays import jl_matrix, jl_vector


class DTW:
    """type DTW (src/dtw.jl:11-17): fstep, bstep, template (D,S), costtable (S,T+1) Float64,
    backpointer (S,T+1) Int (1-based).  DTW(;fstep=0, bstep=1) -- src/dtw.jl:19-21."""

    def __init__(self, fstep=0, bstep=1):
        self.fstep = int(fstep)
        self.bstep = int(bstep)
        self.template = np.zeros((1, 1), order="F")
        self.costtable = np.zeros((1, 1), order="F")
        self.backpointer = np.zeros((1, 1), dtype=np.int64, order="F")


def transition(d, i, j):
    """src/dtw.jl:23-31"""
    if j == i + 1:
        return 0.0
    if i == j:
        return 1.0
    return 2.0


def set_template_(d, template):
    """set_template!(d, template) + lazy_init!(d, S): src/dtw.jl:38-42,53-56"""
    d.template = jl_matrix(template, "template")
    S = d.template.shape[1]
    d.costtable = np.arange(1, S + 1, dtype=np.float64).reshape(S, 1, order="F")
    d.backpointer = np.arange(1, S + 1, dtype=np.int64).reshape(S, 1, order="F")


def fit_(d, template, sequence=None, tables=True):
    """fit!(d, template, sequence) / fit!(d, sequence): src/dtw.jl:93-130.  Returns the path (T,), 1-based
    template index per sequence frame.  With tables=True (the reference's behaviour) d.costtable and
    d.backpointer are filled; tables=False is the path-only mode that keeps the tables out of HBM."""
    if sequence is None:
        template, sequence = d.template, template
    tm = jl_matrix(template, "template")
    sq = jl_matrix(sequence, "sequence")
    if tm.shape[0] != sq.shape[0]:
        raise _lib.DimensionMismatch("template and sequence must have the same feature dimension")
    D, S = tm.shape
    T = sq.shape[1]
    path = np.empty(T, dtype=np.int64)
    d.template = tm
    if tables:
        cost = np.empty((S, T + 1), order="F")
        bp = np.empty((S, T + 1), dtype=np.int64, order="F")
        _lib.check(_lib.lib.vcmi_dtw_fit(_lib.dptr(tm), S, _lib.dptr(sq), T, D, d.fstep, d.bstep, _lib.iptr(path),
                                         _lib.dptr(cost), _lib.iptr(bp)))
        d.costtable, d.backpointer = cost, bp
    else:
        _lib.check(_lib.lib.vcmi_dtw_fit(_lib.dptr(tm), S, _lib.dptr(sq), T, D, d.fstep, d.bstep, _lib.iptr(path), None, None))
    return path


def fit_batch(d, templates, sequences):
    """Batch extension: n independent (template, sequence) pairs in ONE launch, one workgroup per
    pair; returns a list of paths.  Equivalent to [fit!(DTW(...), t, s) for (t, s) in zip(...)]."""
    n = len(templates)
    if n != len(sequences):
        raise ValueError("templates and sequences must have the same length")
    if n == 0:
        return []
    tms = [jl_matrix(t, "template") for t in templates]
    sqs = [jl_matrix(s, "sequence") for s in sequences]
    D = tms[0].shape[0]
    for t, s in zip(tms, sqs):
        if t.shape[0] != D or s.shape[0] != D:
            raise _lib.DimensionMismatch("all matrices of a batch must share the feature dimension")
    S = np.array([t.shape[1] for t in tms], dtype=np.int64)
    T = np.array([s.shape[1] for s in sqs], dtype=np.int64)
    paths = [np.empty(int(t), dtype=np.int64) for t in T]
    dpp = C.POINTER(C.c_double) * n
    ipp = C.POINTER(C.c_int64) * n
    _lib.check(_lib.lib.vcmi_dtw_fit_batch(n, dpp(*[_lib.dptr(t) for t in tms]), _lib.iptr(S), dpp(*[_lib.dptr(s) for s in sqs]),
                                           _lib.iptr(T), D, d.fstep, d.bstep, ipp(*[_lib.iptr(p) for p in paths])))
    return paths


def fit_batch_dev(d, feats, tmpl_off, S, seq_off, T, D, paths=None):
    """Device-resident batch: `feats` is a 1-D float64 torch tensor on the HIP device holding every (D,S) template and
    (D,T) sequence in the Julia memory image at element offsets tmpl_off / seq_off.  Returns (paths, path_off): one
    int64 device tensor with the 1-based paths back to back, and the host offsets of each pair's path in it."""
    import torch

    from ._arrays import current_stream_ptr

    arr = lambda a: np.ascontiguousarray(a, dtype=np.int64)  # noqa: E731
    tmpl_off, S, seq_off, T = arr(tmpl_off), arr(S), arr(seq_off), arr(T)
    n = len(S)
    if not (len(tmpl_off) == len(seq_off) == len(T) == n):
        raise ValueError("offset and length arrays must have one entry per pair")
    if feats.dtype != torch.float64 or not feats.is_cuda or not feats.is_contiguous():
        raise TypeError("feats must be a contiguous float64 tensor on the HIP device")
    path_off = np.concatenate([[0], np.cumsum(T)[:-1]]).astype(np.int64) if n else np.zeros(0, dtype=np.int64)
    if paths is None:
        paths = torch.empty(int(T.sum()), dtype=torch.int64, device=feats.device)
    _lib.check(_lib.lib.vcmi_dtw_fit_batch_dev(n, feats.data_ptr(), _lib.iptr(tmpl_off), _lib.iptr(S), _lib.iptr(seq_off),
                                               _lib.iptr(T), int(D), d.fstep, d.bstep, paths.data_ptr(), _lib.iptr(path_off),
                                               current_stream_ptr()))
    return paths, path_off


def update_(d, v):
    """update!(d, v): one on-line column, src/dtw.jl:61-90.  Host-side by design (the reference
    grows both tables by hcat per call; a single S-cell column is not a GPU target).  Same arithmetic order as
    fit!: (lastcost[j] + obs) + transition."""
    v = jl_vector(v)
    S, T = d.costtable.shape
    last = d.costtable[:, T - 1]
    cur = np.zeros(S)
    curbp = np.zeros(S, dtype=np.int64)
    for i in range(1, S + 1):
        diff = v - d.template[:, i - 1]
        obs = 0.0
        for x in diff:
            obs = obs + x * x
        minindex = i
        mincost = last[i - 1] + obs + transition(d, minindex, i)
        for j in range(i - d.bstep, i + d.fstep + 1):
            if j < 1 or j > S:
                continue
            c = last[j - 1] + obs + transition(d, j, i)
            if c < mincost:
                mincost, minindex = c, j
        cur[i - 1] = mincost
        curbp[i - 1] = minindex
    d.costtable = np.asfortranarray(np.hstack([d.costtable, cur[:, None]]))
    d.backpointer = np.asfortranarray(np.hstack([d.backpointer, curbp[:, None]]))


def backward(d):
    """backward(d): src/dtw.jl:133-145, from the materialised tables."""
    T = d.costtable.shape[1] - 1
    path = np.zeros(T, dtype=np.int64)
    if T == 0:
        return path
    path[T - 1] = int(np.argmin(d.costtable[:, T])) + 1
    for i in range(T, 1, -1):
        path[i - 2] = d.backpointer[path[i - 1] - 1, i]
    return path
