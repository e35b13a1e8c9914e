"""adversarial.py -- frames on which a wrong screen-out WOULD show (VERDICT r5 "Next round" item 1).  TEST INFRASTRUCTURE ONLY:
imported by tests/ and __graft_entry__.smoke(); drives the C oracle (vco_logdens = lpr of src/gmm.jl:25-27) to PLACE the frames.

The frames a model's own p(x) produces have, on a peaked model (eigenvalues down to 1e-5), one live mixture each:
every screening decision is far from its threshold.  The kernels that skip work on a proof (csrc/gmmmap_screen.hpp: fvconvert
shape 3 with the FP64 or the bf16-split screen; gmmmap_screen_argmax_kernel) are wrong exactly where they rule out a mixture
that carries posterior mass, so the frames built here put posterior mass where p(x) never does:

  tie        for a pair (a, b): x_a ~ N_a, x_b ~ N_b; the point of the segment x_a -> x_b where mixture a stops being the
             arg-max (bisection on the oracle's arg-max): a and the mixture that takes over have EQUAL log-weighted density
             there and share the posterior -- on a peaked model as well.
  seg        the point of the segment mu_a -> mu_b where lpr_a = lpr_b (whatever a third mixture does there).
  imb        along the same x_a -> x_b segments, the points where a leads / trails the best other mixture by 1, 20, 45 and 47 nats:
             either side of one posterior share, and either side of the e^-46 line below which fvconvert drops a term
             (vcmi_gmmmap_set_prune, default 46).
  triple     a boundary point a | b slid along the boundary until the mixture that takes over changes: three (nearly) equal
             log-densities (nested bisection; kept when the top three lie within `triple_tol` nats).
  outlier    mu_m + r L_m z for r = 1e2, 1e3, 1e4 ("sigmas"), and ordinary frames shifted by +-2000 in every feature: the
             bf16 screen's certified margin eps = 2^-12 (|P||x| + |c|) is largest there.

adversarial_frames() returns the frames, a category code per frame and, from the oracle, the gap between the best and the second
best log-density at each (what the tests use to know which arg-max decisions rounding may legitimately flip).
"""
import numpy as np

CATEGORIES = ("tie", "seg", "imb", "triple", "outlier", "shift")
IMBALANCES = (1.0, 20.0, 45.0, 47.0)


def _chol_x(sig, D):
    return [np.linalg.cholesky((s[:D, :D] + s[:D, :D].T) / 2.0) for s in sig]


def _top2(L):
    """(arg-max, best, second best) per row of the (n, M) log-density matrix"""
    top = np.argmax(L, axis=1)
    best = L[np.arange(len(L)), top]
    L2 = L.copy()
    L2[np.arange(len(L)), top] = -np.inf
    return top, best, np.max(L2, axis=1)


def _lead(L, a):
    """lpr_a - max_{c != a} lpr_c per row"""
    n = np.arange(len(L))
    la = L[n, a]
    L2 = L.copy()
    L2[n, a] = -np.inf
    return la - np.max(L2, axis=1)


def _bisect(pred, lo, hi, iters):
    """largest s (to 2^-iters of the bracket) with pred(s) true, given pred(lo) true and pred(hi) false, vectorised"""
    lo, hi = lo.copy(), hi.copy()
    for _ in range(iters):
        mid = 0.5 * (lo + hi)
        ok = pred(mid)
        lo = np.where(ok, mid, lo)
        hi = np.where(ok, hi, mid)
    return lo, hi


def adversarial_frames(ref, w, mu, sig, D, seed, npairs=2016, nimb=500, ntriples=150, noutliers=192, triple_tol=1e-3):
    """ref: oracle.c_oracle.GMMMap of (w, mu, sig); returns dict(X (n, D), cat (n,) index into CATEGORIES, pair (n, 2))."""
    rng = np.random.default_rng(seed)
    M = len(w)
    live = np.flatnonzero(np.asarray(w) > 0.0)
    Ls = _chol_x(sig, D)
    mux = np.asarray(mu)[:, :D]

    def draw(m):
        return mux[m] + rng.standard_normal(D) @ Ls[m].T

    # ordered pairs of live mixtures, every unordered pair once while they last, then random ones
    allp = [(a, b) for i, a in enumerate(live) for b in live[i + 1:]]
    rng.shuffle(allp)
    while len(allp) < npairs:
        a, b = rng.choice(live, 2, replace=False)
        allp.append((int(a), int(b)))
    pairs = np.array(allp[:npairs], dtype=np.int64)
    a_idx, b_idx = pairs[:, 0], pairs[:, 1]
    Xa = np.stack([draw(a) for a in a_idx])
    Xb = np.stack([draw(b) for b in b_idx])
    seg = lambda s, P0=Xa, P1=Xb: P0 + s[:, None] * (P1 - P0)          # noqa: E731

    out_X, out_cat, out_pair = [], [], []

    def emit(X, cat, pr):
        out_X.append(np.atleast_2d(X))
        out_cat.append(np.full(len(np.atleast_2d(X)), CATEGORIES.index(cat), dtype=np.int64))
        out_pair.append(np.atleast_2d(pr))

    # ---- tie: where a stops being the arg-max on x_a -> x_b
    a_top = lambda s: np.argmax(ref.logdens(seg(s)), axis=1) == a_idx    # noqa: E731
    ok0 = a_top(np.zeros(npairs)) & ~a_top(np.ones(npairs))              # (a draw of a whose arg-max is not a: dropped)
    lo, hi = _bisect(lambda s: a_top(s) | ~ok0, np.zeros(npairs), np.ones(npairs), 58)
    s0 = lo
    emit(seg(lo)[ok0], "tie", pairs[ok0])
    emit(seg(hi)[ok0], "tie", pairs[ok0])                                  # ... and the first point on the other side

    # ---- imb: a leads / trails the best other mixture by d nats, on the first `nimb` segments
    k = min(nimb, npairs)
    sub = np.arange(k)
    lead = lambda s: _lead(ref.logdens(Xa[sub] + s[:, None] * (Xb[sub] - Xa[sub])), a_idx[sub])     # noqa: E731
    for d in IMBALANCES:
        lo_p, _ = _bisect(lambda s: lead(s) >= d, np.zeros(k), s0[sub], 36)                          # a ahead by d
        _, hi_m = _bisect(lambda s: lead(s) >= -d, s0[sub], np.ones(k), 36)                          # a behind by d
        keep = ok0[sub]
        emit((Xa[sub] + lo_p[:, None] * (Xb[sub] - Xa[sub]))[keep], "imb", pairs[sub][keep])
        emit((Xa[sub] + hi_m[:, None] * (Xb[sub] - Xa[sub]))[keep], "imb", pairs[sub][keep])

    # ---- seg: lpr_a = lpr_b on mu_a -> mu_b (the verdict's literal construction)
    Ma, Mb = mux[a_idx], mux[b_idx]
    n = np.arange(npairs)

    def a_over_b(s):
        L = ref.logdens(Ma + s[:, None] * (Mb - Ma))
        return L[n, a_idx] >= L[n, b_idx]
    lo, _ = _bisect(a_over_b, np.zeros(npairs), np.ones(npairs), 58)
    emit(Ma + lo[:, None] * (Mb - Ma), "seg", pairs)

    # ---- triple: slide the a | . boundary point from x_b towards x_c until the mixture that takes over changes
    if ntriples > 0 and len(live) >= 3:
        tri = np.array([rng.choice(live, 3, replace=False) for _ in range(ntriples)], dtype=np.int64)
        Ta = np.stack([draw(m) for m in tri[:, 0]])
        Tb = np.stack([draw(m) for m in tri[:, 1]])
        Tc = np.stack([draw(m) for m in tri[:, 2]])
        ta = tri[:, 0]

        nt = ntriples

        def boundary(v):
            """v: (nt, G) end-point parameters -> the a-side boundary points on x_a -> x_b + v (x_c - x_b), (nt, G, D), and the
            mixture on their other side (nt, G); all nt * G segments bisected in one batch (one oracle call per step)"""
            G = v.shape[1]
            End = (Tb[:, None, :] + v[:, :, None] * (Tc - Tb)[:, None, :]).reshape(nt * G, D)
            Sta = np.repeat(Ta, G, axis=0)
            aa = np.repeat(ta, G)
            top_a = lambda s: np.argmax(ref.logdens(Sta + s[:, None] * (End - Sta)), axis=1) == aa          # noqa: E731
            lo_s, hi_s = _bisect(top_a, np.zeros(nt * G), np.ones(nt * G), 44)
            other = np.argmax(ref.logdens(Sta + hi_s[:, None] * (End - Sta)), axis=1)
            return (Sta + lo_s[:, None] * (End - Sta)).reshape(nt, G, D), other.reshape(nt, G)
        # the mixture that takes over at v = 0 and at v = 1; where they differ, narrow the v at which it changes on a grid of G
        # points per level (G^levels ~ 1e9: as fine as 30 bisection steps, in a tenth of the oracle calls)
        G, levels = 8, 10
        _, oe = boundary(np.stack([np.zeros(nt), np.ones(nt)], axis=1))
        o0, o1 = oe[:, 0], oe[:, 1]
        cand = (o0 != o1) & (o0 != ta) & (o1 != ta)
        vlo, vhi = np.zeros(nt), np.ones(nt)
        for _ in range(levels):
            grid = vlo[:, None] + (vhi - vlo)[:, None] * (np.arange(1, G + 1) / (G + 1.0))[None, :]
            _, og = boundary(grid)
            same = og == o0[:, None]                                   # (nt, G): still the v = 0 mixture on the other side
            first = np.where(same.all(axis=1), G, np.argmin(same, axis=1))          # first grid point where it is not
            allg = np.concatenate([vlo[:, None], grid, vhi[:, None]], axis=1)      # (nt, G + 2)
            vlo, vhi = allg[np.arange(nt), first], allg[np.arange(nt), first + 1]
        P = boundary(vlo[:, None])[0][:, 0, :]
        Lp = np.sort(ref.logdens(P), axis=1)
        good = cand & (Lp[:, -1] - Lp[:, -3] < triple_tol)
        emit(P[good], "triple", tri[good][:, :2])

    # ---- outliers and shifted frames
    per = max(noutliers // 3, 1)
    for r in (1e2, 1e3, 1e4):
        ms = rng.choice(live, per)
        emit(np.stack([mux[m] + r * (rng.standard_normal(D) @ Ls[m].T) for m in ms]), "outlier", np.stack([ms, ms], 1))
    ms = rng.choice(live, per)
    base = np.stack([draw(m) for m in ms])
    emit(base + 2000.0, "shift", np.stack([ms, ms], 1))
    emit(base - 2000.0, "shift", np.stack([ms, ms], 1))

    X = np.ascontiguousarray(np.concatenate(out_X))
    return {"X": X, "cat": np.concatenate(out_cat), "pair": np.concatenate(out_pair)}


def mixed_call(ref, w, mu, sig, D, seed, ratio=9, min_frames=8192, **kw):
    """The adversarial frames scattered 1 : ratio among draws from the model's own p(x) (at least `min_frames` in all, so that the
    call is grouped and takes the screened kernels).  Returns (X (T, D), is_adversarial (T,), cat (T,) with -1 for p(x) draws,
    gap (T,): best minus second best oracle log-density)."""
    import synthdata
    adv = adversarial_frames(ref, w, mu, sig, D, seed, **kw)
    na = len(adv["X"])
    nplain = max(ratio * na, min_frames - na)
    Xp = synthdata.sample_frames(seed + 1, np.asarray(w), np.asarray(mu), np.asarray(sig), nplain, 0, D)
    X = np.concatenate([adv["X"], Xp])
    cat = np.concatenate([adv["cat"], np.full(nplain, -1, dtype=np.int64)])
    order = np.random.default_rng(seed + 2).permutation(len(X))
    X, cat = np.ascontiguousarray(X[order]), cat[order]
    _, best, second = _top2(ref.logdens(X))
    return X, cat >= 0, cat, best - second
