"""GPU parity on ADVERSARIAL frames: the kernels that skip work on a proof -- fvconvert shape 3 with the bf16-split
and the FP64 screen, 4 / 2 / 1 screening rows per mixture (csrc/gmmmap_screen.hpp), the screened arg-max of predict -- against
the oracle on EVERY frame of calls that put posterior mass where a model's own p(x) never does: exact two-mixture boundaries,
1 / 20 / 45 / 47 nats either side (the e^-46 prune line included), triple points, outliers at 1e2..1e4 sigma and frames shifted by
+-2000 (oracle/adversarial.py builds them with the oracle's log-densities; src/gmmmap.jl:109-117, src/gmm.jl:24-30,44-47).
They are scattered 1:9 among p(x) draws so that the call is grouped and takes the screened kernels (>= 8192 frames).

Tolerance: per-frame relative 1e-9 as everywhere (north_star: 1e-5).  At a tie the posterior is as ill-conditioned as it gets
(an error d in a log-density moves y by d |E_a - E_b| / 4): the two CPU restatements differ by up to ~1e-10 there themselves.
"""
import numpy as np
import pytest

from conftest import julia_model

pytestmark = pytest.mark.gpu

TOL = 1e-9
TIE = 1e-7           # nats: below this gap between the two best log-densities rounding may legitimately pick either


@pytest.fixture(scope="module")
def vc():
    import voiceconversion_jl_amd as m
    assert m.device_count() >= 1
    return m


_CASES = {}


def _case(lam_lo, D=40, M=64, dup=False, zero=False, **kw):
    """model + mixed call + oracle answers, built once per module (the generator's bisections take ~10 s of host time)"""
    key = (lam_lo, D, M, dup, zero, tuple(sorted(kw.items())))
    if key not in _CASES:
        import synthdata as sd
        from oracle import adversarial as adv, c_oracle as co
        w, mu, sig = sd.synth_model(1002, 2 * D, M, lam_lo=lam_lo)
        if dup:          # an exact tie wherever mixture 0 leads: the reference's indmax takes the first (src/gmm.jl:46)
            mu[M - 1], sig[M - 1] = mu[0], sig[0]
            w = w.copy(); w[M - 1] = w[0]; w /= w.sum()
        if zero:
            w = w.copy(); w[5] = 0.0; w /= w.sum()
        ref = co.GMMMap(w, mu, sig)
        X, is_adv, cat, gap = adv.mixed_call(ref, w, mu, sig, D, 7, **kw)
        Yref, _ = ref.fvconvert_mt(X)                      # the per-frame oracle arithmetic (OpenMP over frames)
        L = ref.logdens(X)
        idx = np.argmax(L, axis=1) + 1                     # indmax of src/gmm.jl:46: the first maximum (numpy's rule too) ...
        assert np.array_equal(idx[:3000], ref.predict(X[:3000]))       # ... as the oracle's predict, which runs on one core
        import os, time
        if os.path.isdir("gpurun_out"):
            with open("gpurun_out/adversarial_case_timing.txt", "a") as f:
                f.write(f"{key}: built at {time.strftime('%H:%M:%S')}, {len(X)} frames\n")
        _CASES[key] = dict(w=w, mu=mu, sig=sig, X=X, is_adv=is_adv, cat=cat, gap=gap, Yref=Yref, L=L, idx=idx, names=adv.CATEGORIES)
    return _CASES[key]


def _frame_err(Y, Yref):
    return np.linalg.norm(Y - Yref, axis=1) / np.maximum(np.linalg.norm(Yref, axis=1), 1e-300)


def _report(c, err):
    return {("p(x)" if k < 0 else c["names"][k]): float(err[c["cat"] == k].max()) for k in np.unique(c["cat"])}


def test_the_generator_covers_what_it_claims():
    """ties with two live mixtures, triple points with three, imbalance points on both sides of the 46-nat prune line, far
    outliers -- and the p(x) draws beside them have nothing of the kind (their closest second-best mixture is hundreds of nats
    away on the SURVEY 8(d) model, which is why they cannot catch a wrong screen-out)"""
    c = _case(1e-5)
    cat, gap, names = c["cat"], c["gap"], c["names"]
    assert len(c["X"]) >= 8192 and 0.08 < c["is_adv"].mean() < 0.12
    tie = cat == names.index("tie")
    assert tie.sum() >= 3500 and np.max(gap[tie]) < 1e-6
    Ls = np.sort(c["L"], axis=1)
    tri = cat == names.index("triple")
    assert tri.sum() >= 20 and np.max(Ls[tri][:, -1] - Ls[tri][:, -3]) < 1e-3
    imb = gap[cat == names.index("imb")]
    for lo, hi in ((0.9, 1.1), (19.9, 20.1), (44.9, 45.1), (46.9, 47.1)):
        assert np.sum((imb > lo) & (imb < hi)) >= 300, (lo, hi)
    assert (cat == names.index("seg")).sum() >= 2000
    assert np.min(gap[cat == names.index("outlier")]) > 100.0
    assert np.min(gap[cat == -1]) > 100.0


@pytest.mark.parametrize("lam_lo", [1e-5, 1e-4, 1e-3])
@pytest.mark.parametrize("rows,fp64_screen", [(4, False), (4, True), (2, True), (1, True)])
def test_screened_fvconvert_on_adversarial_frames(vc, lam_lo, rows, fp64_screen):
    """shape 3 forced (and the model's own plan where that is shape 3), every frame of the call against the oracle"""
    import torch
    from voiceconversion_jl_amd import _lib
    c = _case(lam_lo, zero=(lam_lo == 1e-4))
    _lib.debug_force({4: _lib.DBG_SCREEN_ROWS4, 2: _lib.DBG_SCREEN_ROWS2, 1: _lib.DBG_SCREEN_ROWS1}[rows])     # read at creation
    try:
        g = vc.GMMMap(*julia_model(c["w"], c["mu"], c["sig"]))
    finally:
        _lib.debug_force(0)
    Xd = torch.from_numpy(c["X"]).cuda()
    _lib.debug_force(_lib.DBG_CONVERT_SHAPE_SCREENED | (_lib.DBG_SCREEN_FP64 if fp64_screen else 0))
    try:
        assert g.convert_plan()[1] == 3
        Y = vc.fvconvert(g, Xd.t()).t().cpu().numpy()
        Y2 = vc.fvconvert(g, Xd.t()).t().cpu().numpy()
    finally:
        _lib.debug_force(0)
    assert np.array_equal(Y, Y2)                                     # grouped + screened: still a function of the data alone
    err = _frame_err(Y, c["Yref"])
    assert np.all(np.isfinite(Y)) and err.max() < TOL, _report(c, err)
    # the dense loop (every mixture for every frame, no screen, no grouping) on the same frames: the same answer to rounding
    g.set_prune(float("inf"))
    _lib.debug_force(_lib.DBG_CONVERT_NO_GROUPING)
    try:
        Yd = vc.fvconvert(g, Xd.t()).t().cpu().numpy()
    finally:
        _lib.debug_force(0)
        g.set_prune(46.0)
    assert _frame_err(Yd, c["Yref"]).max() < TOL
    # what the 46-nat prune drops is worth e^-46 ~ 1e-20 of a frame: the screened and the dense answer agree far below TOL
    assert _frame_err(Y, Yd).max() < 1e-10, _report(c, _frame_err(Y, Yd))


@pytest.mark.parametrize("lam_lo", [1e-5, 1e-3])
def test_every_loop_shape_on_adversarial_frames(vc, lam_lo):
    """the library's own choice for the model, and the broad / peaked loops forced: the same frames, the same bar"""
    import torch
    from voiceconversion_jl_amd import _lib
    c = _case(lam_lo)
    g = vc.GMMMap(*julia_model(c["w"], c["mu"], c["sig"]))
    Xd = torch.from_numpy(c["X"]).cuda()
    for force in (0, _lib.DBG_CONVERT_SHAPE_BROAD, _lib.DBG_CONVERT_SHAPE_PEAKED, _lib.DBG_CONVERT_WIDE_TILES):
        _lib.debug_force(force)
        try:
            Y = vc.fvconvert(g, Xd.t()).t().cpu().numpy()
        finally:
            _lib.debug_force(0)
        err = _frame_err(Y, c["Yref"])
        assert err.max() < TOL, (force, _report(c, err))
    Yh = vc.fvconvert(g, np.asfortranarray(c["X"].T))               # host pointers: the chunked pipeline groups per chunk
    assert _frame_err(Yh.T, c["Yref"]).max() < TOL


def _check_indices(c, got, what):
    """identical to the oracle's wherever the decision is not within rounding of a tie; at a (near-)tie the chosen mixture must
    be one of the tied ones"""
    got = np.asarray(got)
    clear = c["gap"] > TIE
    assert np.array_equal(got[clear], c["idx"][clear]), (what, int(np.sum(got[clear] != c["idx"][clear])))
    near = ~clear
    L = c["L"]
    chosen = L[np.flatnonzero(near), got[near] - 1]
    assert np.all(chosen >= np.max(L[near], axis=1) - TIE), what
    return int(near.sum())


@pytest.mark.parametrize("lam_lo,dup", [(1e-5, False), (1e-4, True), (1e-3, False)])
def test_screened_predict_on_adversarial_frames(vc, lam_lo, dup):
    """gmmmap_screen_argmax_kernel (forced), the early-exit kernel and the all-tiles kernel: indices identical to co.predict on every
    frame whose decision is not within 1e-7 nats of a tie, one of the tied mixtures on the others -- and with a DUPLICATED
    mixture (exact ties, bit for bit equal log-densities on either side) the smaller index everywhere, as indmax does"""
    import torch
    from voiceconversion_jl_amd import _lib
    c = _case(lam_lo, dup=dup)
    g = vc.GMMMap(*julia_model(c["w"], c["mu"], c["sig"]))
    Xj = np.asfortranarray(c["X"].T)
    M = len(c["w"])
    res = {}
    for name, force in (("auto", 0), ("screen", _lib.DBG_PREDICT_SCREEN), ("early", _lib.DBG_PREDICT_NO_SCREEN), ("full", _lib.DBG_PREDICT_NO_EARLY_EXIT)):
        _lib.debug_force(force)
        try:
            res[name] = vc.predict(g.px, Xj)
            if name == "screen":
                dev = vc.predict(g.px, torch.from_numpy(c["X"]).cuda().t())
                assert np.array_equal(np.asarray(dev.cpu()), res[name])
        finally:
            _lib.debug_force(0)
        n_near = _check_indices(c, res[name], name)
        if dup:
            assert not np.any(res[name] == M)                        # the copy never wins a tie against the original
    assert n_near >= 3500                                            # the ties were really in the call
    # (the kernels sum a mixture's whitened squares in different orders -- last tile first in the early-exit loop -- so at a gap
    # of 1e-10 nats they may pick different members of a tie; everywhere else they agree with the oracle, hence with each other)
    assert np.array_equal(res["auto"], res["screen"]) or np.array_equal(res["auto"], res["early"])


def test_posteriors_on_adversarial_frames(vc):
    """predict_proba (src/gmm.jl:24-30) at the boundaries: the shares themselves (0.5 / 0.5, thirds) against the oracle"""
    from oracle import c_oracle as co
    c = _case(1e-5)
    sel = np.flatnonzero(c["is_adv"])[:4000]
    g = vc.GMMMap(*julia_model(c["w"], c["mu"], c["sig"]))
    P = vc.predict_proba(g.px, np.asfortranarray(c["X"][sel].T))
    Pref = co.GMMMap(c["w"], c["mu"], c["sig"]).predict_proba(c["X"][sel])
    assert np.max(np.abs(P - Pref.T)) < 1e-9
