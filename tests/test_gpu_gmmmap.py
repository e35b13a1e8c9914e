"""GPU parity: GMMMap fvconvert / predict_proba / predict / vc through the C-ABI vs the CPU oracle and the
committed golden vectors.  Tolerance: north_star asks <= 1e-5 relative on converted mel-cepstra; the FP64
kernels are held to 1e-9 here (observed ~1e-12)."""
import numpy as np
import pytest

from conftest import frame_relerr, julia_model, load_golden, relerr

pytestmark = pytest.mark.gpu

TOL = 1e-9


@pytest.fixture(scope="module")
def vc():
    import voiceconversion_jl_amd as m
    assert m.device_count() >= 1
    return m


@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("swap", [False, True])
def test_fixture_model_matches_golden(vc, fixture_model, kernel, swap):
    w, mu, sig = julia_model(*fixture_model)
    z = load_golden("gmmmap_fixture_model.npz")
    k = "swap" if swap else "fwd"
    g = vc.GMMMap(w, mu, sig, swap=swap)
    g.set_kernel(kernel)
    assert len(g) == 1 and vc.dim(g) == 40 and vc.ncomponents(g) == 32 and vc.size(g) == (40, 1)   # test/gmmmap.jl:9-14
    X = z[f"X_{k}"].T                      # Julia shape (D,T)
    Y = vc.fvconvert(g, X)
    assert Y.shape == X.shape
    assert frame_relerr(Y, z[f"Y_{k}"].T) < TOL
    P = vc.predict_proba(g.px, X)
    assert P.shape == (32, X.shape[1])
    assert np.max(np.abs(P - z[f"P_{k}"].T)) < 1e-9
    assert np.array_equal(vc.predict(g.px, X), z[f"idx_{k}"])
    # single-frame signature of the reference, src/gmmmap.jl:101
    y0 = vc.fvconvert(g, X[:, 0].copy())
    assert y0.shape == (40,) and relerr(y0, z[f"Y_{k}"][0]) < TOL


@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("swap", [False, True])
def test_joint_model_matches_golden(vc, joint_model, kernel, swap):
    """The reference's second trained model, clb_and_slt_gmm32_order40 (test/vc.jl:40-51): conversion, posteriors, arg-max
    and the vc() frame loop against the committed vectors (oracle/gen_golden_joint.py)."""
    w, mu, sig = julia_model(*joint_model)
    z = load_golden("gmmmap_joint_model.npz")
    k = "swap" if swap else "fwd"
    g = vc.GMMMap(w, mu, sig, swap=swap)
    g.set_kernel(kernel)
    assert vc.dim(g) == 40 and vc.ncomponents(g) == 32
    X = z[f"X_{k}"].T
    Y = vc.fvconvert(g, X)
    assert frame_relerr(Y, z[f"Y_{k}"].T) < TOL
    assert np.max(np.abs(vc.predict_proba(g.px, X) - z[f"P_{k}"].T)) < 1e-9
    assert np.array_equal(vc.predict(g.px, X), z[f"idx_{k}"])
    if not swap:
        out = vc.vc(g, z["vc_fm"].T)
        assert np.array_equal(out[0], z["vc_fm"][:, 0])                  # src/common.jl:23
        assert frame_relerr(out[1:], z["vc_out"][:, 1:].T) < TOL


def test_vc_keeps_power_row(vc, fixture_model):
    w, mu, sig = julia_model(*fixture_model)
    z = load_golden("gmmmap_fixture_model.npz")
    g = vc.GMMMap(w, mu, sig)
    out = vc.vc(g, z["vc_fm"].T)
    assert np.array_equal(out[0], z["vc_fm"][:, 0])                  # src/common.jl:23
    assert frame_relerr(out[1:], z["vc_out"][:, 1:].T) < TOL


@pytest.mark.parametrize("kernel", [1, 2])
def test_config1_plumbing(vc, kernel):
    """BASELINE.json configs[0]: D=24, M=8, T=1000."""
    z = load_golden("gmmmap_cfg1_D24_M8_T1000.npz")
    g = vc.GMMMap(*julia_model(z["weights"], z["means"], z["covars"]))
    g.set_kernel(kernel)
    Y = vc.fvconvert(g, z["X"].T)
    assert frame_relerr(Y, z["Y"].T) < TOL
    assert np.array_equal(vc.predict(g.px, z["X"].T), z["idx"])


@pytest.mark.parametrize("D,M,T", [(40, 64, 4099), (16, 3, 130), (20, 5, 1), (25, 4, 77), (7, 2, 65), (80, 8, 300),
                                   (33, 6, 200), (42, 4, 100), (50, 5, 333), (55, 3, 64), (57, 3, 70), (70, 4, 129), (66, 3, 90), (75, 2, 40),
                                   # beyond D = 48: eight waves per workgroup (128 frames), two block buffers up to D = 72
                                   (56, 6, 1000), (64, 5, 777), (72, 4, 600), (80, 3, 515)])
def test_random_models_vs_oracle(vc, D, M, T):
    """Seeded synthetic models (SURVEY 8d generator) at sizes the C oracle finishes in seconds.  The MFMA kernel exists
    for the padded dimensions 16..80 in steps of 4 (D = 50 is the static + delta vector of 25-dimensional
    mel-cepstra); D = 7 exercises the generic kernel; ragged T exercises the tile tails."""
    from oracle import c_oracle as co, np_oracle as npo
    w, mu, sig = npo.synth_model(1000 + D + M, 2 * D, M)
    X = npo.sample_frames(2000 + T, w, mu, sig, T, 0, D)
    ref = co.GMMMap(w, mu, sig)
    Yref = ref.fvconvert(X)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    Y = vc.fvconvert(g, X.T)
    assert frame_relerr(Y, Yref.T) < TOL
    assert np.max(np.abs(vc.predict_proba(g.px, X.T) - ref.predict_proba(X).T)) < 1e-9
    assert np.array_equal(vc.predict(g.px, X.T), ref.predict(X))
    assert relerr(g.SyxSxxinv, np.transpose(ref.A, (1, 2, 0))) < 1e-9


def test_two_pass_predict_matches_in_kernel_argmax(vc):
    """predict through the (M,T) log-density matrix + argmax kernel (the generic path's route) against the argmax kept
    inside the MFMA kernel, with M not a multiple of 4 and a zero weight."""
    from oracle import c_oracle as co, np_oracle as npo
    from voiceconversion_jl_amd import _lib
    w, mu, sig = npo.synth_model(4040, 80, 10)
    w = w.copy(); w[7] = 0.0; w /= w.sum()
    X = npo.sample_frames(4041, w, mu, sig, 1000, 0, 40)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    one = vc.predict(g.px, X.T)
    _lib.debug_force(_lib.DBG_PREDICT_TWO_PASS)
    try:
        two = vc.predict(g.px, X.T)
    finally:
        _lib.debug_force(0)
    assert np.array_equal(one, two) and np.array_equal(one, co.GMMMap(w, mu, sig).predict(X))


@pytest.mark.parametrize("D,M,T", [(40, 64, 5000), (80, 32, 3000), (80, 64, 4097), (24, 9, 700), (17, 3, 50), (16, 2, 16), (40, 200, 333)])
def test_predict_early_exit_is_exact(vc, D, M, T):
    """predict(px, X) stops the whitening of a mixture as soon as its partial |z|^2 shows that it cannot be the first
    maximum of any of a tile's 16 frames (MODE 3: the tiles run last first).  The indices must equal those of the kernel
    that evaluates every tile of every mixture, and the oracle's -- also with exact ties (duplicated mixtures: the first
    one wins) and with a mixture of weight zero."""
    from oracle import c_oracle as co, np_oracle as npo
    from voiceconversion_jl_amd import _lib
    w, mu, sig = npo.synth_model(900 + D + M, 2 * D, M)
    if M >= 3:                                     # mixtures 0 and M-1 identical: every frame of either ties exactly
        mu[M - 1], sig[M - 1] = mu[0], sig[0]
        w = w.copy(); w[M - 1] = w[0]; w[1] = 0.0; w /= w.sum()
    X = npo.sample_frames(901, w, mu, sig, T, 0, D)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    fast = vc.predict(g.px, X.T)
    _lib.debug_force(_lib.DBG_PREDICT_NO_EARLY_EXIT)
    try:
        full = vc.predict(g.px, X.T)
    finally:
        _lib.debug_force(0)
    ref = co.GMMMap(w, mu, sig).predict(X)
    assert np.array_equal(fast, full) and np.array_equal(fast, ref)
    if M >= 3:
        assert not np.any(fast == M) and not np.any(fast == 2)      # the duplicate never wins, the zero weight neither


def test_zero_weight_component_has_zero_posterior(vc):
    from oracle import c_oracle as co, np_oracle as npo
    w, mu, sig = npo.synth_model(5, 32, 6)
    w = w.copy(); w[2] = 0.0; w /= w.sum()
    X = npo.sample_frames(6, w, mu, sig, 100, 0, 16)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    P = vc.predict_proba(g.px, X.T)
    assert np.all(P[2] == 0.0)
    assert frame_relerr(vc.fvconvert(g, X.T), co.GMMMap(w, mu, sig).fvconvert(X).T) < TOL


def test_errors(vc, fixture_model):
    w, mu, sig = julia_model(*fixture_model)
    g = vc.GMMMap(w, mu, sig)
    with pytest.raises(vc.DimensionMismatch):
        vc.fvconvert(g, np.zeros(39))                                 # src/gmmmap.jl:102
    bad = sig.copy()
    bad[:40, :40, 3] = -np.eye(40)
    with pytest.raises(vc.PosDefException):
        vc.GMMMap(w, mu, bad)                                         # PosDefException from MvNormal, src/gmm.jl:17


def test_device_resident_and_deterministic(vc, fixture_model):
    import torch
    w, mu, sig = julia_model(*fixture_model)
    z = load_golden("gmmmap_fixture_model.npz")
    g = vc.GMMMap(w, mu, sig)
    Xd = torch.from_numpy(np.ascontiguousarray(z["X_fwd"])).cuda()   # (T,D) contiguous = Julia (D,T) image
    Y1 = vc.fvconvert(g, Xd.t())
    Y2 = vc.fvconvert(g, Xd.t())
    torch.cuda.synchronize()
    assert torch.equal(Y1, Y2)
    assert frame_relerr(Y1.cpu().numpy(), z["Y_fwd"].T) < TOL


def test_linearity_at_full_size(vc):
    """Size-independent property at BASELINE config 2 scale (D=40, M=64, T=10^6): with a single-mixture model
    the map is affine, y = b + A x, so fvconvert(x1 + x2 - x3) = y1 + y2 - y3; and for the full model the
    converted frames of a tiled input repeat exactly."""
    import torch
    from oracle import np_oracle as npo
    w, mu, sig = npo.synth_model(1002, 80, 64)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    base = npo.sample_frames(1002, w, mu, sig, 4096, 0, 40)
    T = 1_000_000
    reps = -(-T // 4096)
    Xd = torch.from_numpy(base).cuda().repeat(reps, 1)[:T].contiguous()
    Y = vc.fvconvert(g, Xd.t()).t()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(Y).all())
    first = Y[:4096]
    for r in (1, reps // 2, reps - 2):
        assert torch.equal(Y[r * 4096:(r + 1) * 4096], first)         # identical inputs -> identical outputs
    from oracle import c_oracle as co
    ref = co.GMMMap(w, mu, sig).fvconvert(base[:256])
    assert frame_relerr(first[:256].cpu().numpy().T, ref.T) < TOL
    # the affine property, at full size: one mixture -> posterior 1, y = b + A x for every frame
    g1 = vc.GMMMap(*julia_model(w[:1] / w[:1], mu[:1], sig[:1]))
    T3 = T // 3
    x1, x2, x3 = Xd[:T3], Xd[T3:2 * T3].flip(0), Xd[2 * T3:3 * T3] * 0.5
    y1, y2, y3 = (vc.fvconvert(g1, x.contiguous().t()).t() for x in (x1, x2, x3))
    yc = vc.fvconvert(g1, (x1 + x2 - x3).contiguous().t()).t()
    torch.cuda.synchronize()
    scale = float(torch.max(torch.abs(yc)))
    assert float(torch.max(torch.abs(yc - (y1 + y2 - y3)))) < 1e-9 * scale


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("D,M,T", [(160, 3, 333), (100, 4, 130), (84, 2, 1), (66, 3, 150), (17, 2, 40)])
def test_posterior_beyond_80_dimensions(vc, D, M, T, kernel):
    """Dimensions without an instantiation of the conversion kernel, above all 80 < D <= 160 (e.g. the 160-dimensional
    joint GMM of delta-augmented features): the log-densities come from the tiled MFMA kernel (whitening blocks streamed through LDS; kernel 0 = automatic choice) -- against the oracle and
    against the generic kernel (kernel 1).  D = 100 and 84 end in a partial row tile."""
    from oracle import np_oracle as npo
    w, mu, sig = npo.synth_model(7000 + D, 2 * D, M, lam_lo=1e-3)
    X = npo.sample_frames(7001 + D, w, mu, sig, T, 0, D)
    ref = npo.GMMMap(w, mu, sig)
    P_ref = ref.predict_proba(X)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    g.set_kernel(kernel)
    P = vc.predict_proba(g.px, X.T)
    assert P.shape == (M, T)
    assert np.max(np.abs(P - P_ref.T)) < 1e-9
    assert np.array_equal(vc.predict(g.px, X.T), ref.predict(X))


def test_posterior_pruning_changes_nothing_visible(vc, fixture_model):
    """vcmi_gmmmap_set_prune: the default (46 nats) skips the regression of mixtures whose posterior is below 1e-20 on a whole
    16-frame tile.  Against the dense loop (prune = inf) on the reference's own model and on synthetic ones the outputs
    agree to rounding; the counter shows that work was really skipped, and that nothing is skipped when nothing may be:
    (a) pruning off, (b) a model of identical mixtures, where every posterior is 1/M."""
    import torch
    from oracle import np_oracle as npo
    cases = [fixture_model + (40,)]
    cases.append(npo.synth_model(1002, 80, 64) + (40,))
    cases.append(npo.synth_model(7, 48, 5) + (24,))
    for w, mu, sig, D in cases:
        M = len(w)
        g = vc.GMMMap(*julia_model(w, mu, sig))
        X = npo.sample_frames(3, w, mu, sig, 3000, 0, D)
        Xd = torch.from_numpy(X).cuda()
        g.prune_stats(True)
        Yp = vc.fvconvert(g, Xd.t()).t().clone()
        n_pruned = g.prune_stats(True)
        g.set_prune(float("inf"))
        Yf = vc.fvconvert(g, Xd.t()).t().clone()
        n_dense = g.prune_stats(False)
        tiles = -(-3000 // 16)
        assert n_dense == tiles * M
        assert 0 < n_pruned < n_dense                       # something was skipped, and not everything
        err = float((torch.linalg.norm(Yp - Yf, dim=1) / torch.linalg.norm(Yf, dim=1)).max())
        assert err < 1e-15, err
    # identical mixtures: every posterior is exactly 1/M, nothing may be skipped, y = the single-mixture map
    w1, mu1, sig1 = npo.synth_model(11, 80, 1)
    Mi = 9
    g = vc.GMMMap(*julia_model(np.full(Mi, 1.0 / Mi), np.repeat(mu1, Mi, 0), np.repeat(sig1, Mi, 0)))
    g1 = vc.GMMMap(*julia_model(w1, mu1, sig1))
    X = npo.sample_frames(5, w1, mu1, sig1, 500, 0, 40)
    g.prune_stats(True)
    Y = vc.fvconvert(g, np.asfortranarray(X.T))
    assert g.prune_stats(False) == -(-500 // 16) * Mi
    assert frame_relerr(Y, vc.fvconvert(g1, np.asfortranarray(X.T))) < 1e-12
    # a threshold that would show in y is refused
    with pytest.raises(vc.VCMIError):
        g.set_prune(20.0)
    with pytest.raises(vc.VCMIError):
        g.set_prune(float("nan"))


@pytest.mark.parametrize("D,M,T", [(82, 5, 300), (96, 8, 1000), (100, 3, 77), (160, 4, 130), (130, 70, 65)])
def test_fvconvert_beyond_the_tile_kernel(vc, D, M, T):
    """80 < D <= 160 (e.g. the 82..96-dimensional static + delta vectors of 41..48 coefficients): MFMA log-densities
    (logdens_tiled_kernel) + the softmax / regression kernel, against the oracle, against the one-lane-per-frame generic
    kernel it replaces on the automatic path, and with the pruning off; also through the host-pointer pipeline and vc()."""
    from oracle import c_oracle as co, np_oracle as npo
    w, mu, sig = npo.synth_model(300 + D, 2 * D, M, lam_lo=1e-3)
    if M >= 3:
        w = w.copy(); w[1] = 0.0; w /= w.sum()
    X = npo.sample_frames(301, w, mu, sig, T, 0, D)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    ref = co.GMMMap(w, mu, sig).fvconvert(X)
    Y = vc.fvconvert(g, X.T)
    assert frame_relerr(Y, ref.T) < TOL
    g.set_prune(float("inf"))
    Yd = vc.fvconvert(g, X.T)
    g.set_prune(46.0)
    assert frame_relerr(Y, Yd) < 1e-14
    g.set_kernel(1)
    Yg = vc.fvconvert(g, X.T)
    g.set_kernel(0)
    assert frame_relerr(Yg, ref.T) < TOL and frame_relerr(Y, Yg) < 1e-9
    fm = np.asfortranarray(np.vstack([np.arange(T, dtype=np.float64)[None], X.T]))
    out = vc.vc(g, fm)
    assert np.array_equal(out[0], fm[0]) and frame_relerr(out[1:], ref.T) < TOL


@pytest.mark.parametrize("D,M,T,lam_lo", [(40, 64, 20000, 1e-5), (24, 9, 8192, 1e-1), (80, 16, 9000, 1e-5), (16, 4, 8193, 3e-1)])
def test_frame_grouping_changes_nothing_visible(vc, D, M, T, lam_lo):
    """From 8192 frames on fvconvert first groups the frames by their nearest source mean (three small kernels, index
    indirection in the MFMA kernel, mixture loop started at the workgroup's group), so that a 16-frame tile holds frames of
    one mixture and the pruning leaves one or two regressions per tile.  Frames are independent of each other: the outputs
    must equal those of the caller's order -- to rounding where several mixtures share a frame (broad covariances: the
    rotated mixture order changes the order of the sum), bit for bit where one mixture owns it --, and the oracle's; the
    counter shows the regressions that were saved; shuffled and time-ordered input give the same frames."""
    import torch
    from oracle import c_oracle as co, np_oracle as npo
    from voiceconversion_jl_amd import _lib
    w, mu, sig = npo.synth_model(77 + D + M, 2 * D, M, lam_lo=lam_lo)
    X = npo.sample_frames(78, w, mu, sig, T, 0, D)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    Xd = torch.from_numpy(X).cuda()
    g.prune_stats(True)
    Yg = vc.fvconvert(g, Xd.t()).t().clone()
    n_grouped = g.prune_stats(True)
    _lib.debug_force(_lib.DBG_CONVERT_NO_GROUPING)
    try:
        Yu = vc.fvconvert(g, Xd.t()).t().clone()
    finally:
        _lib.debug_force(0)
    n_plain = g.prune_stats(False)
    err = float((torch.linalg.norm(Yg - Yu, dim=1) / torch.linalg.norm(Yu, dim=1)).max())
    assert err < 1e-13, err
    if lam_lo <= 1e-4:
        assert torch.equal(Yg, Yu)                      # peaked covariances: one mixture per frame, nothing to reorder
        assert n_grouped < n_plain                      # and the grouping saved regressions
    ref = co.GMMMap(w, mu, sig).fvconvert(X[:300])
    assert frame_relerr(Yg[:300].cpu().numpy().T, ref.T) < TOL
    # the same frames in another order come out the same, frame by frame
    perm = torch.randperm(T, generator=torch.Generator().manual_seed(5)).cuda()
    Ys = vc.fvconvert(g, Xd[perm].contiguous().t()).t()
    back = torch.empty_like(Ys)
    back[perm] = Ys
    assert float((torch.linalg.norm(back - Yg, dim=1) / torch.linalg.norm(Yg, dim=1)).max()) < 1e-13
    # repeat runs are bit-identical: the grouping is a stable counting sort (round 4), so the tiles, the rotated mixture
    # order of every workgroup and with them every sum are a function of the data alone
    for _ in range(4):
        assert torch.equal(vc.fvconvert(g, Xd.t()).t(), Yg)


@pytest.mark.parametrize("which", ["synthetic", "broad", "fixture"])
def test_loop_shapes_agree_and_the_model_picks_one(vc, fixture_model, which):
    """fvconvert has four code shapes (include/vcmi.h: vcmi_gmmmap_convert_plan): 0 dense (set_prune(inf)), 1 "broad" (every
    whitening tile, one branch around the regression), 2 "peaked" (last whitening tile first), 3 "screened" (gmmmap_screen.hpp:
    four mixtures screened per MFMA tile on their last four whitening rows, survivors evaluated in full; grouped calls).
    Which of 1 / 2 / 3 runs is a property of the MODEL, estimated at creation from 256 frames drawn from it; the test
    hook forces any.  All of them give the dense loop's y to rounding and the oracle's to TOL; the counters add up."""
    import torch
    from oracle import c_oracle as co, np_oracle as npo
    from voiceconversion_jl_amd import _lib
    if which == "fixture":
        w, mu, sig = fixture_model
        expect = 1
    else:
        w, mu, sig = npo.synth_model(1002, 80, 64, lam_lo=1e-5 if which == "synthetic" else 1e-1)
        expect = 3 if which == "synthetic" else 1
    M, T, D = len(w), 20000, 40
    X = npo.sample_frames(9, w, mu, sig, T, 0, D)
    Xd = torch.from_numpy(X).cuda()
    g = vc.GMMMap(*julia_model(w, mu, sig))
    _, shape, active, undecided = g.convert_plan()
    assert shape == expect, (shape, active, undecided)
    assert 1.0 / M - 1e-9 <= active <= 1.0 and 0.0 <= undecided <= 1.0
    g.set_prune(float("inf"))
    assert g.convert_plan()[1] == 0
    g.prune_stats(True)
    Yd = vc.fvconvert(g, Xd.t()).t().clone()
    issued, _, _, _ = g.convert_plan()
    tiles = -(-T // 16)
    assert g.prune_stats(False) == tiles * M
    assert issued == 42 * M * 4 * (-(-T // 64))     # 42 MFMAs per (tile, mixture) at D = 40; T <= 32768: one tile per wave, four waves per workgroup
    _lib.debug_force(_lib.DBG_CONVERT_WIDE_TILES)
    try:
        g.prune_stats(True)
        Yw = vc.fvconvert(g, Xd.t()).t().clone()
        assert g.convert_plan()[0] == 2 * 42 * M * 4 * (-(-T // 128))     # the throughput shape: two tiles per wave
        g.prune_stats(False)
    finally:
        _lib.debug_force(0)
    assert float((torch.linalg.norm(Yw - Yd, dim=1) / torch.linalg.norm(Yd, dim=1)).max()) < 1e-13
    g.set_prune(46.0)
    ref = co.GMMMap(w, mu, sig).fvconvert(X[:400])
    for force in (_lib.DBG_CONVERT_SHAPE_BROAD, _lib.DBG_CONVERT_SHAPE_PEAKED, _lib.DBG_CONVERT_SHAPE_SCREENED, 0):
        _lib.debug_force(force)
        try:
            g.prune_stats(True)
            Y = vc.fvconvert(g, Xd.t()).t().clone()
            n_issued, sh, _, _ = g.convert_plan()
            n_reg = g.prune_stats(False)
        finally:
            _lib.debug_force(0)
        assert sh == {_lib.DBG_CONVERT_SHAPE_BROAD: 1, _lib.DBG_CONVERT_SHAPE_PEAKED: 2, _lib.DBG_CONVERT_SHAPE_SCREENED: 3, 0: expect}[force]
        # (shape 3 on a broad model: nearly every mixture survives the four-row screen, so the screen comes on top of the dense work)
        assert 0 < n_issued <= issued * (1.1 if sh == 3 else 1.0) and 0 < n_reg <= tiles * M
        err = float((torch.linalg.norm(Y - Yd, dim=1) / torch.linalg.norm(Yd, dim=1)).max())
        assert err < 1e-13, (force, err)
        assert frame_relerr(Y[:400].cpu().numpy().T, ref.T) < TOL


@pytest.mark.parametrize("T", [1, 17, 64, 65, 1000, 2000, 8191, 30000])
def test_utterance_sized_calls_one_tile_per_wave(vc, T):
    """Calls of up to 32768 frames run one frame tile per wave (64-frame workgroups: twice as many, each with half the
    loop -- an utterance does not fill the chip either way); DBG_CONVERT_WIDE_TILES forces the 128-frame workgroups of the
    throughput path.  Same y to rounding (the workgroups start their mixture loops at different mixtures), the oracle's to
    TOL, in all three loop shapes, through device and host pointers (the host call hands the kernel its pinned slots)."""
    import torch
    from oracle import c_oracle as co, np_oracle as npo
    from voiceconversion_jl_amd import _lib
    w, mu, sig = npo.synth_model(1002, 80, 64)
    X = npo.sample_frames(11, w, mu, sig, T, 0, 40)
    Xd = torch.from_numpy(X).cuda()
    g = vc.GMMMap(*julia_model(w, mu, sig))
    ref = co.GMMMap(w, mu, sig).fvconvert(X[:300])
    for prune in (46.0, float("inf")):
        g.set_prune(prune)
        for shape in (0, _lib.DBG_CONVERT_SHAPE_BROAD, _lib.DBG_CONVERT_SHAPE_PEAKED):
            got = []
            for wide in (0, _lib.DBG_CONVERT_WIDE_TILES):
                _lib.debug_force(shape | wide)
                try:
                    got.append(vc.fvconvert(g, Xd.t()).t().clone())
                finally:
                    _lib.debug_force(0)
            err = float((torch.linalg.norm(got[0] - got[1], dim=1) / torch.linalg.norm(got[1], dim=1)).max())
            assert err < 1e-13, (prune, shape, err)
            assert frame_relerr(got[0][:300].cpu().numpy().T, ref.T) < TOL
    g.set_prune(46.0)
    Yh = vc.fvconvert(g, np.asfortranarray(X.T))                # host pointers: (D,T) Julia image
    assert frame_relerr(Yh[:, :300], ref.T) < TOL
    assert np.array_equal(Yh, got[0].cpu().numpy().T) or frame_relerr(Yh, got[0].cpu().numpy().T) < 1e-13


@pytest.mark.parametrize("T", [1, 3000, 40_000, 700_000])      # 320 B / 960 KB: recorded, not locked; 12.8 MB (heap) / 224 MB (mmap): whole pages locked
def test_pinned_arrays_take_the_direct_path_and_change_nothing(vc, fixture_model, T):
    """vcmi_host_register (include/vcmi.h): a caller that keeps its arrays pins them once; the host-pointer calls then DMA
    straight from / into them.  Results equal the staged path's, for either side alone and for both (compared to rounding:
    a chunk's frames are grouped among themselves, so a path that chose other chunk boundaries would compose its tiles
    differently on a broad model; repeat runs of one path are bit-identical); the registration is visible through vcmi_host_is_registered, overlapping ranges and unknown pointers are refused."""
    w, mu, sig = julia_model(*fixture_model)
    g = vc.GMMMap(w, mu, sig)
    rng = np.random.default_rng(T)
    X = np.asfortranarray(rng.standard_normal((40, T)) * 0.3)
    Y0 = vc.fvconvert(g, X)                                   # staged: nothing pinned
    same = lambda A, B: frame_relerr(A, B) < 1e-13            # noqa: E731
    Xp, Yp = X.copy(order="F"), np.empty_like(X, order="F")
    assert not vc.is_pinned(Xp)
    vc.pin(Xp)
    try:
        assert vc.is_pinned(Xp) and not vc.is_pinned(Yp)
        with pytest.raises(vc.VCMIError):
            vc.pin(Xp)                                        # overlapping range
        assert np.array_equal(vc.fvconvert(g, Xp), Y0)        # input direct, output staged
        vc.pin(Yp)
        try:
            vc.fvconvert(g, Xp, out=Yp)                       # both direct
            assert same(Yp, Y0)
            Yb = Yp.copy()
            vc.fvconvert(g, Xp, out=Yp)
            assert np.array_equal(Yp, Yb)                     # repeat runs of one path: bit-identical
            Yp[:] = 0.0
            vc.fvconvert(g, X, out=Yp)                        # input staged, output direct
            assert np.array_equal(Yp, Y0)
            P0 = vc.predict_proba(g.px, X)
            assert np.array_equal(vc.predict_proba(g.px, Xp), P0)
            assert np.array_equal(vc.predict(g.px, Xp), vc.predict(g.px, X))
        finally:
            vc.unpin(Yp)
    finally:
        vc.unpin(Xp)
    assert not vc.is_pinned(Xp)
    with pytest.raises(vc.VCMIError):
        vc.unpin(Xp)                                          # not registered (any more)
    assert np.array_equal(vc.fvconvert(g, Xp), Y0)            # ... and the array still converts, staged again


def test_torch_pinned_memory_is_recognised(vc, fixture_model):
    """Host memory pinned by the caller's own runtime (here torch's pin_memory = hipHostMalloc) is found through the HIP
    runtime's pointer attributes: no registration call needed."""
    import torch
    w, mu, sig = julia_model(*fixture_model)
    g = vc.GMMMap(w, mu, sig)
    T = 300_000
    xt = (torch.randn(T, 40, dtype=torch.float64) * 0.3).pin_memory()
    X = xt.numpy().T                                          # (40,T) Fortran view of the pinned storage
    assert X.flags.f_contiguous and vc.is_pinned(X)
    Y0 = vc.fvconvert(g, np.asfortranarray(X.copy()))
    assert np.array_equal(vc.fvconvert(g, X), Y0)


@pytest.mark.parametrize("D,M,T,lam_lo", [(40, 64, 200_000, 1e-5), (40, 64, 9000, 1e-5), (40, 67, 40_000, 1e-4), (24, 9, 33_000, 1e-5),
                                          (16, 4, 8200, 1e-5), (47, 33, 50_000, 1e-5), (40, 130, 35_000, 1e-5), (40, 32, 70_000, 1e-1)])
@pytest.mark.parametrize("rows", [4, 2, 1])
def test_screened_shape_against_dense_and_oracle(vc, D, M, T, lam_lo, rows):
    _screened_shape_case(vc, D, M, T, lam_lo, rows, 0.0, False)


@pytest.mark.parametrize("D,M,T,lam_lo,offset", [(40, 64, 40_000, 1e-5, 0.0), (40, 64, 40_000, 1e-5, 300.0), (32, 20, 20_000, 1e-4, -2000.0),
                                                 (40, 32, 30_000, 1e-1, 50.0), (36, 12, 9000, 1e-5, 7.0)])
@pytest.mark.parametrize("fp64_screen", [False, True])
def test_screen_on_the_bf16_pipe_is_certified(vc, D, M, T, lam_lo, offset, fp64_screen):
    """Four rows per mixture, D <= 40: the screen runs on v_mfma_f32_16x16x32_bf16 with P and x split into bf16 hi + lo and a
    CERTIFIED error margin (|a^ - a| <= 2^-12 (|P_i| |x| + |c_i|)), so that what it rules out is ruled out.  Same checks as the
    FP64 screen (DBG_SCREEN_FP64 selects that one), also with every feature shifted by a large constant -- where c_i = P_i mu is
    thousands of times the a_i that decide and the margin grows with it: fewer mixtures are ruled out, none wrongly."""
    _screened_shape_case(vc, D, M, T, lam_lo, 4, offset, fp64_screen)


def _screened_shape_case(vc, D, M, T, lam_lo, rows, offset, fp64_screen):
    """Shape 3 (gmmmap_screen.hpp): on grouped frames the workgroup's own mixture is evaluated first, every other mixture is
    screened on its last four whitening rows (four mixtures per MFMA tile, no cross-lane sum) and only survivors are evaluated.
    A screened-out mixture is below e^-prune of the final maximum for every frame of the workgroup, so y is the dense loop's to
    rounding: checked on both tile widths (T <= 32768: one tile per wave), M not a multiple of 4 / 16 / 64, M > 64, a zero-weight
    mixture, D not a multiple of 4, the broad model (forced: nearly everything survives), with 4 / 2 / 1 screening rows per
    mixture (4 / 8 / 16 mixtures per tile; prepare() picks the cheapest for the model, the hook forces each), against the
    oracle, repeat runs bit-identical, and the counters show the saving on the peaked models."""
    import torch
    from oracle import c_oracle as co, np_oracle as npo
    from voiceconversion_jl_amd import _lib
    w, mu, sig = npo.synth_model(500 + D + M, 2 * D, M, lam_lo=lam_lo)
    if M >= 9:
        w = w.copy(); w[3] = 0.0; w /= w.sum()
    X = npo.sample_frames(501, w, mu, sig, T, 0, D)
    if offset:
        mu = mu + offset
        X = X + offset
    Xd = torch.from_numpy(X).cuda()
    _lib.debug_force({4: _lib.DBG_SCREEN_ROWS4, 2: _lib.DBG_SCREEN_ROWS2, 1: _lib.DBG_SCREEN_ROWS1}[rows])     # read at creation
    try:
        g = vc.GMMMap(*julia_model(w, mu, sig))
    finally:
        _lib.debug_force(0)
    shape3 = _lib.DBG_CONVERT_SHAPE_SCREENED | (_lib.DBG_SCREEN_FP64 if fp64_screen else 0)
    _lib.debug_force(shape3)
    try:
        assert g.convert_plan()[1] == 3
        g.prune_stats(True)
        Y = vc.fvconvert(g, Xd.t()).t().clone()
        issued3 = g.convert_plan()[0]
        nreg3 = g.prune_stats(False)
        for _ in range(3):
            assert torch.equal(vc.fvconvert(g, Xd.t()).t(), Y)
    finally:
        _lib.debug_force(0)
    _lib.debug_force(_lib.DBG_CONVERT_SHAPE_PEAKED)
    try:
        g.prune_stats(True)
        Y2 = vc.fvconvert(g, Xd.t()).t().clone()
        issued2 = g.convert_plan()[0]
        g.prune_stats(False)
    finally:
        _lib.debug_force(0)
    g.set_prune(float("inf"))
    Yd = vc.fvconvert(g, Xd.t()).t().clone()
    g.set_prune(46.0)
    rel = lambda a, b: float((torch.linalg.norm(a - b, dim=1) / torch.linalg.norm(b, dim=1)).max())  # noqa: E731
    assert rel(Y, Yd) < 1e-13 and rel(Y, Y2) < 1e-13
    ref = co.GMMMap(w, mu, sig).fvconvert(X[:300])
    assert frame_relerr(Y[:300].cpu().numpy().T, ref.T) < TOL
    tiles = -(-T // 16)
    assert 0 < nreg3 <= tiles * M
    if lam_lo <= 1e-4:
        # the screen costs a quarter of the last-tile test; with few mixtures the group's own (42 MFMAs per tile) dominates both
        # (fewer rows per mixture screen more mixtures per tile but let more through: prepare() weighs the two; forced here)
        if rows == 4 and not offset:
            assert issued3 < (0.6 if M >= 32 else 1.0) * issued2, (issued3, issued2)
        if rows == 4 and not offset:
            assert nreg3 < max(0.2, 1.5 / M) * tiles * M         # about one regression per tile: the frame's own mixture


@pytest.mark.parametrize("D,M,T,lam_lo", [(80, 64, 40_000, 1e-3), (40, 64, 20_000, 1e-5), (80, 32, 9000, 1e-1), (24, 9, 8200, 1e-2),
                                          (47, 130, 12_000, 1e-3), (16, 4, 8192, 1e-5), (52, 17, 10_000, 1e-1)])
def test_screened_predict_is_exact(vc, fixture_model, D, M, T, lam_lo):
    """From 8192 frames on predict(px, X) groups the frames by nearest source mean and runs the screened arg-max
    (gmmmap_screen_argmax_kernel): the group keys' mixtures in full, every other mixture ruled out -- four per MFMA tile -- as
    soon as an upper bound of its log-density lies below the best one found, survivors in full.  Indices identical to the
    early-exit kernel's (DBG_PREDICT_NO_SCREEN), the all-tiles kernel's and the oracle's: on peaked and broad models,
    D = 80 (the trajectory conversion's static + delta vectors), D not a multiple of 4, M not a multiple of 4 / beyond 128,
    exact ties (duplicated mixtures: the smaller index wins whatever the evaluation order), a mixture of weight zero;
    repeat runs identical; device-resident and host arrays."""
    import torch
    from oracle import c_oracle as co, np_oracle as npo
    from voiceconversion_jl_amd import _lib
    w, mu, sig = npo.synth_model(1900 + D + M, 2 * D, M, lam_lo=lam_lo)
    if M >= 4:
        mu[M - 1], sig[M - 1] = mu[0], sig[0]
        w = w.copy(); w[M - 1] = w[0]; w[1] = 0.0; w /= w.sum()
    X = npo.sample_frames(1901, w, mu, sig, T, 0, D)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    auto = vc.predict(g.px, X.T)                   # the library's own choice (screened where the model's frames leave few survivors)
    _lib.debug_force(_lib.DBG_PREDICT_SCREEN)      # ... and the screen whatever the model: on a broad one nearly every mixture survives
    try:
        fast = vc.predict(g.px, X.T)
        assert np.array_equal(vc.predict(g.px, X.T), fast)
        dev = vc.predict(g.px, torch.from_numpy(X).cuda().t())
        assert np.array_equal(np.asarray(dev.cpu()), fast)
    finally:
        _lib.debug_force(0)
    assert np.array_equal(auto, fast)
    _lib.debug_force(_lib.DBG_PREDICT_NO_SCREEN)
    try:
        early = vc.predict(g.px, X.T)
    finally:
        _lib.debug_force(0)
    _lib.debug_force(_lib.DBG_PREDICT_NO_EARLY_EXIT)
    try:
        full = vc.predict(g.px, X.T)
    finally:
        _lib.debug_force(0)
    assert np.array_equal(fast, early) and np.array_equal(fast, full)
    ref = co.GMMMap(w, mu, sig).predict(X[:3000])
    assert np.array_equal(fast[:3000], ref)
    if M >= 4:
        assert not np.any(fast == M) and not np.any(fast == 2)


def test_screened_predict_on_the_reference_model(vc, fixture_model):
    w, mu, sig = fixture_model
    from oracle import c_oracle as co, np_oracle as npo
    from voiceconversion_jl_amd import _lib
    X = npo.sample_frames(77, w, mu, sig, 30_000, 0, 40)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    early = vc.predict(g.px, X.T)                  # (a broad model: the library keeps the early-exit kernel)
    _lib.debug_force(_lib.DBG_PREDICT_SCREEN)
    try:
        fast = vc.predict(g.px, X.T)
    finally:
        _lib.debug_force(0)
    assert np.array_equal(fast, early) and np.array_equal(fast[:2000], co.GMMMap(w, mu, sig).predict(X[:2000]))
