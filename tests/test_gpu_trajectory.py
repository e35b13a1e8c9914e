"""GPU parity: TrajectoryGMMMap fvconvert / vc / push_delta vs golden vectors and the C oracle.
Tolerance: the reference solves (W'D^-1W) y = W'D^-1E with a sparse direct solver, the GPU with a banded
Cholesky; cond(P) ~ 1e6, so the two agree to ~1e-9; the test holds 1e-6 relative (north_star: 1e-5)."""
import numpy as np
import pytest

from conftest import julia_model, load_golden, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture(scope="module")
def vc():
    import voiceconversion_jl_amd as m
    assert m.device_count() >= 1
    return m


def test_accessors(vc, fixture_model):
    """test/trajectory_gmmmap.jl:36-50 (with the 80-dim fixture read as static 20 + delta 20)"""
    g = vc.GMMMap(*julia_model(*fixture_model))
    t = vc.TrajectoryGMMMap(g, 100)
    assert len(t) == 100 and vc.dim(t) == 40 and vc.ncomponents(t) == 32 and vc.size(t) == (40, 100)


def test_golden_fixture_model(vc, fixture_model):
    z = load_golden("trajectory_fixture_model.npz")
    g = vc.GMMMap(*julia_model(*fixture_model))
    t = vc.TrajectoryGMMMap(g, 100)
    assert np.array_equal(vc.push_delta(z["static"].T), z["X"].T)
    Y = vc.fvconvert(t, z["X"].T)
    assert Y.shape == (20, 100)
    assert relerr(Y, z["Y"].T) < TOL
    assert np.array_equal(vc.predict(g.px, z["X"].T), z["mhat"])
    # vc(): chunks of length(t) = 30 frames, power row kept (src/common.jl:31-63)
    t30 = vc.TrajectoryGMMMap(g, 30)
    out = vc.vc(t30, z["vc_fm"].T)
    assert out.shape == (21, 100)
    assert np.array_equal(out[0], z["vc_fm"][:, 0])
    assert relerr(out[1:], z["vc_out_L30"][:, 1:].T) < TOL
    with pytest.raises(vc.DimensionMismatch):
        vc.fvconvert(t, np.zeros((38, 10)))                          # src/trajectory_gmmmap.jl:68


@pytest.mark.parametrize("D,M,Ts", [(40, 8, [300]), (12, 4, [1, 2, 3, 4, 5, 50]), (20, 6, [64, 7, 129]),
                                    (16, 4, [40, 3]), (25, 4, [33, 2]), (32, 3, [21]),
                                    (13, 3, [45, 1, 6]), (35, 4, [60, 17]), (7, 2, [30, 2]), (27, 3, [25])])
def test_vs_oracle_batch(vc, D, M, Ts):
    """Config-5 shape (static D=40, X dim 80) at a length the oracle finishes in seconds, plus the short-utterance
    edge cases T = 1..5 where the stencil loses neighbours; D = 16 / 32 (the rhs row opens a tile of its own) and the
    odd D = 25 (8-byte stencil / panel accesses) cover the other instantiations of the blocked solver; D = 13, 35, 7, 27
    have none and run padded in the next larger one (16, 40, 12, 30)."""
    from oracle import c_oracle as co, np_oracle as npo
    w, mu, sig = npo.synth_model(500 + D, 4 * D, M, lam_lo=1e-3)
    ref = co.TrajectoryGMMMap(co.GMMMap(w, mu, sig))
    g = vc.GMMMap(*julia_model(w, mu, sig))
    t = vc.TrajectoryGMMMap(g, max(Ts))
    rng = np.random.default_rng(D)
    Xs = []
    for T in Ts:
        static = npo.sample_frames(int(rng.integers(1 << 30)), w, mu, sig, T, 0, D)
        static = np.cumsum(static, axis=0) / np.sqrt(np.arange(1, T + 1))[:, None]
        Xs.append(npo.push_delta(static))
    Ys = t.fvconvert_batch([x.T for x in Xs])
    for x, y in zip(Xs, Ys):
        yref, mh, _ = ref.fvconvert(x)
        assert relerr(y, yref.T) < TOL
    y0 = vc.fvconvert(t, Xs[0].T)
    assert np.array_equal(y0, Ys[0])                                 # batch == single, bit for bit


@pytest.mark.parametrize("D", [13, 35, 41, 46])
def test_padded_blocked_solver_against_the_runtime_dimension_kernel(vc, D):
    """A static dimension without an instantiation of the blocked solver: solved in the next larger one with the extra
    dimensions decoupled (unit diagonal, zero right-hand side) -- the same trajectories as the runtime-D kernel."""
    from voiceconversion_jl_amd import _lib
    from oracle import np_oracle as npo
    M = 4
    w, mu, sig = npo.synth_model(880 + D, 4 * D, M, lam_lo=1e-3)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    t = vc.TrajectoryGMMMap(g, 90)
    rng = np.random.default_rng(D)
    Xs = []
    for T in (90, 31, 2):
        static = npo.sample_frames(int(rng.integers(1 << 30)), w, mu, sig, T, 0, D)
        static = np.cumsum(static, axis=0) / np.sqrt(np.arange(1, T + 1))[:, None]
        Xs.append(npo.push_delta(static).T)
    Yb = t.fvconvert_batch(Xs)
    _lib.debug_force(_lib.DBG_TRAJ_GENERIC)
    try:
        Yr = t.fvconvert_batch(Xs)
    finally:
        _lib.debug_force(0)
    for a, b in zip(Yb, Yr):
        assert a.shape == b.shape and relerr(a, b) < 1e-9


def test_constructW_structure(vc):
    """test/trajectory_gmmmap.jl:1-34, 53: constructW(30, 40)"""
    import scipy.sparse as sp
    D, T = 30, 40
    W = vc.constructW(D, T)
    assert sp.issparse(W) and W.shape == (2 * D * T, D * T)
    z = load_golden("trajectory_fixture_model.npz")
    ref = sp.coo_matrix((z["W_vals"], (z["W_rows"] - 1, z["W_cols"] - 1)), shape=W.shape).tocsc()
    assert abs(W - ref).max() == 0
    Wd = W.toarray()
    I = np.eye(D)
    for t in range(T):
        s = 2 * D * t
        assert np.array_equal(Wd[s:s + D, t * D:(t + 1) * D], I)
        if t >= 1:
            assert np.array_equal(Wd[s + D:s + 2 * D, (t - 1) * D:t * D], -0.5 * I)
        if t < T - 1:
            assert np.array_equal(Wd[s + D:s + 2 * D, (t + 1) * D:(t + 2) * D], 0.5 * I)
    assert W.nnz == D * T + 2 * D * (T - 1)


def test_blocked_and_generic_solvers_agree(vc):
    """The MFMA-blocked banded Cholesky (default) against the runtime-D LDS-window kernel (the path of static dimensions
    without a blocked instantiation) and the MFMA g_t kernel against the one-workgroup-per-frame kernel: two independent
    device implementations of src/trajectory_gmmmap.jl:85-105 on the same utterances."""
    from voiceconversion_jl_amd import _lib
    from oracle import np_oracle as npo
    D, M = 40, 8
    w, mu, sig = npo.synth_model(777, 4 * D, M, lam_lo=1e-3)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    t = vc.TrajectoryGMMMap(g, 400)
    rng = np.random.default_rng(5)
    Xs = []
    for T in (400, 57, 1):
        static = npo.sample_frames(int(rng.integers(1 << 30)), w, mu, sig, T, 0, D)
        static = np.cumsum(static, axis=0) / np.sqrt(np.arange(1, T + 1))[:, None]
        Xs.append(npo.push_delta(static).T)
    Yb = t.fvconvert_batch(Xs)
    _lib.debug_force(_lib.DBG_TRAJ_GENERIC | _lib.DBG_TRAJ_G_SCALAR)
    try:
        Yr = t.fvconvert_batch(Xs)
    finally:
        _lib.debug_force(0)
    for a, b in zip(Yb, Yr):
        assert relerr(a, b) < 1e-9


@pytest.mark.parametrize("solver", ["blk", "generic"])
def test_not_positive_definite_normal_matrix(vc, solver):
    """A joint covariance whose x block is PD but whose conditional covariance Syy - A Sxy is negative definite passes
    the GMMMap constructor (only p(x) is factorised, src/gmm.jl:17) and makes W'D^-1W indefinite: the reference's
    Cholesky-based solve would throw; both device solvers report it through the status flag -> PosDefException."""
    D, M, T = 12, 2, 9
    Dj = 4 * D
    rng = np.random.default_rng(3)
    w = np.array([0.5, 0.5])
    mu = rng.standard_normal((M, Dj))
    I = np.eye(2 * D)
    sig = np.stack([np.block([[I, 2.0 * I], [2.0 * I, I]])] * M)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    t = vc.TrajectoryGMMMap(g, T)
    from voiceconversion_jl_amd import _lib
    _lib.debug_force(_lib.DBG_TRAJ_GENERIC if solver == "generic" else 0)
    try:
        with pytest.raises(vc.PosDefException):
            vc.fvconvert(t, rng.standard_normal((2 * D, T)))
    finally:
        _lib.debug_force(0)


def test_more_utterances_than_compute_units(vc):
    """A batch larger than the device's CU count: workgroups loop over utterances, so every per-utterance piece of
    workgroup state (window buffers, published-column ring, sequence tags, flags) is re-initialised.  Each result must
    equal, bit for bit, the single-utterance call."""
    from oracle import c_oracle as co, np_oracle as npo
    D, M = 12, 4
    w, mu, sig = npo.synth_model(4242, 4 * D, M, lam_lo=1e-3)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    t = vc.TrajectoryGMMMap(g, 16)
    rng = np.random.default_rng(11)
    Ts = rng.integers(1, 17, size=700)
    Xs = []
    for T in Ts:
        static = npo.sample_frames(int(rng.integers(1 << 30)), w, mu, sig, int(T), 0, D)
        Xs.append(npo.push_delta(static).T)
    Ys = t.fvconvert_batch(Xs)
    ref = co.TrajectoryGMMMap(co.GMMMap(w, mu, sig))
    for i in list(range(0, 700, 37)) + [699]:
        assert np.array_equal(vc.fvconvert(t, Xs[i]), Ys[i])
        assert relerr(Ys[i], ref.fvconvert(Xs[i].T)[0].T) < TOL
    # 700 solves on 256 CUs: up to static D = 30 the launch holds TWO workgroups per CU (two scalar chains share its SIMDs);
    # forced back to one per CU the results must not move by a bit
    from voiceconversion_jl_amd import _lib
    _lib.debug_force(_lib.DBG_TRAJ_ONE_WG_PER_CU)
    try:
        Y1 = t.fvconvert_batch(Xs)
    finally:
        _lib.debug_force(0)
    assert all(np.array_equal(a, b) for a, b in zip(Ys, Y1))


def test_repeated_runs_are_bit_identical(vc):
    """The blocked solver's two pivot waves hand columns over through LDS without a barrier (sequence tags), the GV
    ascent has a gather team beside its MFMA team: any ordering slip would show as run-to-run differences.  A ragged
    batch converted six times must give the same bits every time."""
    from oracle import np_oracle as npo
    D, M = 40, 8
    w, mu, sig = npo.synth_model(31, 4 * D, M, lam_lo=1e-3)
    t = vc.TrajectoryGMMMap(vc.GMMMap(*julia_model(w, mu, sig)), 200)
    rng = np.random.default_rng(1)
    Xs = []
    for T in rng.integers(2, 200, size=120):
        st = npo.sample_frames(int(rng.integers(1 << 30)), w, mu, sig, int(T), 0, D)
        Xs.append(npo.push_delta(st).T)
    ref = t.fvconvert_batch(Xs)
    muv = np.var(ref[0], axis=1, ddof=1) * 1.3 + 1e-3
    tgv = vc.TrajectoryGVGMMMap(t, muv, np.diag(muv ** 2 * 0.05))
    refg = tgv.fvconvert_batch(Xs, epochs=4, alpha=1e-5)
    for _ in range(5):
        assert all(np.array_equal(a, b) for a, b in zip(t.fvconvert_batch(Xs), ref))
        assert all(np.array_equal(a, b) for a, b in zip(tgv.fvconvert_batch(Xs, epochs=4, alpha=1e-5), refg))


def test_batch_of_empty_utterances_on_a_fresh_handle(vc):
    """The device-resident batch entry points return OK for a batch whose utterances are all empty, also when nothing has
    run on the handle yet (no status word allocated)."""
    import ctypes as C
    import torch
    from oracle import np_oracle as npo
    from voiceconversion_jl_amd import _lib
    D, M = 12, 4
    w, mu, sig = npo.synth_model(91, 4 * D, M)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    t = vc.TrajectoryGMMMap(g, 50)
    dX = torch.zeros(8, dtype=torch.float64, device="cuda")
    dY = torch.zeros(8, dtype=torch.float64, device="cuda")
    off = np.zeros(3, dtype=np.int64)
    T = np.zeros(3, dtype=np.int64)
    rc = _lib.lib.vcmi_traj_convert_batch_dev(t._h, 3, C.c_void_p(dX.data_ptr()), _lib.iptr(off), _lib.iptr(T),
                                              C.c_void_p(dY.data_ptr()), _lib.iptr(off), None)
    assert rc == 0
    muv = np.ones(D)
    tgv = vc.TrajectoryGVGMMMap(vc.TrajectoryGMMMap(g, 50), muv, np.eye(D))
    rc = _lib.lib.vcmi_trajgv_convert_batch_dev(tgv._h, 3, C.c_void_p(dX.data_ptr()), _lib.iptr(off), _lib.iptr(T), 5, 1e-5,
                                                C.c_void_p(dY.data_ptr()), _lib.iptr(off), None)
    assert rc == 0
    assert t.fvconvert_batch([np.zeros((2 * D, 0), order="F")])[0].shape == (D, 0)


@pytest.mark.parametrize("D,M,Ts", [(47, 3, [40, 1]), (48, 4, [75, 2, 3]), (64, 3, [50]), (57, 2, [33, 4]), (72, 2, [20, 5]), (90, 2, [12])])
def test_static_dimensions_beyond_the_blocked_solver(vc, D, M, Ts):
    """The reference has no limit on the static dimension (src/trajectory_gmmmap.jl:65-110); the blocked solver ends at 46.
    Beyond it `traj_solve_big_kernel` runs: the same banded Cholesky with the window's lower triangle packed in LDS up to
    D = 64 and in HBM above -- a fallback (17 x the blocked solver's time at D = 48), but the same answers: against the oracle,
    batch == single bit for bit, and the not-PD report."""
    from oracle import c_oracle as co, np_oracle as npo
    w, mu, sig = npo.synth_model(500 + D, 4 * D, M, lam_lo=1e-3)
    ref = co.TrajectoryGMMMap(co.GMMMap(w, mu, sig))
    g = vc.GMMMap(*julia_model(w, mu, sig))
    t = vc.TrajectoryGMMMap(g, max(Ts))
    rng = np.random.default_rng(D)
    Xs = []
    for T in Ts:
        static = npo.sample_frames(int(rng.integers(1 << 30)), w, mu, sig, T, 0, D)
        static = np.cumsum(static, axis=0) / np.sqrt(np.arange(1, T + 1))[:, None]
        Xs.append(npo.push_delta(static))
    Ys = t.fvconvert_batch([x.T for x in Xs])
    for x, y in zip(Xs, Ys):
        yref, _, _ = ref.fvconvert(x)
        assert relerr(y, yref.T) < TOL
    assert np.array_equal(vc.fvconvert(t, Xs[0].T), Ys[0])


def test_not_positive_definite_beyond_the_blocked_solver(vc):
    D, M, T = 50, 2, 7
    rng = np.random.default_rng(3)
    I = np.eye(2 * D)
    sig = np.stack([np.block([[I, 2.0 * I], [2.0 * I, I]])] * M)
    g = vc.GMMMap(*julia_model(np.array([0.5, 0.5]), rng.standard_normal((M, 4 * D)), sig))
    t = vc.TrajectoryGMMMap(g, T)
    with pytest.raises(vc.PosDefException):
        vc.fvconvert(t, rng.standard_normal((2 * D, T)))
