"""GPU: one host process driving a device group (vcmi_set_devices, include/vcmi.h) and the pipelined host-pointer path.

The box has one GPU, so the group is [0, 0] (two worker threads sharing the device: exercises the sharding, the replicas
and the join) for the paths without a collective, and [0] for the E-step, whose statistics then go through a real
ncclAllReduce on a one-rank RCCL communicator.  Every result must equal the single-device result: bit for bit where
the arithmetic per frame / pair / utterance is unchanged, to 1e-12 for the E-step sums."""
import numpy as np
import pytest

from conftest import julia_model, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture()
def vc():
    import voiceconversion_jl_amd as m
    assert m.device_count() >= 1
    m.set_devices([])
    yield m
    m.set_devices([])


def _model(npo, seed, Dj, M):
    return npo.synth_model(seed, Dj, M)


def test_group_roundtrip_and_errors(vc):
    assert vc.get_devices() == []
    vc.set_devices([0, 0])
    assert vc.get_devices() == [0, 0]
    with pytest.raises(vc.VCMIError, match="not visible"):
        vc.set_devices([0, 99])
    assert vc.get_devices() == [0, 0]           # a refused list leaves the group as it was
    vc.set_devices([])
    assert vc.get_devices() == []


@pytest.mark.parametrize("T", [5000, 300_001])
def test_frame_paths_sharded_equal_single_device(vc, T):
    from oracle import np_oracle as npo
    D, M = 24, 8
    w, mu, sig = _model(npo, 91, 2 * D, M)
    X = npo.sample_frames(92, w, mu, sig, T, 0, D)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    Xj = np.asfortranarray(X.T)
    fm = np.asfortranarray(np.vstack([np.arange(T, dtype=np.float64)[None, :], X.T]))
    one = (vc.fvconvert(g, Xj), vc.vc(g, fm), vc.predict_proba(g.px, Xj), vc.predict(g.px, Xj))
    vc.set_devices([0, 0])
    two = (vc.fvconvert(g, Xj), vc.vc(g, fm), vc.predict_proba(g.px, Xj), vc.predict(g.px, Xj))
    for a, b in zip(one, two):
        assert np.array_equal(a, b)
    assert np.array_equal(two[1][0], fm[0])    # power row untouched (src/common.jl:23)


def test_host_pipeline_matches_device_resident_path(vc):
    """The chunked pinned-staging pipeline (several chunks, ragged last chunk, strided input) against one launch on
    device-resident data."""
    import torch
    from oracle import np_oracle as npo
    D, M, T = 40, 16, 250_007
    w, mu, sig = _model(npo, 93, 2 * D, M)
    X = npo.sample_frames(94, w, mu, sig, T, 0, D)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    dev = vc.fvconvert(g, torch.from_numpy(X).cuda().t()).t().cpu().numpy()
    host = vc.fvconvert(g, np.asfortranarray(X.T))
    assert np.array_equal(host.T, dev)
    fm = np.asfortranarray(np.vstack([np.full((1, T), 7.5), X.T]))
    out = vc.vc(g, fm)
    assert np.array_equal(out[1:].T, dev) and np.all(out[0] == 7.5)
    # per-frame call of the reference (T = 1) still works through the same path
    assert np.array_equal(vc.fvconvert(g, X[17]), dev[17])


def test_dtw_and_align_batches_sharded(vc):
    rng = np.random.default_rng(5)
    D = 12
    tm, sq = [], []
    for _ in range(23):
        S, T = int(rng.integers(20, 90)), int(rng.integers(20, 90))
        t = rng.standard_normal((D, S))
        tm.append(np.asfortranarray(t))
        sq.append(np.asfortranarray(t[:, np.sort(rng.integers(0, S, T))] + 0.2 * rng.standard_normal((D, T))))
    d = vc.DTW(fstep=0, bstep=2)
    one_paths = vc.fit_batch(d, tm, sq)
    one_al = vc.align_batch(tm, sq)
    vc.set_devices([0, 0])
    two_paths = vc.fit_batch(d, tm, sq)
    two_al = vc.align_batch(tm, sq)
    for a, b in zip(one_paths, two_paths):
        assert np.array_equal(a, b)
    for (s1, n1), (s2, n2) in zip(one_al, two_al):
        assert np.array_equal(n1, n2)


def test_trajectory_batches_sharded(vc):
    from oracle import np_oracle as npo
    D, M = 12, 4
    w, mu, sig = npo.synth_model(4242, 4 * D, M, lam_lo=1e-3)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    t = vc.TrajectoryGMMMap(g, 50)
    rng = np.random.default_rng(8)
    Xs = []
    for T in [50, 31, 2, 77, 50, 50, 9, 64, 13, 40, 50]:
        st = np.cumsum(npo.sample_frames(int(rng.integers(1 << 30)), w, mu, sig, T, 0, D), axis=0) / np.sqrt(np.arange(1, T + 1))[:, None]
        Xs.append(np.asfortranarray(npo.push_delta(st).T))
    one = t.fvconvert_batch(Xs)
    fm = np.asfortranarray(np.vstack([np.ones((1, 777)), np.tile(Xs[3], (1, 11))[:, :777]]))
    one_vc = vc.vc(t, fm)
    assert len(t) == 777 - 15 * 50              # the reference's quirk: W is left at the last chunk's length
    t2 = vc.TrajectoryGMMMap(g, 50)
    muv = one[0].var(axis=1, ddof=1) * 1.2
    Sv = np.diag(muv ** 2 * 0.1)
    tgv = vc.TrajectoryGVGMMMap(t2, muv, Sv)
    one_gv = tgv.fvconvert_batch(Xs[:9], epochs=5, alpha=1e-5)
    vc.set_devices([0, 0])
    two = t.fvconvert_batch(Xs)
    t3 = vc.TrajectoryGMMMap(g, 50)
    two_vc = vc.vc(t3, fm)
    two_gv = tgv.fvconvert_batch(Xs[:9], epochs=5, alpha=1e-5)
    for a, b in zip(one, two):
        assert np.array_equal(a, b)
    assert np.array_equal(one_vc, two_vc)
    for a, b in zip(one_gv, two_gv):
        assert np.array_equal(a, b)


def test_estep_through_rccl_single_rank(vc):
    """Group [0]: the local statistics go through ncclAllReduce on a one-rank communicator (ncclCommInitAll) -- the RCCL
    path a multi-GPU host takes, as far as one GPU can exercise it."""
    from oracle import c_oracle as co, np_oracle as npo
    rng = np.random.default_rng(3)
    Dj, M, N = 80, 32, 30_000
    w, mu, _ = npo.synth_model(21, Dj, M)
    var = np.exp(rng.uniform(np.log(1e-3), 0.0, (M, Dj)))
    comp = rng.choice(M, size=N, p=w)
    X = mu[comp] + rng.standard_normal((N, Dj)) * np.sqrt(var[comp])
    one = vc.estep_diag(X.T, w, mu.T, var.T)
    vc.set_devices([0])
    two = vc.estep_diag(X.T, w, mu.T, var.T)
    for a, b in zip(one, two):
        assert relerr(np.atleast_1d(a), np.atleast_1d(b)) <= 1e-12
    r0, r1, r2, rl = co.estep_diag(X, w, mu, var)
    assert relerr(two[1], r1.T) < 1e-9
    # full covariance through the same path
    wf, muf, sigf = npo.synth_model(22, 16, 3, lam_lo=1e-1)
    Xf = npo.sample_frames(23, wf, muf, sigf, 9000, 0, 16)
    sj = np.asfortranarray(np.transpose(sigf, (2, 1, 0)))
    vc.set_devices([])
    f1 = vc.estep_full(Xf.T, wf, muf.T, sj)
    vc.set_devices([0])
    f2 = vc.estep_full(Xf.T, wf, muf.T, sj)
    for a, b in zip(f1, f2):
        assert relerr(np.atleast_1d(a), np.atleast_1d(b)) <= 1e-12


def test_two_physical_devices_when_the_box_has_them(vc):
    """Runs by itself wherever two GPUs are visible (the 1-GPU boxes of this pool skip it): the group [0, 1] shards
    fvconvert over both devices (bit-identical to one device) and the E-step's local statistics meet in ONE ncclAllReduce
    between two single-process communicators (ncclCommInitAll) -- equal to one device to 1e-12 (the sum's order differs)."""
    if vc.device_count() < 2:
        pytest.skip(f"needs 2 GPUs; this box has {vc.device_count()}")
    from oracle import np_oracle as npo
    rng = np.random.default_rng(5)
    Dj, M, N = 80, 128, 200_000
    w, mu, sig = npo.synth_model(31, Dj, 16)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    X = np.asfortranarray(npo.sample_frames(32, w, mu, sig, 50_000, 0, 40).T)
    vc.set_devices([])
    y1 = vc.fvconvert(g, X)
    we, mue, _ = npo.synth_model(33, Dj, M)
    var = np.exp(rng.uniform(np.log(1e-3), 0.0, (M, Dj)))
    comp = rng.choice(M, size=N, p=we)
    Xe = mue[comp] + rng.standard_normal((N, Dj)) * np.sqrt(var[comp])
    one = vc.estep_diag(Xe.T, we, mue.T, var.T)
    try:
        vc.set_devices([0, 1])
        y2 = vc.fvconvert(g, X)
        two = vc.estep_diag(Xe.T, we, mue.T, var.T)
    finally:
        vc.set_devices([])
    assert np.array_equal(y1, y2)
    for a, b in zip(one, two):
        assert relerr(np.atleast_1d(a), np.atleast_1d(b)) <= 1e-12


def test_estep_refuses_duplicate_devices(vc):
    vc.set_devices([0, 0])
    with pytest.raises(vc.VCMIError, match="distinct devices"):
        vc.estep_diag(np.zeros((4, 100)), np.ones(2) / 2, np.zeros((4, 2)), np.ones((4, 2)))


def test_dtw_dev_calls_on_two_nonblocking_streams(vc):
    """consecutive vcmi_dtw_fit_batch_dev calls on different non-blocking streams share the descriptor and
    observation workspaces; the library orders them itself (event wait + asynchronous descriptor copies)."""
    import torch
    from voiceconversion_jl_amd import _lib
    rng = np.random.default_rng(11)
    D = 8

    def batch(n):
        feats, toff, soff, poff, S, T, fo, po = [], [], [], [], [], [], 0, 0
        for _ in range(n):
            s, t = int(rng.integers(40, 120)), int(rng.integers(40, 120))
            a = rng.standard_normal((s, D)); b = a[np.sort(rng.integers(0, s, t))] + 0.3 * rng.standard_normal((t, D))
            toff.append(fo); feats.append(a.ravel()); fo += a.size
            soff.append(fo); feats.append(b.ravel()); fo += b.size
            poff.append(po); po += t; S.append(s); T.append(t)
        arr = lambda a: np.asarray(a, dtype=np.int64)  # noqa: E731
        return torch.from_numpy(np.concatenate(feats)).cuda(), arr(toff), arr(S), arr(soff), arr(T), arr(poff), po

    A, B = batch(300), batch(300)

    def run(bt, out, stream):
        f, toff, S, soff, T, poff, _ = bt
        _lib.check(_lib.lib.vcmi_dtw_fit_batch_dev(len(S), f.data_ptr(), _lib.iptr(toff), _lib.iptr(S), _lib.iptr(soff), _lib.iptr(T),
                                                   D, 0, 2, out.data_ptr(), _lib.iptr(poff), stream.cuda_stream))

    refA = torch.empty(A[6], dtype=torch.int64, device="cuda"); refB = torch.empty(B[6], dtype=torch.int64, device="cuda")
    s0 = torch.cuda.current_stream()
    run(A, refA, s0); torch.cuda.synchronize(); run(B, refB, s0); torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(5):
        outA = torch.zeros_like(refA); outB = torch.zeros_like(refB)
        torch.cuda.synchronize()
        run(A, outA, s1)
        run(B, outB, s2)          # no host synchronisation in between
        torch.cuda.synchronize()
        assert torch.equal(outA, refA) and torch.equal(outB, refB)
