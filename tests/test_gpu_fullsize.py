"""GPU parity at BASELINE.json's full configuration sizes, through size-independent properties plus oracle
spot checks on a bounded sample (the oracle cannot finish the full sizes in seconds)."""
import numpy as np
import pytest

from conftest import julia_model, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vc():
    import voiceconversion_jl_amd as m
    assert m.device_count() >= 1
    return m


def test_config3_estep_1p25M_frames(vc):
    """configs[2] per-GPU shard: Dj=80, M=128, N=1.25e6.  Responsibilities sum to one per frame (sum S0 = N),
    statistics are additive over frame shards (what the all-reduce relies on), run-to-run bit-identical, and the
    first 20000 frames match the oracle."""
    import torch
    from oracle import c_oracle as co, np_oracle as npo
    Dj, M, N = 80, 128, 1_250_000
    w, mu, _ = npo.synth_model(1003, Dj, M)
    rg = np.random.default_rng(1003)
    var = np.exp(rg.uniform(np.log(1e-3), 0.0, (M, Dj)))
    comp = rg.choice(M, size=N, p=w)
    X = torch.from_numpy(mu[comp] + rg.standard_normal((N, Dj)) * np.sqrt(var[comp])).cuda()
    a = vc.estep_diag_dev(X.t(), w, mu.T, var.T)
    assert torch.equal(a, vc.estep_diag_dev(X.t(), w, mu.T, var.T))
    cut = 777_777
    h = vc.estep_diag_dev(X[:cut].t(), w, mu.T, var.T) + vc.estep_diag_dev(X[cut:].t(), w, mu.T, var.T)
    assert float((a - h).abs().max() / a.abs().max()) < 1e-12
    S0, S1, S2, ll = vc.unpack_stats(a.cpu().numpy(), Dj, M)
    assert abs(S0.sum() - N) < 1e-7 * N and np.all(S0 >= 0) and np.all(S2 >= 0)
    n = 20000
    r0, r1, r2, rl = co.estep_diag(X[:n].cpu().numpy(), w, mu, var)
    g0, g1, g2, gl = vc.estep_diag(X[:n].t(), w, mu.T, var.T)
    assert relerr(g0, r0) < 1e-9 and relerr(g1, r1.T) < 1e-9 and relerr(g2, r2.T) < 1e-9 and abs(gl - rl) < 1e-9 * abs(rl)


def test_config4_dtw_1000_pairs(vc):
    """configs[3]: 1000 pairs of ~500x500 frames, D=40, bstep=2/fstep=0.  Every path is a valid warping path
    (in range, non-decreasing, steps <= bstep), a second run is identical, and 25 sampled pairs are bit-exact
    against the oracle."""
    from oracle import c_oracle as co
    rng = np.random.default_rng(1004)
    pairs = []
    for _ in range(1000):
        S, T = int(rng.integers(450, 551)), int(rng.integers(450, 551))
        t = rng.standard_normal((S, 40))
        idx = np.clip(np.sort(rng.integers(0, S, T)), 0, S - 1)
        pairs.append((t, t[idx] + 0.2 * rng.standard_normal((T, 40))))
    d = vc.DTW(fstep=0, bstep=2)
    paths = vc.fit_batch(d, [t.T for t, _ in pairs], [s.T for _, s in pairs])
    again = vc.fit_batch(d, [t.T for t, _ in pairs[:50]], [s.T for _, s in pairs[:50]])
    for (t, s), p in zip(pairs, paths):
        assert p.shape == (s.shape[0],) and p.min() >= 1 and p.max() <= t.shape[0]
        dp = np.diff(p)
        assert dp.min() >= 0 and dp.max() <= 2
    for p, q in zip(paths[:50], again):
        assert np.array_equal(p, q)
    for k in rng.choice(1000, size=25, replace=False):
        t, s = pairs[k]
        assert np.array_equal(paths[k], co.dtw_fit(t, s, 0, 2, tables=False)), k


def test_config2_convert_1M_distinct_frames_sampled_at_random_offsets(vc):
    """configs[1] at full size with 10^6 DISTINCT frames (not a tile repeated): 512 frames at seeded random positions -- the
    last, partial workgroup's included -- against the oracle, on the device path and through host pointers (chunked pipeline),
    in the caller's order although the kernel works on grouped frames; a second run is bit-identical; and the same 512 frames
    converted on their own (an utterance-sized call: ungrouped, other tiles) agree to rounding."""
    import torch
    from oracle import c_oracle as co, np_oracle as npo
    D, M, T = 40, 64, 1_000_000
    w, mu, sig = npo.synth_model(1002, 2 * D, M)
    X = npo.sample_frames(20_002, w, mu, sig, T, 0, D)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    Xd = torch.from_numpy(X).cuda()
    Y = vc.fvconvert(g, Xd.t()).t()
    assert bool(torch.isfinite(Y).all()) and torch.equal(Y, vc.fvconvert(g, Xd.t()).t())
    rng = np.random.default_rng(77)
    pos = np.unique(np.concatenate([rng.choice(T - 64, size=448, replace=False), np.arange(T - 64, T)]))     # 1e6 % 128 = 64
    ref = co.GMMMap(w, mu, sig).fvconvert(X[pos])
    got = Y[torch.from_numpy(pos).cuda()].cpu().numpy()
    err = np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert err.max() < 1e-9, (float(err.max()), int(pos[np.argmax(err)]))
    small = vc.fvconvert(g, np.asfortranarray(X[pos].T)).T
    assert (np.linalg.norm(small - got, axis=1) / np.linalg.norm(got, axis=1)).max() < 1e-13
    Yh = vc.fvconvert(g, np.asfortranarray(X.T))               # host pointers: 320 MB up, 320 MB down, chunks grouped one by one
    errh = np.linalg.norm(Yh.T[pos] - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert errh.max() < 1e-9
    assert (np.linalg.norm(Yh.T - Y.cpu().numpy(), axis=1) / np.linalg.norm(Yh.T, axis=1)).max() < 1e-13


def test_config5_trajectory_T2000(vc):
    """configs[4]: static D=40 (X dim 80), M=64, T about 2000.  SIXTEEN DIFFERENT utterances of unequal length (1900..2100
    frames) in one batch, each against the oracle; a batch of identical utterances gives identical outputs; one utterance
    alone equals its result in the batch bit for bit."""
    from oracle import c_oracle as co, np_oracle as npo
    D, M = 40, 64
    w, mu, sig = npo.synth_model(1005, 4 * D, M, lam_lo=1e-3)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    tj = vc.TrajectoryGMMMap(g, 2000)
    rng = np.random.default_rng(1005)
    Ts = [2000] + [int(v) for v in rng.integers(1900, 2101, size=15)]
    Xs = []
    for k, T in enumerate(Ts):
        st = npo.sample_frames(7 + k, w, mu, sig, T, 0, D)
        st = np.cumsum(st, axis=0) / np.sqrt(np.arange(1, T + 1))[:, None]
        Xs.append(npo.push_delta(st))
    Ys = tj.fvconvert_batch([X.T for X in Xs])
    ref = co.TrajectoryGMMMap(co.GMMMap(w, mu, sig))
    for X, Y in zip(Xs, Ys):
        assert Y.shape == (D, X.shape[0]) and np.all(np.isfinite(Y))
        Yref, _, _ = ref.fvconvert(X)
        assert relerr(Y, Yref.T) < 1e-6
    same = tj.fvconvert_batch([Xs[0].T] * 6)
    assert all(np.array_equal(same[0], y) for y in same[1:]) and np.array_equal(same[0], Ys[0])
    assert np.array_equal(tj.fvconvert_batch([Xs[5].T])[0], Ys[5])
