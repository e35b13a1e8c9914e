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


def test_config5_trajectory_T2000(vc):
    """configs[4]: static D=40 (X dim 80), M=64, T=2000.  One utterance against the oracle, a batch of identical
    utterances gives identical outputs, and the result is finite."""
    from oracle import c_oracle as co, np_oracle as npo
    D, M, T = 40, 64, 2000
    w, mu, sig = npo.synth_model(1005, 4 * D, M, lam_lo=1e-3)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    tj = vc.TrajectoryGMMMap(g, T)
    st = npo.sample_frames(7, w, mu, sig, T, 0, D)
    st = np.cumsum(st, axis=0) / np.sqrt(np.arange(1, T + 1))[:, None]
    X = npo.push_delta(st)
    Ys = tj.fvconvert_batch([X.T] * 6)
    assert all(np.array_equal(Ys[0], y) for y in Ys[1:]) and np.all(np.isfinite(Ys[0]))
    Yref, _, _ = co.TrajectoryGMMMap(co.GMMMap(w, mu, sig)).fvconvert(X)
    assert relerr(Ys[0], Yref.T) < 1e-6
