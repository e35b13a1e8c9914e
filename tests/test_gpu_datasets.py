"""GPU parity for SURVEY 8(f) rank 3: mc2e, align_mcep (src/align.jl:38-55) and the device-resident joint training matrix
(ParallelDataset, src/datasets.jl:52-98) vs the C oracle.  The aligned frames and the kept-frame selection are exact
(DTW is bit-exact; the energies differ from the oracle by ~1e-15 relative, far from the threshold on this data); the
feature values are copies / exact halves, so the matrices are compared bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vc():
    import voiceconversion_jl_amd as m
    assert m.device_count() >= 1
    return m


def _mcep_pair(rng, S, T, D):
    """mel-cepstrum-like parallel utterances: decaying coefficients, c0 spread so that part of the frames is silence"""
    src = rng.standard_normal((S, D)) * np.exp(-0.3 * np.arange(D)) * 0.3
    src[:, 0] = rng.uniform(-9.0, 1.0, S)
    idx = np.clip(np.sort(rng.integers(0, S, T)), 0, S - 1)
    return src, src[idx] + 0.01 * rng.standard_normal((T, D))


def test_mc2e(vc):
    from oracle import c_oracle as co
    rng = np.random.default_rng(0)
    for D, T, fftlen, alpha in [(25, 200, 256, 0.41), (41, 37, 512, 0.35), (3, 5, 64, 0.0)]:
        mc = _mcep_pair(rng, T, T, D)[0]
        e = vc.mc2e(mc.T, alpha, fftlen)
        ref = co.mc2e(mc, alpha, fftlen)
        assert np.max(np.abs(e - ref) / ref) < 1e-12
        # the HIP path against the frequency-domain evaluation (oracle/crosscheck.py: numpy.fft, no SPTK recursion)
        from oracle import crosscheck as cc
        e_t, _ = cc.mc2e_frequency_domain(mc, alpha, fftlen)
        assert np.max(np.abs(e - e_t) / e_t) < 1e-11
    c = np.zeros((5, 1))
    c[0, 0] = 0.7                                      # only c0: h = [exp(c0), 0, ...] -> e = exp(2 c0)
    assert abs(vc.mc2e(c, 0.35, 64)[0] - np.exp(1.4)) < 1e-12


def test_align_mcep(vc):
    from oracle import c_oracle as co
    rng = np.random.default_rng(1)
    for S, T, D in [(60, 70, 25), (300, 280, 41), (17, 17, 5)]:
        src, tgt = _mcep_pair(rng, S, T, D)
        s_ref, t_ref = co.align_mcep(src, tgt, 0.41, 256)
        s, t = vc.align_mcep(src.T, tgt.T, 0.41, 256)
        assert 0 < s.shape[1] < S                     # some silence was removed, something was kept
        assert np.array_equal(s, s_ref.T) and np.array_equal(t, t_ref.T)
        s2, t2 = vc.align_mcep(src.T, tgt.T, 0.41, 256, remove_silence=False)
        a_ref = co.align(src, tgt)[0]
        assert np.array_equal(s2, src.T) and np.array_equal(t2, a_ref.T)
    with pytest.raises(vc.DimensionMismatch):
        vc.align_mcep(np.zeros((5, 4)), np.zeros((6, 4)), 0.41, 64)


@pytest.mark.parametrize("diff,ignore0th,add_delta", [(False, True, False), (True, True, True), (False, False, True)])
def test_parallel_dataset_on_device(vc, diff, ignore0th, add_delta):
    """raw parallel utterances -> align_mcep -> joint matrix on the device == oracle align_mcep + oracle assembly;
    then straight into the full-covariance E-step"""
    import torch
    from oracle import c_oracle as co
    rng = np.random.default_rng(2)
    shapes = [(120, 130, 25), (90, 80, 25), (257, 300, 25), (33, 40, 25)]
    pairs = [_mcep_pair(rng, S, T, D) for S, T, D in shapes]
    ds = vc.ParallelDataset([(s.T, t.T) for s, t in pairs], diff=diff, ignore0th=ignore0th, add_delta=add_delta,
                            alpha=0.41, fftlen=256)
    refs = []
    for s, t in pairs:
        sa, ta = co.align_mcep(s, t, 0.41, 256)
        refs.append(co.joint_features(sa, ta, ignore0th, add_delta, diff))
    want = np.concatenate(refs, axis=0)
    assert isinstance(ds.X, torch.Tensor) and ds.X.is_cuda and ds.X.stride(0) == 1
    assert list(ds.counts) == [r.shape[0] for r in refs] and len(ds) == want.shape[0]
    assert np.array_equal(ds.X.t().cpu().numpy(), want)
    # already aligned pairs, no silence removal: pure assembly
    al = [co.align_mcep(s, t, 0.41, 256, remove_silence=False) for s, t in pairs]
    ds2 = vc.ParallelDataset([(a.T, b.T) for a, b in al], diff=diff, ignore0th=ignore0th, add_delta=add_delta, align=False,
                             remove_silence=False)
    want2 = np.concatenate([co.joint_features(a, b, ignore0th, add_delta, diff) for a, b in al], axis=0)
    assert np.array_equal(ds2.X.t().cpu().numpy(), want2)
    # joint=false: the source and target halves as views (src/datasets.jl:92-96); standarize=true throws in the reference
    ds3 = vc.ParallelDataset([(a.T, b.T) for a, b in al], diff=diff, ignore0th=ignore0th, add_delta=add_delta, align=False,
                             remove_silence=False, joint=False)
    half = want2.shape[1] // 2
    assert np.array_equal(ds3.X.t().cpu().numpy(), want2[:, :half]) and np.array_equal(ds3.Y.t().cpu().numpy(), want2[:, half:])
    with pytest.raises(vc.DimensionMismatch):
        vc.ParallelDataset([(a.T, b.T) for a, b in al], align=False, standarize=True)
    # the matrix feeds the E-step as it is
    Dj, N = ds.X.shape
    r = vc.train_gmm(ds.X, n_components=2, n_iter=3, n_init=1, min_covar=1e-3, seed=1)
    assert r["means"].shape == (Dj, 2) and np.isfinite(r["loglik"]).all()
