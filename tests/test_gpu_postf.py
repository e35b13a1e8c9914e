"""GPU: the pre / post steps of `vc` as DEVICE-RESIDENT operations: push_delta
(src/datasets.jl:6-13) and the VarianceScaling post-filter (src/gv.jl:10-15) on device tensors, and `vc` with the post-filter
applied before the result leaves HBM (bin/vc.jl:75-82: push_delta -> vc; fvpostf! on the converted mel-cepstrum).
push_delta is bit-exact (a copy, one multiply-add pair with exact halves); the post-filter's sums have a fixed order of their
own, so it is held to 1e-12 against vco_variance_scaling, and bit for bit against the library's host-pointer entry."""
import numpy as np
import pytest

from conftest import frame_relerr, julia_model, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vc():
    import voiceconversion_jl_amd as m
    assert m.device_count() >= 1
    return m


@pytest.mark.parametrize("D,T", [(40, 2000), (25, 1), (24, 2), (7, 3), (40, 300_001)])
def test_push_delta_on_device_tensors(vc, D, T):
    import torch
    from oracle import c_oracle as co
    rng = np.random.default_rng(D + T)
    src = rng.standard_normal((T, D))
    ref = co.push_delta(src)                                   # (T, 2D)
    assert np.array_equal(vc.push_delta(np.asfortranarray(src.T)), ref.T)
    dev = vc.push_delta(torch.from_numpy(src).cuda().t())
    assert dev.is_cuda and tuple(dev.shape) == (2 * D, T) and np.array_equal(dev.cpu().numpy(), ref.T)
    # a view with a leading dimension: rows 2..D+1 of a (D+1,T) feature matrix, the way bin/vc.jl:77 slices src[2:end,:]
    fm = torch.from_numpy(np.hstack([rng.standard_normal((T, 1)), src])).cuda()
    assert np.array_equal(vc.push_delta(fm.t()[1:]).cpu().numpy(), ref.T)


@pytest.mark.parametrize("D,T", [(25, 700), (40, 300_000), (1, 2), (256, 50)])
def test_variance_scaling_on_device_tensors(vc, D, T):
    import torch
    from oracle import c_oracle as co
    rng = np.random.default_rng(D * 3 + T)
    src = rng.standard_normal((T, D)) * rng.uniform(0.1, 3.0, D) + rng.standard_normal(D)
    s2 = rng.uniform(0.5, 2.0, D)
    vs = vc.VarianceScaling(s2)
    ref = co.variance_scaling(src, s2)
    host = vc.fvpostf(vs, src.T)
    assert relerr(host, ref.T) < 1e-12
    d = torch.from_numpy(src).cuda()
    out = vc.fvpostf(vs, d.t())                                # new device tensor
    assert out.is_cuda and np.array_equal(out.cpu().numpy(), host) and np.array_equal(d.cpu().numpy(), src)
    assert vc.fvpostf_(vs, d.t()) is not None and np.array_equal(d.cpu().numpy().T, host)       # in place
    # in place on the feature rows of a (D+1,T) matrix: the power row is not touched
    if D < 256:
        fm = torch.from_numpy(np.hstack([rng.standard_normal((T, 1)), src])).cuda()
        p0 = fm[:, 0].clone()
        vc.fvpostf_(vs, fm.t()[1:])
        assert torch.equal(fm[:, 0], p0) and np.array_equal(fm[:, 1:].cpu().numpy().T, host)


def test_vc_with_the_post_filter_is_vc_then_fvpostf(vc, fixture_model):
    """vc(g, fm; postfilter) == fvpostf(vs, vc(g, fm)[2:end,:]) with the power row kept -- against the oracle's two steps and,
    bit for bit, against the library's own two host-pointer calls (the fused entry runs the same kernels on the same bytes)."""
    from oracle import c_oracle as co, np_oracle as npo
    w, mu, sig = fixture_model
    T = 5000
    X = npo.sample_frames(31, w, mu, sig, T, 0, 40)
    fm = np.asfortranarray(np.vstack([np.arange(T, dtype=np.float64)[None], X.T]))
    s2 = np.random.default_rng(1).uniform(0.5, 2.0, 40)
    vs = vc.VarianceScaling(s2)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    out = vc.vc(g, fm, postfilter=vs)
    two = vc.vc(g, fm)
    two[1:] = vc.fvpostf(vs, two[1:])
    # (to rounding: the fused entry groups the whole matrix's frames, the plain one each pipeline chunk's -- on a broad model
    # the order of a frame's sum over mixtures follows the grouping)
    assert np.array_equal(out[0], fm[0]) and frame_relerr(out[1:], two[1:]) < 1e-12
    ref = co.variance_scaling(co.GMMMap(w, mu, sig).fvconvert(X), s2)
    assert frame_relerr(out[1:], ref.T) < 1e-9
    with pytest.raises(vc.DimensionMismatch):
        vc.vc(g, fm, postfilter=vc.VarianceScaling(s2[:-1]))
    with pytest.raises(vc.DimensionMismatch):
        vc.vc(g, fm[:, :1], postfilter=vs)                     # the variance of one frame is undefined


@pytest.mark.parametrize("T,L", [(260, 100), (90, 100), (400, 400)])
def test_trajectory_vc_with_the_post_filter(vc, T, L):
    """push_delta -> vc(TrajectoryGMMMap) -> fvpostf!, the bin/vc.jl pipeline: the chunks of length(c) frames are converted by
    the device-resident batch path, the filter sees the whole (D,T) result; compared with the oracle's three steps and with
    the library's separate calls"""
    from oracle import c_oracle as co, np_oracle as npo
    Ds, M = 12, 4
    w, mu, sig = npo.synth_model(311, 4 * Ds, M, lam_lo=1e-3)
    st = np.cumsum(npo.sample_frames(32, w, mu, sig, T, 0, Ds), axis=0) / np.sqrt(np.arange(1, T + 1))[:, None]
    X = vc.push_delta(np.asfortranarray(st.T))                 # (2 Ds, T)
    fm = np.asfortranarray(np.vstack([np.linspace(0, 1, T)[None], X]))
    s2 = np.random.default_rng(2).uniform(0.5, 2.0, Ds)
    vs = vc.VarianceScaling(s2)
    tj = vc.TrajectoryGMMMap(vc.GMMMap(*julia_model(w, mu, sig)), L)
    out = vc.vc(tj, fm, postfilter=vs)
    assert len(tj) == (T - 1) % L + 1                          # as vc: W was left at the last chunk's length
    tj2 = vc.TrajectoryGMMMap(vc.GMMMap(*julia_model(w, mu, sig)), L)
    two = vc.vc(tj2, fm)
    plain = two.copy()
    two[1:] = vc.fvpostf(vs, two[1:])
    assert np.array_equal(out[0], fm[0]) and relerr(out, two) < 1e-13
    ref_tj = co.TrajectoryGMMMap(co.GMMMap(w, mu, sig))
    ref = ref_tj.vc(np.ascontiguousarray(fm.T), L)            # (T, Ds + 1)
    assert relerr(plain, ref.T) < 1e-6
    assert relerr(out[1:], co.variance_scaling(np.ascontiguousarray(ref[:, 1:]), s2).T) < 1e-6
