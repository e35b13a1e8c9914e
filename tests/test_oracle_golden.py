"""CPU: the C oracle (oracle/vc_oracle.c) against the committed golden vectors and the reference's own KATs;
numpy restatement spot checks.  No GPU, no /root/reference."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_golden, relerr
from oracle import c_oracle as co
from oracle import np_oracle as npo


def test_dtw_reference_kats():
    """test/dtw.jl:7-31 of the reference"""
    v1 = np.array([[1., 2, 3], [1, 2, 4], [1, 8, 5], [10, 3, 6]])
    v2 = np.array([[1., 2, 3], [1, 2, 4], [1, 2, 5], [1, 8, 5], [10, 3, 6]])
    assert co.dtw_fit(v1, v2, 0, 1)[0].tolist() == [1, 2, 2, 3, 4]
    assert npo.dtw_fit(v1, v2, 0, 1)[0].tolist() == [1, 2, 2, 3, 4]
    a = np.arange(6.)[:, None]
    b = np.array([0., 0, 1, 2, 3, 4, 4, 5])[:, None]
    assert co.dtw_fit(a, b, 0, 1)[0].tolist() == [1, 1, 2, 3, 4, 5, 5, 6]
    assert npo.dtw_fit(a, b, 0, 1)[0].tolist() == [1, 1, 2, 3, 4, 5, 5, 6]


def test_dtw_golden_bit_exact():
    z = load_golden("dtw_cases.npz")
    names = ["kat0", "kat1"] + [f"r{k}" for k in range(int(z["n_random"]))]
    for nm in names:
        fs, bs = (int(x) for x in z[f"{nm}_steps"])
        p, c, b = co.dtw_fit(z[f"{nm}_tmpl"], z[f"{nm}_seq"], fs, bs)
        assert np.array_equal(p, z[f"{nm}_path"]) and np.array_equal(c, z[f"{nm}_cost"]) and np.array_equal(b, z[f"{nm}_bp"])
        assert np.array_equal(co.dtw_fit(z[f"{nm}_tmpl"], z[f"{nm}_seq"], fs, bs, tables=False), z[f"{nm}_path"])
        if nm.startswith("r"):
            nt, ap = co.align(z[f"{nm}_tmpl"], z[f"{nm}_seq"])
            assert np.array_equal(nt, z[f"{nm}_align_newtgt"]) and np.array_equal(ap, z[f"{nm}_align_path"])
    # one case through the pure-numpy restatement as well
    p, c, b = npo.dtw_fit(z["r3_tmpl"], z["r3_seq"], *(int(x) for x in z["r3_steps"]))
    assert np.array_equal(c, z["r3_cost"]) and np.array_equal(b, z["r3_bp"])


def test_constructW_pattern():
    """test/trajectory_gmmmap.jl:1-34 of the reference: constructW(30, 40)"""
    D, T = 30, 40
    r, c, v = co.constructW(D, T)
    W = sp.coo_matrix((v, (r - 1, c - 1)), shape=(2 * D * T, D * T)).toarray()
    I = np.eye(D)
    for t in range(T):
        s = 2 * D * t
        for i in range(T):
            blk = W[s:s + D, i * D:(i + 1) * D]
            assert np.array_equal(blk, I if i == t else np.zeros((D, D)))
            blk = W[s + D:s + 2 * D, i * D:(i + 1) * D]
            want = -0.5 * I if i == t - 1 else (0.5 * I if i == t + 1 else np.zeros((D, D)))
            assert np.array_equal(blk, want)
    z = load_golden("trajectory_fixture_model.npz")
    assert np.array_equal(r, z["W_rows"]) and np.array_equal(c, z["W_cols"]) and np.array_equal(v, z["W_vals"])
    assert abs(npo.constructW(D, T) - sp.csc_matrix(W)).max() == 0


def test_gmmmap_fixture_model(fixture_model):
    w, mu, sig = fixture_model
    z = load_golden("gmmmap_fixture_model.npz")
    for swap, k in ((False, "fwd"), (True, "swap")):
        g = co.GMMMap(w, mu, sig, swap=swap)
        assert g.D == 40 and g.M == 32                                  # test/gmmmap.jl:9-14
        assert relerr(g.fvconvert(z[f"X_{k}"]), z[f"Y_{k}"]) < 1e-9
        assert np.max(np.abs(g.predict_proba(z[f"X_{k}"]) - z[f"P_{k}"])) < 1e-9
        assert np.array_equal(g.predict(z[f"X_{k}"]), z[f"idx_{k}"])
    g = co.GMMMap(w, mu, sig)
    out = g.vc(z["vc_fm"])
    assert np.array_equal(out[:, 0], z["vc_fm"][:, 0]) and relerr(out[:, 1:], z["vc_out"][:, 1:]) < 1e-9
    P = g.predict_proba(z["X_fwd"][:8])
    assert np.allclose(P.sum(axis=1), 1.0, atol=1e-12)
    # numpy restatement on a few frames
    gn = npo.GMMMap(w, mu, sig)
    assert relerr(gn.fvconvert(z["X_fwd"][:5]), z["Y_fwd"][:5]) < 1e-12


def test_gmmmap_joint_model(joint_model):
    """The reference's second trained model (test/vc.jl:40-51): both restatements against the committed vectors."""
    w, mu, sig = joint_model
    z = load_golden("gmmmap_joint_model.npz")
    for swap, k in ((False, "fwd"), (True, "swap")):
        g = co.GMMMap(w, mu, sig, swap=swap)
        assert g.D == 40 and g.M == 32
        assert relerr(g.fvconvert(z[f"X_{k}"]), z[f"Y_{k}"]) < 1e-9
        assert np.max(np.abs(g.predict_proba(z[f"X_{k}"]) - z[f"P_{k}"])) < 1e-9
        assert np.array_equal(g.predict(z[f"X_{k}"]), z[f"idx_{k}"])
        gn = npo.GMMMap(w, mu, sig, swap=swap)
        assert relerr(gn.fvconvert(z[f"X_{k}"][:5]), z[f"Y_{k}"][:5]) < 1e-12
    out = co.GMMMap(w, mu, sig).vc(z["vc_fm"])
    assert np.array_equal(out[:, 0], z["vc_fm"][:, 0]) and relerr(out[:, 1:], z["vc_out"][:, 1:]) < 1e-9


def test_gmmmap_config1():
    z = load_golden("gmmmap_cfg1_D24_M8_T1000.npz")
    g = co.GMMMap(z["weights"], z["means"], z["covars"])
    assert relerr(g.fvconvert(z["X"]), z["Y"]) < 1e-9
    assert np.array_equal(g.predict(z["X"]), z["idx"])


@pytest.mark.parametrize("D,M,T,lam_lo", [(40, 64, 3001, 1e-5), (24, 8, 1000, 1e-5), (25, 5, 77, 1e-2), (7, 3, 31, 1e-1), (40, 32, 1, 1e-3)])
def test_gemm_structured_baseline_matches_the_per_frame_oracle(D, M, T, lam_lo):
    """bench.py's cpu_baseline_strong (the same math as blocked GEMMs + OpenMP, oracle/vc_oracle_gemm.c) is a
    BASELINE, not a reference: it must reproduce the per-frame restatement of src/gmmmap.jl:101-118 to 1e-12 -- ragged block
    tails, D not a multiple of the 4-row register tile, a zero-weight mixture (posterior exactly 0), fewer frames than a block."""
    import synthdata as sd
    w, mu, sig = sd.synth_model(900 + D + M, 2 * D, M, lam_lo=lam_lo)
    if M >= 5:
        w = w.copy(); w[2] = 0.0; w /= w.sum()
    X = sd.sample_frames(901, w, mu, sig, T, 0, D)
    g = co.GMMMap(w, mu, sig)
    Y, nthr = g.fvconvert_gemm(X)
    Yref = g.fvconvert(X)
    assert nthr >= 1 and Y.shape == Yref.shape
    assert float(np.max(np.linalg.norm(Y - Yref, axis=1) / np.linalg.norm(Yref, axis=1))) < 1e-12


def test_gemm_baseline_on_the_reference_model(fixture_model):
    w, mu, sig = fixture_model
    z = load_golden("gmmmap_fixture_model.npz")
    Y, _ = co.GMMMap(w, mu, sig).fvconvert_gemm(z["X_fwd"])
    assert float(np.max(np.linalg.norm(Y - z["Y_fwd"], axis=1) / np.linalg.norm(z["Y_fwd"], axis=1))) < 1e-11


def test_logdens_is_the_posterior_before_the_softmax(fixture_model):
    """vco_logdens (lpr of src/gmm.jl:25-27) -- the quantity oracle/adversarial.py bisects on -- against the golden posteriors
    and against numpy's independent restatement."""
    from scipy.special import logsumexp
    w, mu, sig = fixture_model
    z = load_golden("gmmmap_fixture_model.npz")
    L = co.GMMMap(w, mu, sig).logdens(z["X_fwd"])
    P = np.exp(L - logsumexp(L, axis=1, keepdims=True))
    assert np.max(np.abs(P - z["P_fwd"])) < 1e-10                       # (the golden posteriors come from the numpy restatement)
    assert np.max(np.abs(P - co.GMMMap(w, mu, sig).predict_proba(z["X_fwd"]))) < 1e-13
    assert np.array_equal(np.argmax(L, axis=1) + 1, z["idx_fwd"])


def test_not_positive_definite_is_reported(fixture_model):
    w, mu, sig = fixture_model
    bad = sig.copy()
    bad[3, :40, :40] = -np.eye(40)
    try:
        co.GMMMap(w, mu, bad)
        assert False, "expected LinAlgError"
    except np.linalg.LinAlgError:
        pass


def test_trajectory_fixture_model(fixture_model):
    w, mu, sig = fixture_model
    z = load_golden("trajectory_fixture_model.npz")
    assert np.array_equal(co.push_delta(z["static"]), z["X"])
    assert np.array_equal(npo.push_delta(z["static"]), z["X"])
    t = co.TrajectoryGMMMap(co.GMMMap(w, mu, sig))
    Y, mh, Ey = t.fvconvert(z["X"])
    assert np.array_equal(mh, z["mhat"]) and relerr(Ey, z["Ey"]) < 1e-9 and relerr(Y, z["Y"]) < 1e-6
    assert relerr(t.vc(z["vc_fm"], 30), z["vc_out_L30"]) < 1e-6
    # chunk length is part of the contract (src/common.jl:42-57): another chunking gives another answer
    assert relerr(t.vc(z["vc_fm"], 100)[:, 1:], z["Y"]) < 1e-6
    assert relerr(t.vc(z["vc_fm"], 30), t.vc(z["vc_fm"], 100)) > 1e-6


def test_estep_golden():
    z = load_golden("estep_diag_N2000_D80_M16.npz")
    S0, S1, S2, ll = co.estep_diag(z["X"], z["w"], z["mu"], z["var"])
    assert relerr(S0, z["S0"]) < 1e-10 and relerr(S1, z["S1"]) < 1e-10 and relerr(S2, z["S2"]) < 1e-10
    assert abs(ll - float(z["loglik"])) < 1e-10 * abs(float(z["loglik"]))
    assert abs(S0.sum() - 2000) < 1e-8


def test_estep_full_golden():
    z = load_golden("estep_full_N1000_D80_M8.npz")
    S0, S1, S2, ll = co.estep_full(z["X"], z["w"], z["mu"], z["sigma"])
    assert relerr(S0, z["S0"]) < 1e-10 and relerr(S1, z["S1"]) < 1e-10 and relerr(S2, z["S2"]) < 1e-10
    assert abs(ll - float(z["loglik"])) < 1e-10 * abs(float(z["loglik"]))
    assert abs(S0.sum() - 1000) < 1e-8
    # the full-covariance E-step restricted to diagonal covariances is the diagonal E-step
    zd = load_golden("estep_diag_N2000_D80_M16.npz")
    sig = np.zeros((16, 80, 80))
    sig[:, np.arange(80), np.arange(80)] = zd["var"]
    f0, f1, f2, fl = co.estep_full(zd["X"], zd["w"], zd["mu"], sig)
    assert relerr(f0, zd["S0"]) < 1e-10 and relerr(f1, zd["S1"]) < 1e-10
    assert relerr(f2[:, np.arange(80), np.arange(80)], zd["S2"]) < 1e-10
    assert abs(fl - float(zd["loglik"])) < 1e-10 * abs(float(zd["loglik"]))


def test_gv_golden(fixture_model):
    """TrajectoryGVGMMMap / VarianceScaling / diffgmm: C oracle vs the fixtures the literal numpy restatement
    (explicit sparse W, block-diagonal D^-1) produced"""
    z, zt = load_golden("gv_fixture_model.npz"), load_golden("trajectory_fixture_model.npz")
    w, mu, sig = fixture_model
    t = co.TrajectoryGMMMap(co.GMMMap(w, mu, sig))
    assert relerr(t.fvconvert_gv(zt["X"], z["muv"], z["sigmavv"], 100, 1.0e-5), z["Y_gv"]) < 1e-6
    assert relerr(co.variance_scaling(zt["Y"], z["muv"]), z["Y_scaled"]) < 1e-13
    dm, ds = co.diffgmm(mu[:2], sig[:2])
    assert relerr(dm, z["diff_mu"]) < 1e-15 and relerr(ds, z["diff_sigma"]) < 1e-14


def test_align_mcep_golden():
    z = load_golden("align_mcep_case.npz")
    e = co.mc2e(z["src"], 0.41, 256)
    assert np.max(np.abs(e - z["energy"]) / z["energy"]) < 1e-12
    s, t = co.align_mcep(z["src"], z["tgt"], 0.41, 256)
    assert np.array_equal(s, z["src_kept"]) and np.array_equal(t, z["tgt_kept"]) and 0 < len(s) < 60
    assert np.array_equal(co.joint_features(s, t, True, True, True), z["joint"])


@pytest.mark.parametrize("D,M,T", [(12, 4, 17), (16, 4, 9), (25, 4, 12), (32, 3, 8), (12, 4, 1), (12, 4, 2)])
def test_trajectory_two_restatements_agree(D, M, T):
    """The two independent restatements of src/trajectory_gmmmap.jl:65-110 -- C (stencil + banded Cholesky) and numpy
    (explicit sparse W as the reference builds it, block-diagonal D^-1, sparse direct solve) -- on the dimensions the
    GPU tests run the blocked solver at (one-tile, rhs-row-in-its-own-tile, odd, three-tile), incl. T = 1, 2 where the
    stencil loses its neighbours.  Neither side is the product; this pins the checker."""
    w, mu, sig = npo.synth_model(900 + D, 4 * D, M, lam_lo=1e-3)
    static = npo.sample_frames(7 + T, w, mu, sig, T, 0, D)
    X = npo.push_delta(static)
    yc, mc, ec = co.TrajectoryGMMMap(co.GMMMap(w, mu, sig)).fvconvert(X)
    yn, mn, en = npo.TrajectoryGMMMap(npo.GMMMap(w, mu, sig)).fvconvert(X)
    assert np.array_equal(mc, mn) and relerr(ec, en) < 1e-10 and relerr(yc, yn) < 1e-7
