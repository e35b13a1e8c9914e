"""Generate tests/golden/*.npz -- run in the BUILD container only (needs /root/reference for the model
fixture and /opt/conda/bin/h5dump).  TEST INFRASTRUCTURE ONLY.

    python -m oracle.gen_golden

Expected outputs come from the numpy/scipy restatement (oracle/np_oracle.py); before anything is
written, each is checked against the C oracle (oracle/vc_oracle.c) -- the two independent
restatements must agree (bit-exact for DTW, <=1e-9 relative for floating-point paths) -- AND against
third-party code (oracle/crosscheck.py: sklearn GaussianMixture, scipy multivariate_normal /
logsumexp / solveh_banded, numpy.linalg.solve, 50-digit mpmath).  The only data
taken from the reference tree are the trained-model tensors of
test/models/clb_to_slt_gmm32_order40_diff.jld (the model test/gmmmap.jl:3-8 loads) and the DTW
known-answer vectors of test/dtw.jl:7-31.
"""
import os
import subprocess
import tempfile

import numpy as np

from . import c_oracle as co
from . import crosscheck as cc
from . import np_oracle as npo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
REF_MODEL = "/root/reference/test/models/clb_to_slt_gmm32_order40_diff.jld"
H5DUMP = "/opt/conda/bin/h5dump"


def _relmax(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(a)), 1e-300))


def extract_model():
    """raw HDF5 buffers are the Julia memory image."""
    with tempfile.TemporaryDirectory() as td:
        arrs = {}
        for name in ("weights", "means", "covars"):
            p = os.path.join(td, name + ".bin")
            subprocess.check_call([H5DUMP, "-d", "/" + name, "-b", "LE", "-o", p, REF_MODEL], stdout=subprocess.DEVNULL)
            arrs[name] = np.fromfile(p, dtype="<f8")
    w = arrs["weights"]
    M = len(w)
    mu = arrs["means"].reshape(M, -1)
    Dj = mu.shape[1]
    sig = arrs["covars"].reshape(M, Dj, Dj)      # [m][col][row]
    return w, mu, sig


def main():
    os.makedirs(OUT, exist_ok=True)
    w, mu, sig = extract_model()
    M, Dj = mu.shape
    D = Dj // 2
    np.savez(os.path.join(OUT, "model_clb_to_slt_gmm32_order40_diff.npz"), weights=w, means=mu, covars=sig)

    # ---- (2) frame conversion on the fixture model, both directions
    out = {}
    for swap in (False, True):
        gn, gc = npo.GMMMap(w, mu, sig, swap=swap), co.GMMMap(w, mu, sig, swap=swap)
        lo = D if swap else 0
        X = npo.sample_frames(20261 + swap, w, mu, sig, 256, lo, lo + D)
        Y, P, idx = gn.fvconvert(X), gn.predict_proba(X), gn.predict(X)
        assert _relmax(Y, gc.fvconvert(X)) < 1e-9 and np.max(np.abs(P - gc.predict_proba(X))) < 1e-9
        assert np.array_equal(idx, gc.predict(X))
        assert _relmax(gn.A, gc.A) < 1e-9
        # third-party pins (sklearn, scipy, LAPACK gesv; 50-digit mpmath for the forward direction)
        print("cross-check", "swap" if swap else "fwd", cc.check_conversion(w, mu, sig, X, Y, P, idx, swap=swap))
        if not swap:
            print("cross-check mpmath", cc.check_conversion_mpmath(w, mu, sig, X, Y, P))
        k = "swap" if swap else "fwd"
        out.update({f"X_{k}": X, f"Y_{k}": Y, f"P_{k}": P, f"idx_{k}": idx})
    # vc(): power row passthrough
    fm = np.concatenate([np.linspace(-3, 3, 64)[:, None], out["X_fwd"][:64]], axis=1)
    gn = npo.GMMMap(w, mu, sig)
    out["vc_fm"], out["vc_out"] = fm, npo.vc_frames(gn, fm)
    np.savez(os.path.join(OUT, "gmmmap_fixture_model.npz"), **out)

    # ---- config 1 of BASELINE.json: D=24, M=8, T=1000 synthetic (plumbing case)
    w1, mu1, sig1 = npo.synth_model(1001, 48, 8)
    X1 = npo.sample_frames(1001, w1, mu1, sig1, 1000, 0, 24)
    g1n, g1c = npo.GMMMap(w1, mu1, sig1), co.GMMMap(w1, mu1, sig1)
    Y1 = g1n.fvconvert(X1)
    assert _relmax(Y1, g1c.fvconvert(X1)) < 1e-9
    cc.check_conversion(w1, mu1, sig1, X1, Y1, g1n.predict_proba(X1), g1n.predict(X1))
    np.savez(os.path.join(OUT, "gmmmap_cfg1_D24_M8_T1000.npz"), weights=w1, means=mu1, covars=sig1, X=X1, Y=Y1,
             idx=g1n.predict(X1))

    # ---- (3) DTW: the reference's two KATs + random pairs
    dt = {}
    v1 = np.array([[1., 2, 3], [1, 2, 4], [1, 8, 5], [10, 3, 6]])
    v2 = np.array([[1., 2, 3], [1, 2, 4], [1, 2, 5], [1, 8, 5], [10, 3, 6]])
    a1 = np.array([[0.], [1], [2], [3], [4], [5]])
    a2 = np.array([[0.], [0], [1], [2], [3], [4], [4], [5]])
    kats = [(v1, v2, 0, 1, [1, 2, 2, 3, 4]), (a1, a2, 0, 1, [1, 1, 2, 3, 4, 5, 5, 6])]   # test/dtw.jl:17,29
    for k, (t, s, fs, bs, exp) in enumerate(kats):
        p, c, b = npo.dtw_fit(t, s, fs, bs)
        assert p.tolist() == exp and co.dtw_fit(t, s, fs, bs)[0].tolist() == exp
        dt.update({f"kat{k}_tmpl": t, f"kat{k}_seq": s, f"kat{k}_path": np.array(exp, dtype=np.int64),
                   f"kat{k}_cost": c, f"kat{k}_bp": b, f"kat{k}_steps": np.array([fs, bs])})
    rng = np.random.default_rng(424242)
    n = 0
    for Dd in (1, 3, 40):
        for bs, fs in ((1, 0), (2, 0), (2, 1), (3, 2)):
            for rep in range(2):
                S, T = int(rng.integers(5, 61)), int(rng.integers(5, 61))
                t = rng.standard_normal((S, Dd))
                # sequence: time-warped noisy copy of the template so that paths are non-trivial
                src_idx = np.clip(np.sort(rng.integers(0, S, T)), 0, S - 1)
                s = t[src_idx] + 0.3 * rng.standard_normal((T, Dd))
                if rep == 1:
                    s = np.round(s * 2) / 2      # coarse values -> exact cost ties exercise the strict '<' rule
                    t = np.round(t * 2) / 2
                p, c, b = npo.dtw_fit(t, s, fs, bs)
                p2, c2, b2 = co.dtw_fit(t, s, fs, bs)
                assert np.array_equal(p, p2) and np.array_equal(c, c2) and np.array_equal(b, b2)
                nt, ap = npo.align(t, s)
                nt2, ap2 = co.align(t, s)
                assert np.array_equal(nt, nt2) and np.array_equal(ap, ap2)
                dt.update({f"r{n}_tmpl": t, f"r{n}_seq": s, f"r{n}_path": p, f"r{n}_cost": c, f"r{n}_bp": b,
                           f"r{n}_steps": np.array([fs, bs]), f"r{n}_align_newtgt": nt, f"r{n}_align_path": ap})
                n += 1
    dt["n_random"] = np.array(n)
    np.savez_compressed(os.path.join(OUT, "dtw_cases.npz"), **dt)

    # ---- (4) trajectory: W pattern (test/trajectory_gmmmap.jl:53) + a solve with the fixture model read as
    #      static D=20 + delta D=20 (the *_with_delta model is missing from the reference tree)
    tr = {}
    r, c, v = co.constructW(30, 40)
    Wn = npo.constructW(30, 40).tocoo()
    import scipy.sparse as sp
    assert abs(sp.coo_matrix((v, (r - 1, c - 1)), shape=Wn.shape).tocsc() - Wn.tocsc()).max() == 0
    tr.update(W_rows=r, W_cols=c, W_vals=v)
    gn, gc = npo.GMMMap(w, mu, sig), co.GMMMap(w, mu, sig)
    tn, tc = npo.TrajectoryGMMMap(gn), co.TrajectoryGMMMap(gc)
    static = npo.sample_frames(777, w, mu, sig, 100, 0, D // 2)
    # smooth the static track so that delta features are speech-like, then push_delta
    static = np.cumsum(static, axis=0) / np.sqrt(np.arange(1, 101))[:, None]
    Xd = npo.push_delta(static)
    assert np.array_equal(Xd, co.push_delta(static))
    Y, mh, Ey = tn.fvconvert(Xd)
    Yc, mhc, Eyc = tc.fvconvert(Xd)
    assert np.array_equal(mh, mhc) and _relmax(Y, Yc) < 1e-6 and _relmax(Ey, Eyc) < 1e-9, (_relmax(Y, Yc))
    print("cross-check trajectory", cc.check_trajectory(w, mu, sig, Xd, Y, mh, Ey))   # LAPACK pbsv on the band
    fmt = np.concatenate([np.linspace(0, 1, 100)[:, None], Xd], axis=1)
    vco = npo.vc_traj(tn, fmt, 30)
    assert _relmax(vco, tc.vc(fmt, 30)) < 1e-6
    tr.update(static=static, X=Xd, Y=Y, mhat=mh, Ey=Ey, vc_fm=fmt, vc_out_L30=vco)
    np.savez(os.path.join(OUT, "trajectory_fixture_model.npz"), **tr)

    # ---- (5) diagonal E-step, N=2000, Dj=80, M=16
    wd, mud, _ = npo.synth_model(3003, 80, 16)
    rg = np.random.default_rng(3003)
    var = np.exp(rg.uniform(np.log(1e-3), 0.0, (16, 80)))
    comp = rg.choice(16, size=2000, p=wd)
    Xe = mud[comp] + rg.standard_normal((2000, 80)) * np.sqrt(var[comp])
    S0, S1, S2, ll = npo.estep_diag(Xe, wd, mud, var)
    c0, c1, c2, cl = co.estep_diag(Xe, wd, mud, var)
    assert _relmax(S0, c0) < 1e-10 and _relmax(S1, c1) < 1e-10 and _relmax(S2, c2) < 1e-10 and abs(ll - cl) < 1e-8 * abs(ll)
    cc.check_estep_diag(Xe, wd, mud, var, S0, S1, S2, ll)                              # sklearn responsibilities
    np.savez(os.path.join(OUT, "estep_diag_N2000_D80_M16.npz"), X=Xe, w=wd, mu=mud, var=var, S0=S0, S1=S1, S2=S2,
             loglik=np.array(ll))
    # ---- (6) full-covariance E-step (what bin/train_gmm.jl:84-103 runs), N=1000, Dj=80, M=8; the log-density
    # is additionally checked against scikit-learn's own full-covariance routine when it is importable here.
    wf, muf, sigf = npo.synth_model(3004, 80, 8, lam_lo=1e-3)
    Xf = npo.sample_frames(3004, wf, muf, sigf, 1000, 0, 80)
    S0, S1, S2, ll = npo.estep_full(Xf, wf, muf, sigf)
    c0, c1, c2, cl = co.estep_full(Xf, wf, muf, sigf)
    assert _relmax(S0, c0) < 1e-10 and _relmax(S1, c1) < 1e-10 and _relmax(S2, c2) < 1e-10 and abs(ll - cl) < 1e-8 * abs(ll)
    cc.check_estep_full(Xf, wf, muf, sigf, S0, S1, S2, ll)                             # sklearn, full covariance
    np.savez(os.path.join(OUT, "estep_full_N1000_D80_M8.npz"), X=Xf, w=wf, mu=muf, sigma=sigf, S0=S0, S1=S1, S2=S2,
             loglik=np.array(ll))
    # ---- (7) TrajectoryGVGMMMap on the same fixture utterance (src/trajectory_gmmmap.jl:114-189), VarianceScaling,
    #      diffgmm parameters of the fixture model (first two mixtures)
    rg = np.random.default_rng(7007)
    muv = Y.var(axis=0, ddof=1) * 1.3
    Ar = rg.standard_normal((D // 2, D // 2))
    Sv = Ar @ Ar.T / (D // 2) * np.mean(muv) ** 2 * 0.1 + np.diag(muv ** 2 * 0.05)
    ygv = npo.trajgv_fvconvert(tn, Xd, muv, Sv, epochs=100, alpha=1.0e-5)
    assert _relmax(ygv, tc.fvconvert_gv(Xd, muv, Sv, 100, 1.0e-5)) < 1e-6
    cc.check_gv(w, mu, sig, Xd, muv, Sv, ygv, epochs=100, alpha=1.0e-5)                # dense numpy / scipy.sparse, no shared code
    vs = npo.variance_scaling(Y, muv)
    assert _relmax(vs, co.variance_scaling(Y, muv)) < 1e-13
    dm, ds = npo.diffgmm(mu[:2], sig[:2])
    dmc, dsc = co.diffgmm(mu[:2], sig[:2])
    assert _relmax(dm, dmc) < 1e-15 and _relmax(ds, dsc) < 1e-14
    np.savez(os.path.join(OUT, "gv_fixture_model.npz"), muv=muv, sigmavv=Sv, Y_gv=ygv, Y_scaled=vs, diff_mu=dm, diff_sigma=ds)
    # ---- (8) align_mcep / mc2e / joint features (src/align.jl:38-55, src/datasets.jl:52-98); mc2e is third-party
    #      arithmetic restated from the SPTK recursions -- anchored by the closed form for a c0-only cepstrum
    rg = np.random.default_rng(8008)
    srcm = rg.standard_normal((60, 25)) * np.exp(-0.3 * np.arange(25)) * 0.3
    srcm[:, 0] = rg.uniform(-9.0, 1.0, 60)
    tgtm = srcm[np.sort(rg.integers(0, 60, 70))] + 0.01 * rg.standard_normal((70, 25))
    en = npo.mc2e(srcm, 0.41, 256)
    assert np.max(np.abs(en - co.mc2e(srcm, 0.41, 256)) / en) < 1e-12
    cc.check_mc2e(srcm, 0.41, 256, en)                                                 # frequency-domain evaluation (numpy.fft)
    c0 = np.zeros((1, 5)); c0[0, 0] = 0.7
    assert abs(npo.mc2e(c0, 0.35, 64)[0] - np.exp(1.4)) < 1e-12
    sa, ta = npo.align_mcep(srcm, tgtm, 0.41, 256)
    sac, tac = co.align_mcep(srcm, tgtm, 0.41, 256)
    assert np.array_equal(sa, sac) and np.array_equal(ta, tac)
    jf = npo.parallel_dataset([(sa, ta)], diff=True, ignore0th=True, add_delta=True)
    assert np.array_equal(jf, co.joint_features(sa, ta, True, True, True))
    np.savez(os.path.join(OUT, "align_mcep_case.npz"), src=srcm, tgt=tgtm, energy=en, src_kept=sa, tgt_kept=ta, joint=jf)
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f)) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
