"""One-off robustness sweep on the GPU box (not part of the test suite): random shapes through the paths whose kernels pick an
instantiation by dimension / mixture count, against the oracle.  usage: python tools/fuzz_shapes.py [seed] [cases]"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import voiceconversion_jl_amd as vc  # noqa: E402
from conftest import julia_model, relerr  # noqa: E402
from oracle import c_oracle as co, np_oracle as npo  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rg = np.random.default_rng(seed)
worst = {}


def note(k, v, shape):
    if v > worst.get(k, (0, None))[0]:
        worst[k] = (v, shape)


for _ in range(cases):
    # diagonal E-step
    Dj, M, N = int(rg.integers(1, 81)) * 2, int(rg.integers(1, 129)), int(rg.integers(1, 3000))
    w, mu, _ = npo.synth_model(int(rg.integers(1 << 30)), Dj, M)
    var = np.exp(rg.uniform(np.log(1e-3), 0.0, (M, Dj)))
    comp = rg.choice(M, size=N, p=w)
    X = mu[comp] + rg.standard_normal((N, Dj)) * np.sqrt(var[comp])
    r0, r1, r2, rl = co.estep_diag(X, w, mu, var)
    S0, S1, S2, ll = vc.estep_diag(X.T, w, mu.T, var.T)
    note("estep_diag", max(relerr(S0, r0), relerr(S1, r1.T), relerr(S2, r2.T), abs(ll - rl) / abs(rl)), (Dj, M, N))
    # full-covariance E-step
    Dj, M, N = int(rg.integers(2, 161)), int(rg.integers(1, 12)), int(rg.integers(2, 1500))
    w, mu, sig = npo.synth_model(int(rg.integers(1 << 30)), Dj, M, lam_lo=1e-3)
    X = npo.sample_frames(int(rg.integers(1 << 30)), w, mu, sig, N, 0, Dj)
    ref = co.estep_full(X, w, mu, sig)
    got = vc.estep_full(X.T, w, mu.T, np.transpose(sig, (2, 1, 0)))
    note("estep_full", max(relerr(got[0], ref[0]), relerr(got[1], ref[1].T), relerr(got[2], np.transpose(ref[2], (2, 1, 0))),
                           abs(got[3] - ref[3]) / abs(ref[3])), (Dj, M, N))
    # conversion / posterior
    D, M, T = int(rg.integers(1, 81)), int(rg.integers(1, 9)), int(rg.integers(1, 400))
    w, mu, sig = npo.synth_model(int(rg.integers(1 << 30)), 2 * D, M, lam_lo=1e-3)
    X = npo.sample_frames(int(rg.integers(1 << 30)), w, mu, sig, T, 0, D)
    refg = co.GMMMap(w, mu, sig)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    Y = vc.fvconvert(g, X.T)
    note("fvconvert", float(np.max(np.linalg.norm(Y.T - refg.fvconvert(X), axis=1) / np.linalg.norm(refg.fvconvert(X), axis=1))), (D, M, T))
    note("posterior", float(np.max(np.abs(vc.predict_proba(g.px, X.T) - refg.predict_proba(X).T))), (D, M, T))
    # trajectory (padded and native dimensions)
    D, M, T = int(rg.integers(2, 41)), int(rg.integers(1, 6)), int(rg.integers(1, 120))
    w, mu, sig = npo.synth_model(int(rg.integers(1 << 30)), 4 * D, M, lam_lo=1e-3)
    reft = co.TrajectoryGMMMap(co.GMMMap(w, mu, sig))
    t = vc.TrajectoryGMMMap(vc.GMMMap(*julia_model(w, mu, sig)), T)
    st = npo.sample_frames(int(rg.integers(1 << 30)), w, mu, sig, T, 0, D)
    st = np.cumsum(st, axis=0) / np.sqrt(np.arange(1, T + 1))[:, None]
    x = npo.push_delta(st)
    y = t.fvconvert_batch([x.T])[0]
    yref, _, _ = reft.fvconvert(x)
    note("trajectory", relerr(y, yref.T), (D, M, T))
# DTW: a ragged batch per seed (bit-exact paths and align outputs; strips for S > 512, every fused DMAX, both step windows)
mism = 0
for bs in (1, 2):
    D = int(rg.integers(1, 48))
    pairs = []
    for _ in range(cases):
        S, T = int(rg.integers(1, 1300)), int(rg.integers(1, 700))
        tmpl = rg.standard_normal((S, D))
        idx = np.clip(np.sort(rg.integers(0, S, T)), 0, S - 1)
        pairs.append((tmpl, tmpl[idx] + 0.2 * rg.standard_normal((T, D))))
    d = vc.DTW(fstep=0, bstep=bs)
    paths = vc.fit_batch(d, [t.T for t, _ in pairs], [q.T for _, q in pairs])
    for (t, q), pth in zip(pairs, paths):
        mism += int(not np.array_equal(pth, co.dtw_fit(t, q, 0, bs, tables=False)))
    if bs == 2:
        outs = vc.align_batch([t.T for t, _ in pairs], [q.T for _, q in pairs])
        for (t, q), (src, nt) in zip(pairs, outs):
            mism += int(not np.array_equal(nt, co.align(t, q)[0].T))
print(f"dtw          path / align mismatches {mism}")
worst["dtw"] = (float(mism), None)
for k, (v, shape) in worst.items():
    print(f"{k:12s} worst relative error {v:.2e} at {shape}")
bad = [k for k, (v, _) in worst.items() if not v < (1e-6 if k == "trajectory" else 1e-9)]      # dtw: 0 mismatches
print("FUZZ_OK" if not bad else f"FUZZ_FAIL {bad}")
