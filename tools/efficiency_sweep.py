"""Looking for cliffs (tools, GPU box): fvconvert and the diagonal E-step over dimensions and mixture counts on models whose
frames are SHARED between mixtures (what trained models look like), device-resident; per shape ms per call and the fraction of
the FP64 MFMA roof by algorithmic flops -- an outlier in a row or column is a shape that fell off its fast path."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import voiceconversion_jl_amd as vc, synthdata as sd

PEAK = 78.6e12


def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


what = sys.argv[1] if len(sys.argv) > 1 else "both"
if what in ("convert", "both"):
    T = 500_000
    print("fvconvert, broad synthetic models (eigenvalues in [0.1, 1]), %d frames: ms (fraction of the roof, dense flops)" % T)
    for D in ((16, 24, 25, 32, 40, 48, 64, 80) if len(sys.argv) < 3 else tuple(int(a) for a in sys.argv[2].split(","))):
        row = []
        for M in (8, 32, 64):
            w, mu, sig = sd.synth_model(7, 2 * D, M, lam_lo=1e-1)
            g = vc.GMMMap(w, np.asfortranarray(mu.T), np.asfortranarray(np.transpose(sig, (2, 1, 0))))
            X = torch.from_numpy(sd.sample_frames(8, w, mu, sig, T, 0, D)).cuda()
            Y = torch.empty_like(X)
            dt = timeit(lambda: vc.fvconvert(g, X.t(), out=Y.t()))
            flop = T * M * (D * (D + 1) + 2 * D * D + 2 * D)
            row.append("%7.3f (%.2f)" % (dt * 1e3, flop / dt / PEAK))
        print("  D %3d: M 8 / 32 / 64: %s" % (D, "  ".join(row)))
if what in ("estep", "both"):
    N = 500_000
    print("diagonal E-step, shared frames (means 3 sigma / sqrt(Dj) apart), %d frames: ms (fraction of the roof)" % N)
    for Dj in ((24, 32, 48, 50, 64, 80, 81, 100, 160) if len(sys.argv) < 3 else tuple(int(a) for a in sys.argv[2].split(","))):
        row = []
        for M in (16, 64, 128, 256):
            rg = np.random.default_rng(M + Dj)
            w = rg.dirichlet(2.0 * np.ones(M))
            var = np.exp(rg.uniform(np.log(0.05), 0.0, (M, Dj)))
            mu = 3.0 * rg.standard_normal((M, Dj)) / np.sqrt(Dj)
            comp = rg.choice(M, size=N, p=w)
            X = torch.from_numpy(mu[comp] + rg.standard_normal((N, Dj)) * np.sqrt(var[comp])).cuda()
            muT, varT = np.asfortranarray(mu.T), np.asfortranarray(var.T)
            out = torch.empty(vc.stats_len(Dj, M), dtype=torch.float64, device="cuda")
            dt = timeit(lambda: vc.estep_diag_dev(X.t(), w, muT, varT, out=out))
            flop = N * (2 * M * 2 * Dj + 2 * M * 2 * Dj + 2 * M)
            row.append("%7.3f (%.2f)" % (dt * 1e3, flop / dt / PEAK))
        print("  Dj %3d: M 16 / 64 / 128 / 256: %s" % (Dj, "  ".join(row)))
if what in ("estep_full",):
    N = 200_000
    print("full-covariance E-step, broad synthetic models, %d frames: ms (fraction of the roof)" % N)
    for Dj in (32, 48, 50, 64, 80, 82, 96, 128, 160):
        row = []
        for M in (8, 32, 64):
            w, mu, sig = sd.synth_model(9, Dj, M, lam_lo=1e-1)
            X = torch.from_numpy(sd.sample_frames(10, w, mu, sig, N, 0, Dj)).cuda()
            muT, sgT = np.asfortranarray(mu.T), np.asfortranarray(np.transpose(sig, (2, 1, 0)))
            out = torch.empty(vc.full_stats_len(Dj, M), dtype=torch.float64, device="cuda")
            dt = timeit(lambda: vc.estep_full_dev(X.t(), w, muT, sgT, out=out), n=3)
            flop = N * (2 * M * Dj * (Dj + 1) + 2 * M * Dj)
            row.append("%7.3f (%.2f)" % (dt * 1e3, flop / dt / PEAK))
        print("  Dj %3d: M 8 / 32 / 64: %s" % (Dj, "  ".join(row)))
if what in ("dtw",):
    n, S = 1000, 500
    print("DTW fit + backward, %d pairs of ~%d x %d frames: ms (fraction of the FMA-free FP64 vector roof, 39.3 TFLOP/s)" % (n, S, S))
    rg = np.random.default_rng(3)
    for D in (12, 24, 25, 32, 40, 41, 48, 60, 80):
        tm = [rg.standard_normal((D, S)) for _ in range(8)]
        sq = [rg.standard_normal((D, S + 17)) for _ in range(8)]
        d = vc.DTW()
        ts, ss = [tm[i % 8] for i in range(n)], [sq[i % 8] for i in range(n)]
        t0 = time.perf_counter(); vc.fit_batch(d, ts, ss); t1 = time.perf_counter() - t0
        t0 = time.perf_counter(); vc.fit_batch(d, ts, ss); t1 = time.perf_counter() - t0
        print("  D %3d: host-pointer batch %.2f ms" % (D, t1 * 1e3))
if what in ("posterior",):
    T = 500_000
    print("predict_proba / predict, broad synthetic models, %d frames: ms (fraction of the roof, M D (D + 1) flops per frame)" % T)
    for D in (16, 24, 32, 40, 48, 56, 64, 72, 80):
        row = []
        for M in (8, 32, 64):
            w, mu, sig = sd.synth_model(11, 2 * D, M, lam_lo=1e-1)
            px = vc.GMMMap(w, np.asfortranarray(mu.T), np.asfortranarray(np.transpose(sig, (2, 1, 0)))).px
            X = torch.from_numpy(sd.sample_frames(12, w, mu, sig, T, 0, D)).cuda()
            d1 = timeit(lambda: vc.predict_proba(px, X.t()), n=3)
            d2 = timeit(lambda: vc.predict(px, X.t()), n=3)
            flop = T * M * D * (D + 1)
            row.append("%6.3f (%.2f) / %6.3f (%.2f)" % (d1 * 1e3, flop / d1 / PEAK, d2 * 1e3, flop / d2 / PEAK))
        print("  D %3d: M 8 / 32 / 64: %s" % (D, "   ".join(row)))
if what in ("traj",):
    M = 16
    print("TrajectoryGMMMap fvconvert, 256 utterances: ms per call (ns per frame and static dimension cubed x 1e3)")
    for D in (12, 16, 20, 24, 25, 28, 30, 32, 36, 40, 44, 46, 48):
        w, mu, sig = sd.synth_model(5, 4 * D, M, lam_lo=1e-3)
        g = vc.GMMMap(w, np.asfortranarray(mu.T), np.asfortranarray(np.transpose(sig, (2, 1, 0))))
        row = []
        for T in (100, 500, 2000):
            tj = vc.TrajectoryGMMMap(g, T)
            st = sd.sample_frames(6, w, mu, sig, T, 0, D)
            st = np.cumsum(st, axis=0) / np.sqrt(np.arange(1, T + 1))[:, None]
            X1 = np.ascontiguousarray(vc.push_delta(np.asfortranarray(st.T)).T)
            n = 256
            X = torch.from_numpy(np.tile(X1, (n, 1))).cuda()
            Y = torch.empty((n * T, D), dtype=torch.float64, device="cuda")
            xoff = np.arange(n, dtype=np.int64) * T * 2 * D
            yoff = np.arange(n, dtype=np.int64) * T * D
            Ts = np.full(n, T, dtype=np.int64)
            from voiceconversion_jl_amd import _lib

            def step():
                _lib.check(_lib.lib.vcmi_traj_convert_batch_dev(tj._h, n, X.data_ptr(), _lib.iptr(xoff), _lib.iptr(Ts), Y.data_ptr(), _lib.iptr(yoff),
                                                                torch.cuda.current_stream().cuda_stream))
            dt = timeit(step, n=3)
            row.append("%8.3f (%.2f)" % (dt * 1e3, dt * 1e9 / (n * T) / D ** 3 * 1e3))
        print("  D %3d: T 100 / 500 / 2000: %s" % (D, "  ".join(row)))
