import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import voiceconversion_jl_amd as vc
from oracle import np_oracle as npo
for Dj, M, N in ((80, 64, 100000), (160, 64, 100000), (160, 64, 500000)):
    w, mu, sig = npo.synth_model(1005, Dj, M, lam_lo=1e-3)
    X = npo.sample_frames(1006, w, mu, sig, N, 0, Dj)
    Xd = torch.from_numpy(X).cuda()
    out_t = torch.empty(vc.full_stats_len(Dj, M), dtype=torch.float64, device="cuda")
    muT, sgT = np.asfortranarray(mu.T), np.asfortranarray(np.transpose(sig, (2, 1, 0)))
    for _ in range(2): vc.estep_full_dev(Xd.t(), w, muT, sgT, out=out_t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): vc.estep_full_dev(Xd.t(), w, muT, sgT, out=out_t)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    flop = (2 * M * Dj * (Dj + 1) + 2 * M * Dj) * N
    print(Dj, M, N, "%.2f ms" % (dt * 1e3), "%.2f TF" % (flop / dt / 1e12), "frac %.3f" % (flop / dt / 78.6e12))
