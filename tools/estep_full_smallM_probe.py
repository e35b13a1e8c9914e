"""Full-covariance E-step at the reference's model sizes (tools, GPU box): M = 8, 16, 32 mixtures of the reference's trained joint
model (its first M mixtures, weights renormalised), 5e5 frames drawn from them; ms per call and the fraction of the FP64 MFMA roof
by algorithmic flops (2 M Dj (Dj + 1) + 2 M Dj per frame)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import voiceconversion_jl_amd as vc, synthdata as sd

z = np.load("tests/golden/model_clb_and_slt_gmm32_order40.npz")
N, Dj = 500_000, 80
for M in (8, 16, 32):
    w = z["weights"][:M] / z["weights"][:M].sum()
    mu, sig = np.ascontiguousarray(z["means"][:M]), np.ascontiguousarray(z["covars"][:M])
    X = torch.from_numpy(sd.sample_frames(3, w, mu, sig, N, 0, Dj)).cuda()
    muT, sgT = np.asfortranarray(mu.T), np.asfortranarray(np.transpose(sig, (2, 1, 0)))
    out = torch.empty(vc.full_stats_len(Dj, M), dtype=torch.float64, device="cuda")
    for _ in range(3):
        vc.estep_full_dev(X.t(), w, muT, sgT, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        vc.estep_full_dev(X.t(), w, muT, sgT, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    flop = (2 * M * Dj * (Dj + 1) + 2 * M * Dj) * N
    print(f"M {M:3d}: {ms:.3f} ms  {flop / ms / 1e9 / 78.6:.3f} of the roof")
