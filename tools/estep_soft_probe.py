"""Diagonal E-step on frames that share their mixtures, by construction (tools, GPU box): M mixtures whose means lie `sep`
standard deviations apart per dimension / sqrt(Dj) -- posteriors spread over many mixtures -- against the separable data of
the bench.  Prints ms per 1.25e6 frames and the issued-MFMA fraction is left to bench.py; this is a timing probe."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import os
import voiceconversion_jl_amd as vc
from voiceconversion_jl_amd import _lib

_lib.debug_force(int(os.environ.get("FORCE", "0")))        # e.g. FORCE=1024: M <= 32 through estep_mfma_kernel's shared tiles
MS = [int(m) for m in os.environ.get("MS", "8,16,32,64,128").split(",")]

Dj, N = 80, 1_250_000
for M in MS:
    for sep in (3.0, 40.0):
        rg = np.random.default_rng(M)
        w = rg.dirichlet(2.0 * np.ones(M))
        var = np.exp(rg.uniform(np.log(0.05), 0.0, (M, Dj)))
        mu = sep * rg.standard_normal((M, Dj)) / np.sqrt(Dj)
        comp = rg.choice(M, size=N, p=w)
        X = torch.from_numpy(mu[comp] + rg.standard_normal((N, Dj)) * np.sqrt(var[comp])).cuda()
        muT, varT = np.asfortranarray(mu.T), np.asfortranarray(var.T)
        out = torch.empty(vc.stats_len(Dj, M), dtype=torch.float64, device="cuda")
        for _ in range(6):
            vc.estep_diag_dev(X.t(), w, muT, varT, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            vc.estep_diag_dev(X.t(), w, muT, varT, out=out)
        torch.cuda.synchronize()
        print(f"M {M:4d} sep {sep:5.1f}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
