"""Diagonal E-step outside the one-kernel shape (more than 128 mixtures, odd joint dimension): the MFMA paths against the
generic kernels they replace on the automatic route.  (tools, GPU box.)"""
import os

os.environ.setdefault("VCMI_TEST_HOOKS", "1")       # vcmi_debug_force is inert without it
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import voiceconversion_jl_amd as vc, synthdata as sd
from voiceconversion_jl_amd import _lib
for Dj, M, N in ((80, 256, 1_250_000), (79, 128, 1_250_000), (81, 200, 500_000)):
    w, mu, _ = sd.synth_model(3, Dj, M)
    rg = np.random.default_rng(4)
    var = np.exp(rg.uniform(np.log(1e-3), 0.0, (M, Dj)))
    comp = rg.choice(M, size=N, p=w)
    X = torch.from_numpy(mu[comp] + rg.standard_normal((N, Dj)) * np.sqrt(var[comp])).cuda()
    out = torch.empty(vc.stats_len(Dj, M), dtype=torch.float64, device="cuda")
    muT, varT = np.asfortranarray(mu.T), np.asfortranarray(var.T)
    res = {}
    for name, flag in (("mfma", 0), ("generic", _lib.DBG_ESTEP_GENERIC)):
        _lib.debug_force(flag)
        vc.estep_diag_dev(X.t(), w, muT, varT, out=out); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): vc.estep_diag_dev(X.t(), w, muT, varT, out=out)
        torch.cuda.synchronize(); res[name] = (time.perf_counter() - t0) / 3 * 1e3
        res[name + "_stats"] = out.cpu().numpy().copy()
    _lib.debug_force(0)
    a, b = res["mfma_stats"], res["generic_stats"]
    print("Dj %d M %d N %d: MFMA path %.2f ms, generic kernels %.2f ms, max rel diff %.1e" % (Dj, M, N, res["mfma"], res["generic"], np.max(np.abs(a - b)) / np.max(np.abs(b))))
