"""The whole of BASELINE configs[2] on ONE GPU (tools, GPU box): 10^7 frames, Dj = 80, M = 128 -- the hard-assignment path against
the one-kernel path (DBG_ESTEP_NO_HARD): times and the largest relative difference of the packed statistics."""
import os, sys, time
os.environ["VCMI_TEST_HOOKS"] = "1"
sys.path.insert(0, ".")
import numpy as np, torch
import voiceconversion_jl_amd as vc
from voiceconversion_jl_amd import _lib
import synthdata as npo
Dj, M, N = 80, 128, 10_000_000
w, mu, _ = npo.synth_model(1003, Dj, M)
var = np.exp(np.random.default_rng(1003).uniform(np.log(1e-3), 0.0, (M, Dj)))
g = torch.Generator(device="cuda"); g.manual_seed(1)
comp = torch.multinomial(torch.tensor(w, device="cuda"), N, replacement=True, generator=g)
X = torch.tensor(mu, device="cuda")[comp] + torch.randn((N, Dj), dtype=torch.float64, device="cuda", generator=g) * torch.tensor(np.sqrt(var), device="cuda")[comp]
muT, varT = np.asfortranarray(mu.T), np.asfortranarray(var.T)
a = vc.estep_diag_dev(X.t(), w, muT, varT).clone(); torch.cuda.synchronize()
print("soft", _lib.estep_last_soft())
t0 = time.perf_counter(); a = vc.estep_diag_dev(X.t(), w, muT, varT).clone(); torch.cuda.synchronize(); print("hard ms", (time.perf_counter() - t0) * 1e3)
_lib.debug_force(_lib.DBG_ESTEP_NO_HARD)
o = vc.estep_diag_dev(X.t(), w, muT, varT).clone(); torch.cuda.synchronize()
t0 = time.perf_counter(); o = vc.estep_diag_dev(X.t(), w, muT, varT).clone(); torch.cuda.synchronize(); print("one-kernel ms", (time.perf_counter() - t0) * 1e3)
_lib.debug_force(0)
print("max rel diff", float(((a - o).abs() / o.abs().max()).max()), "S0 sum", float(a[:M].sum()))
