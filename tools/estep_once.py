#!/usr/bin/env python3
"""Profiling target: the diagonal E-step on the reference's trained 32-mixture model (bench workload estep_fixture), a few calls
with the one-kernel path pinned and nothing else (tools/sq_pmc.py --prog tools/estep_once.py runs it under rocprofv3).
    python3 tools/estep_once.py [calls] [debug_force flags]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VCMI_TEST_HOOKS"] = "1"
import numpy as np
import torch

import voiceconversion_jl_amd as vc
from voiceconversion_jl_amd import _lib

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 8
force = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N = 1_250_000
z = np.load(os.path.join(ROOT, "tests", "golden", "model_clb_and_slt_gmm32_order40.npz"))
w, mu, sig = z["weights"] / z["weights"].sum(), np.ascontiguousarray(z["means"]), z["covars"]
M, Dj = mu.shape
var = np.ascontiguousarray(np.stack([np.diag(sig[m]) for m in range(M)]))
chol = np.linalg.cholesky(sig)
rg = np.random.default_rng(2003)
comp = rg.choice(M, size=N, p=w)
X = mu[comp] + np.einsum("nd,nkd->nk", rg.standard_normal((N, Dj)), chol[comp]) if N <= 200_000 else None
if X is None:
    X = np.empty((N, Dj))
    for m in range(M):
        idx = np.flatnonzero(comp == m)
        X[idx] = mu[m] + rg.standard_normal((len(idx), Dj)) @ chol[m].T
Xd = torch.from_numpy(X).cuda()
muT, varT = np.asfortranarray(mu.T), np.asfortranarray(var.T)
out = torch.empty(vc.stats_len(Dj, M), dtype=torch.float64, device="cuda")
_lib.debug_force(force)
vc.estep_set_path(vc.ESTEP_SOFT)
for _ in range(calls):
    vc.estep_diag_dev(Xd.t(), w, muT, varT, out=out)
torch.cuda.synchronize()
