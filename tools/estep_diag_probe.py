import os

os.environ.setdefault("VCMI_TEST_HOOKS", "1")       # vcmi_debug_force is inert without it
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import voiceconversion_jl_amd as vc
from voiceconversion_jl_amd import _lib
from oracle import c_oracle as co
from test_gpu_estep import _hard_case
from conftest import relerr
for overlap in (3.0, 1.0, 0.3):
  for (Dj,M) in ((80,128),(80,37),(48,16)):
    w, mu, var, X = _hard_case(515 + Dj + M, Dj, M, 6000, 10.0, 1e-7, 1e-2, overlap)
    r0, r1, r2, rl = co.estep_diag(X, w, mu, var)
    lp = -0.5 * (((X[:300, None, :] - mu[None]) ** 2 / var[None]).sum(-1) + np.log(var).sum(-1)[None]) + np.log(w)[None]
    g = np.exp(lp - lp.max(1, keepdims=True)); g /= g.sum(1, keepdims=True)
    out=[]
    for generic in (False, True):
        _lib.debug_force(_lib.DBG_ESTEP_GENERIC if generic else 0)
        S0, S1, S2, ll = vc.estep_diag(X.T, w, mu.T, var.T)
        _lib.debug_force(0)
        out.append((relerr(S0, r0), relerr(S1, r1.T), relerr(S2, r2.T), abs(ll-rl)/abs(rl)))
    print(f"overlap {overlap} Dj {Dj} M {M}: mean max posterior {g.max(1).mean():.3f}  mfma {['%.1e'%v for v in out[0]]}  generic {['%.1e'%v for v in out[1]]}")
