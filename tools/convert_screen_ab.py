import sys, time, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import voiceconversion_jl_amd as vc, synthdata as npo
from voiceconversion_jl_amd import _lib
w, mu, sig = npo.synth_model(1002, 80, 64, lam_lo=1e-5)
import os
rows = int(os.environ.get("ROWS", "0"))
if rows: _lib.debug_force({4: _lib.DBG_SCREEN_ROWS4, 2: _lib.DBG_SCREEN_ROWS2, 1: _lib.DBG_SCREEN_ROWS1}[rows])
g = vc.GMMMap(w, np.asfortranarray(mu.T), np.asfortranarray(np.transpose(sig, (2, 1, 0))))
_lib.debug_force(0)
print("rows", rows or "auto", "plan", g.convert_plan())
T = 1_000_000
X = npo.sample_frames(1002, w, mu, sig, T, 0, 40)
Xd = torch.from_numpy(X).cuda(); Yd = torch.empty_like(Xd)
def run(force, n=30):
    _lib.debug_force(force)
    for _ in range(5): vc.fvconvert(g, Xd.t(), out=Yd.t())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): vc.fvconvert(g, Xd.t(), out=Yd.t())
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    g.prune_stats(True); vc.fvconvert(g, Xd.t(), out=Yd.t()); torch.cuda.synchronize()
    iss = g.convert_plan()[0]; nreg = g.prune_stats(False)
    _lib.debug_force(0)
    return dt * 1e3, iss, nreg, Yd.clone()
for name, f in (("peaked", _lib.DBG_CONVERT_SHAPE_PEAKED), ("screened", _lib.DBG_CONVERT_SHAPE_SCREENED), ("screened", _lib.DBG_CONVERT_SHAPE_SCREENED)):
    ms, iss, nreg, Y = run(f)
    print(name, "%.4f ms" % ms, "issued", iss, "frac %.3f" % (iss * 2048 / (ms * 1e-3) / 78.6e12), "nreg", nreg)
    if name == "peaked": Yp = Y
    else: print("  max rel diff vs peaked", float((torch.linalg.norm(Y - Yp, dim=1) / torch.linalg.norm(Yp, dim=1)).max()))
