import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import voiceconversion_jl_amd as vc
import synthdata as npo
D, M, T = 40, 64, 1_000_000
w, mu, sig = npo.synth_model(1002, 2 * D, M)
g = vc.GMMMap(w, np.asfortranarray(mu.T), np.asfortranarray(np.transpose(sig, (2, 1, 0))))
X = npo.sample_frames(1002, w, mu, sig, T, 0, D)
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "benchlike":      # what bench.py has done before the host-path measurement
    Xd = torch.from_numpy(X).cuda(); Yd = torch.empty_like(Xd)
    for _ in range(30): vc.fvconvert(g, Xd.t(), out=Yd.t())
    torch.cuda.synchronize()
    from oracle import c_oracle as co
    ref = co.GMMMap(w, mu, sig); ref.fvconvert(X[:100000])
    Y = Yd[:300000].cpu().numpy()
Xh = np.asfortranarray(X.T)
vc.fvconvert(g, Xh)
fr = []
keep, fk, fd = [], [], []
for rep in range(4):      # results kept alive: the call alone (allocation + first touch + conversion), then the free alone
    t0 = time.perf_counter(); keep.append(vc.fvconvert(g, Xh)); fk.append((time.perf_counter() - t0) * 1e3)
for rep in range(4):
    t0 = time.perf_counter(); keep.pop(); fd.append((time.perf_counter() - t0) * 1e3)
print("   call only (results kept):", " ".join("%.1f" % x for x in fk), "| freeing a 320 MB result:", " ".join("%.1f" % x for x in fd))
for rep in range(4):
    t0 = time.perf_counter(); Y = vc.fvconvert(g, Xh); fr.append((time.perf_counter() - t0) * 1e3)
Yh = np.empty_like(Xh, order="F"); vc.fvconvert(g, Xh, out=Yh)
ru = []
for rep in range(4):
    t0 = time.perf_counter(); vc.fvconvert(g, Xh, out=Yh); ru.append((time.perf_counter() - t0) * 1e3)
print(mode, os.environ.get("VCMI_HOST_NUMA", "-"), os.environ.get("VCMI_HOST_POPULATE", "-"), "fresh", " ".join("%.1f" % x for x in fr), "| reused", " ".join("%.1f" % x for x in ru))
