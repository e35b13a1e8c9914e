import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voiceconversion_jl_amd as vc
import synthdata as npo
D, M, T = 40, 64, 1_000_000
w, mu, sig = npo.synth_model(1002, 2 * D, M)
g = vc.GMMMap(w, np.asfortranarray(mu.T), np.asfortranarray(np.transpose(sig, (2, 1, 0))))
X = npo.sample_frames(1002, w, mu, sig, T, 0, D)
Xh = np.asfortranarray(X.T)
vc.fvconvert(g, Xh)
for rep in range(3):
    t0 = time.perf_counter(); Y = vc.fvconvert(g, Xh); t1 = time.perf_counter()
    print("fresh out: %.2f ms" % ((t1 - t0) * 1e3))
    del Y
Yh = np.empty_like(Xh, order="F")
for rep in range(3):
    t0 = time.perf_counter(); vc.fvconvert(g, Xh, out=Yh); t1 = time.perf_counter()
    print("reused out: %.2f ms" % ((t1 - t0) * 1e3))
fm = np.asfortranarray(np.vstack([np.ones((1, T)), Xh]))
vc.vc(g, fm)
t0 = time.perf_counter(); vc.vc(g, fm); print("vc fresh: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
print(open('/sys/kernel/mm/transparent_hugepage/enabled').read())
