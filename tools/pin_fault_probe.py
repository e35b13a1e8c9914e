"""Does hipHostUnregister of a small (heap) array leave the runtime with a stale pin of a neighbouring pageable buffer?
    python3 tools/pin_fault_probe.py A|B [iterations]      A: pin / unpin small arrays between converter creations; B: no pins"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import synthdata as sd
import voiceconversion_jl_amd as vc

mode, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 300
w, mu, sig = sd.synth_model(7, 80, 64)
args = (w, np.asfortranarray(mu.T), np.asfortranarray(np.transpose(sig, (2, 1, 0))))
rng = np.random.default_rng(0)
for i in range(n):
    T = int(rng.choice([1, 7, 100, 3000]))
    if mode == "A":
        Xp = np.asfortranarray(rng.standard_normal((40, T)))
        Yp = np.empty_like(Xp, order="F")
        vc.pin(Xp)
        vc.pin(Yp)
        g0 = vc.GMMMap(*args)
        vc.fvconvert(g0, Xp, out=Yp)
        vc.unpin(Yp)
        vc.unpin(Xp)
        del Xp, Yp, g0
    junk = [np.empty(int(rng.integers(10, 5000))) for _ in range(20)]
    g = vc.GMMMap(*args)
    del junk, g
    if i % 50 == 0:
        print(mode, i, flush=True)
print(mode, "done", flush=True)
