"""tools/zero_copy_probe.py -- the 10^6-frame host call with the kernels reading / writing the pinned slots themselves
(what calls of up to 2 MB do) instead of the staging ring:

    gpurun -- 'python tools/zero_copy_probe.py; VCMI_HOST_DIRECT_KB=4000000 python tools/zero_copy_probe.py'

Round 4, one box: ring 10.7 ms, direct 23.1 ms -- the grouping key kernel streams x over the link at 52 GB/s (6.1 ms), then
the conversion kernel reads the same rows AGAIN, permuted, and writes y (10.8 ms).  The direct path stays a small-call path.
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, synthdata as sd
import voiceconversion_jl_amd as vc
w, mu, sig = sd.synth_model(1002, 80, 64)
g = vc.GMMMap(*bench.julia_model(w, mu, sig))
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
X = np.asfortranarray(sd.sample_frames(7, w, mu, sig, T, 0, 40).T)
Y = np.empty_like(X)
for i in range(3):
    vc.fvconvert(g, X, out=Y)
ts = []
for i in range(8):
    t = time.perf_counter(); vc.fvconvert(g, X, out=Y); ts.append(time.perf_counter() - t)
print("env", os.environ.get("VCMI_HOST_DIRECT_KB"), "T", T, "ms", [round(t * 1e3, 2) for t in ts], flush=True)
