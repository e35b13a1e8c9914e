"""Timing of the fused DTW kernels on uniform batches (development aid): separates the loop's efficiency from strip /
tail effects.  usage: python tools/dtw_shapes.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import voiceconversion_jl_amd as vc
from voiceconversion_jl_amd import _lib

D = 40


def run(n, S, T, reps=5):
    rng = np.random.default_rng(0)
    t = rng.standard_normal((S, D)); s = rng.standard_normal((T, D))
    feats = torch.from_numpy(np.concatenate([np.concatenate([t.ravel(), s.ravel()])] * n)).cuda()
    per = (S + T) * D
    toff = np.arange(n, dtype=np.int64) * per
    soff = toff + S * D
    poff = np.arange(n, dtype=np.int64) * T
    Sa = np.full(n, S, dtype=np.int64); Ta = np.full(n, T, dtype=np.int64)
    out = torch.empty(n * T, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def step():
        _lib.check(_lib.lib.vcmi_dtw_fit_batch_dev(n, feats.data_ptr(), _lib.iptr(toff), _lib.iptr(Sa), _lib.iptr(soff), _lib.iptr(Ta),
                                                   D, 0, 2, out.data_ptr(), _lib.iptr(poff), st))
    step(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        step()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    waves = n * ((S + 127) // 128 if S <= 512 else 0)
    print(f"n={n:5d} S={S} T={T}: {ms:.3f} ms   cells/s {n*S*T/ms*1e-6:.1f} G")


for n, S, T in [(256, 512, 500), (1024, 512, 500)]:
    run(n, S, T)
