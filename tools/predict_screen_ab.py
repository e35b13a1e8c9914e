#!/usr/bin/env python3
"""predict(px, X) on long inputs: grouped frames + screened arg-max (gmmmap_screen_argmax_kernel) against the early-exit kernel
(MODE 3; DBG_PREDICT_NO_SCREEN), D = 80, M = 64, 512,000 frames -- (a) drawn from the model, (b) the trajectory bench's smooth
random walks (out of distribution for every single mixture).  VCMI_TEST_HOOKS=1 python3 tools/predict_screen_ab.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import synthdata as npo  # noqa: E402
import voiceconversion_jl_amd as vc  # noqa: E402
from voiceconversion_jl_amd import _lib  # noqa: E402


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    D, M, T = 40, 64, 512_000
    for lam in (1e-3, 1e-1):
        w, mu, sig = npo.synth_model(1005, 4 * D, M, lam_lo=lam)
        g = vc.GMMMap(w, np.asfortranarray(mu.T), np.asfortranarray(np.transpose(sig, (2, 1, 0))))
        Xm = npo.sample_frames(5, w, mu, sig, T, 0, 2 * D)
        walk = []
        rng = np.random.default_rng(6)
        for _ in range(8):
            st = npo.sample_frames(int(rng.integers(1 << 30)), w, mu, sig, 2000, 0, D)
            st = np.cumsum(st, axis=0) / np.sqrt(np.arange(1, 2001))[:, None]
            walk.append(np.ascontiguousarray(vc.push_delta(np.asfortranarray(st.T)).T))
        Xw = np.concatenate([walk[i % 8] for i in range(T // 2000)])
        for name, X in (("model-drawn", Xm), ("random walks", Xw)):
            Xd = torch.from_numpy(np.ascontiguousarray(X)).cuda()
            res = {}
            for label, flag in (("screened", _lib.DBG_PREDICT_SCREEN), ("early exit", _lib.DBG_PREDICT_NO_SCREEN)):
                _lib.debug_force(flag)
                res[label] = (timeit(lambda: vc.predict(g.px, Xd.t())), vc.predict(g.px, Xd.t()).cpu().numpy())
                _lib.debug_force(0)
            same = np.array_equal(res["screened"][1], res["early exit"][1])
            print(f"lam_lo {lam:g} {name:13s}: screened {res['screened'][0]:.3f} ms, early exit {res['early exit'][0]:.3f} ms, identical {same}, "
                  f"distinct m-hat {len(np.unique(res['screened'][1]))}")


if __name__ == "__main__":
    main()
