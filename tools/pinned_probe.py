#!/usr/bin/env python3
"""Host-pointer fvconvert on caller-pinned arrays: vcmi_host_register'ed numpy arrays against torch pin_memory
(hipHostMalloc) arrays against pageable ones, 10^6 frames of D = 40, for the pipeline chunk size in VCMI_HOST_CHUNK_MB
(read once per process: run once per value).  Prints ms per call (best and mean of 5)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import synthdata as npo  # noqa: E402
import voiceconversion_jl_amd as vc  # noqa: E402


def timeit(f, n=5):
    f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t0) * 1e3)
    return "%.2f / %.2f ms" % (min(ts), sum(ts) / len(ts))


def main():
    T, D = 1_000_000, 40
    w, mu, sig = npo.synth_model(1002, 80, 64, lam_lo=1e-5)
    g = vc.GMMMap(w, np.asfortranarray(mu.T), np.asfortranarray(np.transpose(sig, (2, 1, 0))))
    X = np.asfortranarray(npo.sample_frames(1002, w, mu, sig, T, 0, D).T)
    Y = np.empty_like(X, order="F")
    print("chunk MB", os.environ.get("VCMI_HOST_CHUNK_MB", "default"))
    print("pageable          ", timeit(lambda: vc.fvconvert(g, X, out=Y)))
    Y0 = Y.copy()
    vc.pin(X), vc.pin(Y)
    print("registered (4K pg)", timeit(lambda: vc.fvconvert(g, X, out=Y)), np.array_equal(Y, Y0))
    vc.unpin(X), vc.unpin(Y)
    xt = torch.from_numpy(np.ascontiguousarray(X.T)).pin_memory()
    yt = torch.empty_like(xt).pin_memory()
    Xp, Yp = xt.numpy().T, yt.numpy().T
    print("hipHostMalloc     ", timeit(lambda: vc.fvconvert(g, Xp, out=Yp)), np.array_equal(Yp, Y0))
    # device-resident: the kernel alone
    xd, yd = xt.cuda(), torch.empty_like(xt, device="cuda")
    torch.cuda.synchronize()

    def dev():
        vc.fvconvert(g, xd.t(), out=yd.t())
        torch.cuda.synchronize()
    print("device-resident   ", timeit(dev))


if __name__ == "__main__":
    main()
