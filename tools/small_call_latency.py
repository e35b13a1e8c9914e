"""tools/small_call_latency.py -- what ONE utterance-sized call costs through the host-pointer entry points (what a Julia
`ccall` passes): fvconvert of T frames (D = 40, M = 64), a DTW pair, a trajectory conversion.  Median of 200 calls each.

    gpurun -- python tools/small_call_latency.py
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402
import synthdata as sd  # noqa: E402
import voiceconversion_jl_amd as vc  # noqa: E402


def med(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    ts = np.sort(ts)
    return {"median_us": round(float(ts[len(ts) // 2]) * 1e6, 1), "p10_us": round(float(ts[len(ts) // 10]) * 1e6, 1),
            "p90_us": round(float(ts[9 * len(ts) // 10]) * 1e6, 1)}


def main():
    out = {}
    w, mu, sig = sd.synth_model(1002, 80, 64)
    g = vc.GMMMap(*bench.julia_model(w, mu, sig))
    for T in (1, 16, 100, 500, 2000, 20000):
        X = sd.sample_frames(7, w, mu, sig, T, 0, 40)
        Xj = np.asfortranarray(X.T)               # Julia (D,T) image
        out[f"fvconvert_T{T}"] = med(lambda: vc.fvconvert(g, Xj))
    # the other users of the direct small-call path: posterior (the finishing kernel re-reads and re-writes the
    # pinned output), predict, vc (a pinned-to-pinned copy in front of the kernel), at the sizes around the direct limit
    for T in (100, 500, 2000, 3000):
        X = sd.sample_frames(7, w, mu, sig, T, 0, 40)
        Xj = np.asfortranarray(X.T)
        fm = np.asfortranarray(np.vstack([np.ones((1, T)), Xj]))
        out[f"predict_proba_T{T}"] = med(lambda: vc.predict_proba(g.px, Xj))
        out[f"predict_T{T}"] = med(lambda: vc.predict(g.px, Xj))
        out[f"vc_T{T}"] = med(lambda: vc.vc(g, fm))
    if os.environ.get("SMALL_CALLS_ONLY"):
        print(json.dumps(out, indent=1))
        return
    wt, mut, sigt = sd.synth_model(1005, 160, 64)
    gt = vc.GMMMap(*bench.julia_model(wt, mut, sigt))
    for T in (100, 500, 2000):
        tg = vc.TrajectoryGMMMap(gt, T)
        X = np.asfortranarray(sd.sample_frames(9, wt, mut, sigt, T, 0, 80).T)
        out[f"traj_fvconvert_T{T}"] = med(lambda: vc.fvconvert(tg, X), n=50, warm=5)
    rng = np.random.default_rng(3)
    for T in (200, 500, 1000):
        a, b = rng.standard_normal((40, T)), rng.standard_normal((40, T + 17))
        d = vc.DTW(fstep=0, bstep=2)
        out[f"dtw_fit_tables_{T}x{T + 17}"] = med(lambda: vc.fit_(d, a, b), n=100, warm=10)
        out[f"dtw_fit_path_only_{T}x{T + 17}"] = med(lambda: vc.fit_(d, a, b, tables=False), n=100, warm=10)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
