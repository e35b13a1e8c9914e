"""tools/small_T_sweep.py -- device-resident fvconvert of utterance-sized inputs (D = 40, M = 64 and the fixture model):
median time of one synchronised call per T, with the library's choice of launch shape (one frame tile per wave up to
32768 frames) and with the throughput shape forced (DBG_CONVERT_WIDE_TILES).  The thresholds in csrc/gmmmap.hip
(kSmallCallFrames, kSortMinFrames) come from this sweep, run with the two as environment knobs of a probe build.

    gpurun -- python tools/small_T_sweep.py
"""
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("VCMI_TEST_HOOKS", "1")       # vcmi_debug_force is inert without it

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import bench  # noqa: E402
import synthdata as sd  # noqa: E402
import voiceconversion_jl_amd as vc  # noqa: E402


def med(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    return round(float(np.median(ts)) * 1e6, 1)


def main():
    from voiceconversion_jl_amd import _lib

    out = {}
    for variant in ("synthetic", "fixture"):
        w, mu, sig = bench.convert_model(variant)
        g = vc.GMMMap(*bench.julia_model(w, mu, sig))
        for T in (1, 64, 256, 512, 1000, 2000, 4000, 8000, 16000, 32000):
            X = torch.from_numpy(sd.sample_frames(7, w, mu, sig, T, 0, 40)).cuda()
            Y = torch.empty_like(X)
            row = []
            for force in (0, _lib.DBG_CONVERT_WIDE_TILES):
                _lib.debug_force(force)
                try:
                    row.append(med(lambda: vc.fvconvert(g, X.t(), out=Y.t())))
                finally:
                    _lib.debug_force(0)
            out[f"{variant}_T{T}"] = {"library_us": row[0], "wide_tiles_us": row[1]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
