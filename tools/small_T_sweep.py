"""tools/small_T_sweep.py -- device-resident fvconvert of utterance-sized inputs (D = 40, M = 64 and the fixture model):
median time of one synchronised call per T.  Environment knobs of the library (read once per process) select the variant:
VCMI_SMALL_CALL_FRAMES (one frame tile per wave up to that many frames), VCMI_GROUP_MIN_FRAMES (grouping from that many on).

    gpurun -- 'for v in "0 8192" "16384 8192" "16384 512"; do set -- $v; python tools/small_T_sweep.py; done'
"""
import json
import os
import sys
import time

import numpy as np

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
    out = {"env": {k: os.environ.get(k) for k in ("VCMI_SMALL_CALL_FRAMES", "VCMI_GROUP_MIN_FRAMES")}}
    for variant in ("synthetic", "fixture"):
        w, mu, sig = bench.convert_model(variant)
        g = vc.GMMMap(*bench.julia_model(w, mu, sig))
        for T in (1, 64, 256, 512, 1000, 2000, 4000, 8000, 16000, 32000):
            X = torch.from_numpy(sd.sample_frames(7, w, mu, sig, T, 0, 40)).cuda()
            Y = torch.empty_like(X)
            out[f"{variant}_T{T}"] = med(lambda: vc.fvconvert(g, X.t(), out=Y.t()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
