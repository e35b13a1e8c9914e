#!/usr/bin/env python3
"""A/B of fvconvert shape 3 on one box: the persistent kernel (gmmmap_screen_pers.hpp, round 6) against round 5's
one-workgroup-per-128-frames kernel (DBG_SCREEN_NO_PERSIST), BASELINE configs[1] (D = 40, M = 64, 10^6 frames): step time by
HIP events, regressions / MFMAs issued, the outputs against each other, against the dense loop and against the oracle.

    python3 tools/pers_ab.py [--frames N] [--steps K]        (run on the GPU box)"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VCMI_TEST_HOOKS"] = "1"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import torch

    import synthdata as sd
    import voiceconversion_jl_amd as vc
    from oracle import c_oracle as co
    from voiceconversion_jl_amd import _lib

    T = args.frames
    w, mu, sig = sd.synth_model(1002, 80, 64)
    X = sd.sample_frames(1002, w, mu, sig, T, 0, 40)
    Xd = torch.from_numpy(X).cuda()
    Yd = torch.empty_like(Xd)
    g = vc.GMMMap(w, np.asfortranarray(mu.T), np.asfortranarray(np.transpose(sig, (2, 1, 0))))
    n0 = 4096
    pos = np.concatenate([np.arange(n0 // 2), T - 1 - np.arange(n0 // 2)])
    Yref = co.GMMMap(w, mu, sig).fvconvert(X[pos])
    g.set_prune(float("inf"))
    vc.fvconvert(g, Xd.t(), out=Yd.t())
    Ydense = Yd.clone()
    g.set_prune(46.0)
    res = {"plan": g.convert_plan()}
    outs = {}
    for label, force in (("persistent", 0), ("round5", _lib.DBG_SCREEN_NO_PERSIST), ("persistent_again", 0)):
        _lib.debug_force(force)
        try:
            for _ in range(5):
                vc.fvconvert(g, Xd.t(), out=Yd.t())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                vc.fvconvert(g, Xd.t(), out=Yd.t())
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.steps
            g.prune_stats(True)
            vc.fvconvert(g, Xd.t(), out=Yd.t())
            issued = g.convert_plan()[0]
            nreg = g.prune_stats(False)
            Y2 = Yd.clone()
            vc.fvconvert(g, Xd.t(), out=Yd.t())
            rep = bool(torch.equal(Y2, Yd))
        finally:
            _lib.debug_force(0)
        outs[label] = Y2
        rel = lambda a, b: float((torch.linalg.norm(a - b, dim=1) / torch.linalg.norm(b, dim=1)).max())  # noqa: E731
        Yh = Y2[torch.from_numpy(pos).cuda()].cpu().numpy()
        res[label] = {"ms": ms, "frames_per_s": T / ms * 1e3, "hbm_frac_algorithmic": 640.0 * T / (ms * 1e-3) / 8e12, "mfma_issued": issued,
                      "regressions": nreg, "repeat_identical": rep, "vs_dense": rel(Y2, Ydense),
                      "vs_oracle_%d_frames_head_and_tail" % len(pos): float(np.max(np.linalg.norm(Yh - Yref, axis=1) / np.linalg.norm(Yref, axis=1)))}
    res["persistent_vs_round5_max_rel"] = float((torch.linalg.norm(outs["persistent"] - outs["round5"], dim=1) /
                                                 torch.linalg.norm(outs["round5"], dim=1)).max())
    res["persistent_bitwise_equal_round5"] = bool(torch.equal(outs["persistent"], outs["round5"]))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
