#!/usr/bin/env python3
"""A/B of the fvconvert loop shapes on one box: for each model (SURVEY 8d synthetic, broad synthetic lam_lo = 1e-1, the
reference's trained 32-mixture model) and each shape (auto / broad / peaked / dense) the kernel time of 10^6 frames, the
regressions and MFMAs actually issued, and the output against the dense loop and the C oracle.

    python3 tools/convert_ab.py [--frames N] [--steps K] [--libs a.so,b.so]   (run on the GPU box)

With --libs every library build is measured in its own child process (the library is loaded once per process)."""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def models():
    import synthdata as sd

    out = {}
    w, mu, sig = sd.synth_model(1002, 80, 64)
    out["synthetic"] = (w, mu, sig)
    w, mu, sig = sd.synth_model(1002, 80, 64, lam_lo=1e-1)
    out["broad"] = (w, mu, sig)
    z = np.load(os.path.join(ROOT, "tests", "golden", "model_clb_to_slt_gmm32_order40_diff.npz"))
    out["fixture"] = (z["weights"], z["means"], z["covars"])
    return out


def run(args):
    os.environ["VCMI_TEST_HOOKS"] = "1"
    import torch

    import synthdata as sd
    import voiceconversion_jl_amd as vc
    from voiceconversion_jl_amd import _lib
    from oracle import c_oracle as co

    T = args.frames
    res = {}
    for name, (w, mu, sig) in models().items():
        M, Dj = mu.shape
        D = Dj // 2
        X = sd.sample_frames(7, w, mu, sig, T, 0, D)
        Xd = torch.from_numpy(X).cuda()
        Yd = torch.empty_like(Xd)
        g = vc.GMMMap(w, np.asfortranarray(mu.T), np.asfortranarray(np.transpose(sig, (2, 1, 0))))
        n0 = 2000
        Yref = co.GMMMap(w, mu, sig).fvconvert(X[:n0])
        g.set_prune(float("inf"))
        vc.fvconvert(g, Xd.t(), out=Yd.t())
        Ydense = Yd.clone()
        res[name] = {"model_active_frac": g.convert_plan()[2], "model_undecided_frac": g.convert_plan()[3], "M": M}
        for label, force, prune in (("auto", 0, 46.0), ("broad", 4096, 46.0), ("peaked", 8192, 46.0), ("nogroup_broad", 4096 | 2048, 46.0),
                                    ("dense", 0, float("inf")), ("dense_r3loop", 8192, float("inf"))):
            _lib.debug_force(force)
            g.set_prune(prune)
            for _ in range(3):
                vc.fvconvert(g, Xd.t(), out=Yd.t())
            g.prune_stats(True)
            vc.fvconvert(g, Xd.t(), out=Yd.t())
            torch.cuda.synchronize()
            nmf, shape, _, _ = g.convert_plan()
            nreg = g.prune_stats(False)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
            for a, b in evs:
                a.record()
                vc.fvconvert(g, Xd.t(), out=Yd.t())
                b.record()
            torch.cuda.synchronize()
            ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
            d = float((torch.linalg.norm(Yd - Ydense, dim=1) / torch.linalg.norm(Ydense, dim=1)).max())
            Y = Yd[:n0].cpu().numpy()
            err = float(np.max(np.linalg.norm(Y - Yref, axis=1) / np.linalg.norm(Yref, axis=1)))
            tiles = -(-T // 16)
            res[name][label] = {"ms": round(ms, 4), "shape": shape, "regressions_frac": round(nreg / (tiles * M), 4),
                                "mfma_issued": nmf, "issued_frac_of_roof": round(nmf * 2048 / (ms * 1e-3) / 78.6e12, 4),
                                "algorithmic_frac": round(M * (3 * D * D + 6 * D + 25) * T / (ms * 1e-3) / 78.6e12, 4),
                                "vs_dense": d, "vs_oracle": err}
        _lib.debug_force(0)
    print(json.dumps(res))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--libs", default="")
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.libs and not args.child:
        for lib in args.libs.split(","):             # selected through LIBVCMI_PROBE: the in-tree library is never overwritten
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--frames", str(args.frames), "--steps", str(args.steps)],
                               capture_output=True, text=True, env=dict(os.environ, LIBVCMI_PROBE=os.path.join(ROOT, lib)))
            print("==", lib)
            try:
                d = json.loads(p.stdout.strip().splitlines()[-1])
                for model, r in d.items():
                    print(" ", model, "active_frac %.3f undecided_frac %.3f" % (r["model_active_frac"], r["model_undecided_frac"]))
                    for k, v in r.items():
                        if isinstance(v, dict):
                            print("    %-14s %8.4f ms  shape %d  reg %.4f  issued %.3f  alg %.3f  vs_dense %.1e  vs_oracle %.1e" %
                                  (k, v["ms"], v["shape"], v["regressions_frac"], v["issued_frac_of_roof"], v["algorithmic_frac"], v["vs_dense"], v["vs_oracle"]))
            except Exception as e:  # noqa: BLE001
                print("FAILED", e, p.stdout[-2000:], p.stderr[-3000:])
        return
    run(args)


if __name__ == "__main__":
    main()
