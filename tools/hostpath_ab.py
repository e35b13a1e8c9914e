#!/usr/bin/env python3
"""Host-pointer fvconvert (what a ccall passes) on one box, per library build: fresh / reused output, and the link's own
rate through the same pinned slots (vcmi_debug_pcie_probe).   python3 tools/hostpath_ab.py tools/_lib_a.so,tools/_lib_b.so"""
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import bench
    import synthdata as sd
    import voiceconversion_jl_amd as vc

    w, mu, sig = sd.synth_model(1002, 80, 64)
    X = sd.sample_frames(1002, w, mu, sig, 1_000_000, 0, 40)
    g = vc.GMMMap(*bench.julia_model(w, mu, sig))
    Xh = np.asfortranarray(X.T)
    vc.fvconvert(g, Xh)
    res = {}
    for rep in range(2):
        keep = []
        t0 = time.perf_counter()
        for _ in range(4):
            keep.append(vc.fvconvert(g, Xh))
        res["fresh_ms_%d" % rep] = (time.perf_counter() - t0) / 4 * 1e3
        del keep
        Yh = np.empty_like(Xh, order="F")
        vc.fvconvert(g, Xh, out=Yh)
        t0 = time.perf_counter()
        for _ in range(4):
            vc.fvconvert(g, Xh, out=Yh)
        res["reused_ms_%d" % rep] = (time.perf_counter() - t0) / 4 * 1e3
    try:
        res["pcie"] = bench.pcie_roof(Xh.nbytes)
    except Exception as e:  # noqa: BLE001
        res["pcie"] = repr(e)
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        for lib in sys.argv[1].split(","):           # selected through LIBVCMI_PROBE: the in-tree library is never overwritten
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], capture_output=True, text=True,
                               env=dict(os.environ, LIBVCMI_PROBE=os.path.join(ROOT, lib)))
            print("==", lib, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-1500:])
