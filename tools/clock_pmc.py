#!/usr/bin/env python3
"""Effective shader clock and MFMA-pipe occupancy of a bench workload's kernels from one rocprofv3 PMC pass.

    python3 tools/clock_pmc.py <workload> [--out DIR] [extra bench args]

effective clock = GRBM_GUI_ACTIVE / 8 / kernel wall time (rocprofv3 sums the counter over the 8 XCDs; MI355X_MICROARCH.md,
'DVFS give-back': within 3 % of the in-kernel clock on dispatches of 10 ms or more, reads high below ~0.3 ms).  The FP64
roof of 78.6 TFLOP/s assumes 2.4 GHz; what the chip sustains under an FP64-MFMA-dense kernel is lower, and a kernel's
fraction of the roof AT THE CLOCK IT RAN AT is achieved / (78.6 * clock / 2.4).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES /
(SQ_BUSY_CYCLES summed over SIMDs) as rocprofv3 reports them.  rocprofv3 and the bench are child processes."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTERS = ["GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU",
            "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY"]


def main():
    argv = sys.argv[1:]
    w = argv.pop(0)
    out = os.path.join(ROOT, "gpurun_out", "clock_pmc", w)
    if "--out" in argv:
        i = argv.index("--out")
        out = argv[i + 1]
        del argv[i:i + 2]
    os.makedirs(out, exist_ok=True)
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + COUNTERS + ["--output-format", "csv", "-d", out, "--", sys.executable,
           os.path.join(ROOT, "bench.py"), "--workload", w, "--steps", "20", "--warmup", "5", "--profile-run", "--pmc", "off"] + argv     # (default --clock-warm-ms: the steady pass is in the averages)
    with open(os.path.join(out, "err.txt"), "w") as err:
        subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=err, timeout=400, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), check=True)
    cc = glob.glob(os.path.join(out, "*", "*counter_collection.csv"))[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k][r["Dispatch_Id"]] = float(r.get("End_Timestamp", 0) or 0) - float(r.get("Start_Timestamp", 0) or 0)
    res = {}
    for k, d in acc.items():
        n = len(disp[k])
        dur_ns = sum(disp[k].values()) / n
        if dur_ns < 2e5:
            continue
        gui = d.get("GRBM_GUI_ACTIVE", 0.0) / n
        res[k] = {"launches": n, "ms_per_launch_profiled": dur_ns / 1e6,
                  "effective_clock_GHz": gui / 8.0 / dur_ns if dur_ns else None,
                  "mfma_busy_frac": d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(d.get("SQ_BUSY_CYCLES", 1.0), 1.0),
                  "per_launch": {c: d.get(c, 0.0) / n for c in COUNTERS}}
    json.dump(res, open(os.path.join(out, "clock.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
