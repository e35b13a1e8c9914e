#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE against KNOWN traffic, per access pattern (tools/microbench_fetch.hip: every kernel reads or writes
each byte of a 320 MB buffer exactly once).  Two rocprofv3 passes (one counter each, kernel-trace only), counters in KB.
Writes <out>/fetch_calibration.json: counter x 1024 / bytes per kernel = the factor by which the counter under-/over-reports
that pattern on this GPU.   python3 tools/fetch_calibration.py [--out DIR]
Never touches the GPU itself: rocprofv3 and the microbenchmark are child processes."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BYTES = 1000000 * 40 * 8


def main():
    out = os.path.join(ROOT, "gpurun_out", "fetch_calibration")
    if "--out" in sys.argv:
        out = sys.argv[sys.argv.index("--out") + 1]
    exe = os.path.join(ROOT, "tools", "microbench_fetch")
    res = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out, c)
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "err.txt"), "w") as err:
            subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", d, "--", exe],
                           stdout=subprocess.DEVNULL, stderr=err, timeout=300, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), check=True)
        f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
        acc, n = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                k = r["Kernel_Name"].split("(")[0]
                acc[k] += float(r["Counter_Value"])
                n[k] += 1
        for k in acc:
            res[k][c + "_KB_per_launch"] = acc[k] / n[k]
            res[k][c + "_over_bytes"] = acc[k] / n[k] * 1024 / BYTES
    res = dict(res)
    res["_meta"] = {"bytes_per_kernel": BYTES, "note": "FETCH_over_bytes of a read kernel / WRITE_over_bytes of a write kernel = counter x 1024 / bytes moved"}
    os.makedirs(out, exist_ok=True)
    json.dump(res, open(os.path.join(out, "fetch_calibration.json"), "w"), indent=1)
    for k, v in res.items():
        if k != "_meta":
            print("%-14s FETCH x1024/bytes %.3f   WRITE x1024/bytes %.3f" % (k, v.get("FETCH_SIZE_over_bytes", 0), v.get("WRITE_SIZE_over_bytes", 0)))


if __name__ == "__main__":
    main()
