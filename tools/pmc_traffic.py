#!/usr/bin/env python3
"""HBM traffic of a bench workload's kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (they do not fit one
pass: MI355X_MICROARCH.md, HBM section), plus a third pass for SQ_INSTS_MFMA (the matrix instructions each kernel issued:
what bench.py prices the MFMA pipe with), kernel-trace only, of `python3 bench.py --workload <w> --steps 2 --warmup 1
--profile-run --clock-warm-ms 0` (with --cpu-seconds 0 the bench launches nothing but the W warm-up steps, the K timed
steps and a small parity sample; --clock-warm-ms 0 switches the steady-clock pass off, so W + K = 3 steps reach the
counters -- round 4 divided by 3 while ~60 clock-warm launches were counted too: every per-step figure was 4-23x high.
Per-step figures are now per-launch averages x the launches of ONE step, and a launch count that is not a whole multiple
of the 3 steps is recorded in `_meta.problems`, which makes bench.py refuse the table).  Writes <out>/traffic.json: per kernel the KB counters per launch and per bench step, plus `_meta` with the hash
of the library sources the passes were collected from -- bench.py refuses a table whose hash is not the current one.

    python3 tools/pmc_traffic.py <workload> [--out DIR] [extra bench args]

Never touches the GPU itself: rocprofv3 and the bench are child processes (the program after `--` is python3 itself)."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NSTEPS = 3            # 2 timed + 1 warm-up


def main():
    argv = sys.argv[1:]
    w = argv.pop(0)
    out = os.path.join(ROOT, "gpurun_out", "pmc_traffic", w)
    if "--out" in argv:
        i = argv.index("--out")
        out = argv[i + 1]
        del argv[i:i + 2]
    import bench

    res = collections.defaultdict(dict)
    cmd_tail = ["--", sys.executable, os.path.join(ROOT, "bench.py"), "--workload", w, "--steps", "2", "--warmup", "1",
                "--profile-run", "--clock-warm-ms", "0", "--pmc", "off"] + argv
    problems = {}            # kernel -> why its per-step figures cannot be trusted (bench.py refuses a table for the kernels named here)
    for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_MFMA"):
        d = os.path.join(out, c)
        os.makedirs(d, exist_ok=True)
        env = dict(os.environ, TMPDIR="/tmp")
        with open(os.path.join(d, "err.txt"), "w") as err:
            subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", d] + cmd_tail,
                           stdout=subprocess.DEVNULL, stderr=err, timeout=400, cwd="/tmp", env=env, check=True)
        f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
        acc, n = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if r["Counter_Name"] == c:
                acc[k] += float(r["Counter_Value"])
                n[k] += 1
        for k in acc:
            if c == "SQ_INSTS_MFMA":      # matrix instructions per launch (per wave: 2048 flop each for v_mfma_f64_16x16x4)
                res[k]["SQ_INSTS_MFMA_per_launch"] = acc[k] / n[k]
                continue
            per_step = n[k] / NSTEPS
            if n[k] % NSTEPS and n[k] > NSTEPS:      # (kernels launched once, e.g. by the parity sample, are not per-step work)
                problems[k] = "%d launches in the %s pass are not a multiple of the %d steps" % (n[k], c, NSTEPS)
            res[k][c + "_KB_per_launch"] = acc[k] / n[k]
            res[k][c + "_KB_per_step"] = acc[k] / n[k] * per_step
            res[k]["launches_per_step"] = per_step
    for k, d in res.items():
        d["hbm_bytes_per_launch_raw"] = (d.get("FETCH_SIZE_KB_per_launch", 0) + d.get("WRITE_SIZE_KB_per_launch", 0)) * 1024
        d["hbm_bytes_per_step_raw"] = (d.get("FETCH_SIZE_KB_per_step", 0) + d.get("WRITE_SIZE_KB_per_step", 0)) * 1024
    res = dict(res)
    res["_meta"] = {"source_hash": bench.source_hash(), "collected": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
                    "command": "bench.py --workload %s --steps 2 --warmup 1 --profile-run --clock-warm-ms 0 %s" % (w, " ".join(argv)),
                    "steps_per_pass": NSTEPS, "problems": problems}
    json.dump(res, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
