#!/usr/bin/env python3
"""Where a kernel's wave cycles go: several rocprofv3 PMC passes (<= 8 SQ counters each) over tools/convert_once.py (or --prog), per-launch
averages per kernel.   python3 tools/sq_pmc.py [--prog tools/estep_once.py] [--force FLAGS] [--out DIR]        (run on the GPU box)"""
import collections, csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [
    ["GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"],
    ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH"],
    ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INST_CYCLES_SALU", "SQ_THREAD_CYCLES_VALU"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_LDS", "SQ_INST_LEVEL_SMEM", "SQ_INSTS_FLAT", "SQ_IFETCH", "SQ_WAVES_EQ_64"],
]


def main():
    argv = sys.argv[1:]
    force, out, prog = "0", os.path.join(ROOT, "gpurun_out", "sq_pmc"), os.path.join("tools", "convert_once.py")
    if "--prog" in argv:                       # e.g. tools/estep_once.py
        prog = argv[argv.index("--prog") + 1]
    if "--force" in argv:
        force = argv[argv.index("--force") + 1]
    if "--out" in argv:
        out = argv[argv.index("--out") + 1]
    out = os.path.abspath(out)
    os.makedirs(out, exist_ok=True)
    res = collections.defaultdict(dict)
    for i, counters in enumerate(PASSES):
        d = os.path.join(out, f"pass{i}")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.join(ROOT, prog), "8", force]
        with open(os.path.join(out, f"err{i}.txt"), "w") as err:
            r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=err, timeout=400, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
        if r.returncode != 0:
            res["_errors"][f"pass{i}"] = open(os.path.join(out, f"err{i}.txt")).read()[-600:]
            continue
        cc = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
        if not cc:
            res["_errors"][f"pass{i}"] = "no counter file"
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(dict)
        for r in csv.DictReader(open(cc[0])):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][r["Dispatch_Id"]] = float(r.get("End_Timestamp", 0) or 0) - float(r.get("Start_Timestamp", 0) or 0)
        for k, dd in acc.items():
            n = len(disp[k])
            res[k]["launches"] = n
            res[k].setdefault("ms_per_launch_profiled", []).append(sum(disp[k].values()) / n / 1e6)
            for c, v in dd.items():
                res[k][c] = v / n
    json.dump(res, open(os.path.join(out, "sq.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
