#!/bin/bash
# MFMA pipe occupancy vs other VALU work of a bench workload's kernels (one PMC pass).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
w=$1; shift
out=$R/gpurun_out/mpmc/$w
mkdir -p $out
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --cpu-seconds 0 "$@" > /dev/null 2>$out/err.txt
python3 - "$(find $out -name '*counter_collection.csv' | head -1)" "$(find $out -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); n[k] += 1
dur = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Kernel_Name"].split("(")[0][:48]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, d in acc.items():
    if d.get("SQ_WAVE_CYCLES", 0) < 1e6: continue
    print(k, "launches", n[k], "ms/launch %.3f" % (dur[k] / n[k] / 1e6), {c.replace("SQ_", ""): "%.4g" % (v / n[k]) for c, v in d.items()})
PY
