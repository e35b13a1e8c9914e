#!/bin/bash
# SQ activity breakdown of a bench workload's kernels (one PMC pass).  usage: tools/kernel_pmc.sh <workload> [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
w=$1; shift
out=$R/gpurun_out/kpmc/$w
mkdir -p $out
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU --output-format csv -d $out -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 --cpu-seconds 0 "$@" > /dev/null 2>$out/err.txt
python3 - "$(find $out -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); n[k] += 1
for k, d in acc.items():
    wc = d.get("SQ_WAVE_CYCLES", 1)
    if wc < 1e6: continue
    print(k, "launches", n[k], {c.replace("SQ_", ""): round(v / wc, 3) for c, v in d.items() if c not in ("SQ_WAVE_CYCLES", "SQ_INSTS_VALU")},
          "wave_quadcycles/launch %.3g" % (wc / n[k]), "valu_insts/launch %.3g" % (d.get("SQ_INSTS_VALU", 0) / n[k]))
PY
