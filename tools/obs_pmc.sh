#!/bin/bash
# SQ stall breakdown of the DTW kernels (one PMC pass, no tracing besides kernel-trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in ${VARIANTS:-0 21}; do
  export VCMI_OBS_VARIANT=$v
  out=$R/gpurun_out/obspmc/v$v
  mkdir -p $out
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $out -- python3 $R/bench.py --workload dtw --steps 2 --warmup 1 --cpu-seconds 0 > $out/bench.json 2>$out/err.txt
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    if "dtw" not in k: continue
    wc = d.get("SQ_WAVE_CYCLES", 1)
    print(k, {c: round(v / wc, 3) for c, v in d.items() if c != "SQ_WAVE_CYCLES"}, "wave_cycles=%.3g" % wc)
PY
done
