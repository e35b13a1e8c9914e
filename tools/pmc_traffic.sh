#!/bin/bash
# HBM traffic of a bench workload's kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md: they
# do not fit one pass), kernel-trace only.  usage: tools/pmc_traffic.sh <workload> [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
w=$1; shift
for c in FETCH_SIZE WRITE_SIZE; do
  out=$R/gpurun_out/pmc_traffic/$w/$c
  mkdir -p $out
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 --cpu-seconds 0 "$@" > /dev/null 2>$out/err.txt
done
python3 - "$R/gpurun_out/pmc_traffic/$w" <<'PY'
import csv, sys, glob, collections, json
root = sys.argv[1]
NSTEPS = 3
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{root}/{c}/*/*counter_collection.csv")[0]
    acc, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if r["Counter_Name"] == c:
            acc[k] += float(r["Counter_Value"]); n[k] += 1
    for k in acc:
        res[k][c + "_KB_per_launch"] = acc[k] / n[k]
        res[k][c + "_KB_per_step"] = acc[k] / NSTEPS        # the bench ran NSTEPS steps (2 timed + 1 warm-up)
        res[k]["launches_per_step"] = n[k] / NSTEPS
for k, d in res.items():
    d["hbm_bytes_per_launch_raw"] = (d.get("FETCH_SIZE_KB_per_launch", 0) + d.get("WRITE_SIZE_KB_per_launch", 0)) * 1024
    d["hbm_bytes_per_step_raw"] = (d.get("FETCH_SIZE_KB_per_step", 0) + d.get("WRITE_SIZE_KB_per_step", 0)) * 1024
print(json.dumps(res, indent=1))
json.dump(res, open(f"{root}/traffic.json", "w"), indent=1)
PY
