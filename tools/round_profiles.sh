#!/bin/bash
# One round's evidence, collected on the GPU box:  bash tools/round_profiles.sh r02
#   per workload: the bench JSON line, the rocprofv3 --kernel-trace --stats summary of the same command, the per-CALL
#   kernel durations of that run (so that averages can be taken over the timed calls only), and the HBM traffic from
#   separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (tools/pmc_traffic.sh).
# Results land in gpurun_out/<tag>/; copy what is to be judged into profiles/.
tag=${1:-rXX}; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out $out/pmc
cd /tmp && export TMPDIR=/tmp
for w in ${@:-convert estep estep_full em_full dtw traj trajgv}; do
  steps=10; [ $w = trajgv ] && steps=3
  extra=""
  timeout 400 python3 $R/bench.py --workload $w --steps $steps --warmup 2 $extra 2>/dev/null | tail -1 > $out/${w}_bench.json
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -- python3 $R/bench.py --workload $w --steps $steps --warmup 2 --cpu-seconds 0 > /dev/null 2>&1
  f=$(find $out/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $out/${w}_kernel_stats.csv
  t=$(find $out/prof_$w -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python3 - "$t" > $out/${w}_kernel_calls.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
print("call,kernel,start_us,duration_us")
for i, r in enumerate(rows):
    print(f'{i},"{r["Kernel_Name"].split("(")[0][:70]}",{(int(r["Start_Timestamp"]) - t0) / 1e3:.1f},{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:.2f}')
PY
  rm -rf $out/prof_$w
  bash $R/tools/pmc_traffic.sh $w > /dev/null 2>&1
  [ -f $R/gpurun_out/pmc_traffic/$w/traffic.json ] && cp $R/gpurun_out/pmc_traffic/$w/traffic.json $out/pmc/${w}_traffic.json
  echo "$w: $(cut -c1-200 $out/${w}_bench.json | grep -o '"value": [0-9.e+]*\|ms_per_step": [0-9.]*' | tr '\n' ' ')"
done
