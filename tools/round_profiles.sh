#!/bin/bash
# One round's evidence: for every bench workload the JSON line and the rocprofv3 kernel statistics of the same command.
# usage (on the GPU box): bash tools/round_profiles.sh r01c
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for w in convert estep estep_full em_full dtw traj trajgv; do
  steps=10; [ $w = trajgv ] && steps=3
  timeout 400 python3 $R/bench.py --workload $w --steps $steps --warmup 2 2>/dev/null | tail -1 > $out/${w}_bench.json
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -- python3 $R/bench.py --workload $w --steps $steps --warmup 2 --cpu-seconds 0 > /dev/null 2>&1
  f=$(find $out/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $out/${w}_kernel_stats.csv
  echo "$w: $(cut -c1-200 $out/${w}_bench.json | grep -o '"value": [0-9.e+]*\|ms_per_step": [0-9.]*' | tr '\n' ' ')"
done
