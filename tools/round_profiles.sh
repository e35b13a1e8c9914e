#!/bin/bash
# One round's evidence, collected on the GPU box:  bash tools/round_profiles.sh r03
#   the driver's default command (`bench.py` with no workload: headline + the `workloads` table over configs[1..4]);
#   per workload: the bench JSON line, the rocprofv3 --kernel-trace --stats summary of the same command, the per-CALL
#   kernel durations of that run (so that averages can be taken over the timed calls only), the HBM traffic from
#   separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (tools/pmc_traffic.py: stamped with the library's source hash)
#   and the effective clock / MFMA-pipe occupancy pass (tools/clock_pmc.py).
# Results land in gpurun_out/<tag>/; copy what is to be judged into profiles/.
tag=${1:-rXX}; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out $out/pmc $out/clock
cd /tmp && export TMPDIR=/tmp
for spec in ${@:-convert convert_fixture convert_joint convert_broad estep estep_fixture estep_full estep_full_fixture em_full dtw dtw:d41 traj traj:chunk100 trajgv}; do
  w=${spec%%:*}; var=${spec#*:}; [ "$var" = "$spec" ] && var=""
  name=$w; extra=""
  [ "$var" = d41 ] && { name=dtw_d41; extra="--dim 41"; }
  [ "$var" = chunk100 ] && { name=traj_chunk100; extra="--chunk 100"; }
  # the PMC passes first: installed into this box's copy of profiles/<tag>_pmc/, so that the bench lines below (and the
  # default command at the end) quote traffic stamped with the hash of the sources they run
  python3 $R/tools/pmc_traffic.py $w --out $out/pmc_$name $extra > /dev/null 2>&1
  [ -f $out/pmc_$name/traffic.json ] && { cp $out/pmc_$name/traffic.json $out/pmc/${name}_traffic.json; mkdir -p $R/profiles/${tag}_pmc; cp $out/pmc_$name/traffic.json $R/profiles/${tag}_pmc/${name}_traffic.json; }
  rm -rf $out/pmc_$name
  steps=20; warm=5; [ $w = trajgv ] && { steps=5; warm=2; }     # as the driver's default run: two warm-up steps leave the clock ramping
  timeout 400 python3 $R/bench.py --workload $w --steps $steps --warmup $warm --pmc off $extra 2>/dev/null | tail -1 > $out/${name}_bench.json
  [ -f $R/gpurun_out/bench_detail_${w}_n1.json ] && cp $R/gpurun_out/bench_detail_${w}_n1.json $out/${name}_bench_detail.json      # the full record behind the line
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -- python3 $R/bench.py --workload $w --steps $steps --warmup $warm --profile-run --clock-warm-ms 0 $extra > /dev/null 2>&1
  f=$(find $out/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $out/${name}_kernel_stats.csv
  t=$(find $out/prof_$name -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python3 - "$t" > $out/${name}_kernel_calls.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
print("call,kernel,start_us,duration_us")
for i, r in enumerate(rows):
    print(f'{i},"{r["Kernel_Name"].split("(")[0][:70]}",{(int(r["Start_Timestamp"]) - t0) / 1e3:.1f},{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:.2f}')
PY
  rm -rf $out/prof_$name
  python3 $R/tools/clock_pmc.py $w --out $out/clk_$name $extra > /dev/null 2>&1
  [ -f $out/clk_$name/clock.json ] && cp $out/clk_$name/clock.json $out/clock/${name}_clock.json
  rm -rf $out/clk_$name
  echo "$name: $(cut -c1-200 $out/${name}_bench.json | grep -o '"value": [0-9.e+]*\|ms_per_step": [0-9.]*' | tr '\n' ' ')"
done
( time python3 $R/bench.py --steps 20 --warmup 5 > $out/all_bench.json 2> $out/all_bench.err ) 2> $out/all_bench.time
[ -f $R/gpurun_out/bench_detail_all_n1.json ] && cp $R/gpurun_out/bench_detail_all_n1.json $out/all_bench_detail.json
echo "default command: $(cat $out/all_bench.time | tr '\n' ' '); line: $(wc -c < $out/all_bench.json) bytes"
