#!/bin/bash
# per-kernel average durations of one bench workload (rocprofv3 --kernel-trace --stats, profile run: no CPU baseline, no
# clock-warm pass): tools/kstats.sh estep [extra bench.py flags]
w=${1:-convert}; shift
d=$(mktemp -d /tmp/kstats.XXXX)
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $d -o k -- python3 $OLDPWD/bench.py --workload $w --steps 20 --warmup 5 --profile-run --clock-warm-ms 0 "$@" > $d/log 2>&1 )
f=$(find $d -name "*kernel_stats.csv" | head -1)
[ -z "$f" ] && { tail -5 $d/log; exit 1; }
python3 - "$f" <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f'{r["Name"][:70]:70s} {r["Calls"]:>6s} {float(r["AverageNs"]) / 1e3:9.1f} us {r["Percentage"]:>6s} %')
P
