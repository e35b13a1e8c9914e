#!/bin/bash
# usage: tools/variants.sh "0 1 2 3"  -- run bench.py for each VCMI_VARIANT and print kernel ms / roofline fraction
for v in $1; do
  VCMI_VARIANT=$v python bench.py --steps 10 --warmup 2 --cpu-seconds 1 2>&1 | tail -1 > /tmp/bench_v.json
  python3 - "$v" <<'PY'
import sys, json
d = json.load(open('/tmp/bench_v.json'))
print("variant", sys.argv[1], "kernel_ms %.3f" % d["roofline"]["kernel_ms"], "frac %.3f" % d["roofline"]["frac"], "err %.2e" % d["parity_max_rel_err_vs_oracle"])
PY
done
