#!/bin/bash
# A/B of dtw_obs_kernel variants (VCMI_OBS_VARIANT = 10*rows_per_lane + columns_per_iteration) -- run on the GPU box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in ${VARIANTS:-0 21 51 52}; do
  export VCMI_OBS_VARIANT=$v
  out=$R/gpurun_out/obsvar/v$v
  mkdir -p $out
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/bench.py --workload dtw --steps 3 --warmup 1 --cpu-seconds 0 > $out/bench.json 2>/dev/null
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "variant $v: $(grep dtw_obs $f | cut -d, -f1-4 | sed 's/.*dtw_obs_kernel/obs/') | $(grep dtw_rec $f | awk -F, '{print "rec avg ns", $(NF-4)}')"
done
