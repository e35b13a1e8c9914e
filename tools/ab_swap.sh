#!/bin/bash
# A/B (or A/B/C...) timing of library builds inside ONE gpurun call, i.e. on one box, one clock state:
#   here:        make -C voiceconversion.jl_amd/csrc; cp voiceconversion.jl_amd/libvcmi.so tools/_lib_base.so
#                (edit) make ...;                       cp voiceconversion.jl_amd/libvcmi.so tools/_lib_new.so
#   on the box:  gpurun -- 'bash tools/ab_swap.sh "base new base new" --workload traj --steps 10 --warmup 2'
# Box-to-box spread is ~3 %, run-to-run on one box ~0.2 %: differences of a per cent are only visible this way.
names=$1; shift
for v in $names; do
  echo -n "$v: "
  LIBVCMI_PROBE=tools/_lib_$v.so python bench.py "$@" --cpu-seconds 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms' % d['ms_per_step'], 'steady %s' % (d.get('steady') or {}).get('steady_ms_per_step'), 'frac %.4f' % d['roofline']['frac'], 'err', d.get('parity_max_rel_err_vs_oracle'))"
done
