#!/bin/bash
# per-phase shader cycles of the fvconvert loop, one workgroup (probe build: make EXTRA=-DVCMI_CONVERT_PROF2 -> tools/_lib_prof2.so)
export LIBVCMI_PROBE=tools/_lib_prof2.so      # selected, not copied over the in-tree library
for w in convert convert_fixture; do
  echo "== $w"; python bench.py --workload $w --steps 1 --warmup 0 --pmc off --cpu-seconds 0 2>&1 | grep "convert prof" | tail -4
  echo "== $w dense"; python bench.py --workload $w --steps 1 --warmup 0 --pmc off --cpu-seconds 0 --prune inf 2>&1 | grep "convert prof" | tail -4
done
