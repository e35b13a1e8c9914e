#!/bin/bash
# per-phase s_memtime counts of the diagonal E-step kernel (probe build tools/_lib_eprof.so = make EXTRA=-DVCMI_ESTEP_PROF)
export LIBVCMI_PROBE=tools/_lib_eprof.so      # selected, not copied over the in-tree library
python bench.py --workload estep --steps 1 --warmup 0 --pmc off --cpu-seconds 0 2>&1 | grep "estep prof" | tail -8
