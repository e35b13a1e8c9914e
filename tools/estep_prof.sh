#!/bin/bash
# per-phase s_memtime counts of the diagonal E-step kernels (probe build tools/_lib_eprof.so = make EXTRA=-DVCMI_ESTEP_PROF):
# the one-barrier kernel (estep_wave.hpp) and, with the test hook, the three-barrier one
export LIBVCMI_PROBE=tools/_lib_eprof.so      # selected, not copied over the in-tree library
python3 bench.py --workload estep --steps 1 --warmup 0 --profile-run --clock-warm-ms 0 2>&1 | grep "estep.*prof" | sort | tail -8
VCMI_TEST_HOOKS=1 python3 bench.py --workload estep --steps 1 --warmup 0 --profile-run --clock-warm-ms 0 --debug-force 4194304 2>&1 | grep "estep.*prof" | sort | tail -8
