#!/bin/bash
# per-phase s_memtime counts of the diagonal E-step kernel (probe build tools/_lib_eprof.so = make EXTRA=-DVCMI_ESTEP_PROF)
cp voiceconversion.jl_amd/libvcmi.so /tmp/_keep.so
cp tools/_lib_eprof.so voiceconversion.jl_amd/libvcmi.so
python bench.py --workload estep --steps 1 --warmup 0 --pmc off --cpu-seconds 0 2>&1 | grep "estep prof" | tail -8
cp /tmp/_keep.so voiceconversion.jl_amd/libvcmi.so
