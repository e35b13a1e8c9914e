#!/bin/bash
# like ab_swap.sh, for tools/dtw_shapes.py
for v in $1; do cp tools/_lib_$v.so voiceconversion.jl_amd/libvcmi.so; echo "== $v"; timeout 60 python tools/dtw_shapes.py 2>&1 | grep "n= " ; done
