#!/bin/bash
# like ab_swap.sh, for tools/dtw_shapes.py
for v in $1; do echo "== $v"; LIBVCMI_PROBE=tools/_lib_$v.so timeout 60 python tools/dtw_shapes.py 2>&1 | grep "n= " ; done
