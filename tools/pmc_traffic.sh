#!/bin/bash
# HBM traffic of a bench workload's kernels (see tools/pmc_traffic.py).  usage: tools/pmc_traffic.sh <workload> [extra bench args]
cd /tmp && export TMPDIR=/tmp
exec python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py "$@"
