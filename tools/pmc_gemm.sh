#!/bin/bash
# PMC pass for the forward pass (diagnostic).  Usage: tools/pmc_gemm.sh <outdir> <counters...>
out=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/profile_layers.py --batch 50 --reps 1 --top 0
