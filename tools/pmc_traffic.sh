#!/bin/bash
# HBM traffic of every kernel of the hot path from the L2 fabric counters, as MI355X_MICROARCH.md
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes, counters together with --kernel-trace only.
# Usage (on the GPU box): tools/pmc_traffic.sh <outdir> [bench config: 1 (default) | 2 | 4]
set -e
out=${1:-$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic}
repo=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cfg=${2:-1}
mkdir -p "$out"
out=$(cd "$out" && pwd)                     # absolute: the passes run from /tmp
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$out/$ctr" -- python3 "$repo/bench.py" --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-f32-line --no-extra-configs > "$out/$ctr.log" 2>&1
done
python3 "$repo/tools/pmc_summarize.py" "$out" > "$out/summary.json"
