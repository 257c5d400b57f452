#!/bin/bash
# One profiling pass of the current build on the GPU box; leaves everything under gpurun_out/prof_<tag>/:
#   kernel_stats_lanes1.csv / kernel_stats_lanes2.csv   rocprofv3 --kernel-trace --stats of bench.py --config 1 (--lanes 1: one
#                                                       image at a time, kernels never overlap -> per-kernel averages are the
#                                                       stand-alone durations; --lanes 2: the default, two images in flight)
#   kernel_stats_cfg2.csv / kernel_stats_cfg4.csv       the same (lanes 1) for BASELINE configs[2] and configs[4]
#   pmc/summary.json, pmc_cfg{2,4}/summary.json         HBM bytes per launch (FETCH_SIZE / WRITE_SIZE in separate passes)
#   pmc_sq/summary.json                                 matrix-pipe occupancy, wave-cycle split, LDS bank conflicts
#   bench_default.json.log                              the default bench line (with its configs[2] / configs[4] objects)
#   csrc.sha256                                         one hash per translation unit (source + headers + flags) these summaries were
#                                                       taken on (+ commit if given as $2): bench.py quotes a kernel family's PMC
#                                                       figures only while ITS unit matches
# The program sits directly after "--" (no wrapper), counters are collected with --kernel-trace only.
# Usage: tools/profile_round.sh <tag> [commit] [parts: all | stats | pmc | sq | bench (comma separated)]
tag=${1:-r03}
commit=${2:-unknown}
parts=${3:-all}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
want() { [ "$parts" = all ] || echo ",$parts," | grep -q ",$1,"; }
python3 - "$commit" > $out/csrc.sha256 <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench                      # unit_hashes(): one hash per translation unit (source + shared headers + compile flags)
print(json.dumps({"commit": sys.argv[1], "units": bench.unit_hashes()}))
PY
common="--no-cpu-baseline --no-roofline --no-f32-line --no-extra-configs"
stats() {   # name, bench arguments
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$1 -o $1 -- python3 bench.py $2 $common > $out/stats_$1.log 2>&1
  f=$(find $out/stats_$1 -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/kernel_stats_$1.csv
  echo "== $1: $(grep -c . $out/kernel_stats_$1.csv 2>/dev/null) rows"; python3 tools/print_stats.py $out/kernel_stats_$1.csv 14
  rm -rf $out/stats_$1
}
if want stats; then
  stats lanes1 "--steps 8 --lanes 1"
  stats lanes2 "--steps 8 --lanes 2"
  stats cfg2 "--config 2 --steps 6 --lanes 1"
  stats cfg4 "--config 4 --steps 3 --warmup 1 --lanes 1"
fi
if want pmc; then
  bash tools/pmc_traffic.sh $out/pmc 1 > $out/pmc.log 2>&1; echo "pmc rc=$?"
  bash tools/pmc_traffic.sh $out/pmc_cfg2 2 > $out/pmc_cfg2.log 2>&1; echo "pmc cfg2 rc=$?"
  bash tools/pmc_traffic.sh $out/pmc_cfg4 4 > $out/pmc_cfg4.log 2>&1; echo "pmc cfg4 rc=$?"
  rm -rf $out/pmc/FETCH_SIZE $out/pmc/WRITE_SIZE $out/pmc_cfg2/FETCH_SIZE $out/pmc_cfg2/WRITE_SIZE $out/pmc_cfg4/FETCH_SIZE $out/pmc_cfg4/WRITE_SIZE
fi
if want sq; then
  bash tools/pmc_sq.sh $out/pmc_sq 1 > $out/pmc_sq.log 2>&1; echo "sq rc=$?"
  rm -rf $out/pmc_sq/sq
fi
if want bench; then
  python3 bench.py > $out/bench_default.json.log 2>&1; echo "bench rc=$?"; tail -1 $out/bench_default.json.log | cut -c1-300
fi
