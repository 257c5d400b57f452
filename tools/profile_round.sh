#!/bin/bash
# One profiling pass of the current build on the GPU box; leaves everything under gpurun_out/prof_<tag>/:
#   kernel_stats_lanes1.csv / kernel_stats_lanes2.csv   rocprofv3 --kernel-trace --stats of bench.py (--lanes 1: one image at
#                                                       a time, kernels never overlap -> per-kernel averages are the
#                                                       stand-alone durations; --lanes 2: the default, two images in flight)
#   pmc/summary.json                                    HBM bytes per launch (FETCH_SIZE / WRITE_SIZE in separate passes)
#   bench_default.json.log                              the default bench line
# The program sits directly after "--" (no wrapper), counters are collected with --kernel-trace only.
tag=${1:-r02}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
for lanes in 1 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_lanes$lanes -o l$lanes -- python3 bench.py --steps 8 --lanes $lanes --no-cpu-baseline --no-roofline --no-f32-line > $out/stats_lanes$lanes.log 2>&1
  f=$(find $out/stats_lanes$lanes -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/kernel_stats_lanes$lanes.csv
  echo "== lanes $lanes: $(grep -c . $out/kernel_stats_lanes$lanes.csv 2>/dev/null) rows"; python3 tools/print_stats.py $out/kernel_stats_lanes$lanes.csv 12
done
bash tools/pmc_traffic.sh $out/pmc > $out/pmc.log 2>&1; echo "pmc rc=$?"; python3 - <<PY
import json
d=json.load(open("$out/pmc/summary.json"))
for k,v in list(d.items())[:10]: print(k[:70].ljust(70), v["launches_sampled"], v["hbm_bytes_per_launch"], v["avg_launch_us_profiled"])
PY
python3 bench.py > $out/bench_default.json.log 2>&1; echo "bench rc=$?"; tail -1 $out/bench_default.json.log | cut -c1-300
# keep only the summaries (the raw traces are large)
rm -rf $out/stats_lanes1 $out/stats_lanes2 $out/pmc/FETCH_SIZE $out/pmc/WRITE_SIZE
