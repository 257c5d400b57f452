#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench (run on the GPU box); prints the top kernels and leaves the CSV in
# gpurun_out/prof_<tag>/ for copying into profiles/.
tag=${1:-stats}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 bench.py --steps 8 --no-cpu-baseline --no-roofline --no-f32-line --no-extra-configs > gpurun_out/prof_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/prof_{tag}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.reader(open(f)))
print(f)
for r in rows[1:28]:
    print(r[0].replace("(anonymous namespace)::", "")[:64].ljust(64), r[1].rjust(6), r[3][:10].rjust(11), r[4])
PY
