#!/bin/bash
# Matrix-pipe and LDS counters of every kernel of the hot path (one rocprofv3 pass, counters with --kernel-trace only):
# SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs) = the fraction of the matrix pipes' cycles that held an MFMA
# (rocprofv3's MfmaUtil formula), the wave-cycle split (parked / issue-stalled / issuing) and LDS bank conflicts.
# Usage (on the GPU box): tools/pmc_sq.sh <outdir> [bench config: 1 (default) | 2 | 4]
set -e
out=${1:-$GRAFT_REPO_ROOT/gpurun_out/pmc_sq}
repo=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cfg=${2:-1}
mkdir -p "$out"
out=$(cd "$out" && pwd)                     # absolute: the pass runs from /tmp
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --output-format csv -d "$out/sq" -- python3 "$repo/bench.py" --config $cfg --lanes 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-f32-line --no-extra-configs > "$out/sq.log" 2>&1
python3 "$repo/tools/pmc_sq_summarize.py" "$out" > "$out/summary.json"
