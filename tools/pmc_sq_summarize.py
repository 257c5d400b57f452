#!/usr/bin/env python3
"""Summarise the SQ / GRBM pass of tools/pmc_sq.sh per kernel (means over the sampled launches).
What the counters mean on gfx950 (checked against launches of known size):
  * SQ_VALU_MFMA_BUSY_CYCLES = the cycles an MFMA occupies its pipe, summed over the chip: 16 per v_mfma_f32_16x16x32_f16
    (the (102400, 728, 728) launches of the pre-split GEMM read 325 017 600 = 16 x 20 313 600 MFMAs: 1200 tiles x 23 K-steps
    x 768, less the skipped padding column tiles; the (1638400, 128, 256) launch 314 572 800 = 16 x 6400 x 4 x 768), 32 per
    32x32x16 MFMA (MI355X_MICROARCH.md).  Busy fraction of the 1024 matrix pipes = counter / (1024 x kernel cycles):
    "mfma_pipe_busy_at_2p4ghz" prices the kernel's wall time at the 2.4 GHz the roofline uses (comparable with
    roofline.frac, which counts algorithmic flops: this one includes K / N padding); "mfma_util_rocprof_formula" is
    rocprofv3's MfmaUtil (counter / (GRBM_GUI_ACTIVE x SIMDs), gfx94x section -- ROCm 7.2 ships none for gfx950).
  * GRBM_GUI_ACTIVE arrives summed over the 8 XCDs.  counter / 8 / duration reads 2.2-2.5 GHz on every kernel, while the
    shader clock measured inside the same GEMM (s_memtime against s_memrealtime, profiles/r02_gemm_phase_profile.txt) is
    1.5-1.9 GHz and 192 MFMAs take 3100 s_memtime ticks: it is not the clock the CUs hold under DVFS -- reported, not used.
  * SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY count quad-cycles summed over waves: only their ratios are used."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
SIMDS = 256 * 4
per_dispatch = collections.defaultdict(lambda: collections.defaultdict(float))
meta = {}
for f in glob.glob(f"{root}/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r["Dispatch_Id"], r["Kernel_Name"])
        c = r["Counter_Name"]
        v = float(r["Counter_Value"])
        if c == "GRBM_GUI_ACTIVE":
            per_dispatch[key][c] = max(per_dispatch[key][c], v)
        else:
            per_dispatch[key][c] += v
        meta[key] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for (did, kname), c in per_dispatch.items():
    name = kname.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
    a = acc[name]
    a["n"] += 1
    a["dur_ns"] += meta[(did, kname)]
    for k, v in c.items():
        a[k] += v
out = {}
for name, a in acc.items():
    n = a["n"]
    gui = a["GRBM_GUI_ACTIVE"] / n / 8.0                        # per XCD
    wave = a["SQ_WAVE_CYCLES"] or 1.0
    out[name] = {
        "launches_sampled": int(n),
        "avg_launch_us_profiled": round(a["dur_ns"] / n / 1e3, 2),
        "gpu_cycles_per_launch": round(gui),
        "grbm_clock_ghz": round(gui / (a["dur_ns"] / n), 3) if a["dur_ns"] else None,
        "mfma_busy_counter_per_launch": round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / n),
        "mfma_util_rocprof_formula": round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (gui * SIMDS), 4) if gui else None,
        "mfma_pipe_busy_at_2p4ghz": round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (SIMDS * (a["dur_ns"] / n) * 2.4), 4) if a["dur_ns"] else None,
        "wave_cycles_parked": round(a["SQ_WAIT_ANY"] / wave, 3),
        "wave_cycles_issue_stalled": round(a["SQ_WAIT_INST_ANY"] / wave, 3),
        "wave_cycles_issuing": round(a["SQ_ACTIVE_INST_ANY"] / wave, 3),
        "lds_bank_conflict_per_active": round(a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"], 3) if a["SQ_LDS_IDX_ACTIVE"] else None,
    }
print(json.dumps(dict(sorted(out.items(), key=lambda kv: -kv[1]["avg_launch_us_profiled"] * kv[1]["launches_sampled"])), indent=1))
