#!/usr/bin/env python3
"""Summarise the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh per kernel.
gfx950 corrections (MI355X_MICROARCH.md, HBM): both counters are in KiB; FETCH_SIZE reports half the bytes
of a wide coalesced streaming read -> doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n_f": 0, "n_w": 0, "dur_ns": 0.0})
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{root}/{ctr}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr:
                continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            name = name.split("(")[0].strip()
            a = acc[name]
            a[ctr] += float(r["Counter_Value"])
            a["n_f" if ctr == "FETCH_SIZE" else "n_w"] += 1
            if ctr == "FETCH_SIZE":
                a["dur_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
out = {}
for name, a in acc.items():
    if not a["n_f"] or not a["n_w"]:
        continue
    fetch = 2.0 * 1024.0 * a["FETCH_SIZE"] / a["n_f"]          # bytes per launch, gfx950 x2 correction
    write = 1024.0 * a["WRITE_SIZE"] / a["n_w"]
    out[name] = {"launches_sampled": a["n_f"], "hbm_read_bytes_per_launch": round(fetch), "hbm_write_bytes_per_launch": round(write),
                 "hbm_bytes_per_launch": round(fetch + write), "avg_launch_us_profiled": round(a["dur_ns"] / a["n_f"] / 1e3, 2)}
print(json.dumps(dict(sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_sampled"])), indent=1))
