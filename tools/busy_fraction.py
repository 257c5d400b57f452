#!/usr/bin/env python3
"""GPU busy fraction of a rocprofv3 --kernel-trace CSV: union of the kernel intervals over the span between the first
and the last kernel of the window (the last `--tail` fraction of the trace, i.e. the timed steps, not the warm-up), and
the mean number of kernels in flight.  Usage: busy_fraction.py <kernel_trace.csv> [--tail 0.5]"""
import csv, sys
path = sys.argv[1]
tail = float(sys.argv[sys.argv.index("--tail") + 1]) if "--tail" in sys.argv else 0.5
rows = list(csv.DictReader(open(path)))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t_lo = iv[0][0] + (iv[-1][1] - iv[0][0]) * (1.0 - tail)
iv = [x for x in iv if x[0] >= t_lo]
span = max(e for _, e, _ in iv) - iv[0][0]
busy, cur_s, cur_e, tot = 0, iv[0][0], iv[0][1], 0
gaps = []
for s, e, _ in iv:
    tot += e - s
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"kernels {len(iv)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms ({100.0 * busy / span:.1f} %)  "
      f"sum of kernel durations {tot / 1e6:.2f} ms (mean {tot / busy:.2f} in flight while busy)")
gaps.sort(reverse=True)
print("idle gaps: n=%d total %.2f ms; largest (us): %s" % (len(gaps), sum(g for g, _ in gaps) / 1e6,
      " ".join(f"{g / 1e3:.0f}" for g, _ in gaps[:12])))
hist = {}
for g, _ in gaps:
    b = 1
    while b * 1000 < g:
        b *= 2
    hist[b] = hist.get(b, 0) + g
print("idle by gap size (<= us: ms):", " ".join(f"{b}:{v / 1e6:.2f}" for b, v in sorted(hist.items())))
