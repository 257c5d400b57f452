#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 *kernel_stats.csv: name, calls, average ns, percentage."""
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for r in rows[1:1 + top]:
    print(r[0].replace("(anonymous namespace)::", "")[:72].ljust(72), r[1].rjust(6), f"{float(r[3]):12.0f}", r[4])
