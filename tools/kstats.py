#!/usr/bin/env python3
"""Prints a rocprofv3 kernel_stats.csv with short kernel names (development aid)."""
import csv, re, sys
for path in sys.argv[1:]:
    print("==", path)
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:28]:
        n = r["Name"]
        m = re.search(r"(\w+_kernel)", n)
        short = (m.group(1) if m else n[:40]) + " " + re.sub(r".*_kernel", "", n)[:70]
        print(f"{short:110s} n={int(r['Calls']):5d} avg={float(r['AverageNs'])/1e3:8.1f}us tot={float(r['TotalDurationNs'])/1e6:8.2f}ms {100*float(r['TotalDurationNs'])/tot:5.1f}%")
