#!/usr/bin/env python3
"""Reads igemm8_ab.py's lines and prints, per shape, the heuristic's time beside the three fastest variants."""
import re
import sys

for line in open(sys.argv[1]):
    parts = line.split("|")
    res = []
    for p in parts[1:]:
        m = re.match(r"\s*(-?\d+):\s+([\d.]+) us", p)
        if m:
            res.append((float(m.group(2)), m.group(1)))
    auto = [r for r in res if r[1] == "-1"]
    if not auto:
        print(line.strip())
        continue
    best = min(res)
    print(f"{parts[0].strip()}  auto {auto[0][0]:.1f}  best v{best[1]} {best[0]:.1f}  ({100 * (auto[0][0] - best[0]) / auto[0][0]:.1f}% better)  top3 {sorted(res)[:3]}")
