#!/usr/bin/env python3
"""Prints the figures of bench.py JSON lines that matter when comparing runs (development aid)."""
import json
import sys

for p in sys.argv[1:]:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    r, dec = d.get("roofline") or {}, d.get("roofline_decode") or {}
    print(f"{p}: {d['value']:.0f} {d['unit']}  {d['ms_per_step']} ms/step  igemm {r.get('achieved')} TF frac {r.get('frac')} "
          f"({r.get('launches_per_step')} launches x {r.get('avg_launch_us')} us)  decode {dec.get('ms')} ms  "
          f"lanes3 {(d.get('concurrent_lanes') or {}).get('value')}")
