#!/usr/bin/env python3
"""avg / calls of the kernels whose names contain one of the given substrings, from a rocprofv3 *kernel_stats.csv
   python tools/kstat.py FILE name [name ...]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
  if any(n in r["Name"] for n in sys.argv[2:]):
    print(f"  {r['Name'].split('(')[0][:60]:60s} calls={r['Calls']:>3s} avg_us={float(r['AverageNs'])/1e3:8.1f} min_us={float(r['MinNs'])/1e3:8.1f}")
