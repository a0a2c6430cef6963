#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats run: `python tools/trace_top.py DIR [N]` (reads *_kernel_stats.csv)."""
import csv, glob, os, sys
d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25
files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
if not files:
    sys.exit("no *_kernel_stats.csv under " + d)
rows = list(csv.DictReader(open(files[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(f"{len(rows)} kernels, {calls} dispatches, {tot / 1e6:.2f} ms of GPU time")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:n]:
    print(f"{float(r['TotalDurationNs']) / 1e6:9.3f} ms {100 * float(r['TotalDurationNs']) / tot:5.1f} %  {int(r['Calls']):6d} x {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:110]}")
