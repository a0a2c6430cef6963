#!/usr/bin/env python3
"""GPU busy time against span per training iteration from a rocprofv3 --kernel-trace csv: windows start at every dispatch of the
kernel whose name contains KEY.   python tools/trace_gaps.py DIR KEY"""
import csv, glob, os, sys
d, key = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
starts = [i for i, r in enumerate(rows) if key in r[2]]
for a, b in zip(starts, starts[1:] + [None]):
    win = rows[a:b] if b is not None else rows[a:a + (starts[1] - starts[0] if len(starts) > 1 else len(rows))]
    span = (max(r[1] for r in win) - win[0][0]) / 1e6
    busy = sum(r[1] - r[0] for r in win) / 1e6
    big = sum(r[1] - r[0] for r in win if r[1] - r[0] > 200_000) / 1e6
    gaps = sorted(((win[i + 1][0] - win[i][1]) / 1e3, win[i][2][:50], win[i + 1][2][:50]) for i in range(len(win) - 1))[-4:]
    print(f"{len(win)} dispatches, span {span:.3f} ms, busy {busy:.3f} ms ({busy / span:.2f}), kernels over 0.2 ms {big:.3f} ms; largest gaps us: " +
          "; ".join(f"{g:.0f} ({x} -> {y})" for g, x, y in gaps))
