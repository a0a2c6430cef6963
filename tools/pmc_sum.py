#!/usr/bin/env python3
"""Per-kernel sums of a rocprofv3 --pmc run written as csv (-f csv): python tools/pmc_sum.py <output dir>  (compositor / resampler kernels only)."""
import csv, glob, collections, sys
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:64]
        if "composite" not in k and "resample" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
    for k, d in acc.items():
        print(k, "dispatches", len(calls[k]), {c: f"{v / len(calls[k]):.4g}" for c, v in d.items()})
