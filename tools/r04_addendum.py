#!/usr/bin/env python3
"""Summary of tools/r04_addendum.sh's outputs (runs on the GPU box: the rocprofv3 csv files are too large to travel)."""
import collections
import csv
import glob
import json
import sys

out = sys.argv[1]
print(open(out + "/env.txt").read().strip())
print()
print("parity suite / smoke:")
print("```")
for f in ("pytest.log", "smoke.log"):
    print("\n".join(l for l in open(f"{out}/{f}").read().strip().splitlines()[-3:] if "amdgpu.ids" not in l))
print("```\n")
b = json.loads([l for l in open(out + "/bench.json").read().splitlines() if l.startswith("{")][-1])
print(f"`python bench.py --steps {b['steps']} --warmup {b['warmup']} --no-config-legs` ({b['config']['workload']}):\n")
for k, e in b.get("precision_legs", {}).items():
    if not isinstance(e, dict):
        continue
    r = e["roofline"]
    print(f"* **{e['precision']}**: {e['value']:.4g} rays/s, {e['ray_samples_per_s']:.4g} ray-samples/s, {e['ms_per_step']:.1f} ms per frame; stage kernel "
          f"{r['algorithmic_tflops']:.1f} algorithmic TF/s, frac {r['frac']:.4f} of {r['peak']} TF/s")
    for kk, h in e.get("hbm_kernels", {}).items():
        print(f"    * {kk}: {h['ms_per_step']:.2f} ms per frame, {h['algorithmic_GBps']:.0f} GB/s of its algorithmic bytes = {h['frac']:.3f} of 8 TB/s"
              + (f" ({h['frac_of_measured_peak']:.3f} of the measured {h['measured_peak_GBps']:.0f} GB/s)" if "frac_of_measured_peak" in h else ""))
print()


def stnerf_rows(pattern):
    stats = glob.glob(f"{out}/{pattern}/**/*kernel_stats.csv", recursive=True)
    rows = []
    if stats:
        for row in csv.DictReader(open(stats[0])):
            if "stnerf::" in row["Name"]:
                rows.append((row["Name"].replace("void ", "").split("(")[0][:72], int(row["Calls"]), float(row["TotalDurationNs"]) / 1e6, float(row["AverageNs"]) / 1e6))
    return rows


fam_ms = {}
for tag, title in (("trace", "C3 taekwondo-1080p-64+64"), ("trace_c5", "C5 synthetic-4k-L8-128+64")):
    rows = stnerf_rows(tag)
    print(f"kernel trace, one step (pose 0, bf16x3), {title}:")
    print("```")
    for nm, c, tot, avg in rows:
        print(f"{nm:72s} calls {c:4d}  total {tot:10.3f} ms  avg {avg:9.4f} ms")
    for fam in ("composite", "resample_kernel", "sample_coarse"):
        ms = sum(r[2] for r in rows if fam in r[0])
        fam_ms[(tag, fam)] = ms
        print(f"  -> {fam}: {ms:.2f} ms per step")
    print("```\n")
acc = collections.defaultdict(dict)
for ctr in ("SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/pmc_{ctr}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].replace("void ", "").split("(")[0][:72]
            if "stnerf::" in k and row["Counter_Name"] == ctr:
                acc[k][ctr] = acc[k].get(ctr, 0.0) + float(row["Counter_Value"])
print("PMC passes (own runs) of the C3 step, per kernel over the step:")
print("```")
comp = {"SQ_INSTS_VALU": 0.0, "gb": 0.0}
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    gb = 1024 * (2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)) / 1e9
    print(f"{k:72s} SQ_INSTS_VALU {d.get('SQ_INSTS_VALU', 0):.4g}   HBM (2 x FETCH + WRITE) {gb:.3f} GB")
    if "composite" in k:
        comp["SQ_INSTS_VALU"] += d.get("SQ_INSTS_VALU", 0)
        comp["gb"] += gb
print("```\n")
try:
    alg = b["precision_legs"][b["config"]["precision"]]["hbm_kernels"]["composite"]["algorithmic_bytes_per_step"] / 1e9
    ms = fam_ms.get(("trace", "composite"), 0.0)
    print(f"compositor at C3 (pose 0): {ms:.2f} ms, {comp['gb']:.2f} GB on the counters = {comp['gb'] / ms:.2f} TB/s; algorithmic bytes of the bench's sweep "
          f"{alg:.2f} GB per step -> counter / algorithmic = {comp['gb'] / alg:.2f}; SQ_INSTS_VALU {comp['SQ_INSTS_VALU']:.4g} "
          f"(x 4 cycles / 1024 SIMDs / 2.4 GHz = {comp['SQ_INSTS_VALU'] * 4 / 1024 / 2.4e9 * 1e3:.2f} ms of vector issue)\n")
except Exception as e:  # noqa: BLE001
    print(f"(no compositor summary: {type(e).__name__}: {e})\n")
print("compositor alone (tools/bench_composite.py, hint bits set as the sampler sets them):")
print("```")
print("\n".join(l for l in open(out + "/bench_composite.txt").read().splitlines() if "amdgpu.ids" not in l))
print("```")
