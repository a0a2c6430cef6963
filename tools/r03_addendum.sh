#!/bin/bash
# Late addendum to tools/r03_evidence.sh (run through gpurun, ~7 GPU-minutes): the resampler changed after the evidence pass
# (every other kernel's ISA is identical, see profiles/r03_addendum_resampler.md), so: the whole parity suite + smoke on the final
# build, the driver's command, one traced + counted step of C3, the resampler alone.
out=gpurun_out/${1:-r03d}
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
{ echo "build: $(ls -la --time-style=full-iso st-nerf_amd/libstnerf_hip.so)"; echo "rev: $(cat .git_rev 2>/dev/null)"; } > $out/env.txt
timeout 900 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; echo rc=$? >> $out/pytest.log
timeout 200 python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo rc=$? >> $out/smoke.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo rc=$? >> $out/bench.err
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
CMD="python bench.py --steps 1 --warmup 0 --cpu-baseline-rays 0 --eager-gpu-baseline-rays 0 --no-psnr-check --no-second-precision"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o p -- $CMD > $out/trace.log 2>&1
for ctr in SQ_INSTS_VALU FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr -f csv -d $out/pmc_$ctr -o p -- $CMD > $out/pmc_$ctr.log 2>&1
done
timeout 200 python tools/resample_phase_prof.py > $out/bench_resample.txt 2>&1
python - "$out" <<'PY' > $out/summary.md 2>&1
import csv, glob, json, sys, collections
out = sys.argv[1]
print(open(out + "/env.txt").read().strip()); print()
print("parity suite / smoke:"); print("```")
for f in ("pytest.log", "smoke.log"):
    print("\n".join(l for l in open(f"{out}/{f}").read().strip().splitlines()[-3:] if "amdgpu.ids" not in l))
print("```\n")
b = json.loads([l for l in open(out + "/bench.json").read().splitlines() if l.startswith("{")][-1])
r = b["roofline"]; hk = b["hbm_kernels"]
print(f"driver's command: {b['value']:.4g} rays/s, {b['ms_per_step']:.1f} ms per frame, stage kernel {r['achieved']:.2f} TF/s = {r['frac']:.4f} of {r['peak']}; "
      f"other_precision {b['other_precision']['precision']}: {b['other_precision']['value']:.4g} rays/s")
for k, e in hk.items():
    print(f"* {k}: {e['ms_per_step']:.2f} ms per frame, {e['algorithmic_GBps']:.0f} GB/s of its algorithmic bytes = {e['frac']:.3f} of 8 TB/s, "
          f"{e.get('frac_of_measured_peak', float('nan')):.3f} of the measured {e.get('measured_peak_GBps', 0):.0f} GB/s")
print()
stats = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
if stats:
    print("kernel trace, one step (pose 0), stnerf kernels:"); print("```")
    for row in csv.DictReader(open(stats[0])):
        if "stnerf::" in row["Name"]:
            print(f"{row['Name'][:70]:70s} calls {row['Calls']:>4s}  total {float(row['TotalDurationNs']) / 1e6:9.3f} ms  avg {float(row['AverageNs']) / 1e6:8.4f} ms")
    print("```\n")
acc = collections.defaultdict(dict)
for ctr in ("SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/pmc_{ctr}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:60]
            if "stnerf::" in k and row["Counter_Name"] == ctr:
                acc[k][ctr] = acc[k].get(ctr, 0.0) + float(row["Counter_Value"])
print("PMC passes (own runs), per kernel over the step:"); print("```")
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    gb = 1024 * (2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)) / 1e9
    print(f"{k:60s} SQ_INSTS_VALU {d.get('SQ_INSTS_VALU', 0):.4g}   HBM (2 x FETCH + WRITE) {gb:.3f} GB")
print("```\n")
print("resampler alone (tools/resample_phase_prof.py):"); print("```")
print("\n".join(l for l in open(out + "/bench_resample.txt").read().splitlines() if "amdgpu.ids" not in l)); print("```")
PY
find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*.db" -delete
cat $out/summary.md
