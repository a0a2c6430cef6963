#!/bin/bash
# Where do the wave kernel's cycles go?  Variant builds (wrong results, timing only) + SQ / cache counters.
out=gpurun_out/${1:-wave2}
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1 KERNELS=wave CASES="bkgd,performer fused" ITERS=3
for tag in "" _noloadw _noepi _noheads _nope _lbar _none; do
  echo "== variant '${tag}'" >> $out/variants.log
  STNERF_LIB=$PWD/st-nerf_amd/libstnerf_hip${tag}.so timeout 120 python tools/ab_wave.py time >> $out/variants.log 2>&1
done
KERNELS=lds timeout 120 python tools/ab_wave.py time >> $out/variants.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 -L > $out/counters_list.txt 2>&1
export CASES="bkgd" ITERS=1
for k in wave lds; do
  KERNELS=$k rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/pmc_sq_$k -o p --output-format csv -- python tools/ab_wave.py time > $out/pmc_sq_$k.log 2>&1
  KERNELS=$k rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM -d $out/pmc_in_$k -o p --output-format csv -- python tools/ab_wave.py time > $out/pmc_in_$k.log 2>&1
  KERNELS=$k rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $out/pmc_tc_$k -o p --output-format csv -- python tools/ab_wave.py time > $out/pmc_tc_$k.log 2>&1
done
cat $out/variants.log
python - <<'PY'
import csv, glob, os, sys
out = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("OUTDIR", "")
PY
for f in $(find $out -name "*counter_collection.csv"); do
  echo "== $f"
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "")
    if "mlp" not in k:
        continue
    acc[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:36s} {v:.6g}")
PY
done
