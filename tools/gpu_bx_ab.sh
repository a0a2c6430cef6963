#!/bin/bash
# A/B of two BUILDS of the bf16x3 stage kernel at the socket's power limit: the product library against a variant
# (st-nerf_amd/libstnerf_hip_<tag>.so, built with STNERF_LIB_TAG=<tag> ...), alternating on ONE box, several seconds per case,
# with the socket's power / clock sampled beside (tools/ab_bx.py prints TF/s and a checksum: outputs must be bit-identical).
#   gpurun --timeout 600 -- 'tools/gpu_bx_ab.sh noslp'
tag=${1:-noslp}; rounds=${2:-3}
out=gpurun_out/bxab_$tag; mkdir -p $out
export SECONDS_PER_CASE=${SECONDS_PER_CASE:-4}
( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n'; echo; sleep 1; done ) > $out/smi.jsonl &
smi=$!
for r in $(seq $rounds); do
  python tools/ab_bx.py 2>/dev/null | sed "s/^/main   /" | tee -a $out/time.log
  STNERF_LIB=$PWD/st-nerf_amd/libstnerf_hip_$tag.so python tools/ab_bx.py 2>/dev/null | sed "s/^/$tag /" | tee -a $out/time.log
done
kill $smi
python - <<PY
import json, re
rows = []
for ln in open("$out/smi.jsonl"):
    try:
        d = json.loads(ln)["card0"]
    except Exception:
        continue
    w = next((float(v) for k, v in d.items() if "Power" in k and "W" in k), None)
    c = next((float(re.sub(r"[^0-9.]", "", v.split("(")[1])) for k, v in d.items() if k.startswith("sclk") and "(" in v), None)
    if w and w > 600:
        rows.append((w, c))
if rows:
    print(f"socket under load: {len(rows)} samples, {sum(w for w, _ in rows) / len(rows):.0f} W, {sum(c for _, c in rows if c) / max(1, sum(1 for _, c in rows if c)):.0f} MHz")
PY
