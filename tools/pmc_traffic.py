#!/usr/bin/env python3
"""HBM traffic of every kernel of the path from the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs,
as /opt/skills/guides/MI355X_MICROARCH.md prescribes) of ONE bench step -> profiles/r02_pmc_hbm_traffic.json, which
bench.py reads for `roofline.traffic` and `hbm_kernels.*.counter_bytes_per_step`.  FETCH_SIZE / WRITE_SIZE are in KB; on
gfx950 FETCH_SIZE counts 128-B requests as 64 B, hence the factor 2 (WRITE_SIZE was checked against the resampler's
known byte count in round 1: exact).

    python tools/pmc_traffic.py --fetch gpurun_out/X/fetch/p_results.db --write gpurun_out/X/write/p_results.db \
        --workload taekwondo-1080p-64+64 --command "..." > profiles/r02_pmc_hbm_traffic.json
"""
import argparse
import json
import sqlite3

KERNELS = {"spacenet": "%spacenet_kernel%", "motionnet": "%motionnet_kernel%", "mlp_stage": "%mlp%stage_kernel%",
           "composite": "%composite%kernel%", "resample": "%resample_kernel%", "sample_coarse": "%sample_coarse_kernel%"}


def total(path, counter, like):
    cur = sqlite3.connect(path).cursor()
    n, s = cur.execute("select count(*), sum(value) from counters_collection where counter_name=? and kernel_name like ?",
                       (counter, like)).fetchone()
    return n, (s or 0.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--workload", default="taekwondo-1080p-64+64")
    ap.add_argument("--steps", type=int, default=1, help="bench steps the profiled command ran (warm-up included)")
    ap.add_argument("--command", default="")
    a = ap.parse_args()
    out = {"workload": a.workload, "command": a.command, "steps": a.steps, "gfx950_fetch_correction": 2.0,
           "note": "hbm bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM "
                   "(gfx950 counts 128-B requests as 64 B for wide coalesced reads); WRITE_SIZE exact.", "kernels": {}}
    for name, like in KERNELS.items():
        nf, fkb = total(a.fetch, "FETCH_SIZE", like)
        nw, wkb = total(a.write, "WRITE_SIZE", like)
        if nf == 0 and nw == 0:
            continue
        hbm = 1024.0 * (2.0 * fkb + wkb)
        out["kernels"][name] = {"launches_per_step": nf / a.steps, "fetch_size_kb_per_step": fkb / a.steps,
                                "write_size_kb_per_step": wkb / a.steps, "hbm_bytes_per_step": hbm / a.steps,
                                "hbm_bytes_per_launch": hbm / max(nf, 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
