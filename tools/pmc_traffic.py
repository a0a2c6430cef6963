#!/usr/bin/env python3
"""HBM traffic per launch of the SpaceNet kernel from the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate
runs, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) -> profiles/r01_pmc_spacenet_traffic.json, which bench.py
reads for `roofline.traffic`.  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B,
hence the factor 2 (WRITE_SIZE was checked against the resampler's known byte count: exact).

    python tools/pmc_traffic.py --fetch gpurun_out/X/fetch/p_results.db --write gpurun_out/X/write/p_results.db \
        --workload taekwondo-1080p-64+64 --rays-per-launch 524288 --command "..." > profiles/r01_pmc_spacenet_traffic.json
"""
import argparse
import json
import sqlite3


def per_launch(path, counter, like):
    cur = sqlite3.connect(path).cursor()
    n, s = cur.execute("select count(*), sum(value) from counters_collection where counter_name=? and kernel_name like ?",
                       (counter, like)).fetchone()
    return n, (s or 0.0) / max(n, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--workload", default="taekwondo-1080p-64+64")
    ap.add_argument("--rays-per-launch", type=int, default=1 << 19)
    ap.add_argument("--command", default="")
    ap.add_argument("--kernel-like", default="%spacenet_kernel%")
    a = ap.parse_args()
    nf, fetch_kb = per_launch(a.fetch, "FETCH_SIZE", a.kernel_like)
    nw, write_kb = per_launch(a.write, "WRITE_SIZE", a.kernel_like)
    print(json.dumps({
        "workload": a.workload, "rays_per_launch": a.rays_per_launch, "command": a.command,
        "kernel": "stnerf::spacenet_kernel<128,8,*>", "launches": nf,
        "fetch_size_kb_per_launch": fetch_kb, "write_size_kb_per_launch": write_kb, "gfx950_fetch_correction": 2.0,
        "hbm_bytes_per_launch": 1024.0 * (2.0 * fetch_kb + write_kb),
        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM (gfx950 counts 128-B requests as 64 B for wide "
                "coalesced reads); WRITE_SIZE checked against known byte counts of the resampler (exact).",
    }, indent=1))


if __name__ == "__main__":
    main()
