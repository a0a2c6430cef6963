#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs into the text tables committed under profiles/.

    python tools/rocpd_summary.py --trace gpurun_out/prof/trace/r01_results.db \
        [--pmc FETCH_SIZE=...db --pmc WRITE_SIZE=...db] > profiles/r01_....md
"""
import argparse
import sqlite3


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0] if "stnerf::" in name else (name[:70] + "...") if len(name) > 70 else name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace")
    ap.add_argument("--pmc", action="append", default=[])
    ap.add_argument("--title", default="rocprofv3 summary")
    args = ap.parse_args()
    print(f"# {args.title}\n")
    if args.trace:
        cur = sqlite3.connect(args.trace).cursor()
        print("## kernel trace (`rocprofv3 --kernel-trace --stats`), durations in ms\n")
        print("| kernel | calls | total ms | avg ms | % | VGPR | AGPR | LDS B | grid x wg |")
        print("|---|---|---|---|---|---|---|---|---|")
        rows = cur.execute("select name, count(*), sum(duration), avg(duration), max(vgpr_count), max(accum_vgpr_count), "
                           "max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows)
        for nm, c, s, a, vg, ag, lds, gx, wx in rows[:14]:
            print(f"| `{short(nm)}` | {c} | {s / 1e6:.3f} | {a / 1e6:.4f} | {100 * s / tot:.2f} | {vg} | {ag} | {lds} | {gx} x {wx} |")
        print(f"\ntotal GPU kernel time {tot / 1e6:.1f} ms over {sum(r[1] for r in rows)} dispatches\n")
    for spec in args.pmc:
        cname, path = spec.split("=", 1)
        cur = sqlite3.connect(path).cursor()
        print(f"## PMC pass `rocprofv3 --pmc {cname}` (own run)\n")
        print(f"| kernel | dispatches | sum {cname} | avg per dispatch | max per dispatch |")
        print("|---|---|---|---|---|")
        for nm, c, s, a, mx in cur.execute(
                "select kernel_name, count(*), sum(value), avg(value), max(value) from counters_collection "
                "where counter_name=? group by kernel_name order by 3 desc limit 10", (cname,)):
            print(f"| `{short(nm)}` | {c} | {s:.1f} | {a:.1f} | {mx:.1f} |")
        print()


if __name__ == "__main__":
    main()
