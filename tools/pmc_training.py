#!/usr/bin/env python3
"""HBM bytes and MFMA utilisation of the training kernels from separate rocprofv3 --pmc passes (rocpd databases):
`python tools/pmc_training.py DIR` with DIR/pmc_{FETCH_SIZE,WRITE_SIZE,MfmaUtil}/p_results.db of
`ONLY_NETS=1 ONLY_FUSED=1 python tools/bench_backward.py`.  Bytes as profiles/rNN_pmc_hbm_traffic.json counts them:
1024 * (2 * FETCH_SIZE + WRITE_SIZE) (MI355X_MICROARCH.md, HBM section)."""
import os, sqlite3, sys
d = sys.argv[1]


def rows(counter):
    p = os.path.join(d, "pmc_" + counter, "p_results.db")
    if not os.path.exists(p):
        return {}
    cur = sqlite3.connect(p).cursor()
    return {nm: (n, s) for nm, n, s in cur.execute(
        "select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name", (counter,))}


f, w, u = rows("FETCH_SIZE"), rows("WRITE_SIZE"), rows("MfmaUtil")
print("| kernel | dispatches | HBM GB per dispatch (2 x fetch + write) | fetch KB | write KB | MfmaUtil avg % |")
print("|---|---|---|---|---|---|")
for nm in sorted(f, key=lambda k: -(2 * f[k][1] + w.get(k, (0, 0))[1])):
    if "stnerf" not in nm:
        continue
    n = f[nm][0]
    wr = w.get(nm, (n, 0.0))[1]
    gb = 1024 * (2 * f[nm][1] + wr) / n / 1e9
    mu = u.get(nm)
    short = nm.replace("stnerf::", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
    print(f"| `{short}` | {n} | {gb:.3f} | {f[nm][1] / n:.0f} | {wr / n:.0f} | {mu[1] / mu[0]:.1f} |" if mu else f"| `{short}` | {n} | {gb:.3f} | {f[nm][1] / n:.0f} | {wr / n:.0f} | |")
