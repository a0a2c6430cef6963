#!/usr/bin/env python3
"""Turn the outputs of tools/evidence.sh (gpurun_out/<tag>/) into the small files committed under profiles/ (RND = the round's
prefix, e.g. r05; the bench command's last line is the contract record, its detail -- `precision_legs` / `config_legs` -- sits in
the --detail-out file beside it):
    RND_final.md            suite / smoke, the driver line's key figures, rocprofv3 kernel traces (f32 and bf16x3, pose 0),
                            PMC passes (FETCH_SIZE, WRITE_SIZE, MfmaUtil, SQ_*) per kernel
    RND_workloads.md        C2 / C3 64+64 / C3 90+30 / C4 / C5: bench line + per-kernel ms
    RND_pmc_hbm_traffic.json   what bench.py reads for roofline.traffic / hbm_kernels.*.counter_bytes_per_step
    RND_bench*.json         the bench records themselves (final line + detail)
Runs ON THE GPU BOX at the end of the evidence pass (the .db files are too large to travel) and writes into
gpurun_out/<tag>/summary/.      python tools/summarise.py gpurun_out/<tag> [RND]"""
import glob
import json
import os
import sqlite3
import sys

src = sys.argv[1]
RND = next((a for a in sys.argv[2:] if not a.startswith("--")), os.path.basename(src.rstrip("/"))[:3])
dst = os.path.join(src, "summary")
os.makedirs(dst, exist_ok=True)


def power_report(src, dst):
    """socket power / clock over the whole 20-step bench run (rocm-smi samples, tools/power_trace.sh)"""
    pw = ["# " + RND + ": socket power and shader clock over `python bench.py --steps 20 --warmup 5` (rocm-smi, ~7 Hz; `tools/power_trace.sh`)\n",
          "The run renders 25 poses in bf16x3 (headline leg), then up to 5 in exact f32 (the short cross-check leg), then the short config legs, the PSNR check and the CPU baseline: "
          "the first plateau of the trace is the split-bf16 stage kernel, the second the exact-f32 one.\n"]
    pcsv = os.path.join(src, "power_bench.csv")
    if os.path.exists(pcsv):
        rows_p = []
        for line in open(pcsv):
            # card0,(fclk),level,(mclk),level,(sclk),level,(socclk),level,power  -- levels may be digits or "S"
            f = line.strip().split(",")
            try:
                rows_p.append((float(f[5].strip("()Mhz")), float(f[-1])))
            except (ValueError, IndexError):
                pass
        n = len(rows_p)
        pw.append(f"{n} samples.  Per sample: shader clock (sclk, MHz) and socket power (W).\n")
        if n:
            peak = max(p_ for _, p_ in rows_p)
            med = lambda c: sorted(c)[len(c) // 2]
            # the bf16x3 plateau draws > 93 % of the peak power, the f32 plateau 70 - 93 %
            hi = [r for r in rows_p if r[1] > 0.93 * peak]
            mid = [r for r in rows_p if 0.70 * peak < r[1] <= 0.93 * peak]
            for name, sel in (("above 93 % of the peak power (the split-bf16 legs)", hi), ("70 - 93 % of the peak power (the exact-f32 legs)", mid)):
                if sel:
                    pw.append(f"* samples {name}: {len(sel)} of {n}; sclk min / median / max {min(r[0] for r in sel):.0f} / {med([r[0] for r in sel]):.0f} / "
                              f"{max(r[0] for r in sel):.0f} MHz, power {min(r[1] for r in sel):.0f} / {med([r[1] for r in sel]):.0f} / {max(r[1] for r in sel):.0f} W")
            step = max(1, n // 60)
            pw.append(f"\n| sample | median sclk MHz | median power W |   (medians over {step} consecutive samples)\n|---|---|---|")
            for i in range(0, n, step):
                seg = rows_p[i:i + step]
                pw.append(f"| {i} | {med([r[0] for r in seg]):.0f} | {med([r[1] for r in seg]):.0f} |")
        open(os.path.join(dst, RND + "_power_bench.csv"), "w").write(open(pcsv).read())
    open(os.path.join(dst, RND + "_power_clock_trace.md"), "w").write("\n".join(pw) + "\n")


if "--power-only" in sys.argv:
    power_report(src, dst)
    print(open(os.path.join(dst, RND + "_power_clock_trace.md")).read())
    sys.exit(0)
KERN = {"mlp_stage (f32 wave)": "%mlp_wave_stage_kernel%", "mlp_stage (bf16x3)": "%mlp_bf16x3_stage_kernel%", "ray_bias": "%ray_bias_kernel%",
        "composite_single": "%composite_single_kernel%", "composite": "%composite_kernel%", "resample": "%resample_kernel%",
        "sample_coarse": "%sample_coarse_kernel%", "compact_rays": "%compact_rays_kernel%", "generate_rays": "%generate_rays_kernel%"}


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0] if "stnerf::" in name else (name[:60] + "...") if len(name) > 60 else name


def last_json(path):
    """The contract record (last stdout line) of a bench run, with the legs of its detail file (<path minus .json>_detail.json)."""
    try:
        lines = [l for l in open(path).read().strip().splitlines() if l.startswith("{")]
        rec = json.loads(lines[-1])
        dpath = path[:-5] + "_detail.json"
        if os.path.exists(dpath):
            det = json.load(open(dpath))["detail"]
            for k in ("precision_legs", "config_legs", "cpu_baseline", "eager_gpu_baseline", "psnr_vs_reference", "share_emulation", "device", "config"):
                if k in det and (k not in rec or k in ("cpu_baseline", "config", "share_emulation")):
                    rec[k if k != "config" else "config_detail"] = det[k]
            head = det["precision_legs"].get(rec["config"]["precision"], {})
            for k in ("kernels", "hbm_kernels", "mask_fraction", "per_rank_compute_s", "ray_samples_per_step_rank0"):
                rec.setdefault(k, head.get(k))
            rec["roofline"] = dict(head.get("roofline", {}), **rec["roofline"])
        return rec
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}


def trace_table(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), max(vgpr_count), max(accum_vgpr_count), max(lds_size), "
                       "max(grid_x), max(workgroup_x) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    out.append("| kernel | calls | total ms | avg ms | % | VGPR | AGPR | LDS B | grid x wg |")
    out.append("|---|---|---|---|---|---|---|---|---|")
    for nm, c, s, a, vg, ag, lds, gx, wx in rows[:12]:
        out.append(f"| `{short(nm)}` | {c} | {s / 1e6:.3f} | {a / 1e6:.4f} | {100 * s / tot:.2f} | {vg} | {ag} | {lds} | {gx} x {wx} |")
    out.append(f"\ntotal GPU kernel time {tot / 1e6:.1f} ms over {sum(r[1] for r in rows)} dispatches\n")


def pmc_rows(db, counters):
    cur = sqlite3.connect(db).cursor()
    res = {}
    for c in counters:
        for nm, n, s in cur.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name", (c,)):
            res.setdefault(short(nm), {})[c] = (n, s)
    return res


md = ["# " + RND + ": evidence for HEAD (one build, one GPU box)\n", "```", open(os.path.join(src, "env.txt")).read().strip(), "```\n"]
md.append("## parity suite, smoke\n```")
for f in ("pytest.log", "smoke.log"):
    md += open(os.path.join(src, f)).read().strip().splitlines()[-3:]
md.append("```\n")
b = last_json(os.path.join(src, "bench.json"))
json.dump(b, open(os.path.join(dst, RND + "_bench.json"), "w"), indent=1)


def leg_line(tag, e):
    r = e["roofline"]
    return (f"* {tag} **{e['precision']}**: **{e['value']:.4g} rays/s, {e['ray_samples_per_s']:.4g} ray-samples/s, {e['ms_per_step']:.1f} ms per frame** "
            f"({e['steps']} timed poses, {e['warmup']} warm-up); stage kernel {r['launches']} launches, avg {r['avg_launch_ms']:.2f} ms, "
            f"**{r['achieved']:.2f} algorithmic TF/s = {r['frac']:.4f} of {r['peak']:.1f} TF/s** ({r.get('peak_note', '')}); {r['executed_mfma_tflops']:.1f} executed MFMA TF/s = "
            f"{r['executed_frac_of_instruction_peak']:.4f} of the instruction's dense peak"
            + (f"; HBM traffic {r['traffic'] / 1e9:.2f} GB per launch (counters) vs {r['algorithmic_bytes_per_launch'] / 1e9:.2f} GB algorithmic" if r.get("traffic") else ""))


if "value" in b:
    md.append(f"## the driver's command: `python bench.py --steps {b['steps']} --warmup {b['warmup']}` (C3, {b['config']['workload']}), scaling label `{b['scaling']}`\n")
    legs = b.get("precision_legs", {})
    for k, e in legs.items():
        if isinstance(e, dict):
            md.append(leg_line("headline" if k == b["config"]["precision"] else "second leg", e))
            for kk, h in e.get("hbm_kernels", {}).items():
                md.append(f"    * {kk}: {h['ms_per_step']:.2f} ms per frame, {h['algorithmic_GBps']:.0f} GB/s of its algorithmic bytes = {h['frac']:.3f} of 8 TB/s"
                          + (f" ({h['frac_of_measured_peak']:.3f} of the measured {h['measured_peak_GBps']:.0f} GB/s)" if "frac_of_measured_peak" in h else "")
                          + (f"; counter / algorithmic bytes {h['counter_over_algorithmic']:.2f}" if "counter_over_algorithmic" in h else ""))
    for wl_, e in b.get("config_legs", {}).items():
        md.append(f"* config leg {wl_} ({e['precision']}, {e['steps']} poses): {e['value']:.4g} rays/s, {e['ray_samples_per_s']:.4g} ray-samples/s, "
                  f"{e['ms_per_step']:.1f} ms per frame, stage {e['roofline']['algorithmic_tflops']:.1f} algorithmic TF/s = {e['roofline']['frac']:.3f} of {e['roofline']['peak']:.1f} (SURVEY 8(d))")
    cb, eg = b.get("cpu_baseline"), b.get("eager_gpu_baseline")
    if cb:
        md.append(f"* cpu_baseline: {cb['value']:.1f} rays/s on {cb['cores']} threads ({cb['host']['cpu']}); frame extrapolates to {cb['extrapolated_frame_seconds']:.0f} s")
    if eg:
        md.append(f"* eager PyTorch-ROCm on the same GPU (oracle restatement, one reference chunk): {eg['value']:.0f} rays/s")
    if b.get("psnr_vs_reference"):
        md.append("* PSNR of the device-RNG render vs the reference: " + json.dumps(b["psnr_vs_reference"]))
    md.append("")
def _build_identity():
    """(rev, .hip_fatbin sha256) of the library the passes ran on: bench.py prints roofline.traffic only when the loaded library's
    kernels are these (tools/evidence.sh writes .git_rev into the snapshot; the digest is read from the .so itself)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from stnerf_amd import hip
    rev = None
    for cand in (os.path.join(src, "git_rev.txt"), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".git_rev")):
        if os.path.exists(cand):
            rev = open(cand).read().strip()[:12]
            break
    return rev, hip.fatbin_sha256()


_rev, _sha = _build_identity()
traffic = {"workload": "taekwondo-1080p-64+64", "pose": "pose 0 of the bench's sweep (orbit 10 deg), one step", "pose_short": "pose 0 (orbit 10 deg), 1 step",
           "rev": _rev, "fatbin_sha256": _sha, "gfx950_fetch_correction": 2.0,
           "note": "hbm bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE), MI355X_MICROARCH.md section HBM; separate rocprofv3 --pmc runs", "kernels": {}, "kernels_bf16x3": {}}
for prec in ("bf16x3", "fp32"):
    db = os.path.join(src, f"trace_{prec}", "p_results.db")
    md.append(f"## rocprofv3 --kernel-trace --stats, ONE step (pose 0), --precision {prec}\n")
    if os.path.exists(db):
        trace_table(db, md)
    else:
        md.append("(missing)\n")
    rows = {}
    for ctr, names in (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]), ("MfmaUtil", ["MfmaUtil"]),
                       ("SQ", ["SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS"])):
        p = os.path.join(src, f"pmc_{ctr}_{prec}", "p_results.db")
        if os.path.exists(p):
            for k, v in pmc_rows(p, names).items():
                rows.setdefault(k, {}).update(v)
    md.append(f"### PMC passes (own runs), --precision {prec}: per kernel, summed over its dispatches of the step\n")
    md.append("| kernel | dispatches | FETCH_SIZE KB | WRITE_SIZE KB | HBM GB (2 x fetch + write) | MfmaUtil avg % | SQ_INSTS_VALU | SQ_INSTS_MFMA | SQ_INSTS_LDS | active / wave cycles |")
    md.append("|---|---|---|---|---|---|---|---|---|---|")
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", (0, 0))[1]):
        if "stnerf::" not in k:
            continue
        f, w = v.get("FETCH_SIZE", (0, 0)), v.get("WRITE_SIZE", (0, 0))
        mu = v.get("MfmaUtil")
        act, wc = v.get("SQ_ACTIVE_INST_ANY", (0, 0))[1], v.get("SQ_WAVE_CYCLES", (0, 1))[1]
        md.append(f"| `{k}` | {f[0] or w[0]} | {f[1]:.0f} | {w[1]:.0f} | {1024 * (2 * f[1] + w[1]) / 1e9:.3f} | "
                  f"{(mu[1] / mu[0]) if mu else float('nan'):.1f} | {v.get('SQ_INSTS_VALU', (0, 0))[1]:.3g} | {v.get('SQ_INSTS_MFMA', (0, 0))[1]:.3g} | "
                  f"{v.get('SQ_INSTS_LDS', (0, 0))[1]:.3g} | {act / max(wc, 1):.3f} |")
    md.append("")
    # traffic json
    fdb, wdb = os.path.join(src, f"pmc_FETCH_SIZE_{prec}", "p_results.db"), os.path.join(src, f"pmc_WRITE_SIZE_{prec}", "p_results.db")
    if os.path.exists(fdb) and os.path.exists(wdb):
        cf, cw = sqlite3.connect(fdb).cursor(), sqlite3.connect(wdb).cursor()
        for name, like in (("spacenet", "%spacenet_kernel%"), ("motionnet", "%motionnet_kernel%"), ("mlp_stage", "%mlp%stage_kernel%"), ("composite", "%composite%kernel%"),
                           ("resample", "%resample_kernel%"), ("sample_coarse", "%sample_coarse_kernel%")):
            nf, fkb = cf.execute("select count(*), sum(value) from counters_collection where counter_name='FETCH_SIZE' and kernel_name like ?", (like,)).fetchone()
            nw, wkb = cw.execute("select count(*), sum(value) from counters_collection where counter_name='WRITE_SIZE' and kernel_name like ?", (like,)).fetchone()
            if not nf and not nw:
                continue
            hbm = 1024.0 * (2.0 * (fkb or 0) + (wkb or 0))
            traffic["kernels" if prec == "fp32" else "kernels_bf16x3"][name] = {
                "launches_per_step": nf, "fetch_size_kb_per_step": fkb or 0, "write_size_kb_per_step": wkb or 0, "hbm_bytes_per_step": hbm,
                "hbm_bytes_per_launch": hbm / max(nf, 1)}
# ---- compositor / resampler / sampler against the HBM rates measured on this box (tools/micro/hbm_copy): time from the kernel
# trace, bytes from the FETCH_SIZE / WRITE_SIZE passes, one step (pose 0) per configuration
try:
    hb = json.load(open(os.path.join(src, "hbm_copy.json")))
except Exception:  # noqa: BLE001
    hb = {}
md.append("## HBM-bound kernels on counter bytes (one step at pose 0; measured rates of this box: "
          + ", ".join(f"{k} {hb[k]:.0f} GB/s" for k in ("read_GBps", "write_GBps", "copy_GBps") if k in hb) + ")\n")
md.append("| config | kernel family | dispatches | ms per step | counter GB (2 x FETCH + WRITE) | TB/s | of 8 TB/s | of the measured rate |")
md.append("|---|---|---|---|---|---|---|---|")
hbm_table = {}
for cfg, tag in (("C3 taekwondo-1080p-64+64 (bf16x3)", "bf16x3"), ("C4 walking-1080p-L4-64+64 (bf16x3)", "c4"), ("C5 synthetic-4k-L8-128+64 (bf16x3)", "c5")):
    tdb, fdb, wdb = (os.path.join(src, f"{k}_{tag}", "p_results.db") for k in ("trace", "pmc_FETCH_SIZE", "pmc_WRITE_SIZE"))
    if not (os.path.exists(tdb) and os.path.exists(fdb) and os.path.exists(wdb)):
        md.append(f"| {cfg} | (missing) | | | | | | |")
        continue
    ct, cf, cw = (sqlite3.connect(x).cursor() for x in (tdb, fdb, wdb))
    for fam, like, ref in (("composite (single + merge)", "%composite%kernel%", "read_GBps"), ("resample", "%resample_kernel%", "copy_GBps"),
                           ("sample_coarse", "%sample_coarse_kernel%", "write_GBps")):
        nd, dur = ct.execute("select count(*), sum(duration) from kernels where name like ?", (like,)).fetchone()
        fkb = cf.execute("select sum(value) from counters_collection where counter_name='FETCH_SIZE' and kernel_name like ?", (like,)).fetchone()[0] or 0
        wkb = cw.execute("select sum(value) from counters_collection where counter_name='WRITE_SIZE' and kernel_name like ?", (like,)).fetchone()[0] or 0
        if not nd:
            continue
        gb, ms = 1024.0 * (2 * fkb + wkb) / 1e9, dur / 1e6
        rate = gb / ms   # TB/s
        frac_m = rate * 1e3 / hb[ref] if ref in hb else float("nan")
        md.append(f"| {cfg} | {fam} | {nd} | {ms:.2f} | {gb:.2f} | {rate:.2f} | {rate / 8:.3f} | {frac_m:.3f} of {ref.split('_')[0]} |")
        hbm_table.setdefault(cfg, {})[fam] = {"dispatches": nd, "ms_per_step": ms, "counter_GB": gb, "TBps": rate, "frac_of_measured": frac_m, "measured": ref}
md.append("")
traffic["hbm_kernels_on_counter_bytes"] = hbm_table
json.dump(traffic, open(os.path.join(dst, RND + "_pmc_hbm_traffic.json"), "w"), indent=1)
power_report(src, dst)
open(os.path.join(dst, RND + "_final.md"), "w").write("\n".join(md) + "\n")

wl = ["# " + RND + ": every BASELINE configuration on the round's build (1 x MI355X, `tools/evidence.sh`)\n",
      "| config | workload | arithmetic | rays/s | ray-samples/s | s per frame | stage kernel TF/s (algorithmic) | frac (SURVEY 8(d): of 416.7 bf16x3 / 157.3 f32) | composite ms | resample ms | sample_coarse ms |",
      "|---|---|---|---|---|---|---|---|---|---|---|"]
for cfg, fn in (("C2", "bench_c2.json"), ("C3", "bench.json"), ("C3 (yml 90+30)", "bench_c3_90_30.json"), ("C4", "bench_c4.json"), ("C5 (one GPU)", "bench_c5.json"),
                ("C3-small, 2 ranks on ONE GPU (gloo; code path only)", "bench_2ranks_one_device.json")):
    bb = last_json(os.path.join(src, fn))
    json.dump(bb, open(os.path.join(dst, RND + "_" + fn), "w"), indent=1)
    if "value" not in bb:
        wl.append(f"| {cfg} | {fn} | failed: {bb.get('error')} | | | | | | | | |")
        continue
    for k, e in bb.get("precision_legs", {}).items():
        if not isinstance(e, dict):
            continue
        hk = e.get("hbm_kernels", {})
        ms = lambda kk: f"{hk[kk]['ms_per_step']:.2f} ({hk[kk]['frac']:.2f} of 8 TB/s)" if kk in hk else ""
        wl.append(f"| {cfg} | {bb['config']['workload']} | {e['precision']} | {e['value']:.4g} | {e['ray_samples_per_s']:.4g} | {e['ms_per_step'] / 1e3:.3f} | "
                  f"{e['roofline']['algorithmic_tflops']:.1f} | {e['roofline']['frac']:.3f} | {ms('composite')} | {ms('resample')} | {ms('sample_coarse')} |")
wl.append("")
for f in ("bench_stage.txt", "bxab/time.log", "bx_prof.txt", "bench_composite.txt", "bench_resample.txt"):
    p = os.path.join(src, f)
    if os.path.exists(p):
        wl += [f"## {f}\n", "```", open(p).read().strip(), "```\n"]
open(os.path.join(dst, RND + "_workloads.md"), "w").write("\n".join(wl) + "\n")
# ---- a rank's share, emulated on the one GPU (bench.py --emulate-share)
se = ["# " + RND + ": what one rank of an N-GPU job computes, EMULATED on one GPU (`bench.py --emulate-share 2,4,8`)\n",
      "Not a scaling measurement: no second GPU, no process group, no collective.  Rank r's interleaved single-row stripes of the view are rendered "
      "alone through `stnerf_amd.parallel.render_view_share` -- the code `render_view` runs before its all-gather -- for the first, middle and last "
      "rank of each N; t(1) is the same function with (rank, N) = (0, 1) over the same poses.  `N t_share / t(1)` exposes what does not shrink with the "
      "share: the persistent stage kernel's tail at 1/N of the items, per-view host work, stripe imbalance.  Gather payload = bytes ONE rank "
      "contributes to the frame's single all-gather for `model.gather` = all / fine / final.\n",
      "| config | N | t_share ms (rank: ms) | worst | N t_share / t(1) | predicted compute efficiency | rays per rank | payload per rank MB (all / fine / final) |",
      "|---|---|---|---|---|---|---|---|"]
for cfg, fn in (("C3 taekwondo-1080p-64+64", "bench.json"), ("C4 walking-1080p-L4-64+64", "bench_c4.json"), ("C5 synthetic-4k-L8-128+64", "bench_c5.json")):
    bb = last_json(os.path.join(src, fn))
    sh = bb.get("share_emulation") if isinstance(bb.get("share_emulation"), dict) and "shares" in bb.get("share_emulation", {}) else None
    if not sh:
        se.append(f"| {cfg} | (no emulation in {fn}) | | | | | | |")
        continue
    se.append(f"| {cfg} | 1 | {sh['t1_ms']:.1f} | | | | | |")
    for e in sh["shares"]:
        pl = e["gather_payload_bytes_per_rank"]
        se.append(f"| {cfg} | {e['ranks']} | " + ", ".join(f"{r}: {ms:.1f}" for r, ms in e["t_share_ms"].items()) + f" | {e['t_share_max_ms']:.1f} | "
                  f"{e['n_times_t_share_over_t1']:.4f} | **{e['predicted_compute_efficiency']:.4f}** | {e['rays_per_rank']} | "
                  f"{pl['all'] / 1e6:.1f} / {pl['fine'] / 1e6:.1f} / {pl['final'] / 1e6:.1f} |")
open(os.path.join(dst, RND + "_share_emulation.md"), "w").write("\n".join(se) + "\n")
# ---- training kernels
tk = ["# " + RND + ": the training kernels (csrc/train.hip, SURVEY 8(f)4) on the round's build\n", "## tools/bench_backward.py\n", "```"]
p = os.path.join(src, "bench_backward.txt")
tk += [l for l in (open(p).read().strip().splitlines() if os.path.exists(p) else ["(missing)"]) if "amdgpu.ids" not in l] + ["```\n"]
tk.append("## rocprofv3 --kernel-trace --stats of `ONLY_NETS=1 ONLY_FUSED=1 python tools/bench_backward.py` (4 SpaceNet + 4 MotionNet iterations; `tools/trace_top.py`)\n")
tt = os.path.join(src, "trace_backward_top.txt")
tk += ["```", open(tt).read().strip() if os.path.exists(tt) else "(missing)", "```\n"]
for title, rel in (("HBM bytes / MfmaUtil per training kernel (separate `rocprofv3 --pmc` passes of the same command; `tools/pmc_training.py`)", "pmc_training.md"),
                   ("`stnerf_train_dw_batch` alone (`tools/bench_dw.py`: operands laid out as `modeling/autograd.py` hands them over)", os.path.join("dw", "bench_dw.txt")),
                   ("... its counters and kernel trace per network (`tools/gpu_dw_prof.sh`)", os.path.join("dw", "pmc.md")),
                   ("one iteration of the reference trainer's inner loop (`tools/bench_train_step.py`)", "train_step.txt")):
    q = os.path.join(src, rel)
    if os.path.exists(q):
        tk += ["## " + title + "\n", "```" if rel.endswith(".txt") else "", open(q).read().strip(), "```\n" if rel.endswith(".txt") else ""]
open(os.path.join(dst, RND + "_training_kernels.md"), "w").write("\n".join(tk) + "\n")
for f in ("hbm_copy.json",):
    p = os.path.join(src, f)
    if os.path.exists(p) and os.path.getsize(p):
        open(os.path.join(dst, RND + "_hbm_copy_microbench.json"), "w").write(open(p).read())
print(open(os.path.join(dst, RND + "_final.md")).read()[:6000])
print(open(os.path.join(dst, RND + "_workloads.md")).read()[:3000])
print(open(os.path.join(dst, RND + "_training_kernels.md")).read()[:2500])
print(open(os.path.join(dst, RND + "_power_clock_trace.md")).read()[:2000])
print(open(os.path.join(dst, RND + "_share_emulation.md")).read())
