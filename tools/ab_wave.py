#!/usr/bin/env python3
"""A/B of two BUILDS of the exact-f32 stage kernel (csrc/mlp_wave.hip) behind stnerf_mlp_stage (GPU box).  Round 2 used
this against the LDS-organised kernel (csrc/mlp_stage.hip, removed in round 3: tests/test_gpu_stage.py checks the stage
against the oracle directly); what is left is the comparison of the tree's build with a variant build of the same source
(STNERF_LIB=.../libstnerf_hip_<tag>.so, built with STNERF_LIB_TAG=<tag>), one library per process:

    STNERF_LIB=.../libstnerf_hip_base.so python tools/ab_wave.py check --save /tmp/base.pt
    python tools/ab_wave.py check --against /tmp/base.pt     bitwise comparison on a set of scenarios (exit code 1 on
                                                             any difference); without --against: determinism only
    python tools/ab_wave.py layers     development build (STNERF_LIB=.../libstnerf_hip_dbg.so): the wave kernel's
                                       activations after every stage against an fp64 evaluation -> which layer is wrong
    python tools/ab_wave.py time       TF/s on 131072 x 64 rows per scenario
"""
import os
import sys
import ctypes as C

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from stnerf_amd import hip, ops, synthetic as syn

FLOP_SPACE, FLOP_SPACE_TIME, FLOP_MOTION = 924_672, 930_048, 153_344


def run(layers, dirs, ns, **kw):
    ops.mlp_stage(layers, dirs, ns, **kw)
    torch.cuda.synchronize()


def scenario(name, n, ns, deep, bkgd_deform, with_perf, with_motion, seed=0):
    torch.manual_seed(41 + ns + seed)
    rs = np.random.RandomState(9 + seed)
    l = 3
    sd_b = syn.spacenet_state("net", rs, bkgd_deform, deep_rgb=deep)
    sd_p = [syn.spacenet_state("net", rs, True, deep_rgb=deep) for _ in range(l - 1)]
    sd_m = [syn.motionnet_state("net", rs) for _ in range(l)]
    xyz = ((torch.rand(n, l, ns, 3) - 0.5) * 5.0).cuda()
    dirs = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1)
    times = torch.where(torch.rand(n, l) < 0.5, torch.floor(torch.rand(n, l) * 30), torch.rand(n, l) * 30) + 1
    rays = torch.cat([torch.zeros(n, 3), dirs, times], -1).cuda()
    mask = (torch.rand(n, l) < 0.45).to(torch.uint8)
    mask[:, 0] = 1
    dm = mask.cuda()
    lst, cnt = ops.compact_rays(dm)
    bk = ops.pack_spacenet(sd_b, "net")
    sp = [ops.pack_spacenet(s_, "net") for s_ in sd_p]
    mo = [ops.pack_motionnet(s_, "net") for s_ in sd_m]

    def layers_for(raw):
        ls = []
        if with_perf:
            for i in range(1, l):
                ls.append(dict(space=sp[i - 1], motion=mo[i] if with_motion else None, xyz=xyz[:, i], raw=raw[:, i],
                               times=rays[:, 6 + i], ray_list=lst[i], ray_count=cnt[i:i + 1]))
        ls.append(dict(space=bk, motion=mo[0] if bkgd_deform else None, xyz=xyz[:, 0], raw=raw[:, 0],
                       times=rays[:, 6] if bkgd_deform else None, plain_time=True))
        return ls

    return dict(name=name, n=n, ns=ns, l=l, deep=deep, layers_for=layers_for, dirs=rays[:, 3:6], mask=mask,
                sd_b=sd_b, sd_p=sd_p, sd_m=sd_m, xyz=xyz, rays=rays, bkgd_deform=bkgd_deform)


SCENARIOS = [
    ("bkgd only, ns=64", dict(n=700, ns=64, deep=False, bkgd_deform=False, with_perf=False, with_motion=False)),
    ("bkgd + timed performers, no motion, ns=13", dict(n=1100, ns=13, deep=False, bkgd_deform=False, with_perf=True, with_motion=False)),
    ("bkgd + performers + motion, ns=64", dict(n=1100, ns=64, deep=False, bkgd_deform=False, with_perf=True, with_motion=True)),
    ("deformed timed bkgd + performers + motion, ns=128", dict(n=600, ns=128, deep=False, bkgd_deform=True, with_perf=True, with_motion=True)),
    ("deep_rgb, deformed bkgd, ns=9", dict(n=1100, ns=9, deep=True, bkgd_deform=True, with_perf=True, with_motion=True)),
    ("ragged: ns=90, n=517", dict(n=517, ns=90, deep=False, bkgd_deform=False, with_perf=True, with_motion=True)),
    ("few items: ns=16, n=768", dict(n=768, ns=16, deep=False, bkgd_deform=False, with_perf=True, with_motion=True)),
    ("few items: ns=24, n=768", dict(n=768, ns=24, deep=False, bkgd_deform=False, with_perf=True, with_motion=True)),
    ("tiny: ns=16, n=40", dict(n=40, ns=16, deep=False, bkgd_deform=False, with_perf=True, with_motion=True)),
]


def check(save=None, against=None):
    bad_total = 0
    saved = torch.load(against) if against else {}
    keep = {}
    for name, kw in SCENARIOS:
        sc = scenario(name, **kw)
        n, l, ns = sc["n"], sc["l"], sc["ns"]
        b = torch.full((n, l, ns, 4), 7.0, device="cuda")
        run(sc["layers_for"](b), sc["dirs"], ns, deep_rgb=sc["deep"], sigmoid_rgb=bool(os.environ.get("SIGMOID")))
        keep[name] = b.cpu()
        if name in saved:
            a = saved[name].cuda()
            diff = (a != b)
            nbad = int(diff.any(-1).sum())
            bad_total += nbad
            line = f"[{name}] rows differing from {against}: {nbad} of {n * l * ns}"
            if nbad:
                d = (a - b).abs()
                comp = [int(diff[..., c].sum()) for c in range(4)]
                line += f"; per component r,g,b,sigma: {comp}; max |d| {float(d.max()):.3e}; finite: {bool(torch.isfinite(b).all())}"
                for layer in range(l):
                    dl = diff[:, layer].any(-1)
                    if bool(dl.any()):
                        idx = dl.nonzero()[:6].tolist()
                        line += f"\n    layer {layer}: {int(dl.sum())} rows, first (ray, sample): {idx}"
                        r, s_ = idx[0]
                        line += f"\n      saved {a[r, layer, s_].tolist()}\n      this  {b[r, layer, s_].tolist()}"
                untouched = (b.cpu()[~sc["mask"].bool()] == 7.0).all() if kw["with_perf"] else True
                line += f"\n    rows of unlisted rays untouched: {bool(untouched)}"
            print(line, flush=True)
        # twice the same bits (dynamic scheduling does not touch the arithmetic)
        raw2 = torch.full((n, l, ns, 4), 7.0, device="cuda")
        run(sc["layers_for"](raw2), sc["dirs"], ns, deep_rgb=sc["deep"], sigmoid_rgb=bool(os.environ.get("SIGMOID")))
        if not torch.equal(raw2, b):
            print(f"[{name}] not deterministic: {int((raw2 != b).any(-1).sum())} rows differ between two runs", flush=True)
            bad_total += 1
    if save:
        torch.save(keep, save)
    print("CHECK", "OK" if bad_total == 0 else f"FAILED ({bad_total})", flush=True)
    return bad_total == 0


# ---------------------------------------------------------------------------------------------- per-layer localisation
def pe(x, n_freq):
    out = [x]
    for f in range(n_freq):
        out += [torch.sin(x * 2.0 ** f), torch.cos(x * 2.0 ** f)]
    return torch.cat(out, -1)


def space_reference(sd, pos, dirs, times, use_time):
    """fp64 activations after every stage of modeling/spacenet.py:101-160 -> {stage id: (rows, width)}"""
    W = lambda k: sd[f"net.{k}.weight"].double()
    B = lambda k: sd[f"net.{k}.bias"].double()
    acts = {}
    p = pe(pos.double(), 10)
    acts[100] = torch.cat([p, torch.zeros(p.shape[0], 1, dtype=torch.float64)], -1)
    h = p
    for i, k in enumerate(["stage1.0", "stage1.2", "stage1.4", "stage1.6"]):
        h = torch.relu(h @ W(k).T + B(k))
        acts[i] = h
    h = torch.cat([h, p], -1)
    for i, k in enumerate(["stage2.0", "stage2.2", "stage2.4"]):
        h = torch.relu(h @ W(k).T + B(k))
        acts[4 + i] = h
    enc = [pe(dirs.double(), 4)]
    if use_time:
        enc.append(pe(times.double().reshape(-1, 1), 10))
    e = torch.relu(torch.cat(enc, -1))
    x = torch.cat([h, e], -1)
    acts[7] = torch.relu(x @ W("rgb_net.1").T + B("rgb_net.1"))
    return acts


def motion_reference(sd, pos, times):
    W = lambda k: sd[f"net.{k}.weight"].double()
    B = lambda k: sd[f"net.{k}.bias"].double()
    t = times.double().reshape(-1, 1)
    lo = torch.floor(t)
    w = t - lo
    e = (1 - w) * pe(torch.cat([pos.double(), lo], -1), 10) + w * pe(torch.cat([pos.double(), lo + 1], -1), 10)
    acts = {199: torch.cat([e, torch.zeros(e.shape[0], 4, dtype=torch.float64)], -1)}
    h = e
    for i, k in enumerate(["motion_net.0", "motion_net.2", "motion_net.4", "motion_net.6", "motion_net.8"]):
        h = torch.relu(h @ W(k).T + B(k))
        acts[200 + i] = h
    return acts


def layers():
    lib = hip.lib()
    if not hasattr(lib, "stnerf_debug_wave_dump"):
        print("layers: needs the development build (STNERF_LIB=st-nerf_amd/libstnerf_hip_dbg.so)")
        return False
    lib.stnerf_debug_wave_dump.argtypes = [C.c_void_p, C.c_int]
    torch.manual_seed(5)
    rs = np.random.RandomState(3)
    n, ns = 300, 16
    ok = True
    # slot 0 = a timed SpaceNet with its MotionNet in front, every ray listed (row = ray * ns + sample)
    sd_s, sd_m = syn.spacenet_state("net", rs, True), syn.motionnet_state("net", rs)
    sp, mo = ops.pack_spacenet(sd_s, "net"), ops.pack_motionnet(sd_m, "net")
    xyz = ((torch.rand(n, ns, 3) - 0.5) * 5.0)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1)
    times = torch.where(torch.rand(n) < 0.5, torch.floor(torch.rand(n) * 30), torch.rand(n) * 30) + 1
    dx, dd, dt = xyz.cuda(), dirs.cuda(), times.cuda()
    raw = torch.zeros(n, ns, 4, device="cuda")
    pos = xyz.reshape(-1, 3)
    tt = times.reshape(n, 1).expand(n, ns).reshape(-1)
    dr = dirs.reshape(n, 1, 3).expand(n, ns, 3).reshape(-1, 3)
    for with_motion in (False, True):
        mref = motion_reference(sd_m, pos, tt) if with_motion else {}
        if with_motion:
            W, B = sd_m["net.motion_net.10.weight"].double(), sd_m["net.motion_net.10.bias"].double()
            pos_s = pos.double() + mref[204] @ W.T + B
        else:
            pos_s = pos.double()
        sref = space_reference(sd_s, pos_s, dr, tt, True)
        ref = dict(mref)
        ref.update(sref)
        for stage in sorted(ref):
            buf = torch.full((n * ns, 256), float("nan"), device="cuda")
            lib.stnerf_debug_wave_dump(C.c_void_p(buf.data_ptr()), stage)
            run([dict(space=sp, motion=mo if with_motion else None, xyz=dx, raw=raw, times=dt)], dd, ns)
            lib.stnerf_debug_wave_dump(None, -1)
            want = ref[stage]
            got = buf[:, :want.shape[1]].cpu().double()
            err = (got - want).abs()
            scale = float(want.abs().max()) + 1e-30
            nanrows = int(torch.isnan(got).any(-1).sum())
            worst = int(err.nan_to_num(1e30).max(-1)[0].argmax())
            wcol = int(err[worst].nan_to_num(1e30).argmax())
            flag = "ok " if (nanrows == 0 and float(err.max()) <= 2e-4 * scale) else "BAD"
            ok = ok and flag == "ok "
            print(f"  motion={with_motion} stage {stage:3d}: {flag} max |err| {float(err.nan_to_num(1e30).max()):.3e} (scale {scale:.2e}), "
                  f"rows with NaN {nanrows}, worst row {worst} (lane c {worst % 32}, wave {worst % 128 // 32}) feature {wcol}", flush=True)
            if flag == "BAD":
                badcols = (err.nan_to_num(1e30) > 2e-4 * scale).any(0).nonzero().flatten().tolist()
                badrows = (err.nan_to_num(1e30) > 2e-4 * scale).any(1).nonzero().flatten().tolist()
                print(f"      bad features ({len(badcols)}): {badcols[:48]}\n      bad rows ({len(badrows)}): {badrows[:32]}", flush=True)
    print("LAYERS", "OK" if ok else "FAILED", flush=True)
    return ok


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def time_():
    n, ns = int(os.environ.get("RAYS", 131072)), 64
    iters = int(os.environ.get("ITERS", 3))
    rs = np.random.RandomState(0)
    bk = ops.pack_spacenet(syn.spacenet_state("net", rs, False), "net")
    sp = ops.pack_spacenet(syn.spacenet_state("net", rs, True), "net")
    mo = ops.pack_motionnet(syn.motionnet_state("net", rs), "net")
    xyz = (torch.rand(n, ns, 3, device="cuda") - 0.5) * 4
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, device="cuda"), dim=-1)
    times = torch.rand(n, device="cuda") * 20 + 1
    raw = torch.empty(n, ns, 4, device="cuda")
    rows = n * ns
    cases = {
        "bkgd only": ([dict(space=bk, motion=None, xyz=xyz, raw=raw)], FLOP_SPACE),
        "performer, no motion": ([dict(space=sp, motion=None, xyz=xyz, raw=raw, times=times)], FLOP_SPACE_TIME),
        "performer fused with motion": ([dict(space=sp, motion=mo, xyz=xyz, raw=raw, times=times)], FLOP_MOTION + FLOP_SPACE_TIME),
    }
    only = os.environ.get("CASES")
    for name, (ls, flop) in cases.items():
        if only and not any(name.startswith(o) for o in only.split(",")):
            continue
        ms = timeit(lambda: ops.mlp_stage(ls, dirs, ns, sigmoid_rgb=True), iters)
        if True:
            print(f"{name:30s} {ms:9.3f} ms  {rows * flop / (ms * 1e-3) / 1e12:7.2f} TF/s", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "check"
    if what == "check":
        opt = dict(zip(sys.argv[2::2], sys.argv[3::2]))
        sys.exit(0 if check(opt.get("--save"), opt.get("--against")) else 1)
    elif what == "layers":
        sys.exit(0 if layers() else 1)
    elif what == "time":
        time_()
