#!/usr/bin/env python3
"""Sustained timing of the bf16x3 stage kernel for ONE build of the library (STNERF_LIB selects a variant build): the kernel
is power-bound, so a variant is judged on several seconds of load, alternating builds on the same box (tools/gpu_bx_ab.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stnerf_amd import ops, synthetic as syn
FLOP_SPACE, FLOP_SPACE_TIME, FLOP_MOTION = 924_672, 930_048, 153_344
n, ns = int(os.environ.get("RAYS", 262144)), 64
secs = float(os.environ.get("SECONDS_PER_CASE", 4))
rs = np.random.RandomState(0)
torch.manual_seed(0)                                   # (the checksum compares BUILDS: same inputs in every process)
bk = ops.pack_spacenet(syn.spacenet_state("net", rs, False), "net", precision="bf16x3")
sp = ops.pack_spacenet(syn.spacenet_state("net", rs, True), "net", precision="bf16x3")
mo = ops.pack_motionnet(syn.motionnet_state("net", rs), "net", precision="bf16x3")
xyz = (torch.rand(n, ns, 3, device="cuda") - 0.5) * 4
dirs = torch.nn.functional.normalize(torch.randn(n, 3, device="cuda"), dim=-1)
times = torch.rand(n, device="cuda") * 20 + 1
raw = torch.empty(n, ns, 4, device="cuda")
cases = {"bkgd": ([dict(space=bk, motion=None, xyz=xyz, raw=raw)], FLOP_SPACE),
         "performer+motion": ([dict(space=sp, motion=mo, xyz=xyz, raw=raw, times=times)], FLOP_SPACE_TIME + FLOP_MOTION)}
for name, (ls, flop) in cases.items():
    ops.mlp_stage(ls, dirs, ns, sigmoid_rgb=True); torch.cuda.synchronize()
    t0, it = time.perf_counter(), 0
    while time.perf_counter() - t0 < secs:
        ops.mlp_stage(ls, dirs, ns, sigmoid_rgb=True); torch.cuda.synchronize(); it += 1
    dt = time.perf_counter() - t0
    print(f"{os.environ.get('STNERF_LIB', 'main').split('_')[-1].replace('.so', ''):8s} {name:18s} {it} launches, {1e3 * dt / it:.2f} ms each, "
          f"{n * ns * flop * it / dt / 1e12:.1f} TF/s algorithmic, checksum {float(raw.double().sum()):.6f}")
