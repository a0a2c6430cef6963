#!/usr/bin/env python3
"""A/B of the persistent stage kernel (stnerf_mlp_stage) against the per-network launches on synthetic rows.
    python tools/bench_stage.py            (on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stnerf_amd import ops, synthetic as syn

FLOP_SPACE, FLOP_SPACE_TIME, FLOP_MOTION = 924_672, 930_048, 153_344
n, ns = int(os.environ.get("RAYS", 131072)), 64
iters = int(os.environ.get("ITERS", 5))
rs = np.random.RandomState(0)
bk = ops.pack_spacenet(syn.spacenet_state("net", rs, False), "net")
sp = ops.pack_spacenet(syn.spacenet_state("net", rs, True), "net")
mo = ops.pack_motionnet(syn.motionnet_state("net", rs), "net")
xyz = (torch.rand(n, ns, 3, device="cuda") - 0.5) * 4
dirs = torch.nn.functional.normalize(torch.randn(n, 3, device="cuda"), dim=-1)
times = torch.rand(n, device="cuda") * 20 + 1
raw = torch.empty(n, ns, 4, device="cuda")


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


rows = n * ns
cases = {
    "spacenet standalone (bkgd)": (lambda: ops.spacenet_fwd(bk, xyz, dirs, None, raw), FLOP_SPACE),
    "stage: bkgd only": (lambda: ops.mlp_stage([dict(space=bk, motion=None, xyz=xyz, raw=raw)], dirs, ns), FLOP_SPACE),
    "spacenet standalone (time)": (lambda: ops.spacenet_fwd(sp, xyz, dirs, times, raw), FLOP_SPACE_TIME),
    "stage: performer without motion": (lambda: ops.mlp_stage([dict(space=sp, motion=None, xyz=xyz, raw=raw, times=times)], dirs, ns), FLOP_SPACE_TIME),
    "motionnet standalone": (lambda: ops.motionnet_fwd(mo, xyz, times, add_to_xyz=True), FLOP_MOTION),
    "motion + space standalone": (lambda: (ops.motionnet_fwd(mo, xyz, times, add_to_xyz=True), ops.spacenet_fwd(sp, xyz, dirs, times, raw)), FLOP_MOTION + FLOP_SPACE_TIME),
    "stage: performer fused": (lambda: ops.mlp_stage([dict(space=sp, motion=mo, xyz=xyz, raw=raw, times=times)], dirs, ns), FLOP_MOTION + FLOP_SPACE_TIME),
}
if os.environ.get("STAGE_ONLY"):
    cases = {k: v for k, v in cases.items() if k.startswith("stage")}
# the split-bf16 arithmetic of the same launch (csrc/mlp_bf16x3.hip): algorithmic TF/s; executed bf16 MFMA = 6 x
rs = np.random.RandomState(0)
bk_x = ops.pack_spacenet(syn.spacenet_state("net", rs, False), "net", precision="bf16x3")
sp_x = ops.pack_spacenet(syn.spacenet_state("net", rs, True), "net", precision="bf16x3")
mo_x = ops.pack_motionnet(syn.motionnet_state("net", rs), "net", precision="bf16x3")
cases.update({
    "bf16x3 stage: bkgd only": (lambda: ops.mlp_stage([dict(space=bk_x, motion=None, xyz=xyz, raw=raw)], dirs, ns), FLOP_SPACE),
    "bf16x3 stage: performer fused": (lambda: ops.mlp_stage([dict(space=sp_x, motion=mo_x, xyz=xyz, raw=raw, times=times)], dirs, ns), FLOP_MOTION + FLOP_SPACE_TIME),
})
only = os.environ.get("CASES")   # comma-separated name prefixes
for name, (fn, flop) in cases.items():
    if only and not any(name.startswith(o) for o in only.split(",")):
        continue
    ms = timeit(fn)
    print(f"{name:36s} {ms:9.3f} ms  {rows * flop / (ms * 1e-3) / 1e12:7.2f} TF/s")
