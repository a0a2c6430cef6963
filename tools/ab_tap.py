#!/usr/bin/env python3
"""What the split-bf16 training forward's tap costs: the launch alone on 262,144 samples (STNERF_LIB selects a development build of
csrc/mlp_bf16x3.hip with -DSTNERF_DEV_TAP_NO_STORES / -DSTNERF_DEV_TAP_NO_BITS; profiles/retired_designs.md).
    STNERF_LIB=st-nerf_amd/libstnerf_hip_nostores.so python tools/ab_tap.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stnerf_amd
from stnerf_amd import ops, synthetic as syn
from stnerf_amd.modeling.spacenet import SpaceNet
from stnerf_amd.modeling import autograd as A

net = SpaceNet(use_time=True)
net.load_state_dict({k[4:]: v for k, v in syn.spacenet_state("net", np.random.RandomState(1), True).items()})
net = net.cuda()
n, ns = 4096, 64
M = n * ns
g = torch.Generator().manual_seed(0)
pos = ((torch.rand(n, ns, 3, generator=g) - 0.5) * 4).cuda()
dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda()
tm = (torch.rand(n, generator=g) * 20 + 1).cuda()
bufs = A._activation_buffers(M, 48, "cuda")
raw = torch.empty(n, ns, 4, device="cuda")
out = []
for prec in ("bf16x3", "fp32"):
    packed = net._packed(prec)
    run = lambda: ops.train_spacenet_fwd(packed, pos, dirs, tm, raw, A._act_views(bufs), bufs[0][:, 256:320], bufs[8])
    for _ in range(3):
        run()
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10)
    out.append(f"{prec} {1e3 * best:.3f} ms")
    if prec == "bf16x3":
        want = torch.empty_like(raw)
        ops.spacenet_fwd(packed, pos, dirs, tm, want)
        t0 = time.perf_counter()
        for _ in range(10):
            ops.spacenet_fwd(packed, pos, dirs, tm, want)
        torch.cuda.synchronize()
        out.append(f"(no tap: {(time.perf_counter() - t0) * 100:.3f} ms)")
print(os.environ.get("STNERF_LIB", "product"), f"train forward of {M} samples:", ", ".join(out))
