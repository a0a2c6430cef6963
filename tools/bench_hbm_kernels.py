#!/usr/bin/env python3
"""Microbenchmarks of the HBM-side kernels (compositor, resampler, sampler) on synthetic rays with a controlled fraction
of performer hits.    python tools/bench_hbm_kernels.py   (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stnerf_amd import ops

n, l = int(os.environ.get("RAYS", 262144)), 3
iters = int(os.environ.get("ITERS", 10))


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


torch.manual_seed(0)
for S, fine in ((64, False), (128, True)):
    for frac in (0.0, 0.4, 1.0):
        t = torch.sort(torch.rand(n, l, S, device="cuda") * 6, -1)[0]
        hit = torch.rand(n, l, device="cuda") < frac
        hit[:, 0] = True
        t[~hit] = -1000.0
        raw = torch.randn(n, l, S, 4, device="cuda")
        mask = hit.to(torch.uint8)
        live = int(hit.sum())
        kw = dict(fine=fine, cut_negative_t=not fine, thresholds=[0.0 if fine else None, 0.1, 0.1], evaluated=[2, 1, 1],
                  want_weights=not fine, rgb_activated=True)
        ms = timeit(lambda: ops.composite(t, raw, mask, **kw))
        by = 20 * S * live + (l + 20 * (l + 1) + (0 if fine else 4 * l * S)) * n
        print(f"composite S={S:3d} fine={int(fine)} performer hit fraction {frac:.1f}: {ms:7.3f} ms  {by / ms / 1e6:8.1f} GB/s (live bytes)  {1e3 * ms / n:6.3f} us/kray")
rays = torch.cat([torch.rand(n, 3, device="cuda"), torch.nn.functional.normalize(torch.randn(n, 3, device="cuda"), dim=-1)], -1)
for n1, n2 in ((64, 64), (90, 30)):
    for frac in (0.0, 0.4, 1.0):
        t = torch.sort(torch.rand(n, l, n1, device="cuda") * 6, -1)[0]
        hit = torch.rand(n, l, device="cuda") < frac
        hit[:, 0] = True
        t[~hit] = -1000.0
        w = torch.rand(n, l, n1, device="cuda") ** 8
        ms = timeit(lambda: ops.resample(t, w, n2, rays, seed=1))
        by = 8 * n1 * int(hit.sum()) + 16 * (n1 + n2) * l * n
        print(f"resample {n1}+{n2} performer hit fraction {frac:.1f}: {ms:7.3f} ms  {by / ms / 1e6:8.1f} GB/s  {1e3 * ms / n:6.3f} us/kray")
bx = torch.tensor([[[-3, -3, -3], [3, -3, -3], [3, 3, -3], [-3, 3, -3], [-3, -3, 3], [3, -3, 3], [3, 3, 3], [-3, 3, 3]]] * l,
                  dtype=torch.float32, device="cuda")
r2 = torch.cat([rays[:, :3] * 0 + torch.tensor([0.0, 0.0, -4.0], device="cuda"), rays[:, 3:], torch.ones(n, l, device="cuda")], -1).contiguous()
for n1 in (64, 90):
    ms = timeit(lambda: ops.sample_coarse(r2, bx, n1, seed=1))
    print(f"sample_coarse n1={n1}: {ms:7.3f} ms  {16 * n1 * l * n / ms / 1e6:8.1f} GB/s")
