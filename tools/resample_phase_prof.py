#!/usr/bin/env python3
"""Resampler alone (GPU box): ms per call and TB/s at the BASELINE shapes, and -- with a library built with
STNERF_EXTRA_FLAGS=-DSTNERF_COMP_PROF STNERF_LIB_TAG=cprof (STNERF_LIB=.../libstnerf_hip_cprof.so) -- the per-phase
cycle split of resample_kernel.  Pairs a layer misses (all depths -1000) are `miss` of the performer pairs."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from stnerf_amd import hip, ops

NAMES = ["missed-pair check (+ its stores)", "stage t / w in LDS", "ATen-order sum, pdf, fp64 cdf scan", "bins", "draws + inverse cdf",
         "sort + ranks", "output (t, xyz)", "loop head"]
prof = hasattr(hip.lib(), "stnerf_debug_composite_phases") and "cprof" in os.environ.get("STNERF_LIB", "")
for name, l, n1, n2, miss in (("C3 3 x 64+64", 3, 64, 64, 0.6), ("C3 3 x 64+64, no misses", 3, 64, 64, 0.0), ("C4 5 x 64+64", 5, 64, 64, 0.7),
                              ("C5 9 x 128+64", 9, 128, 64, 0.8), ("yml 3 x 90+30", 3, 90, 30, 0.6)):
    n = 262144 if l * (n1 + n2) <= 640 else 65536
    g = torch.Generator(device="cuda").manual_seed(1)
    t = torch.sort(torch.rand(n, l, n1, device="cuda", generator=g) * 4 + 0.5, -1)[0]
    hit = torch.rand(n, l, device="cuda", generator=g) >= miss
    hit[:, 0] = True
    t[~hit] = -1000.0
    w = torch.rand(n, l, n1, device="cuda", generator=g) ** 4 * hit[..., None]
    rays = torch.cat([torch.rand(n, 3, device="cuda"), torch.nn.functional.normalize(torch.randn(n, 3, device="cuda"), dim=-1)], -1)
    for _ in range(2):
        ops.resample(t, w, n2, rays, seed=5)
    torch.cuda.synchronize()
    if prof:
        buf = (C.c_ulonglong * 8)()
        hip.lib().stnerf_debug_composite_phases(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.resample(t, w, n2, rays, seed=5)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    S = n1 + n2
    per_pair = 8 * n1 + 16 * S
    live = int(hit.sum())
    line = f"{name:26s} n={n} live pairs {live / (n * l):.2f}: {ms:7.3f} ms  {n * l * per_pair / ms / 1e9:6.3f} TB/s algorithmic  {ms * 1e6 / live:6.2f} ns / live pair"
    if prof:
        hip.lib().stnerf_debug_composite_phases(buf, 1)
        tot = sum(buf)
        line += f"\n    {tot / 5 / live:.0f} wave-cycles per live pair: " + ", ".join(f"{nm} {100 * v / tot:.1f}%" for nm, v in zip(NAMES, buf))
    print(line, flush=True)
