#!/usr/bin/env python3
"""Compositor alone at the BASELINE shapes (GPU box): ms per call and algorithmic TB/s, production kernels
(composite_single_kernel + composite_merge_kernel) or, with STNERF_COMPOSITE_KERNEL=staged, the LDS-staged kernel.

    python tools/bench_composite.py [rays]

Depth lists as the sampler leaves them (ascending inside each layer's interval, -1000 on the rays a layer misses); `hit`
is the probability that a ray crosses a performer's box (the taekwondo / walking views: ~0.4 of the rays cross any; the
9-layer orbit pose that looks along the row of boxes: every crossing ray crosses all of them)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from stnerf_amd import ops

CASES = [  # name, l, S, fine, performer hit pattern
    ("C3 coarse 3x64", 3, 64, False, "indep 0.4"),
    ("C3 fine 3x128", 3, 128, True, "indep 0.4"),
    ("C3 fine 3x128, all hit", 3, 128, True, "all 1.0"),
    ("C4 coarse 5x64", 5, 64, False, "indep 0.3"),
    ("C4 fine 5x128", 5, 128, True, "indep 0.3"),
    ("C5 coarse 9x128, side view", 9, 128, False, "indep 0.12"),
    ("C5 fine 9x192, side view", 9, 192, True, "indep 0.12"),
    ("C5 fine 9x192, along the row", 9, 192, True, "all 0.37"),
]


def make(n, l, S, pattern, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    kind, p = pattern.split()
    p = float(p)
    lo = torch.rand(n, l, 1, device="cuda", generator=g) * 3.0
    t = torch.sort(lo + (0.3 + 2.0 * torch.rand(n, l, 1, device="cuda", generator=g)) * torch.rand(n, l, S, device="cuda", generator=g), -1)[0]
    if kind == "indep":
        hit = torch.rand(n, l, device="cuda", generator=g) < p
    else:
        hit = (torch.rand(n, 1, device="cuda", generator=g) < p).expand(n, l).clone()
    hit[:, 0] = True
    t[~hit] = -1000.0
    raw = torch.randn(n, l, S, 4, device="cuda", generator=g)
    raw[..., :3] = torch.sigmoid(raw[..., :3])
    mask = hit.to(torch.uint8)
    if os.environ.get("HINT", "1") == "1":      # bit 1 = the sampler's "this layer's depths are all -1000" hint (include/stnerf.h)
        mask = mask | ((~hit).to(torch.uint8) << 1)
    return t, raw, mask


def main():
    n_arg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    which = os.environ.get("STNERF_COMPOSITE_KERNEL", "merge")
    only = os.environ.get("CASES")
    for name, l, S, fine, pattern in CASES:
        if only and not any(name.startswith(o) for o in only.split(",")):
            continue
        n = n_arg or (262144 if l * S <= 640 else 65536)
        t, raw, mask = make(n, l, S, pattern)
        kw = dict(fine=fine, cut_negative_t=not fine, thresholds=[0.0 if fine else None] + [0.1] * (l - 1),
                  evaluated=[2] + [1] * (l - 1), want_weights=not fine, rgb_activated=True, near=0.05)
        for _ in range(2):
            ops.composite(t, raw, mask, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 5
        e0.record()
        for _ in range(iters):
            ops.composite(t, raw, mask, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        # algorithmic bytes as the launch profiler counts them: every depth + every float4 + mask + outputs (+ weights)
        per_ray = 20 * l * S + l + 20 * (l + 1) + (4 * l * S if not fine else 0)
        live = float((t[:, :, 0] > -999).float().sum(1).mean())
        print(f"{which:6s} {name:32s} n={n:7d} live layers/ray {live:4.2f}: {ms:8.3f} ms  {n * per_ray / ms / 1e9:6.3f} TB/s algorithmic"
              f"  {ms * 1e6 / n:7.2f} ns/ray", flush=True)


if __name__ == "__main__":
    main()
