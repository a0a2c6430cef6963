#!/usr/bin/env python3
"""Per-phase cycle split of composite_kernel (library built with STNERF_EXTRA_FLAGS=-DSTNERF_COMP_PROF)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stnerf_amd import hip, ops
names = ["flags", "live mask", "single-layer path", "staging", "per-layer composites", "rank merge", "merged composite", "order/tail"]
n, l = 262144, 3
torch.manual_seed(0)
for S, fine, frac in ((128, True, 1.0), (128, True, 0.4), (64, False, 1.0)):
    t = torch.sort(torch.rand(n, l, S, device="cuda") * 6, -1)[0]
    hit = torch.rand(n, l, device="cuda") < frac
    hit[:, 0] = True
    t[~hit] = -1000.0
    raw = torch.randn(n, l, S, 4, device="cuda")
    mask = hit.to(torch.uint8)
    kw = dict(fine=fine, cut_negative_t=not fine, thresholds=[0.0 if fine else None, 0.1, 0.1], evaluated=[2, 1, 1],
              want_weights=not fine, rgb_activated=True)
    buf = (C.c_ulonglong * 8)()
    ops.composite(t, raw, mask, **kw)
    torch.cuda.synchronize()
    hip.lib().stnerf_debug_composite_phases(buf, 1)
    ops.composite(t, raw, mask, **kw)
    torch.cuda.synchronize()
    hip.lib().stnerf_debug_composite_phases(buf, 1)
    tot = sum(buf)
    print(f"S={S} fine={fine} hit fraction {frac}: " + ", ".join(f"{nm} {100 * v / tot:.1f}%" for nm, v in zip(names, buf)))
