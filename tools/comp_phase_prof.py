#!/usr/bin/env python3
"""Per-phase cycle split of the compositor's multi-layer kernel (library built with STNERF_EXTRA_FLAGS=-DSTNERF_COMP_PROF
STNERF_LIB_TAG=cprof; run with STNERF_LIB=.../libstnerf_hip_cprof.so): composite_merge_kernel, or composite_kernel (the
LDS-staged one) with STNERF_COMPOSITE_KERNEL=staged.  L / S / HIT from the environment (default 3 / 128,64 / 1.0,0.4)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stnerf_amd import hip, ops
if os.environ.get("STNERF_COMPOSITE_KERNEL") == "staged":
    names = ["flags", "live mask", "single-layer path", "staging", "per-layer composites", "rank merge", "merged composite", "order/tail"]
else:
    names = ["flags", "live mask", "loads + edits + per-layer composites", "insertion merge", "merged composite", "-", "-", "-"]
l = int(os.environ.get("L", 3))
n = 262144 if l <= 5 else 65536
S_fine, S_coarse = (int(x) for x in os.environ.get("S", "128,64").split(","))
torch.manual_seed(0)
for S, fine, frac in ((S_fine, True, 1.0), (S_fine, True, 0.4), (S_coarse, False, 1.0)):
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
    multi = int((hit.sum(1) > 1).sum())
    print(f"S={S} fine={fine} hit fraction {frac}: {tot / max(multi, 1):.0f} wave-cycles (100 MHz ticks) per multi-layer ray: "
          + ", ".join(f"{nm} {100 * v / tot:.1f}%" for nm, v in zip(names, buf) if nm != "-"))
