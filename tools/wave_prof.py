#!/usr/bin/env python3
"""Per-phase cycle breakdown of the wave stage kernel (library built with -DSTNERF_WAVE_PROF; STNERF_LIB=...)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stnerf_amd import hip, ops, synthetic as syn
lib = hip.lib()
lib.stnerf_debug_wave_phases.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
n, ns = int(os.environ.get("RAYS", 131072)), 64
rs = np.random.RandomState(0)
bk = ops.pack_spacenet(syn.spacenet_state("net", rs, False), "net")
sp = ops.pack_spacenet(syn.spacenet_state("net", rs, True), "net")
mo = ops.pack_motionnet(syn.motionnet_state("net", rs), "net")
xyz = (torch.rand(n, ns, 3, device="cuda") - 0.5) * 4
dirs = torch.nn.functional.normalize(torch.randn(n, 3, device="cuda"), dim=-1)
times = torch.rand(n, device="cuda") * 20 + 1
raw = torch.empty(n, ns, 4, device="cuda")
names = ["top (pop, row lookup)", "motion: encoding", "motion: 5 layers", "motion: head", "space: PE(pos)", "space: stage1.0",
         "space: 6 x 256 layers", "space: sigma + dir/time enc", "space: rgb_net.1", "space: rgb head", "store + barrier + shift"]
# ideal MFMA cycles per phase (64 per v_mfma_f32_32x32x2_f32)
ideal = {2: (11 * 16 + 4 * 16 * 16) * 64, 5: 256 * 64, 6: (6 * 1024 + 256) * 64}
cases = {"bkgd only": ([dict(space=bk, motion=None, xyz=xyz, raw=raw)], 36 * 16 * 64),
         "performer fused with motion": ([dict(space=sp, motion=mo, xyz=xyz, raw=raw, times=times)], 38 * 16 * 64)}
buf = (C.c_ulonglong * 16)()
for name, (ls, rgb1_ideal) in cases.items():
    ops.mlp_stage(ls, dirs, ns, sigmoid_rgb=True); torch.cuda.synchronize()
    lib.stnerf_debug_wave_phases(buf, 1)
    ops.mlp_stage(ls, dirs, ns, sigmoid_rgb=True); torch.cuda.synchronize()
    lib.stnerf_debug_wave_phases(buf, 1)
    items = buf[11]
    buf[6] = buf[12] + buf[13]   # the layer loop = its K segments + its ReLU / bias passes
    tot = sum(buf[i] for i in range(11))
    print(f"{name}: {items} item-waves, {tot / items:.0f} cycles per item-wave")
    idl = dict(ideal); idl[8] = rgb1_ideal
    for i, nm in enumerate(names):
        c = buf[i] / items
        extra = f"   ideal MFMA {idl[i]:7d}  -> overhead {c - idl[i]:8.0f}" if i in idl and c > 0 else ""
        print(f"  {nm:30s} {c:10.0f} cycles  {100.0 * buf[i] / tot:5.1f} %{extra}")
    print(f"  layer loop split: K segments {buf[12] / items:.0f} (ideal {idl[6]}), ReLU + bias passes {buf[13] / items:.0f} cycles per item-wave")
