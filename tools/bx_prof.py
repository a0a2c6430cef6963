#!/usr/bin/env python3
"""Per-phase cycle breakdown of the bf16x3 stage kernel (development build of the library with -DSTNERF_BX_PROF):
    STNERF_LIB_TAG=bxprof STNERF_EXTRA_FLAGS=-DSTNERF_BX_PROF python st-nerf_amd/build.py
    STNERF_LIB=st-nerf_amd/libstnerf_hip_bxprof.so python tools/bx_prof.py          (on the GPU box)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stnerf_amd import hip, ops, synthetic as syn
lib = hip.lib()
lib.stnerf_debug_bx_phases.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
n, ns = int(os.environ.get("RAYS", 131072)), 64
rs = np.random.RandomState(0)
bk = ops.pack_spacenet(syn.spacenet_state("net", rs, False), "net", precision="bf16x3")
sp = ops.pack_spacenet(syn.spacenet_state("net", rs, True), "net", precision="bf16x3")
mo = ops.pack_motionnet(syn.motionnet_state("net", rs), "net", precision="bf16x3")
xyz = (torch.rand(n, ns, 3, device="cuda") - 0.5) * 4
dirs = torch.nn.functional.normalize(torch.randn(n, 3, device="cuda"), dim=-1)
times = torch.rand(n, device="cuda") * 20 + 1
raw = torch.empty(n, ns, 4, device="cuda")
names = ["top (consts DMA, barrier)", "motion: encoding + first C", "motion: K passes", "motion: boundaries + head", "space: PE(pos) -> planes",
         "space: K passes", "space: pass A -> park", "space: pass B -> planes + unpark", "space: C operand loads", "space: sigma head (+ next item's loads)",
         "space: rgb tail + colour head", "store + barrier + queue"]
# ideal MFMA cycles (32 per v_mfma_f32_32x32x16_bf16): 48 per slot
ideal = {2: 19 * 48 * 32, 5: 112 * 48 * 32}
cases = {"bkgd only": [dict(space=bk, motion=None, xyz=xyz, raw=raw)],
         "performer fused with motion": [dict(space=sp, motion=mo, xyz=xyz, raw=raw, times=times)]}
# (round 6) the training forward: the same kernel with the activation tap, on the performer SpaceNet alone -- beside the plain launch of it
from stnerf_amd.modeling import autograd as A
bufs = A._activation_buffers(n * ns, 48, "cuda")
cases["performer SpaceNet alone"] = [dict(space=sp, motion=None, xyz=xyz, raw=raw, times=times)]
cases["performer SpaceNet alone, WITH THE TRAINING TAP"] = "tap"
buf = (C.c_ulonglong * 16)()
for name, ls in cases.items():
    if ls == "tap":
        run = lambda: ops.train_spacenet_fwd(sp, xyz, dirs, times, raw, A._act_views(bufs), bufs[0][:, 256:320], bufs[8])
    else:
        run = lambda: ops.mlp_stage(ls, dirs, ns, sigmoid_rgb=True)
    run(); torch.cuda.synchronize()
    lib.stnerf_debug_bx_phases(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    lib.stnerf_debug_bx_phases(buf, 1)
    items = buf[12]
    tot = sum(buf[i] for i in range(12))
    ms = e0.elapsed_time(e1)
    print(f"{name}: {items} item-waves, {tot / items:.0f} cycles per item-wave, launch {ms:.2f} ms -> {tot / items * (items / 1024) / (ms * 1e3):.0f} MHz effective")
    for i, nm in enumerate(names):
        c = buf[i] / items
        if c == 0:
            continue
        extra = f"   ideal MFMA {ideal[i]:7d}  -> x{c / ideal[i]:.3f}" if i in ideal else ""
        print(f"  {nm:42s} {c:10.0f} cycles  {100.0 * buf[i] / tot:5.1f} %{extra}")
