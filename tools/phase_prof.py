"""Per-phase cycle breakdown of the SpaceNet kernel (needs a library built with -DSTNERF_PHASE_PROF)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stnerf_amd import hip, ops, synthetic as syn
lib = hip.lib()
reader = lib.stnerf_debug_read_phases_h if os.environ.get("PRECISION") == "fp16x3" else lib.stnerf_debug_read_phases
reader.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
n, ns = 131072, 64
torch.manual_seed(0)
xyz = ((torch.rand(n, ns, 3) - 0.5) * 6).cuda()
dirs = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1).cuda()
raw = torch.empty(n, ns, 4, device="cuda")
net = ops.pack_spacenet(syn.spacenet_state("net", np.random.RandomState(0), False), "net", precision=os.environ.get("PRECISION", "fp32"))
ops.spacenet_fwd(net, xyz, dirs, None, raw); torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
reader(buf, 1)
ops.spacenet_fwd(net, xyz, dirs, None, raw); torch.cuda.synchronize()
reader(buf, 1)
names = ["PE", "MMA", "BAR1(pre-epilogue)", "EPI", "BAR2(post-epilogue)", "ENC2", "HEAD", "MISC"]
wgs = buf[8]; tiles = n * ns // 128
tot = sum(buf[i] for i in range(8))
print(os.environ.get("PRECISION", "fp32"), "workgroups", wgs, "tiles", tiles, "cycles/tile", tot / tiles)
for i, nm in enumerate(names):
    print(f"  {nm:22s} {buf[i] / tiles:10.0f} cycles/tile  {100.0 * buf[i] / tot:5.1f} %")
