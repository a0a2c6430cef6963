"""Microbenchmark of the fused per-network MLP kernels (mlp.hip: the op-level entry points)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stnerf_amd import ops, synthetic as syn

n, ns = int(os.environ.get("N_RAYS", 131072)), int(os.environ.get("NS", 64))
iters = int(os.environ.get("ITERS", 5))
rs = np.random.RandomState(0)
torch.manual_seed(0)
xyz = ((torch.rand(n, ns, 3) - 0.5) * 6).cuda()
dirs = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1).cuda()
times = (torch.rand(n) * 50).cuda()
raw = torch.empty(n, ns, 4, device="cuda")
res = {}
prec = os.environ.get("PRECISION", "fp32")
for name, kind_time, flop in (("space", False, 924672), ("space_time", True, 930048)):
    net = ops.pack_spacenet(syn.spacenet_state("net", rs, kind_time), "net", precision=prec)
    ops.spacenet_fwd(net, xyz, dirs, times if kind_time else None, raw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.spacenet_fwd(net, xyz, dirs, times if kind_time else None, raw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    res[name] = (ms, n * ns * flop / ms / 1e9)
mot = ops.pack_motionnet(syn.motionnet_state("net", rs), "net", precision=prec)
x2 = xyz.clone()
ops.motionnet_fwd(mot, x2, times, add_to_xyz=True); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ops.motionnet_fwd(mot, x2, times, add_to_xyz=True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
res["motion"] = (ms, n * ns * 153344 / ms / 1e9)
print(prec, " ".join(f"{k}: {v[0]:.2f} ms {v[1]:.1f} TF/s" for k, v in res.items()), "rows", n * ns)
