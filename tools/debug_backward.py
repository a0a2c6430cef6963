"""Per-parameter gradient errors of the SpaceNet / MotionNet backward against fp64 autograd (development aid)."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import stnerf_oracle as O
from stnerf_amd import ops, synthetic as syn
from stnerf_amd.modeling.spacenet import SpaceNet

def rel(got, ref):
    return float((got.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))

# GEMM shapes of the SpaceNet path with more than one sample slice
g = torch.Generator().manual_seed(0)
for (m, n, k, ldx, col0) in [(2970, 256, 63, 320, 256), (2970, 256, 256, 256, 0), (5120, 256, 319, 320, 0), (2970, 128, 283, 284, 0), (4100, 3, 128, 128, 0)]:
    xb = torch.randn(m, ldx, generator=g).cuda()
    x = xb[:, col0:col0 + k]
    dyb = torch.randn(m, (n + 3) // 4 * 4, generator=g).cuda()
    dy = dyb[:, :n]
    dw, db = torch.empty(n, k, device="cuda"), torch.empty(n, device="cuda")
    ops.train_linear_dw(dy, x, dw, db, False)
    print(f"dw m={m} n={n} k={k} ld={ldx} col0={col0}: rel err {rel(dw, dy.double().cpu().T @ x.double().cpu()):.2e}, db {rel(db, dy.double().cpu().sum(0)):.2e}")

for (use_time, n, ns) in [(True, 30, 64), (True, 45, 66), (True, 200, 64)]:
    sd = syn.spacenet_state("net", np.random.RandomState(1), use_time)
    net = SpaceNet(use_time=use_time)
    net.load_state_dict({k[4:]: v for k, v in sd.items()})
    net = net.cuda()
    pos = (torch.rand(n, ns, 3, generator=g) - 0.5) * 4.0
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    times = torch.rand(n, 1, generator=g) * 20 + 1
    c_rgb, c_sig = torch.randn(n, ns, 3, generator=g), torch.randn(n, ns, 1, generator=g) * 0.1
    rays = torch.cat([torch.zeros(n, 3), dirs], -1).cuda()
    pd = pos.cuda().requires_grad_(True)
    rgb, sig = net(pd, rays, times.cuda())
    ((rgb * c_rgb.cuda()).sum() + (sig * c_sig.cuda()).sum()).backward()
    ps = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    p64 = pos.double().requires_grad_(True)
    r64, s64 = O.space_net(ps, "net", p64, dirs.double(), times.double())
    ((r64 * c_rgb.double()).sum() + (s64 * c_sig.double()).sum()).backward()
    print(f"--- SpaceNet {n} x {ns} = {n * ns} samples; forward rel err rgb {rel(rgb, r64.detach()):.2e} sigma {rel(sig, s64.detach()):.2e}")
    for k, p in net.named_parameters():
        print(f"   {k:22s} {rel(p.grad, ps['net.' + k].grad):.2e}")
    print(f"   {'pos':22s} {rel(pd.grad, p64.grad):.2e}")
