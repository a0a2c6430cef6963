#!/usr/bin/env python3
"""Why the split-bf16 training forward takes 1.63 ms inside a step and 1.32 ms back to back: the launch timed with HIP events
(a) alone in a loop, (b) behind the network's weight-gradient batch, (c) behind the gradient chain, (d) behind a 2 GB memset,
(e) behind 20 ms of idle stream (a host sleep).  262,144 samples."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stnerf_amd
from stnerf_amd import ops, synthetic as syn
from stnerf_amd.modeling.spacenet import SpaceNet
from stnerf_amd.modeling import autograd as A

net = SpaceNet(use_time=True)
net.load_state_dict({k[4:]: v for k, v in syn.spacenet_state("net", np.random.RandomState(1), True).items()})
net = net.cuda()
n, ns = 4096, 64
M = n * ns
g = torch.Generator().manual_seed(0)
pos = ((torch.rand(n, ns, 3, generator=g) - 0.5) * 4).cuda()
dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda()
tm = (torch.rand(n, generator=g) * 20 + 1).cuda()
bufs = A._activation_buffers(M, 48, "cuda")
raw = torch.empty(n, ns, 4, device="cuda")
packed = net._packed("bf16x3")
params = [p.detach() for p in net.training_parameters()]
blob = A.dx_blob_bf16x3(net, params, True)
d_raw = torch.randn(M, 4, device="cuda")
dys = [A._buf(M, 256, "cuda")[:, :256] for _ in range(7)] + [A._buf(M, 128, "cuda")[:, :128]]
dpe, dpe_skip = A._buf(M, 64, "cuda")[:, :64], A._buf(M, 64, "cuda")[:, :64]
acts = A._act_views(bufs)
gW = [torch.empty_like(params[2 * i]) for i in range(10)]
gB = [torch.empty_like(params[2 * i + 1]) for i in range(10)]
Cc, R = bufs[0], bufs[6]
xin = [Cc[:, 256:319], acts[0], acts[1], acts[2], Cc[:, :319], acts[4], acts[5]]
dS = A._buf(M, 1, "cuda")
batch = ([(dys[i], xin[i], gW[i], gB[i]) for i in range(7)] + [(dS[:, :1], acts[6], gW[7], gB[7]), (dys[7], R[:, :304], gW[8], gB[8]), (d_raw[:, :3], acts[7], gW[9], gB[9])])
big = torch.empty(1 << 29, dtype=torch.float32, device="cuda")

fwd = lambda: ops.train_spacenet_fwd(packed, pos, dirs, tm, raw, acts, bufs[0][:, 256:320], bufs[8])
dx = lambda: ops.train_spacenet_dx_bf16x3(blob, d_raw, bufs[8], dys, dpe, dpe_skip)
dw = lambda: ops.train_dw_batch(batch, False)


def timed(first, before, reps=12):
    """median ms of `first` when `before` runs in front of it each time"""
    ts = []
    for _ in range(reps):
        before()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        first()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


for _ in range(3):
    fwd(); dx(); dw()
torch.cuda.synchronize()
nothing = lambda: None
for name, k in (("forward with tap", fwd), ("gradient chain", dx), ("weight-gradient batch", dw)):
    print(f"{name}: alone {timed(k, k):.3f} ms | behind the dW batch {timed(k, dw):.3f} | behind the chain {timed(k, dx):.3f} | behind the forward {timed(k, fwd):.3f} | "
          f"behind a 2 GB fill {timed(k, lambda: big.fill_(1.0)):.3f} | behind 20 ms of idle {timed(k, lambda: (torch.cuda.synchronize(), time.sleep(0.02))):.3f}")
