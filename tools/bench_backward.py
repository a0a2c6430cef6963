#!/usr/bin/env python3
"""Throughput of the training kernels (csrc/train.hip) on the GPU box: the three GEMM flavours at the networks' shapes, and a
whole SpaceNet / MotionNet backward (recompute + dX + dW) in TF/s of f32 MFMA work against the 157.3 TF/s peak."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stnerf_amd import ops, synthetic as syn
from stnerf_amd.modeling.spacenet import SpaceNet
from stnerf_amd.modeling.motion_net import MotionNet
PEAK = 157.3

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

m = int(os.environ.get("SAMPLES", 1 << 18))
print(f"{m} samples")
for n, k in ([] if os.environ.get("ONLY_NETS") else [(256, 256), (256, 64), (256, 320), (128, 304), (128, 128)]):
    x, w, dy = torch.randn(m, k, device="cuda"), torch.randn(n, k, device="cuda"), torch.randn(m, n, device="cuda")
    y, dx, dw, db = torch.empty(m, n, device="cuda"), torch.empty(m, k, device="cuda"), torch.empty(n, k, device="cuda"), torch.empty(n, device="cuda")
    fl = 2.0 * m * n * k / 1e12
    t1 = timed(lambda: ops.train_linear_fwd(x, w, db, y, True))
    t2 = timed(lambda: ops.train_linear_dx(dy, w, dx, mask=x))
    t3 = timed(lambda: ops.train_linear_dw(dy, x, dw, db, False))
    tb = timed(lambda: torch.mm(x, w.t()))
    print(f"  {n:3d} x {k:3d}: forward {fl / t1:6.1f} TF/s ({fl / t1 / PEAK:.2f})  dX {fl / t2:6.1f} ({fl / t2 / PEAK:.2f})  dW {fl / t3:6.1f} ({fl / t3 / PEAK:.2f})"
          f"   [rocBLAS sgemm forward via torch.mm: {fl / tb:6.1f}]")
# every weight / bias gradient of a SpaceNet: ten launch groups against stnerf_train_dw_batch
shapes = [] if os.environ.get("ONLY_NETS") else [(256, 63), (256, 256), (256, 256), (256, 256), (256, 319), (256, 256), (256, 256), (1, 256), (128, 304), (3, 128)]
pad = lambda c: (c + 3) // 4 * 4
layers = [(torch.randn(m, pad(n_), device="cuda")[:, :n_], torch.randn(m, pad(k_), device="cuda")[:, :k_], torch.empty(n_, k_, device="cuda"),
           torch.empty(n_, device="cuda")) for n_, k_ in shapes]
fl = sum(2.0 * m * n_ * k_ for n_, k_ in shapes) / 1e12
ta = timed(lambda: [ops.train_linear_dw(*l, False) for l in layers]) if shapes else 1.0
tb = timed(lambda: ops.train_dw_batch(layers, False)) if shapes else 1.0
if shapes:
  print(f"  SpaceNet's ten dW + db: per layer {1e3 * ta:.3f} ms ({fl / ta:.1f} TF/s), one batch {1e3 * tb:.3f} ms ({fl / tb:.1f} TF/s = {fl / tb / PEAK:.2f})")
del layers
rs = np.random.RandomState(0)
n, ns = m // 64, 64
for name, net, flop in [("SpaceNet (time)", SpaceNet(use_time=True), 930_048), ("MotionNet", MotionNet(c_input=4, input_time=True), 153_344)]:
    sd = syn.spacenet_state("net", rs, True) if "Space" in name else syn.motionnet_state("net", rs)
    net.load_state_dict({k[4:]: v for k, v in sd.items()})
    net = net.cuda()
    if "Space" in name:
        pos = ((torch.rand(n, ns, 3, device="cuda") - 0.5) * 4).requires_grad_(True)
        rays = torch.cat([torch.zeros(n, 3, device="cuda"), torch.nn.functional.normalize(torch.randn(n, 3, device="cuda"), dim=-1)], -1)
        tm = torch.rand(n, 1, device="cuda") * 20 + 1
        def step():
            net.zero_grad(set_to_none=True)
            rgb, sig = net(pos, rays, tm)
            (rgb.sum() + sig.sum()).backward()
    else:
        xt = torch.cat([(torch.rand(m, 3, device="cuda") - 0.5) * 4, torch.rand(m, 1, device="cuda") * 20 + 1], -1).requires_grad_(True)
        def step():
            net.zero_grad(set_to_none=True)
            net(xt).sum().backward()
    t = timed(step, 3)
    if not os.environ.get("ONLY_FUSED"):                          # A/B: one dW / db launch group per layer
        from stnerf_amd.modeling import autograd as A
        A.DW_BATCH = False
        t_layers = timed(step, 3)
        A.DW_BATCH = True
        print(f"{name}: weight gradients per layer (STNERF_TRAIN_DW_BATCH=0) {1e3 * t_layers:.2f} ms; batched {1e3 * t:.2f} ms")
    if not os.environ.get("ONLY_FUSED"):     # A/B: the round-4 per-layer backward
        from stnerf_amd.modeling import autograd as A
        A.FUSED_BACKWARD = False
        t_old = timed(step, 3)
        A.FUSED_BACKWARD = True
        print(f"{name}: per-layer backward (STNERF_TRAIN_FUSED=0) {1e3 * t_old:.1f} ms; fused {1e3 * t:.1f} ms")
    # forward (with the tap: the activations are kept at this size) + d x + d W = 3 x the network's FLOPs; with STNERF_TRAIN_KEEP_GB=0 the
    # backward runs the forward once more: 4 x.  (Up to round 6's first profiles this line counted 4 x in both cases.)
    from stnerf_amd.modeling import autograd as A
    passes = 3 if A.KEEP_BYTES >= m * 4 * (A.ACT_FLOATS_PER_SAMPLE if "Space" in name else A.MOTION_ACT_FLOATS_PER_SAMPLE) else 4
    arith = (A.TRAIN_FWD or net.precision) if "Space" in name else "fp32"
    print(f"{name}: forward + backward of {m} samples {1e3 * t:.2f} ms = {m / t / 1e6:.2f} M samples/s, {passes * flop * m / t / 1e12:.1f} TF/s of network work "
          f"({passes} x the forward's FLOPs: {'activations kept' if passes == 3 else 'recomputed'}; {passes * flop * m / t / 1e12 / PEAK:.2f} of the f32 MFMA peak; "
          f"forward / d x arithmetic: {arith})")
    if "Space" in name and not os.environ.get("ONLY_FUSED") and not A.TRAIN_FWD:     # A/B: the exact-f32 forward tap and d x chain (round 5's)
        A.TRAIN_FWD = "fp32"
        t32 = timed(step, 3)
        A.TRAIN_FWD = ""
        print(f"{name}: exact-f32 forward tap + d x chain (STNERF_TRAIN_FWD=fp32) {1e3 * t32:.2f} ms; split bf16 {1e3 * t:.2f} ms")
