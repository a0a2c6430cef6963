#!/usr/bin/env python3
"""stnerf_train_dw_batch alone, on operands laid out as modeling/autograd.py hands them over (layers that share an activation matrix
share its storage, so the cross-tile reuse is the real one): SpaceNet's ten layers and MotionNet's six, ms per call from HIP events,
TF/s of useful work against the 157.3 TF/s f32 MFMA peak, unique operand bytes.  Under `rocprofv3 --pmc FETCH_SIZE` etc. set
ONLY=space / motion so that every dispatch of the kernel is the same problem (tools/pmc_training.py sums per kernel name)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stnerf_amd import ops
PEAK = 157.3
m = int(os.environ.get("SAMPLES", 1 << 18))
reps = int(os.environ.get("REPS", 10))
dev = "cuda"
buf = lambda c: torch.randn(m, (c + 3) // 4 * 4, device=dev)


def space():
    Cc, h0, h1, h2, g0, g1, R, t0 = buf(320), buf(256), buf(256), buf(256), buf(256), buf(256), buf(304), buf(128)
    dys = [buf(256) for _ in range(7)] + [buf(128)]
    d_raw, dS = buf(4), buf(1)
    xin = [Cc[:, 256:319], h0, h1, h2, Cc[:, :319], g0, g1]
    layers = [(dys[i], xin[i]) for i in range(7)] + [(dS[:, :1], R[:, :256]), (dys[7], R[:, :304]), (d_raw[:, :3], t0)]
    unique = 4 * m * (sum(t.shape[1] for t in (Cc, h0, h1, h2, g0, g1, R, t0)) + 7 * 256 + 128 + 4 + 4)
    return layers, unique


def motion():
    E, A = buf(96), [buf(128) for _ in range(5)]
    dys, dO = [buf(128) for _ in range(5)], buf(3)
    layers = [(dys[0], E[:, :84])] + [(dys[j], A[j - 1]) for j in range(1, 5)] + [(dO[:, :3], A[4])]
    return layers, 4 * m * (96 + 10 * 128 + 4)


for name, make in (("space", space), ("motion", motion)):
    if os.environ.get("ONLY", name) != name:
        continue
    layers, unique = make()
    full = [(dy, x, torch.empty(dy.shape[1], x.shape[1], device=dev), torch.empty(dy.shape[1], device=dev)) for dy, x in layers]
    flop = sum(2.0 * m * dy.shape[1] * x.shape[1] for dy, x in layers)
    ops.train_dw_batch(full, False)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        ops.train_dw_batch(full, False)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    med = ms[len(ms) // 2]
    print(f"{name:6s} m = {m}: {med:.3f} ms per call (min {ms[0]:.3f}, max {ms[-1]:.3f}; kernel + reduction), {flop / med / 1e9:.1f} TF/s = "
          f"{flop / med / 1e9 / PEAK:.3f} of the f32 MFMA peak; unique operand bytes {unique / 1e9:.3f} GB")
