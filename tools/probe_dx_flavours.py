#!/usr/bin/env python3
"""What dX = (dY W) * mask costs beyond a forward-flavoured GEMM of the same size (GPU box): the weight operand read across its
rows (W[n][k], k = the OUTPUT index) against a pre-transposed copy read along them, with and without the mask."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stnerf_amd import ops


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


m = 1 << 18
for n, k in [(256, 256), (256, 320), (128, 304), (256, 64)]:
    dy, w, x = torch.randn(m, n, device="cuda"), torch.randn(n, k, device="cuda"), torch.randn(m, k, device="cuda")
    wt = w.t().contiguous()
    dx = torch.empty(m, k, device="cuda")
    fl = 2.0 * m * n * k / 1e12
    a = timed(lambda: ops.train_linear_dx(dy, w, dx, mask=x))
    b = timed(lambda: ops.train_linear_dx(dy, w, dx, mask=None))
    c = timed(lambda: ops.train_linear_fwd(dy, wt, None, dx, False))
    print(f"{n:3d} -> {k:3d}: dX with mask {fl / a:6.1f} TF/s, without {fl / b:6.1f}, as a forward GEMM on W^T (no mask) {fl / c:6.1f}")
