import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stnerf_amd import ops
m = 1 << 18
for n, k in [(256, 256), (128, 128)]:
    x, w, dy = torch.randn(m, k, device="cuda"), torch.randn(n, k, device="cuda"), torch.randn(m, n, device="cuda")
    y, dx, dw, db = torch.empty(m, n, device="cuda"), torch.empty(m, k, device="cuda"), torch.empty(n, k, device="cuda"), torch.empty(n, device="cuda")
    def timed(fn, reps=10):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
    fl = 2.0 * m * n * k / 1e12
    print(os.environ.get("STNERF_LIB", "main")[-8:], n, k, "fwd %.1f  dx %.1f  dw %.1f TF/s" % (fl / timed(lambda: ops.train_linear_fwd(x, w, db, y, True)),
          fl / timed(lambda: ops.train_linear_dx(dy, w, dx, mask=x)), fl / timed(lambda: ops.train_linear_dw(dy, x, dw, db, False))))
