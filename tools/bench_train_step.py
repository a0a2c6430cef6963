#!/usr/bin/env python3
"""One iteration of the reference trainer's inner loop (engine/layered_trainer.py:186-283: model.train(), forward, MSE of both mixed
colours, loss.backward(), Adam step) on the MI355X at the reference's batch size (SOLVER.BUNCH = 4096 rays, config/defaults.py:131):
ms per iteration, rays/s (the figure the trainer logs, :307-309) and where the GPU time goes.  `python tools/bench_train_step.py
[--rays 4096] [--workload taekwondo-1080p-64+64] [--iters 5]`."""
import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench                                             # noqa: E402  (scene construction)
from stnerf_amd import ops, synthetic as syn             # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--workload", default="taekwondo-1080p-64+64")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--fused", type=int, default=1)
ap.add_argument("--eager", action="store_true", help="also time the same iteration through eager PyTorch-ROCm on this GPU (the oracle "
                "restatement of the reference's modules under torch.autograd: what pointing the reference's own code at the GPU gives)")
args = ap.parse_args()
from stnerf_amd.modeling import autograd as A            # noqa: E402
A.FUSED_BACKWARD = bool(args.fused)
dev = torch.device("cuda")
model, (H, W, L, n1, n2, st, dt) = bench.build_scene(args.workload, dev)
model.train()
K, T = syn.camera(H, W, 10.0)
g = torch.Generator().manual_seed(0)
# training-style rays: 7 columns, one integer frame id per ray (data/datasets/ray_dataset.py), drawn from all over the view
all_rays = ops.generate_rays(K, T, H, W, frame_ids=[1.0], device=dev)[:, :6]
pick = torch.randperm(H * W, generator=g)[:args.rays].to(dev)
rays = torch.cat([all_rays[pick], torch.randint(1, 4, (args.rays, 1), generator=g).float().to(dev)], 1).contiguous()
rgbs = torch.rand(args.rays, 3, generator=g).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=4e-4, betas=(0.9, 0.999))
mse = torch.nn.MSELoss()


def iteration():
    opt.zero_grad()
    stage2, stage1, _, _, masks = model(rays, None, None, False)
    loss = mse(stage1[0], rgbs) + mse(stage2[0], rgbs)
    loss.backward()
    opt.step()
    return loss, masks


loss, masks = iteration()
torch.cuda.synchronize()
hits = [float(m.float().mean()) for m in masks]
evals = args.rays * sum(hits) * (2 * n1 + n2)
each = []
import gc                                                # noqa: E402
gc_log = []
gc.callbacks.append(lambda phase, info: gc_log.append((phase, info.get("generation"), time.perf_counter())))
for it in range(args.iters):
    before = torch.cuda.memory_stats()
    n_gc = len(gc_log)
    t0 = time.perf_counter()
    loss, _ = iteration()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    each.append(time.perf_counter() - t0)
    if each[-1] > 1.5 * min(each):                       # an outlier: where did it go?
        after = torch.cuda.memory_stats()
        gcs = [(g, round(1e3 * (e[2] - s_[2]), 1)) for s_, e in zip(gc_log[n_gc::2], gc_log[n_gc + 1::2]) for g in [s_[1]]]
        print(f"  outlier iteration {it}: {1e3 * each[-1]:.1f} ms, host part {1e3 * (t1 - t0):.1f} ms; hipMallocs {after['num_device_alloc'] - before['num_device_alloc']}, "
              f"hipFrees {after['num_device_free'] - before['num_device_free']}, alloc retries {after['num_alloc_retries'] - before['num_alloc_retries']}; "
              f"python gc (generation, ms): {gcs}")
dt_s = sorted(each)[len(each) // 2]                       # the median iteration (the first ones still grow the allocator's pools)
print("  iterations, ms:", " ".join(f"{1e3 * t:.1f}" for t in each))
flop = 3 * evals * ((bench.FLOP_SPACE_TIME if st else bench.FLOP_SPACE) + (bench.FLOP_MOTION if dt else 0) * (sum(hits[1:]) / max(sum(hits), 1e-9)))
print(f"{args.workload}: {args.rays} rays per iteration (hit fractions {[round(h, 3) for h in hits]}), {'fused' if args.fused else 'per-layer'} backward: "
      f"{1e3 * dt_s:.1f} ms per iteration = {args.rays / dt_s:.0f} rays/s, {evals / dt_s / 1e6:.1f} M network evaluations/s trained, "
      f"~{flop / dt_s / 1e12:.1f} TF/s of forward + dX + dW work; loss {float(loss):.5f}")

if args.eager:
    # Informative baseline (like bench.py's eager_gpu_baseline): the reference's algorithm op by op through ATen / rocBLAS with
    # torch.autograd, same weights, same rays, same loss, same optimiser.  (oracle/ is test infrastructure: it is the thing measured
    # AGAINST here, never part of the product path.)
    from oracle import stnerf_oracle as O
    bk, per = syn.scene_boxes(L)
    sd = {k: v.to(dev).requires_grad_(True) for k, v in syn.make_state_dict(L, st, dt, seed=0).items()}
    m = O.OracleModel(layer_num=L, n_coarse=n1, n_fine=n2, params=sd, use_deform_time=dt, use_space_time=st, bkgd_bbox=bk.to(dev),
                      bboxes=per.to(dev))
    opt_e = torch.optim.Adam(list(sd.values()), lr=4e-4, betas=(0.9, 0.999))

    def eager_iteration():
        opt_e.zero_grad()
        with torch.device(dev):
            out = O.render_chunk(m, rays)
        loss = mse(out[1][0], rgbs) + mse(out[0][0], rgbs)
        loss.backward()
        opt_e.step()
        return loss

    eager_iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(max(2, args.iters // 2)):
        le = eager_iteration()
    torch.cuda.synchronize()
    de = (time.perf_counter() - t0) / max(2, args.iters // 2)
    print(f"  eager PyTorch-ROCm on the same GPU (oracle restatement, torch.autograd, rocBLAS): {1e3 * de:.1f} ms per iteration = {args.rays / de:.0f} rays/s "
          f"({de / dt_s:.1f} x this library's iteration); peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB; loss {float(le):.5f}")
