#!/usr/bin/env python3
"""Benchmark of the layered ray-march render path on MI355X (driver contract: one JSON line on rank 0).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one synthetic 1080p view per GPU: device ray generation
(a1/a2) -> coarse sampler -> mask compaction -> MotionNet/SpaceNet (fp32 MFMA) -> composite + merge ->
inverse-CDF resample -> fine MotionNet/SpaceNet -> composite + merge, followed (N > 1) by the RCCL
all-gather of the rendered tiles.  Weak scaling: every rank renders its own full view of a novel-view
sweep; value = total rays of all ranks / max-over-ranks time.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): Taekwondo-shaped
scene, 2 performer layers + background, 1920x1080, 64 coarse + 64 fine samples (128/ray/layer),
USE_SPACE_TIME + USE_DEFORM_TIME, synthetic boxes, random 'trained-like' weights (no dataset or
checkpoint exists for the reference).
"""
import argparse
import json
import os
import sys
import time
import types

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from stnerf_amd import ops, synthetic as syn          # noqa: E402
from stnerf_amd.modeling import build_layered_model   # noqa: E402
from stnerf_amd.utils import layered_batchify_ray     # noqa: E402
from stnerf_amd.parallel import gather_tiles, make_row_renderer, render_view_striped  # noqa: E402

# Algorithmic work per network evaluation: 2 * MACs of every nn.Linear.
FLOP_SPACE, FLOP_SPACE_TIME, FLOP_MOTION = 924_672, 930_048, 153_344
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (spec, no sparsity)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz

WORKLOADS = {
    # name: (H, W, L, N1, N2, space_time, deform_time)
    "taekwondo-1080p-64+64": (1080, 1920, 2, 64, 64, True, True),
    "taekwondo-1080p-90+30": (1080, 1920, 2, 90, 30, True, True),
    "single-512-64+64": (512, 512, 1, 64, 64, True, False),
    "walking-1080p-L4-64+64": (1080, 1920, 4, 64, 64, False, True),
    "synthetic-4k-L8-128+64": (2160, 3840, 8, 128, 64, False, True),   # BASELINE configs[4] on one GPU (use --rays-per-launch 131072)
    "tiny-64-32+0": (64, 64, 1, 32, 0, True, False),
}


def make_cfg(L, n1, n2, st, dt):
    m = types.SimpleNamespace(BOARDER_WEIGHT=1e10, SAMPLE_METHOD="BBOX", SAME_SPACENET=False, TKERNEL_INC_RAW=True,
                              POSE_REFINEMENT=False, USE_DIR=True, USE_DEFORM_VIEW=False, USE_DEFORM_TIME=dt,
                              USE_SPACE_TIME=st, BKGD_USE_DEFORM_TIME=False, BKGD_USE_SPACE_TIME=False, DEEP_RGB=False,
                              COARSE_RAY_SAMPLING=n1, FINE_RAY_SAMPLING=n2)
    return types.SimpleNamespace(MODEL=m, DATASETS=types.SimpleNamespace(LAYER_NUM=L))


def build_scene(workload, device):
    H, W, L, n1, n2, st, dt = WORKLOADS[workload]
    model = build_layered_model(make_cfg(L, n1, n2, st, dt), camera_num=1)
    model.load_state_dict(syn.make_state_dict(L, st, dt, seed=0))
    bk, per = syn.scene_boxes(L)
    model.set_bkgd_bbox(bk)
    model.set_bboxes(per)
    return model.to(device).eval(), (H, W, L, n1, n2, st, dt)


class KernelTimer:
    """Per-launch timings from the library's own profiler (HIP events recorded on the launch stream around every
    kernel launch, stnerf_profile_begin/_end) joined with the hit masks of each stnerf_render_rays call, which
    give the number of rows a masked performer launch really processed."""

    def __init__(self):
        self.masks = []
        self._orig = None

    def start(self):
        self._orig = ops.render_rays

        def wrapped(*a, **k):
            out = self._orig(*a, **k)
            self.masks.append(out[4])
            return out
        ops.render_rays = wrapped
        ops.profile_begin()

    def stop(self):
        ops.render_rays = self._orig
        self.records = ops.profile_end()

    def summarise(self):
        out, call = {}, -1
        counts = [m.sum(0).tolist() for m in self.masks]          # hit rays per layer of every pipeline call
        for r in self.records:
            name = r["kernel"]
            if name == "sample_coarse":
                call += 1                                          # every pipeline call starts with the sampler
            if name in ("spacenet", "motionnet"):
                rays = r["n_rays"] if r["tag"] <= 0 else int(counts[call][r["tag"]])
                flop = FLOP_MOTION if name == "motionnet" else (FLOP_SPACE_TIME if r["kind"] in (1, 4) else FLOP_SPACE)
                if name == "spacenet" and r["kind"] in (3, 4):     # deep_rgb: two more 128x128 layers
                    flop += 2 * 2 * 128 * 128
                d = out.setdefault(name, dict(launches=0, ms=0.0, evals=0, flop=0))
                d["evals"] += rays * r["ns"]
                d["flop"] += rays * r["ns"] * flop
            else:                                                  # HBM-bound kernels: algorithmic bytes per ray
                d = out.setdefault(name, dict(launches=0, ms=0.0, bytes=0))
                d["bytes"] += r["bytes_per_ray"] * r["n_rays"]
            d["launches"] += 1
            d["ms"] += r["ms"]
        return out


def cpu_baseline(workload, budget_rays):
    """The CPU oracle (a restatement of the reference algorithm, oracle/stnerf_oracle.py) timed on this
    box's host cores on a bounded sample of the same workload: whole 3584-ray reference chunks taken
    from the centre rows of the view (where rays hit the performers)."""
    from oracle import stnerf_oracle as O
    H, W, L, n1, n2, st, dt = WORKLOADS[workload]
    K, T = syn.camera(H, W, 10.0)
    bk, per = syn.scene_boxes(L)
    m = O.OracleModel(layer_num=L, n_coarse=n1, n_fine=n2, params=syn.make_state_dict(L, st, dt, seed=0),
                      use_deform_time=dt, use_space_time=st, bkgd_bbox=bk, bboxes=per)
    chunk = 3584
    n = max(chunk, (budget_rays // chunk) * chunk)
    n = min(n, H * W)
    full = O.generate_rays(K, T, H, W) if H * W <= 1 << 19 else None
    if full is None:  # rows around the image centre only (generate the window analytically via the oracle on a crop)
        r0 = (H // 2) * W - n // 2
        rows0, rows1 = r0 // W, (r0 + n + W - 1) // W
        Kc = K.clone()
        Kc[1, 2] -= rows0
        part = O.generate_rays(Kc, T, rows1 - rows0, W)
        rays = part[r0 - rows0 * W: r0 - rows0 * W + n]
    else:
        r0 = max(0, (H * W - n) // 2)
        rays = full[r0:r0 + n]
    rays = torch.cat([rays, syn.frame_id_columns(rays.shape[0], L)], -1)
    torch.manual_seed(0)
    with torch.no_grad():
        O.render_chunk(m, rays[:256])  # warm the allocator / thread pool
        t0 = time.perf_counter()
        O.layered_batchify_ray(m, rays, chuncks=chunk)
        dt_s = time.perf_counter() - t0
    return dict(value=rays.shape[0] / dt_s, unit="rays/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{rays.shape[0]} rays ({rays.shape[0] // chunk} reference chunks of {chunk}) from the centre "
                       f"rows of the {W}x{H} view, oracle/stnerf_oracle.py on torch {torch.__version__} CPU fp32, "
                       f"{dt_s:.1f} s", seconds=dt_s)


def eager_gpu_baseline(workload, n_rays, device):
    """Informative only: the same restatement of the reference run op by op through eager
    PyTorch-ROCm on the MI355X (rocBLAS GEMMs, one launch per elementwise op, activations through HBM) --
    what pointing the reference's own code at the GPU would give.  Not part of the default run."""
    from oracle import stnerf_oracle as O
    H, W, L, n1, n2, st, dt = WORKLOADS[workload]
    K, T = syn.camera(H, W, 10.0)
    bk, per = syn.scene_boxes(L)
    sd = {k: v.to(device) for k, v in syn.make_state_dict(L, st, dt, seed=0).items()}
    m = O.OracleModel(layer_num=L, n_coarse=n1, n_fine=n2, params=sd, use_deform_time=dt, use_space_time=st,
                      bkgd_bbox=bk.to(device), bboxes=per.to(device))
    chunk = 3584
    n = max(chunk, (n_rays // chunk) * chunk)
    rays = ops.generate_rays(K, T, H, W, frame_ids=[1.0] + [2.5] * L, device=device)
    r0 = (H * W - n) // 2
    rays = rays[r0:r0 + n].contiguous()
    torch.manual_seed(0)
    with torch.no_grad(), torch.device(device):
        O.layered_batchify_ray(m, rays[:chunk], chuncks=chunk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        O.layered_batchify_ray(m, rays, chuncks=chunk)
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
    return dict(value=n / dt_s, unit="rays/s", kind="eager PyTorch-ROCm (oracle restatement on cuda:0)",
                sample=f"{n} rays ({n // chunk} reference chunks of {chunk}) from the centre rows, {dt_s:.2f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="taekwondo-1080p-64+64", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-baseline-rays", type=int, default=7168, help="0 disables the CPU baseline leg")
    ap.add_argument("--rays-per-launch", type=int, default=1 << 19)
    ap.add_argument("--partition", default="views", choices=["views", "stripes"],
                    help="N>1: 'views' = one whole view per GPU per step (weak scaling, the default the driver runs); "
                         "'stripes' = ONE view per step, interleaved 8-row stripes over the GPUs (strong scaling)")
    ap.add_argument("--eager-gpu-baseline-rays", type=int, default=0,
                    help=">0: also time the oracle restatement through eager PyTorch-ROCm on this GPU (informative)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16x3"],
                    help="arithmetic of the headline run: exact f32 MFMA (default) or fp32-accurate split-fp16 MFMA")
    ap.add_argument("--no-second-precision", action="store_true",
                    help="skip the extra (untimed-for-`value`) leg that measures the other precision mode")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the render path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)  # RCCL over xGMI

    from stnerf_amd import hip
    info = hip.device_info()
    model, (H, W, L, n1, n2, st, dt) = build_scene(args.workload, device)
    model.max_rays_per_launch = args.rays_per_launch
    l = L + 1
    n_rays = H * W
    frame_ids = [1.0] + [2.5] * L

    def step_striped(i):
        # strong scaling: all ranks share ONE view; rank r renders stripes r, r+N, ... of 8 image rows
        K, T = syn.camera(H, W, orbit_deg=10.0 + 1.5 * i)
        model.seed = i
        rows = make_row_renderer(model, K, T, H, W, frame_ids, device=device)
        tile = render_view_striped(rows, n_rays, 8 * W)
        return tile, [torch.ones(1, device=device)] * l   # (per-layer hit masks are not gathered in this mode)

    def step(i, gather=True):
        if args.partition == "stripes":
            return step_striped(i)
        # novel-view sweep, a new pose every step.  Weak scaling: each GPU renders one whole view per step, and all
        # ranks take the SAME camera for step i (distinct RNG streams), so the per-GPU work does not depend on N --
        # a different pose per rank would change the performer coverage and with it the work of the slowest rank.
        K, T = syn.camera(H, W, orbit_deg=10.0 + 1.5 * i)
        rays = ops.generate_rays(K, T, H, W, frame_ids=frame_ids, device=device)
        model.seed = i * world + rank
        with torch.no_grad():
            fine, coarse, fine_layers, _, masks = layered_batchify_ray(model, rays, None, None)
        tile = torch.cat(list(fine), dim=1).contiguous()      # (H*W, 5): colour, depth, acc of the final image
        if world > 1 and gather:
            tile = gather_tiles(tile, world * n_rays)          # ONE RCCL all-gather of the rendered tiles
        return tile, masks

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(precision, steps, warmup):
        """K timed steps (barrier + synchronize on both sides, MAX over ranks) in one precision mode."""
        model.set_precision(precision)
        for i in range(warmup):
            step(-1 - i)
        timer = KernelTimer()
        timer.start()
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            tile, masks = step(i)
        fence()
        elapsed = time.perf_counter() - t0
        timer.stop()
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        ksum = timer.summarise()  # this rank's launches over the timed steps
        evals = sum(d["evals"] for name, d in ksum.items() if name == "spacenet")
        if world > 1:
            ev = torch.tensor([evals], dtype=torch.float64, device=device)
            dist.all_reduce(ev)
            evals_all = float(ev.item())
        else:
            evals_all = float(evals)
        assert bool(torch.isfinite(tile).all()), "non-finite pixels in the rendered tile"
        return elapsed, ksum, evals, evals_all, tile, masks

    elapsed, ksum, evals, evals_all, tile, masks = measure(args.precision, args.steps, args.warmup)
    other = None
    if not args.no_second_precision:
        op = "fp16x3" if args.precision == "fp32" else "fp32"
        o_el, o_ks, o_ev, o_eva, o_tile, _ = measure(op, max(1, min(args.steps, 2)), 1)
        # same poses / seeds as the headline run's first steps: image agreement between the two arithmetic modes
        other = dict(precision=op, steps=max(1, min(args.steps, 2)), elapsed=o_el, ksum=o_ks, evals_all=o_eva)

    if rank == 0:
        sp = ksum["spacenet"]
        achieved = sp["flop"] / (sp["ms"] * 1e-3) / 1e12
        if args.precision == "fp16x3":   # executed MFMA work is 3 fp16 terms per algorithmic product
            achieved, peak_used = 3.0 * achieved, PEAK_F16_MFMA_TFLOPS
        else:
            peak_used = PEAK_F32_MFMA_TFLOPS
        # HBM traffic cannot be counted inside this process: it comes from the committed rocprofv3 PMC passes
        # of the same command (profiles/), per launch, with the gfx950 FETCH_SIZE correction applied.
        traffic, traffic_src = None, None
        pmc_path = os.path.join(REPO, "profiles", "r01_pmc_spacenet_traffic.json")
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            if pmc.get("workload") == args.workload:
                traffic, traffic_src = pmc["hbm_bytes_per_launch"], "profiles/r01_pmc_spacenet_traffic.json"
        rec = {
            "metric": "rendered rays/s (and ray-samples/s) per GPU, 1080p x 128-sample layered render",
            "value": (1 if args.partition == "stripes" else world) * n_rays * args.steps / elapsed,
            "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong" if args.partition == "stripes" else "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f32-accurate products as 3 fp16 MFMA terms (22-bit split operands), f32 accumulate",
            "data": "synthetic",
            "config": {"workload": args.workload, "precision": args.precision, "height": H, "width": W, "performer_layers": L,
                       "coarse_samples": n1, "fine_samples": n2, "use_space_time": st, "use_deform_time": dt,
                       "rays_per_gpu_per_step": n_rays, "rays_per_launch": args.rays_per_launch,
                       "weights": "random, density head scaled (synthetic.make_state_dict seed 0)",
                       "parallelism": (f"1 view per step in interleaved 8-row stripes over {world} GPUs, one RCCL all-gather per step"
                                       if args.partition == "stripes" else
                                       f"ray tiles: 1 view per GPU per step x {world} GPUs (same camera, own RNG stream), "
                                       "one RCCL all-gather of the rendered tiles per step")},
            "ray_samples_per_s": evals_all / elapsed,
            "ray_samples_per_step_per_gpu": evals / args.steps,
            "mask_fraction": [float(m.float().mean()) for m in masks],
            "roofline": {"kernel": "stnerf::spacenet_kernel (fused PE + 9-layer MLP, v_mfma_f32_32x32x2_f32)",
                         "bound": "mfma", "achieved": achieved, "peak": peak_used, "unit": "TFLOP/s",
                         "frac": achieved / peak_used, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": 28 * sp["evals"] / sp["launches"],
                         "launches": sp["launches"], "avg_launch_ms": sp["ms"] / sp["launches"],
                         "algorithmic_flop_per_launch": sp["flop"] / sp["launches"],
                         "note": "algorithmic FLOPs = network evaluations x 924,672 (930,048 with time), "
                                 "HIP events recorded by the library on the launch stream around every launch of the timed steps"},
            "kernels": {k: {"launches": d["launches"], "ms_per_step": d["ms"] / args.steps,
                            "tflops": d["flop"] / (d["ms"] * 1e-3) / 1e12} for k, d in ksum.items() if "flop" in d},
            "hbm_kernels": {k: {"launches": d["launches"], "ms_per_step": d["ms"] / args.steps,
                                "algorithmic_GBps": d["bytes"] / (d["ms"] * 1e-3) / 1e9, "peak_GBps": 8000.0,
                                "frac": d["bytes"] / (d["ms"] * 1e-3) / 8e12} for k, d in ksum.items() if "bytes" in d},
            "device": info,
        }
        if other is not None:
            osp = other["ksum"]["spacenet"]
            o_ach = osp["flop"] / (osp["ms"] * 1e-3) / 1e12
            peak_o = PEAK_F16_MFMA_TFLOPS if other["precision"] == "fp16x3" else PEAK_F32_MFMA_TFLOPS
            mult = 3.0 if other["precision"] == "fp16x3" else 1.0
            rec["other_precision"] = {
                "precision": other["precision"],
                "note": "same workload and poses, measured after the headline run; fp16x3 = every product a*b evaluated as "
                        "ah*bh + ah*bl + al*bh on the fp16 MFMA pipe with f32 accumulation: passes the same parity tests "
                        "and tolerances as the exact-f32 kernels (tests/test_gpu_f16x3.py)",
                "value": (1 if args.partition == "stripes" else world) * n_rays * other["steps"] / other["elapsed"], "unit": "rays/s",
                "ms_per_step": 1e3 * other["elapsed"] / other["steps"], "steps": other["steps"],
                "ray_samples_per_s": other["evals_all"] / other["elapsed"],
                "roofline": {"bound": "mfma", "algorithmic_tflops": o_ach, "executed_mfma_tflops": mult * o_ach,
                             "peak": peak_o, "unit": "TFLOP/s", "frac": mult * o_ach / peak_o},
                "kernels": {k: {"launches": d["launches"], "ms_per_step": d["ms"] / other["steps"],
                                "algorithmic_tflops": d["flop"] / (d["ms"] * 1e-3) / 1e12}
                            for k, d in other["ksum"].items() if "flop" in d},
            }
        if world == 1 and args.eager_gpu_baseline_rays > 0:
            rec["eager_gpu_baseline"] = eager_gpu_baseline(args.workload, args.eager_gpu_baseline_rays, device)
        if world == 1 and args.cpu_baseline_rays > 0:
            rec["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_baseline_rays)
        else:
            rec["cpu_baseline"] = None
        print(json.dumps(rec))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
