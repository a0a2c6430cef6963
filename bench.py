#!/usr/bin/env python3
"""Benchmark of the layered ray-march render path on MI355X.  Driver contract: the LAST stdout line of rank 0 is the record,
< 3 KB (tests/test_bench_record.py); the line before it is a digest of the detail, whose full form goes to bench_detail.json.

    python bench.py --gpus N --steps K --warmup W          # N > 1: launches its own N ranks (torch.distributed.run, 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over ONE synthetic 1080p view: device ray generation (a1/a2) -> coarse sampler ->
mask compaction -> MotionNet/SpaceNet (split-bf16 MFMA, the library default; the exact f32 MFMA arithmetic is timed as a
short cross-check leg of <= 5 steps) -> composite + merge -> inverse-CDF resample -> fine MotionNet/SpaceNet -> composite + merge,
through the call surface a user of the reference calls: ``stnerf_amd.parallel.render_view`` = what ``render_pose`` runs
(device ray generation + ``layered_batchify_ray``).

The SAME function is the step at every N.  N = 1: the whole view on the one GPU.  N > 1 (the split BASELINE.json
configs[3]/[4] describe): the view is cut into interleaved single-row stripes over the N ranks (the performers cover only
part of the picture; contiguous tiles would differ ~2x in cost), every rank generates and renders its 1/N of the rows
as one launch sequence (striped ray window), one RCCL all-gather rebuilds the WHOLE 5-tuple (mixed + per-layer colour /
depth / acc, fine and coarse, hit masks as one bit column: 11 + 10 l floats per ray; --gather fine / final move less) on
every rank.  `scaling` = "strong" at every N:
value = rays of the view x steps / max-over-ranks time.  `--partition views` (one whole view per rank per step, weak
scaling) stays available as an explicit alternative.

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
from stnerf_amd import parallel                         # noqa: E402
from stnerf_amd.parallel import gather_tiles, render_view  # noqa: E402

PEAK_HBM_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec
def _latest_profile(suffix):
    """The newest round's profiles/rNN_<suffix> (tools/summarise.py writes one per round)."""
    import glob
    found = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_" + suffix)))
    return found[-1] if found else os.path.join(REPO, "profiles", "r00_" + suffix)


def _load_profile(path):
    """A committed profile file, or {} when it is missing or unreadable: side information never costs the run its record."""
    try:
        with open(path) as f:
            d = json.load(f)
        return d if isinstance(d, dict) else {}
    except (OSError, ValueError):
        return {}


MEASURED_HBM_JSON = _latest_profile("hbm_copy_microbench.json")   # tools/micro/hbm_copy on the GPU box
PMC_TRAFFIC_JSON = _latest_profile("pmc_hbm_traffic.json")         # tools/summarise.py (pose 0 of the sweep; carries the build it ran on)

# Algorithmic work per network evaluation: 2 * MACs of every nn.Linear.
FLOP_SPACE, FLOP_SPACE_TIME, FLOP_MOTION = 924_672, 930_048, 153_344
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (spec, no sparsity)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz
# What a register-only MFMA loop on all 256 CUs sustains over seconds on this part (tools/micro/mfma_rate.hip,
# profiles/r01_mfma_rate_microbench.md): the 16-bit MFMA pipe is POWER-bound as soon as the operands carry real (random) data --
# 1.67 PF/s at ~1.6 - 1.7 GHz and 1.3 kW, not the 2.5 PF/s of the 2.4 GHz spec (measured with f16 operands; the bf16 instruction
# runs on the same datapath at the same rate); the f32 MFMA pipe is issue-bound at 154 TF/s whatever the data.
SUSTAINED_16BIT_MFMA_TFLOPS, SUSTAINED_F32_MFMA_TFLOPS = 1670.0, 154.0

WORKLOADS = {
    # name: (H, W, L, N1, N2, space_time, deform_time)
    "taekwondo-1080p-64+64": (1080, 1920, 2, 64, 64, True, True),
    "taekwondo-1080p-90+30": (1080, 1920, 2, 90, 30, True, True),
    "single-512-64+64": (512, 512, 1, 64, 64, True, False),
    "walking-1080p-L4-64+64": (1080, 1920, 4, 64, 64, False, True),
    "synthetic-4k-L8-128+64": (2160, 3840, 8, 128, 64, False, True),   # BASELINE configs[4] on one GPU (use --rays-per-launch 131072)
    "tiny-64-32+0": (64, 64, 1, 32, 0, True, False),
    "taekwondo-192x256-32+32": (192, 256, 2, 32, 32, True, True),     # the C3 scene small enough for the N-rank bitwise tests
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

    def __init__(self, n1=64, layer_flops=None):
        self.masks = []
        self._orig = None
        self.n1 = n1
        self.layer_flops = layer_flops or []     # per layer: FLOPs of one evaluation in the fused stage kernel

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
            if name == "mlp_stage":                                # one persistent launch = every layer of a stage
                hits = [r["n_rays"]] + [int(c) for c in counts[call][1:]]
                d = out.setdefault(name, dict(launches=0, ms=0.0, evals=0, flop=0))
                d["evals"] += sum(hits) * r["ns"]
                d["flop"] += sum(h * r["ns"] * f for h, f in zip(hits, self.layer_flops))
            elif name in ("spacenet", "motionnet"):
                rays = r["n_rays"] if r["tag"] <= 0 else int(counts[call][r["tag"]])
                flop = FLOP_MOTION if name == "motionnet" else (FLOP_SPACE_TIME if r["kind"] in (1, 4) else FLOP_SPACE)
                if name == "spacenet" and r["kind"] in (3, 4):     # deep_rgb: two more 128x128 layers
                    flop += 2 * 2 * 128 * 128
                d = out.setdefault(name, dict(launches=0, ms=0.0, evals=0, flop=0))
                d["evals"] += rays * r["ns"]
                d["flop"] += rays * r["ns"] * flop
            else:                                                  # HBM-bound kernels: algorithmic bytes per ray
                d = out.setdefault(name, dict(launches=0, ms=0.0, bytes=0, bytes_dense=0))
                d["bytes_dense"] += r["bytes_per_ray"] * r["n_rays"]
                if name == "composite":
                    # bytes the kernel HAS to move: t + raw (20 B/sample) only of the layers a ray hits (the background:
                    # every ray), the mask, the outputs, and the per-layer weights of the coarse stage (all layers: the
                    # resampler reads them).  The library's own figure (bytes_dense) charges every layer.
                    l = len(counts[call])
                    hits = r["n_rays"] + sum(int(c) for c in counts[call][1:])
                    dense = r["bytes_per_ray"]
                    weights = dense - (20 * l * r["ns"] + l + 20 * (l + 1))       # 4*l*S if weights were written
                    d["bytes"] += 20 * r["ns"] * hits + (l + 20 * (l + 1) + weights) * r["n_rays"]
                elif name == "resample":
                    # 8 B per coarse sample read and 16 B per fine sample written for hit layers; a missed (ray, layer)
                    # pair is skipped since round 4 (the sampler's hint: nobody reads its constant fill)
                    l = len(counts[call])
                    hits = r["n_rays"] + sum(int(c) for c in counts[call][1:])
                    n1 = self.n1
                    d["bytes"] += (8 * n1 + 16 * r["ns"]) * hits + (24 + l) * r["n_rays"]
                else:
                    d["bytes"] += r["bytes_per_ray"] * r["n_rays"]
            d["launches"] += 1
            d["ms"] += r["ms"]
        return out


def _oracle_ray_window(O, K, T, H, W, first, n):
    """Rays [first, first + n) of the H x W view from the oracle's ray generator (on the image rows they lie in)."""
    rows0, rows1 = first // W, (first + n + W - 1) // W
    Kc = K.clone()
    Kc[1, 2] -= rows0                                    # v = row - rows0 in the crop: same K^-1 [u, row, 1]
    part = O.generate_rays(Kc, T, rows1 - rows0, W)
    return part[first - rows0 * W: first - rows0 * W + n]


def cpu_baseline(workload, budget_rays, threads=0):
    """The CPU oracle (a restatement of the reference algorithm, oracle/stnerf_oracle.py) timed on this box's host
    cores on a bounded sample of the same workload: whole 3584-ray reference chunks spread
    evenly over the image height (the performer boxes span it: every chunk crosses them), each timed on its own; `value` = rays of all chunks / their total time = the whole-frame rate they extrapolate to."""
    import platform
    from oracle import stnerf_oracle as O
    # a fair CPU figure needs a sensible thread count: on the 2 x 64-core host of the MI355X box the chunk-sized GEMMs of
    # the reference peak at 32 threads (555 rays/s) and lose more than half of that at torch's default of 128
    # (tools/cpu_threads_probe.py, profiles/r02_cpu_threads_probe.md)
    torch.set_num_threads(threads if threads > 0 else min(32, os.cpu_count() or 1))
    H, W, L, n1, n2, st, dt = WORKLOADS[workload]
    K, T = syn.camera(H, W, 10.0)
    bk, per = syn.scene_boxes(L)
    m = O.OracleModel(layer_num=L, n_coarse=n1, n_fine=n2, params=syn.make_state_dict(L, st, dt, seed=0),
                      use_deform_time=dt, use_space_time=st, bkgd_bbox=bk, bboxes=per)
    chunk = 3584
    n_chunks = max(1, min(budget_rays // chunk, (H * W) // chunk))
    torch.manual_seed(0)
    per_chunk, evals = [], 0
    with torch.no_grad():
        O.render_chunk(m, torch.cat([_oracle_ray_window(O, K, T, H, W, (H // 2) * W, 256),
                                     syn.frame_id_columns(256, L)], -1))            # warm the allocator / thread pool
        for i in range(n_chunks):
            first = min(H * W - chunk, max(0, int((i + 0.5) / n_chunks * H * W) - chunk // 2))
            rays = torch.cat([_oracle_ray_window(O, K, T, H, W, first, chunk), syn.frame_id_columns(chunk, L)], -1)
            t0 = time.perf_counter()
            out = O.layered_batchify_ray(m, rays, chuncks=chunk)
            sec = time.perf_counter() - t0
            hits = chunk + sum(int(mk.sum()) for mk in out[4][1:])
            evals += hits * (2 * n1 + n2)
            per_chunk.append(dict(first_row=first // W, seconds=round(sec, 3), rays_per_s=round(chunk / sec, 1),
                                  performer_hit_fraction=round((hits - chunk) / (chunk * max(L, 1)), 3)))
    total = sum(c["seconds"] for c in per_chunk)
    rate = n_chunks * chunk / total
    cpu = platform.processor() or ""
    try:
        cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except Exception:
        pass
    return dict(value=rate, unit="rays/s", cores=torch.get_num_threads(), kind="port",
                sample_short=f"{n_chunks} reference chunks of {chunk} rays spread over the {W}x{H} view, oracle/stnerf_oracle.py, "
                             f"torch {torch.__version__} CPU fp32",
                sample=f"{n_chunks} reference chunks of {chunk} rays spread evenly over the rows of the {W}x{H} view "
                       f"(the performer boxes span the image height: every chunk crosses them, see chunks[].performer_hit_fraction), "
                       f"oracle/stnerf_oracle.py on torch "
                       f"{torch.__version__} CPU fp32, {total:.1f} s",
                seconds=total, ray_samples_per_s=evals / total, extrapolated_frame_seconds=H * W / rate,
                host=dict(nproc=os.cpu_count(), torch_threads=torch.get_num_threads(), cpu=cpu), chunks=per_chunk)


def psnr_vs_reference(precision, device):
    """PSNR parity of the production (device Philox) RNG mode with the reference, in the arithmetic of the headline leg:
    tests/golden/psnr_view.npz holds the REFERENCE rendered with two torch seeds on a 128 x 128 view of this scene
    (64+64, seed-0 weights; written by tests/golden/make_golden.py from /root/reference).  PSNR(reference B, reference A)
    is its own run-to-run spread; the HIP render must land on it."""
    import numpy as np
    path = os.path.join(REPO, "tests", "golden", "psnr_view.npz")
    if not os.path.exists(path):
        return None
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    A, B = torch.from_numpy(z["color_a"]).float(), torch.from_numpy(z["color_b"]).float()
    model = build_layered_model(make_cfg(meta["L"], meta["n1"], meta["n2"], True, True), camera_num=1)
    model.load_state_dict(syn.make_state_dict(meta["L"], True, True, seed=meta["weight_seed"]))
    bk, per = syn.scene_boxes(meta["L"])
    model.set_bkgd_bbox(bk)
    model.set_bboxes(per)
    model = model.to(device).eval()
    model.shard_views = False
    K, T = syn.camera(meta["h"], meta["w"], meta["orbit"])
    rays = ops.generate_rays(K, T, meta["h"], meta["w"], frame_ids=[1.0] + [meta["frame"]] * meta["L"], device=device)
    psnr = lambda a, b: float(-10 * torch.log10(torch.mean((a - b) ** 2)))
    out = {"reference_seed_b_vs_seed_a_dB": psnr(B, A), "view": f"{meta['w']}x{meta['h']}, L={meta['L']}, {meta['n1']}+{meta['n2']}",
           "fixture": "tests/golden/psnr_view.npz"}
    for prec in dict.fromkeys([precision, "bf16x3", "fp32"]):
        model.set_precision(prec)
        model.seed = 5
        with torch.no_grad():
            img = layered_batchify_ray(model, rays, None, None)[0][0].cpu()
        out[f"hip_device_rng_{prec}_vs_reference_seed_a_dB"] = psnr(img, A)
    return out


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


def emulate_share(model, dims, rank_counts, stripe_rows, device, steps):
    """An EMULATION on this one GPU of what one rank of an N-GPU job computes -- not a scaling measurement: rank 0's
    interleaved row stripes of the view (parallel.render_view_share = the code render_view runs before its all-gather, with
    a made-up rank (0, N), no process group) timed for each N over the poses of the first timed steps.  t_share(N) against
    t(1) / N exposes what does not shrink with the share: the persistent stage kernel's tail at 1/N of the items, per-view
    host work, stripe imbalance.  predicted_compute_efficiency = t(1) / (N x max-over-emulated-ranks t_share(N)); the
    all-gather (payload reported) is not in it."""
    H, W, L, n1, n2, st, dt = dims
    frame_ids = [1.0] + [2.5] * L
    out = []
    for N in [1] + [n for n in rank_counts if n > 1]:
        per_rank_ms = []
        for r in sorted({0, N // 2, N - 1}):                  # first, middle, last rank: stripe imbalance
            for i in (-1,) + tuple(range(steps)):
                if i == 0:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                K, T = syn.camera(H, W, orbit_deg=10.0 + 1.5 * i)
                model.seed = i
                packed = parallel.render_view_share(model, K, T, H, W, frame_ids, r, N, stripe_rows=stripe_rows, device=device)
            torch.cuda.synchronize()
            per_rank_ms.append((r, 1e3 * (time.perf_counter() - t0) / steps))
        worst = max(ms for _, ms in per_rank_ms)
        if N == 1:
            t1_ms = worst                                     # the same function, the same poses, the whole view
            continue
        out.append({"ranks": N, "t_share_ms": {str(r): ms for r, ms in per_rank_ms}, "t_share_max_ms": worst,
                    "n_times_t_share_over_t1": N * worst / t1_ms, "predicted_compute_efficiency": t1_ms / (N * worst),
                    "rays_per_rank": packed.shape[0],
                    "gather_payload_bytes_per_rank": {mode: packed.shape[0] * 4 * parallel.packed_width(L + 1, mode)
                                                      for mode in parallel.GATHER_MODES}})
    return {"note": "EMULATION on one GPU, not a scaling measurement: rank r's stripes of the view rendered alone (no process group, "
                    "no collective); t1_ms = the same function with (rank, N) = (0, 1) over the same poses",
            "t1_ms": t1_ms, "steps": steps, "stripe_rows": stripe_rows, "shares": out}


def fail(message, code=2):
    """A run that cannot measure anything still ends with ONE parsable JSON line as the last stdout line (rank 0 only), and a
    non-zero exit status."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps({"error": message, "metric": "rendered rays/s (and ray-samples/s) per GPU, 1080p x 128-sample layered render",
                          "value": None}))
        sys.stdout.flush()
    sys.stderr.write("bench.py: " + message + "\n")
    sys.exit(code)


def _self_launch(args):
    """`python bench.py --gpus N` (N > 1) without torch.distributed.run's environment: become the launcher of N ranks of
    this very command (one process per GPU, rendezvous on 127.0.0.1, a free port)."""
    import socket
    if not args.debug_single_device and torch.cuda.device_count() < args.gpus:
        fail(f"--gpus {args.gpus} but this node exposes {torch.cuda.device_count()} GPU(s): one process per GPU "
             "(check HIP_VISIBLE_DEVICES / the compute partition mode)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


PRECISION_NOTES = {
    "bf16x3": "the library default: every fp32 operand = three bf16 pieces (8+8+8 significand bits: exact for finite values in "
              "bf16's exponent range = fp32's), a*b = its six leading cross terms on v_mfma_f32_32x32x16_bf16, a0*b0 and the "
              "five small terms in separate f32 accumulators, heads in fp64: closer to an fp64 evaluation than the fp32 CPU "
              "chain on every layer (tests/test_gpu_stage.py)",
    "fp32": "exact f32 MFMA (v_mfma_f32_32x32x2_f32), model.set_precision('fp32')"}
DTYPE_NOTES = {"fp32": "f32",
               "bf16x3": "f32 (operands as 3 bf16 pieces each: 24-bit significand, f32 exponent range; 6 bf16 MFMA terms per product, f32 accumulate)"}
EXECUTED_TERMS = {"fp32": 1.0, "bf16x3": 6.0}     # MFMA products executed per algorithmic product
STAGE_KERNEL = {
    "bf16x3": "stnerf::mlp_bf16x3_stage_kernel (one persistent launch per stage: MotionNet + SpaceNet of every layer, a wave owns 32 "
              "samples and keeps their activations in registers as three bf16 planes, weights through an LDS-DMA ring, "
              "v_mfma_f32_32x32x16_bf16; achieved = algorithmic rate, executed MFMA rate = 6 x that)",
    "fp32": "stnerf::mlp_wave_stage_kernel (one persistent launch per stage: MotionNet + SpaceNet of every layer, a wave owns 32 "
            "samples and keeps their activations in registers, v_mfma_f32_32x32x2_f32)"}


# ----------------------------------------------------------------------------------------------------------------------
# The record.  The driver keeps the last 8 KB of stdout and parses the LAST line: that line is the compact contract record
# (`final`, < 3 KB, tests/test_bench_record.py); everything else -- both arithmetics' full roofline blocks, the HBM-side
# kernels, the other configs' legs, the CPU leg's per-chunk list, notes -- is `detail`: written to bench_detail.json beside
# this script (and to gpurun_out/ when that exists, so that a gpurun call brings it home) and printed, trimmed, on the line
# BEFORE the final one.
FINAL_LINE_LIMIT = 3072
DETAIL_LINE_LIMIT = 4096
DETAIL_PATH = os.path.join(REPO, "bench_detail.json")
DTYPE_SHORT = {"fp32": "f32", "bf16x3": "f32 (operands split into 3 bf16 pieces, 6 bf16 MFMA terms per product, f32 accumulate)"}
KERNEL_SHORT = {"fp32": "stnerf::mlp_wave_stage_kernel", "bf16x3": "stnerf::mlp_bf16x3_stage_kernel"}


def _stage_of(leg):
    return leg["ksum"]["mlp_stage"] if "mlp_stage" in leg["ksum"] else leg["ksum"]["spacenet"]


def pmc_matches_loaded_library(pmc):
    """Counter bytes are a property of the kernels they were measured on: the committed PMC file names the build
    (`fatbin_sha256` of libstnerf_hip.so's .hip_fatbin, `rev`); anything else loaded -> no traffic figure."""
    from stnerf_amd import hip
    want = pmc.get("fatbin_sha256")
    return bool(want) and want == hip.fatbin_sha256()


def roofline_of(leg, workload, ctx):
    """MFMA roofline of the stage kernel of one leg, as SURVEY.md section 8(d) defines it: achieved = ALGORITHMIC FLOPs (network
    evaluations x FLOPs per evaluation of the reference's arithmetic) / the kernel's time; peak = the ceiling for the arithmetic that
    is delivered -- fp32-faithful products: the f32 MFMA peak for the exact kernel, the dense bf16 MFMA peak / 6 for bf16x3 (six bf16
    terms per product is the minimum for 24-bit significands on an 8-bit-significand MFMA).  The EXECUTED MFMA rate (what the matrix
    pipe does) stays beside it under names that say so."""
    prec = leg["precision"]
    pmc, world = ctx["pmc"], ctx["world"]
    sp = _stage_of(leg)
    alg = sp["flop"] / (sp["ms"] * 1e-3) / 1e12
    terms = EXECUTED_TERMS[prec]
    executed = terms * alg
    instr_peak = PEAK_F32_MFMA_TFLOPS if prec == "fp32" else PEAK_BF16_MFMA_TFLOPS
    peak = instr_peak / terms
    sustained = SUSTAINED_F32_MFMA_TFLOPS if prec == "fp32" else SUSTAINED_16BIT_MFMA_TFLOPS
    # HBM traffic cannot be counted inside this process: it comes from the committed rocprofv3 PMC passes of the same command
    # (profiles/), per launch, with the gfx950 FETCH_SIZE correction applied -- and only when those passes ran on the kernels
    # that are loaded now (the .hip_fatbin digest recorded beside them), on this workload, on one GPU.
    pk = pmc.get({"bf16x3": "kernels_bf16x3", "fp32": "kernels"}.get(prec, "-"), {})
    dom = "mlp_stage" if "mlp_stage" in leg["ksum"] else "spacenet"
    same_build = ctx.get("pmc_same_build")
    if same_build is None:
        same_build = pmc_matches_loaded_library(pmc)
    usable = pmc.get("workload") == workload and world == 1 and dom in pk
    traffic = pk[dom]["hbm_bytes_per_launch"] if (usable and same_build) else None
    return {"kernel": STAGE_KERNEL[prec] if dom == "mlp_stage" else "stnerf::spacenet_kernel (fused PE + 9-layer MLP)",
            "bound": "mfma", "achieved": alg, "peak": peak, "unit": "TFLOP/s", "frac": alg / peak,
            "peak_note": (f"{instr_peak:g} dense bf16 MFMA / {terms:g} bf16x3 terms per fp32-faithful product = {peak:.1f}" if prec == "bf16x3"
                          else f"{instr_peak:g} v_mfma_f32_32x32x2_f32"),
            "algorithmic_tflops": alg, "executed_mfma_tflops": executed, "mfma_terms_per_product": terms,
            "executed_frac_of_instruction_peak": executed / instr_peak, "instruction_peak": instr_peak,
            "measured_sustained_peak": sustained, "frac_of_measured_sustained_peak": executed / sustained,
            "measured_sustained_peak_source": "profiles/r01_mfma_rate_microbench.md: register-only MFMA loop, all 256 CUs, ~3 s, random operands "
                                              "(16-bit MFMA: power-bound at 1.3 kW; f32 MFMA: issue-bound)",
            "traffic": traffic, "traffic_rev": pmc.get("rev") if usable else None, "traffic_pose": pmc.get("pose_short") if usable else None,
            "traffic_same_build": bool(same_build) if usable else None,
            "traffic_source": ("profiles/" + os.path.basename(PMC_TRAFFIC_JSON) + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over ONE "
                               "step of this workload at pose 0 of the sweep (the timed steps sweep the orbit: +- 10 % evaluations), on the build "
                               "whose .hip_fatbin digest the file records; null when another build is loaded") if usable else None,
            "algorithmic_bytes_per_launch": 28 * sp["evals"] / sp["launches"],   # 12 B point in + 16 B raw out per SpaceNet evaluation
            "launches": sp["launches"], "avg_launch_ms": sp["ms"] / sp["launches"],
            "algorithmic_flop_per_launch": sp["flop"] / sp["launches"],
            "note": "algorithmic FLOPs = network evaluations x 924,672 (930,048 with time; + 153,344 per MotionNet evaluation in the fused "
                    "stage kernel); executed = algorithmic x MFMA terms per product (1 for f32, 6 for bf16x3); HIP events recorded by the "
                    "library on the launch stream around every launch of the timed steps (rank 0)"}


def hbm_entry(leg, k, d, ctx):
    pmc, measured = ctx["pmc"], ctx["hbm_microbench"]
    hbm_meas = {"composite": measured.get("read_GBps"), "resample": measured.get("copy_GBps"),
                "sample_coarse": measured.get("write_GBps")}
    sec = d["ms"] * 1e-3
    e = {"launches": d["launches"], "ms_per_step": d["ms"] / leg["steps"],
         "algorithmic_GBps": d["bytes"] / sec / 1e9, "peak_GBps": PEAK_HBM_GBPS, "frac": d["bytes"] / sec / (PEAK_HBM_GBPS * 1e9),
         "algorithmic_bytes_per_step": d["bytes"] / leg["steps"],
         "dense_bytes_per_step": d["bytes_dense"] / leg["steps"]}
    if hbm_meas.get(k):
        e["measured_peak_GBps"] = hbm_meas[k]
        e["frac_of_measured_peak"] = d["bytes"] / sec / (hbm_meas[k] * 1e9)
    pk = pmc.get("kernels_bf16x3" if leg["precision"] == "bf16x3" else "kernels", {})
    if (pmc.get("workload") == ctx["workload"] and ctx["world"] == 1 and k in pk
            and leg.get("workload", ctx["workload"]) == ctx["workload"]):
        cb = pk[k]["hbm_bytes_per_step"]
        e["counter_bytes_per_step"] = cb
        e["counter_GBps"] = cb / (d["ms"] / leg["steps"] * 1e-3) / 1e9
        e["counter_over_algorithmic"] = cb / (d["bytes"] / leg["steps"])
        e["counter_source"] = "profiles/" + os.path.basename(PMC_TRAFFIC_JSON)
    return e


def leg_record(leg, workload, ctx):
    per_rank = leg["per_rank_compute_s"]
    return {"precision": leg["precision"], "dtype": DTYPE_NOTES[leg["precision"]], "note": PRECISION_NOTES[leg["precision"]],
            "workload": workload, "value": leg["rays"] / leg["elapsed"], "unit": "rays/s", "steps": leg["steps"], "warmup": leg["warmup"],
            "ms_per_step": 1e3 * leg["elapsed"] / leg["steps"],
            "ray_samples_per_s": leg["evals_all"] / leg["elapsed"],
            "ray_samples_per_step_rank0": leg["evals"] / leg["steps"],
            "mask_fraction": leg["mask_fraction"],
            "per_rank_compute_s": {"min": min(per_rank), "mean": sum(per_rank) / len(per_rank), "max": max(per_rank), "all": per_rank,
                                   "note": "render time of each rank's share over the timed steps, before the all-gather"},
            "roofline": roofline_of(leg, workload, ctx),
            "kernels": {k: {"launches": d["launches"], "ms_per_step": d["ms"] / leg["steps"],
                            "algorithmic_tflops": d["flop"] / (d["ms"] * 1e-3) / 1e12} for k, d in leg["ksum"].items() if "flop" in d},
            "hbm_kernels": {k: hbm_entry(leg, k, d, ctx) for k, d in leg["ksum"].items() if "bytes" in d}}


def _sig(x, n=6):
    """Numbers of the final line keep n significant digits (a 17-digit double is noise at 1 % run-to-run)."""
    if isinstance(x, float):
        return float(f"{x:.{n}g}")
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return x


def build_records(head, second, config_legs, ctx):
    """(detail, final): `final` is the driver's contract line (< FINAL_LINE_LIMIT bytes as JSON), `detail` everything else.
    head / second / config_legs[] are measure() results; ctx carries the run's arguments and the side legs."""
    H, W, L, n1, n2, st, dt = ctx["dims"]
    world, partition, precision, workload = ctx["world"], ctx["partition"], ctx["precision"], ctx["workload"]
    hr = leg_record(head, workload, ctx)
    roof = hr["roofline"]
    cpu = ctx.get("cpu")
    final = {
        "metric": "rendered rays/s (and ray-samples/s) per GPU, 1080p x 128-sample layered render",
        "value": hr["value"], "unit": "rays/s", "n_gpus": world, "steps": ctx["steps"], "warmup": ctx["warmup"],
        "ms_per_step": hr["ms_per_step"], "higher_is_better": True,
        "scaling": "strong" if partition == "stripes" else "weak", "vs_baseline": None,
        "dtype": DTYPE_SHORT[precision], "data": "synthetic",
        "config": {"workload": workload, "precision": precision, "height": H, "width": W, "performer_layers": L,
                   "coarse_samples": n1, "fine_samples": n2,
                   "parallelism": ("1 GPU, whole view" if world == 1 else
                                   (f"{world} GPUs: interleaved {ctx['stripe_rows']}-row stripes of one view, one RCCL all-gather per view"
                                    if partition == "stripes" else f"{world} GPUs: one whole view each, final tiles all-gathered"))},
        "ray_samples_per_s": hr["ray_samples_per_s"],
        "roofline": {"kernel": KERNEL_SHORT[head["precision"]] if "mlp_stage" in head["ksum"] else "stnerf::spacenet_kernel",
                     "bound": "mfma", "achieved": roof["achieved"], "peak": roof["peak"], "unit": "TFLOP/s", "frac": roof["frac"],
                     "peak_note": roof["peak_note"], "algorithmic_tflops": roof["algorithmic_tflops"],
                     "executed_mfma_tflops": roof["executed_mfma_tflops"], "mfma_terms_per_product": roof["mfma_terms_per_product"],
                     "executed_frac_of_instruction_peak": roof["executed_frac_of_instruction_peak"],
                     "traffic": roof["traffic"], "traffic_rev": roof["traffic_rev"], "traffic_pose": roof["traffic_pose"],
                     "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"],
                     "avg_launch_ms": roof["avg_launch_ms"], "launches": roof["launches"]},
        "cpu_baseline": None if cpu is None else {
            "value": cpu["value"], "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"],
            "sample": cpu.get("sample_short", cpu["sample"])[:200], "chunks": len(cpu.get("chunks") or []),
            "chunks_asked_by_baseline_md": ">= 8 (BASELINE.md 3.3)", "seconds": cpu["seconds"], "cpu": cpu["host"]["cpu"][:64]},
    }
    if second is not None:
        sr = leg_record(second, workload, ctx)
        final["other_precision"] = {"precision": second["precision"], "value": sr["value"], "ms_per_step": sr["ms_per_step"],
                                    "steps": second["steps"], "warmup": second["warmup"], "frac": sr["roofline"]["frac"],
                                    "peak": sr["roofline"]["peak"], "algorithmic_tflops": sr["roofline"]["algorithmic_tflops"]}
    if ctx.get("psnr"):
        p = ctx["psnr"]
        final["psnr_vs_reference_dB"] = {"reference_seed_b_vs_a": p.get("reference_seed_b_vs_seed_a_dB"),
                                         "hip": p.get(f"hip_device_rng_{precision}_vs_reference_seed_a_dB")}
    if ctx.get("share"):
        final["share_emulation"] = {str(e["ranks"]): e["predicted_compute_efficiency"] for e in ctx["share"]["shares"]}
    final["detail"] = os.path.basename(DETAIL_PATH)
    final = _sig(final)

    detail = {
        "record": "detail", "workload": workload, "n_gpus": world,
        "config": {"use_space_time": st, "use_deform_time": dt, "rays_per_view": H * W, "rays_per_launch": ctx["rays_per_launch"],
                   "weights": "random, density head scaled (synthetic.make_state_dict seed 0)", "packed_floats_per_ray": parallel.packed_width(L + 1)},
        "precision_legs": {"note": "the same workload and poses in both arithmetics, one after the other in this process; the second leg is "
                                   "a short cross-check (<= 5 steps)", precision: hr},
        "hbm_bytes_note": "algorithmic = bytes the kernel has to move (t, raw and weights only of the layers a ray hits); "
                          "dense = every layer charged (round-1 accounting)",
        "hbm_microbench": ctx["hbm_microbench"] or None, "psnr_vs_reference": ctx.get("psnr"), "device": ctx["device"],
        "eager_gpu_baseline": ctx.get("eager"), "cpu_baseline": cpu, "share_emulation": ctx.get("share"),
    }
    if second is not None:
        detail["precision_legs"][second["precision"]] = sr
    if config_legs:
        detail["config_legs"] = {leg["workload"]: {k: v for k, v in leg_record(leg, leg["workload"], ctx).items()
                                                   if k not in ("note", "dtype", "per_rank_compute_s")} for leg in config_legs}
    return detail, final


def detail_line(detail):
    """The one-line digest of `detail` printed before the final line (<= DETAIL_LINE_LIMIT bytes): the per-leg headline numbers."""
    def leg(r):
        return _sig({"value": r["value"], "ms_per_step": r["ms_per_step"], "ray_samples_per_s": r["ray_samples_per_s"],
                     "frac": r["roofline"]["frac"], "algorithmic_tflops": r["roofline"]["algorithmic_tflops"],
                     "hbm": {k: [round(e["ms_per_step"], 3), round(e["frac"], 3)] for k, e in r["hbm_kernels"].items()}}, 5)
    d = {"record": "detail", "file": os.path.basename(DETAIL_PATH),
         "precision_legs": {k: leg(v) for k, v in detail["precision_legs"].items() if k != "note"},
         "config_legs": {k: leg(v) for k, v in detail.get("config_legs", {}).items()},
         "eager_gpu_baseline": _sig({k: v for k, v in (detail.get("eager_gpu_baseline") or {}).items() if k in ("value", "unit")}),
         "hbm_columns": "[ms_per_step, frac of 8 TB/s on algorithmic bytes]"}
    if detail.get("share_emulation"):
        d["share_emulation"] = _sig(detail["share_emulation"]["shares"], 5)
    line = json.dumps(d)
    if len(line) > DETAIL_LINE_LIMIT:
        line = json.dumps({"record": "detail", "file": d["file"], "precision_legs": d["precision_legs"]})
    return line


def emit(detail, final, detail_path=None):
    detail_path = detail_path or DETAIL_PATH
    final["detail"] = os.path.basename(detail_path)
    line = json.dumps(final)
    if len(line) >= FINAL_LINE_LIMIT:                       # never let verbosity cost the driver its record again
        for k in ("share_emulation", "psnr_vs_reference_dB", "other_precision"):
            final.pop(k, None)
            line = json.dumps(final)
            if len(line) < FINAL_LINE_LIMIT:
                break
    for path in (detail_path, os.path.join(REPO, "gpurun_out", os.path.basename(detail_path))):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, "w") as f:
                    json.dump({"final": final, "detail": detail}, f, indent=1)
        except OSError:
            pass
    print(detail_line(detail))
    print(line)
    sys.stdout.flush()



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="taekwondo-1080p-64+64", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-baseline-rays", type=int, default=8 * 3584,
                    help="0 disables the CPU baseline leg; default = 8 reference chunks spread over the image, ~50 s of host work "
                         "(what BASELINE.md 3.3 asks for: >= 8 chunks; the driver's run is ~3.5 min with it)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="torch threads of the CPU baseline leg (0 = min(32, host cores))")
    ap.add_argument("--rays-per-launch", type=int, default=1 << 19)
    ap.add_argument("--partition", default="stripes", choices=["views", "stripes"],
                    help="'stripes' (default, every N): ONE view per step, in interleaved row stripes over the GPUs when N > 1 "
                         "(strong scaling, the split BASELINE configs[3]/[4] describe); 'views' = one whole view per GPU "
                         "per step (weak scaling)")
    ap.add_argument("--stripe-rows", type=int, default=1, help="image rows per stripe of the striped partition")
    ap.add_argument("--gather", default="all", choices=list(parallel.GATHER_MODES),
                    help="N > 1: what the one all-gather per view carries (stnerf_amd.parallel): the whole 5-tuple (default, the "
                         "worst case), the fine images render_pose returns, or the two mixed images")
    ap.add_argument("--debug-single-device", action="store_true",
                    help="N>1 with every rank on cuda:0 over gloo: exercises the multi-rank code path on a 1-GPU box (not a measurement)")
    ap.add_argument("--no-psnr-check", action="store_true",
                    help="skip the PSNR-vs-reference check of the device RNG mode (a 128x128 view, ~0.1 s)")
    ap.add_argument("--eager-gpu-baseline-rays", type=int, default=16 * 3584,
                    help=">0 (default: 16 reference chunks, ~0.5 s): also time the oracle restatement through eager PyTorch-ROCm on this GPU -- "
                         "the 'stock ATen on the same GPU' figure of BASELINE.md 3.5 (informative); 0 skips it")
    ap.add_argument("--precision", default="bf16x3", choices=["fp32", "bf16x3"],
                    help="arithmetic of the headline leg: split-bf16 (the library default: three bf16 pieces per fp32 operand, six MFMAs, "
                         "two accumulators: fp32's significand and range) or exact f32 MFMA")
    ap.add_argument("--mlp-schedule", default="stage", choices=["stage", "per_net"],
                    help="exact-f32 MLP scheduling: one persistent launch per stage (default) or one launch per network (round 1)")
    ap.add_argument("--no-second-precision", action="store_true",
                    help="skip the short cross-check leg (<= 5 steps) in the other arithmetic (fp32 <-> bf16x3)")
    ap.add_argument("--no-config-legs", action="store_true",
                    help="skip the short legs over the other BASELINE configs (C4 walking-1080p-L4, C2 single-512), N = 1 only")
    ap.add_argument("--emulate-share", default="",
                    help="comma list of rank counts, e.g. 2,4,8 (N = 1 only): time rank 0's interleaved-stripe share of the view for "
                         "each count on this one GPU -- an EMULATION of a rank's compute, not a scaling measurement "
                         "(profiles/r05_share_emulation.md)")
    ap.add_argument("--detail-out", default=DETAIL_PATH, help="where the full record goes (default: bench_detail.json beside this script)")
    ap.add_argument("--dump-outputs", default=None,
                    help="rank 0 writes the LAST timed step's whole 5-tuple (every rank holds it after the all-gather) to this "
                         "torch file: tests compare an N-rank run with the 1-rank run bit for bit")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_launch(args)                               # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if rank != 0:
        # the record is rank 0's LAST stdout line: whatever another rank (or a library under it: RCCL / torchrun notices) writes to
        # stdout goes to stderr instead, under the driver's own torch.distributed.run as under _self_launch
        sys.stdout.flush()
        os.dup2(2, 1)
    if world != args.gpus:
        fail(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        fail("bench.py needs an MI355X (the render path has no CPU fallback)")
    try:
        parallel.init_from_env(single_device=args.debug_single_device)   # cuda:LOCAL_RANK, RCCL (gloo in the one-device debug mode)
    except RuntimeError as e:
        fail(str(e))
    device = torch.device("cuda", torch.cuda.current_device())
    dist = None
    if world > 1:
        import torch.distributed as dist

    from stnerf_amd import hip
    info = hip.device_info()
    frame_ids_of = lambda L: [1.0] + [2.5] * L
    partition = args.partition

    def scene(workload):
        model, dims = build_scene(workload, device)
        model.max_rays_per_launch = args.rays_per_launch
        model.mlp_schedule = args.mlp_schedule
        model.shard_views = partition == "stripes"      # sharding is opt-in (a collective call): the bench opts in
        model.gather = args.gather
        return model, dims

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(model, dims, precision, steps, warmup):
        """`steps` timed steps (barrier + synchronize on both sides, MAX over ranks) in one arithmetic.  A step is
        parallel.render_view -- the function render_pose runs -- at every N."""
        H, W, L, n1, n2, st, dt = dims
        l, n_rays, frame_ids = L + 1, H * W, frame_ids_of(L)
        model.set_precision(precision)
        clock = {"compute": 0.0, "t0": 0.0}
        inner = type(model).render_rays_raw.__get__(model)

        def timed_raw(*a, **k):                           # this rank's share of the view, before the all-gather
            out = inner(*a, **k)
            torch.cuda.synchronize()
            clock["compute"] += time.perf_counter() - clock["t0"]
            return out
        if world > 1:                                     # (N = 1: no extra synchronise inside the step; compute = elapsed)
            model.render_rays_raw = timed_raw

        def step(i):
            """Novel-view sweep, a new pose every step; every rank uses the same camera for step i."""
            K, T = syn.camera(H, W, orbit_deg=10.0 + 1.5 * i)
            model.seed = i if partition == "stripes" else i * world + rank
            clock["t0"] = time.perf_counter()
            out = render_view(model, K, T, H, W, frame_ids, stripe_rows=args.stripe_rows, device=device)
            if partition == "views" and world > 1:          # weak scaling: every rank its own view, final tiles all-gathered
                gather_tiles(torch.cat(list(out[0]), 1).contiguous(), world * n_rays)
            return out

        try:
            for i in range(warmup):
                step(-1 - i)
            timer = KernelTimer(n1, [FLOP_SPACE] + [(FLOP_SPACE_TIME if st else FLOP_SPACE) + (FLOP_MOTION if dt else 0)] * L)
            timer.start()
            clock["compute"] = 0.0
            fence()
            t0 = time.perf_counter()
            for i in range(steps):
                out = step(i)
            fence()
            elapsed = time.perf_counter() - t0
            timer.stop()
        finally:
            model.__dict__.pop("render_rays_raw", None)
        per_rank = [clock["compute"] if world > 1 else elapsed]
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
            allc = torch.zeros(world, dtype=torch.float64, device=device)
            allc[rank] = clock["compute"]
            dist.all_reduce(allc)
            per_rank = allc.tolist()
        ksum = timer.summarise()  # this rank's launches over the timed steps
        evals = sum(d["evals"] for name, d in ksum.items() if name in ("spacenet", "mlp_stage"))
        if out[4] is not None:
            hit = torch.stack(out[4]).float().mean(1).double()   # the gathered masks of the last view (every rank holds them)
        else:                                                    # --gather final: this rank's rays of the last view
            hit = timer.masks[-1].ne(0).float().mean(0).double()
        evals_all = float(evals)
        if world > 1:
            ev = torch.tensor([evals], dtype=torch.float64, device=device)
            dist.all_reduce(ev)
            evals_all = float(ev.item())
        flat = torch.cat(list(out[0]) + (list(out[1]) if out[1] is not None else []), 1)
        assert bool(torch.isfinite(flat).all()), "non-finite pixels in the rendered view"
        return dict(elapsed=elapsed, ksum=ksum, evals=evals, evals_all=evals_all, out=out, mask_fraction=hit.tolist(),
                    per_rank_compute_s=per_rank, steps=steps, warmup=warmup, precision=precision, dims=dims,
                    rays=(n_rays if partition == "stripes" else world * n_rays) * steps)

    model, dims = scene(args.workload)
    H, W, L, n1, n2, st, dt = dims
    n_rays = H * W
    head = measure(model, dims, args.precision, args.steps, args.warmup)
    second = None
    if not args.no_second_precision:
        # the other arithmetic as a short cross-check leg (<= 5 steps after <= 2 warm-up steps, the same first poses)
        second = measure(model, dims, "fp32" if args.precision != "fp32" else "bf16x3", min(args.steps, 5), min(args.warmup, 2))
    model.set_precision(args.precision)
    config_legs = []
    if world == 1 and not args.no_config_legs and partition == "stripes":
        for wl, k in (("walking-1080p-L4-64+64", 2), ("single-512-64+64", 2)):
            if wl == args.workload:
                continue
            m2, d2 = scene(wl)
            leg = measure(m2, d2, args.precision, k, 1)
            leg["workload"] = wl
            leg.pop("out")
            config_legs.append(leg)
            del m2
            torch.cuda.empty_cache()
    psnr_check = psnr_vs_reference(args.precision, device) if (rank == 0 and not args.no_psnr_check) else None
    share = None
    if world == 1 and args.emulate_share:
        share = emulate_share(model, dims, [int(x) for x in args.emulate_share.split(",")], args.stripe_rows, device,
                              steps=min(args.steps, 3))

    if args.dump_outputs and rank == 0:
        o = head["out"]
        torch.save({"mixed_fine": [t.cpu() for t in o[0]], "mixed_coarse": [t.cpu() for t in o[1]],
                    "layer_fine": [[t.cpu() for t in trip] for trip in o[2]],
                    "layer_coarse": [[t.cpu() for t in trip] for trip in o[3]], "masks": [t.cpu() for t in o[4]],
                    "world": world, "partition": partition, "precision": args.precision}, args.dump_outputs)

    if rank == 0:
        ctx = dict(workload=args.workload, precision=args.precision, steps=args.steps, warmup=args.warmup, world=world,
                   partition=partition, stripe_rows=args.stripe_rows, rays_per_launch=args.rays_per_launch, dims=dims,
                   device=info, psnr=psnr_check,
                   pmc=_load_profile(PMC_TRAFFIC_JSON), hbm_microbench=_load_profile(MEASURED_HBM_JSON),
                   eager=(eager_gpu_baseline(args.workload, args.eager_gpu_baseline_rays, device)
                          if world == 1 and args.eager_gpu_baseline_rays > 0 else None),
                   cpu=(cpu_baseline(args.workload, args.cpu_baseline_rays, args.cpu_threads)
                        if world == 1 and args.cpu_baseline_rays > 0 else None),
                   share=share)
        detail, final = build_records(head, second, config_legs, ctx)
        emit(detail, final, args.detail_out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
