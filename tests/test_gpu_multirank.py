"""More than one rank on the ONE GPU of the test box (gloo instead of RCCL: RCCL refuses two ranks on one device): the
multi-GPU code path exactly as a node runs it -- ``python bench.py --gpus N`` launching its own ranks, and the call
surface (``layered_batchify_ray`` / ``render_pose`` / ``LayeredNeuralRenderer.render_pose``) sharding a view by itself
once a process group exists -- must give, on every rank, every tensor of the single-rank render bit for bit
(render/layered_neural_renderer.py:364-391 returns and :467-485 writes the per-layer images too, not just the mix)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

BENCH_COMMON = ["--steps", "2", "--warmup", "1", "--no-second-precision", "--no-config-legs", "--no-psnr-check",
                "--cpu-baseline-rays", "0", "--eager-gpu-baseline-rays", "0"]


def _run_bench(extra, dump):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):     # a plain `python bench.py`, as the driver runs it
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + BENCH_COMMON + ["--dump-outputs", dump,
                        "--detail-out", dump + ".detail.json"] + extra,
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 2 and len(lines[-1]) < 3072, r.stdout[-2000:]      # the detail digest, then the contract record LAST
    assert [ln for ln in r.stdout.splitlines() if ln.strip()][-1] == lines[-1], r.stdout[-2000:]   # nothing of any rank behind it on stdout
    rec = json.loads(lines[-1])
    side = json.load(open(dump + ".detail.json"))
    assert side["final"] == rec
    rec["_leg"] = side["detail"]["precision_legs"][rec["config"]["precision"]]   # the full leg record (per-rank times, counts)
    return rec, torch.load(dump)


def _flat(d):
    out = d["mixed_fine"] + d["mixed_coarse"] + d["masks"]
    for trip in d["layer_fine"] + d["layer_coarse"]:
        out += trip
    return out


@pytest.mark.parametrize("workload, precision, world", [("taekwondo-192x256-32+32", "bf16x3", 2),
                                                        ("taekwondo-192x256-32+32", "fp32", 3),
                                                        ("single-512-64+64", "bf16x3", 2),
                                                        ("taekwondo-192x256-32+32", "bf16x3", 8)])
def test_bench_launches_its_own_ranks_and_matches_one_rank(tmp_path, workload, precision, world):
    """`python bench.py --gpus N` (no torchrun) = N ranks of the same step function as N = 1; the gathered 5-tuple of the
    last step is the 1-rank one, bit for bit; both lines carry the same partition and scaling label."""
    one, a = _run_bench(["--gpus", "1", "--workload", workload, "--precision", precision], str(tmp_path / "one.pt"))
    many, b = _run_bench(["--gpus", str(world), "--debug-single-device", "--workload", workload, "--precision", precision],
                         str(tmp_path / "many.pt"))
    assert one["n_gpus"] == 1 and many["n_gpus"] == world and b["world"] == world
    assert one["scaling"] == many["scaling"] == "strong"
    assert one["config"]["workload"] == many["config"]["workload"] == workload
    fa, fb = _flat(a), _flat(b)
    assert len(fa) == len(fb) and len(fa) > 10
    for x, y in zip(fa, fb):
        assert x.shape == y.shape and x.dtype == y.dtype and torch.equal(x, y)
    assert float(a["mixed_fine"][0].std()) > 0.01                                  # a picture, not zeros
    assert len(many["_leg"]["per_rank_compute_s"]["all"]) == world
    assert abs(many["_leg"]["ray_samples_per_step_rank0"] * world - one["_leg"]["ray_samples_per_step_rank0"]) <= 0.35 * one["_leg"]["ray_samples_per_step_rank0"]
    assert many["roofline"]["frac"] > 0 and many["cpu_baseline"] is None


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _surface_worker(rank, world, port, precision, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import types
    import torch.distributed as dist
    from stnerf_amd import parallel, synthetic as syn
    from stnerf_amd.render import LayeredNeuralRenderer, render_pose
    from stnerf_amd.utils import layered_batchify_ray
    from oracle import stnerf_oracle as O
    import test_gpu_render as R
    parallel.init_from_env(single_device=True)
    try:
        L, n1, n2, H, W, far = 2, 16, 8, 48, 80, 20.0                  # 3840 rays: one full 3584-ray chunk + a ragged tail
        meta = dict(L=L, n1=n1, n2=n2, space_time=True, deform_time=True, weight_seed=71, edit={})
        model = R.build_model(meta).set_precision(precision)
        K, T = syn.camera(H, W, 12.0)
        pairs = [(0, 1), (1, 2.5), (2, 1)]
        N = H * W
        rays = O.append_frame_ids(O.generate_rays(K, T, H, W), pairs, L)        # CPU rays, as the dataset builds them
        labels, bbox, near_far = torch.zeros(N), torch.zeros(N, 8, 3), torch.tensor([[-1.0, -1.0]]).repeat(N, 1)
        model.shift, model.scale, model.alpha = [[0.0, 0.0, 0.0], [0.1, 0.0, 0.05], None], [1.0, 1.1, 0.9], 0.7
        checks = {}

        def reference_render_pose():
            """render/layered_neural_renderer.py:364-391 with the HIP model in the model's seat"""
            with torch.no_grad():
                stage2, stage1, stage2_layer, stage1_layer, masks = layered_batchify_ray(
                    model, rays.cuda(), labels.cuda(), bbox.cuda(), near_far=near_far.cuda(), density_threshold=0.05,
                    bkgd_density_threshold=0.02)
                color = stage2[0].reshape(H, W, 3)
                depth = stage2[1].reshape(H, W, 1)
                depth[depth < 0] = 0
                depth = depth / far
                color_layer = [i[0].reshape(H, W, 3) for i in stage2_layer]
                depth_layer = []
                for temp in stage2_layer:
                    d1 = temp[1].reshape(H, W, 1)
                    d1[depth < 0] = 0
                    depth_layer.append(d1 / far)
            return [color, depth] + color_layer + depth_layer + list(stage1) + [t for trip in stage1_layer for t in trip] + list(masks)

        cfg = types.SimpleNamespace(DATASETS=types.SimpleNamespace(LAYER_NUM=L, FRAME_NUM=3, FRAME_OFFSET=0),
                                    INPUT=types.SimpleNamespace(SIZE_TEST=[W, H]), OUTPUT_DIR="")
        renderer = LayeredNeuralRenderer(cfg, model=model, gt_poses=torch.stack([T, T]), gt_Ks=[K, K])
        model.shift, model.scale = [[0.0, 0.0, 0.0], [0.1, 0.0, 0.05], None], [1.0, 1.1, 0.9]

        def ours():
            a = render_pose(model, T, K, H, W, pairs, far, 0.05, 0.02)
            b = renderer.render_pose(T, K, pairs, 0.05, 0.02)
            return [a[0], a[1]] + a[2] + a[3] + [b[0], b[1]] + b[2] + b[3]

        for name, fn in (("reference call sequence", reference_render_pose), ("render_pose", ours)):
            for fresh in (False, True):
                model.fresh_draws_per_call = fresh
                model.shard_views, model.seed = False, 3
                whole = [t.clone() for t in fn()]
                seed_after = model.seed
                model.shard_views, model.seed = True, 3
                split = fn()
                same = all(x.shape == y.shape and x.dtype == y.dtype and torch.equal(x, y) for x, y in zip(whole, split))
                checks[f"{name}, fresh draws {fresh}"] = bool(same and len(whole) == len(split) and model.seed == seed_after
                                                              and float(whole[0].std()) > 0.01)
        q.put((rank, checks))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("precision, world", [("bf16x3", 2), ("fp32", 3)])
def test_call_surface_shards_itself_under_a_process_group(precision, world):
    """The reference's render_pose call sequence, stnerf_amd.render.render_pose and LayeredNeuralRenderer.render_pose on
    `world` ranks sharing cuda:0: colour, depth, every layer's colour and depth, the coarse outputs and the masks are the
    single-rank tensors bit for bit on every rank, and the draw counter advances the same way."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_surface_worker, args=(r, world, port, precision, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r for r, _ in results) == list(range(world))
    for r, checks in results:
        assert len(checks) == 4 and all(checks.values()), (r, checks)


def _rccl_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    from stnerf_amd import parallel, synthetic as syn
    from stnerf_amd.utils import layered_batchify_ray
    from oracle import stnerf_oracle as O
    import test_gpu_render as R
    try:                                    # the call init_from_env makes on a node, with the one rank this box can hold
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
        probe = torch.arange(12, device="cuda", dtype=torch.float32).reshape(4, 3)
        echoed = parallel._all_gather_rows(probe, 4, 1)
        torch.cuda.synchronize()
    except Exception as e:                  # no RCCL transport on this box: nothing of ours has run yet
        q.put(("unavailable", repr(e)))
        return
    q.put(("initialised", None))            # from here on a failure is ours
    try:
        checks = {"backend": dist.get_backend() == "nccl", "echo": bool(torch.equal(echoed, probe))}
        L, H, W = 2, 48, 80                                             # 3840 rays: one full chunk + a ragged tail
        model = R.build_model(dict(L=L, n1=16, n2=8, space_time=True, deform_time=True, weight_seed=71, edit={}))
        K, T = syn.camera(H, W, 12.0)
        rays = O.append_frame_ids(O.generate_rays(K, T, H, W), [(0, 1), (1, 2.5), (2, 1)], L).cuda()

        def flat(out):
            stage2, stage1, stage2_layer, stage1_layer, masks = out
            return list(stage2) + list(stage1) + [t for trip in list(stage2_layer) + list(stage1_layer) for t in trip] + list(masks)

        for fresh in (False, True):
            model.fresh_draws_per_call, model.seed = fresh, 5
            with torch.no_grad():
                whole = [t.clone() for t in flat(layered_batchify_ray(model, rays, None, None, density_threshold=0.05,
                                                                      bkgd_density_threshold=0.02))]
                model.seed = 5
                split = flat(parallel.render_rays_sharded(model, rays, 512 * 7, 0.05, 0.02, act=(0, 1, None)))
            checks[f"5-tuple through RCCL, fresh draws {fresh}"] = bool(
                len(whole) == len(split) == 6 + 6 * (L + 1) + (L + 1) and float(whole[0].std()) > 0.01
                and all(x.shape == y.shape and x.dtype == y.dtype and x.is_cuda and torch.equal(x, y) for x, y in zip(whole, split)))
        q.put(("ran", checks))
    finally:
        dist.destroy_process_group()


def test_packed_tuple_goes_through_rccl_itself():
    """What the gloo runs above cannot show: the one collective of the path on the backend a node uses.  A 1-rank "nccl"
    group (RCCL refuses a second rank on the same device) initialised the way ``parallel.init_from_env`` does it; the
    packed 5-tuple of this rank's stripes goes through ``all_gather_into_tensor`` on DEVICE tensors and comes back as the
    unsharded render, bit for bit.  A box whose RCCL cannot initialise at all skips (nothing of ours has run by then)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()

    def next_message(seconds):
        import queue
        for _ in range(seconds):
            try:
                return q.get(timeout=1)
            except queue.Empty:
                if not p.is_alive() and q.empty():
                    return None
        return None

    first = next_message(600)
    if first is None or first[0] == "unavailable":      # (an RCCL that aborts the process instead of raising lands here too)
        p.join(timeout=30)
        pytest.skip(f"RCCL did not initialise on this box: {first[1] if first else f'worker exit code {p.exitcode}'}")
    assert first[0] == "initialised"
    result = next_message(600)
    p.join(timeout=120)
    assert result is not None and result[0] == "ran", (result, p.exitcode)
    assert p.exitcode == 0
    payload = result[1]
    assert len(payload) == 4 and all(payload.values()), payload
