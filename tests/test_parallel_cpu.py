"""The N > 1 path on CPU: 2, 3 and 4 gloo processes shard a view, 'render' their tiles and all-gather them.
(The renderer itself needs a GPU; here render_rows is a deterministic stand-in, so what is under test is
the sharding / collective logic that bench.py and stnerf_amd.parallel run on RCCL.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stnerf_amd.parallel import gather_tiles, render_view_sharded, render_view_striped, shard_range, stripe_spans, unstripe


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 2073600, 2073601):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_stripe_spans_cover_every_ray_once():
    for n in (0, 5, 64, 101, 2073600):
        for world in (1, 2, 3, 8):
            for stripe in (1, 8, 1920 * 8):
                got = sorted(sp for r in range(world) for sp in stripe_spans(n, stripe, r, world))
                assert sum(e - s for s, e in got) == n
                assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
                if n:
                    assert got[0][0] == 0 and got[-1][1] == n
    with pytest.raises(ValueError):
        stripe_spans(10, 0, 0, 2)


def _fake_render(first, n):
    i = torch.arange(first, first + n, dtype=torch.float32)
    return torch.stack([torch.sin(i), torch.cos(i), i * 0.5, i, torch.ones_like(i)], 1)


def _global_index(first, n, stripe, period):
    i = torch.arange(n)
    return first + i if stripe <= 0 else first + (i // stripe) * period + i % stripe


class _StripedFakeRender:
    """A stand-in with the GPU renderer's one-call interface for a rank's stripes (parallel.make_row_renderer):
    ``striped(first, n, stripe, period)`` renders the rays of the window in one go."""

    def __init__(self):
        self.calls = []

    def __call__(self, first, n):
        self.calls.append(("rows", first, n))
        return _fake_render(first, n)

    def striped(self, first, n, stripe, period):
        self.calls.append(("striped", first, n, stripe, period))
        i = _global_index(first, n, stripe, period).float()
        return torch.stack([torch.sin(i), torch.cos(i), i * 0.5, i, torch.ones_like(i)], 1)


def test_unstripe_inverts_the_interleaving():
    for n_rays, stripe, world in [(64, 8, 2), (101, 8, 3), (1080 * 16, 16, 8), (50, 7, 4), (5, 8, 3), (96, 8, 4)]:
        spans = [stripe_spans(n_rays, stripe, r, world) for r in range(world)]
        m = max(sum(e - s for s, e in sp) for sp in spans)
        gathered = torch.full((world * m, 2), -1.0)
        for r in range(world):
            rows = torch.cat([torch.arange(s, e) for s, e in spans[r]] or [torch.zeros(0, dtype=torch.long)]).float()
            gathered[r * m: r * m + rows.numel(), 0] = rows
            gathered[r * m: r * m + rows.numel(), 1] = 2 * rows
        out = unstripe(gathered, n_rays, stripe, world, m)
        assert torch.equal(out[:, 0], torch.arange(n_rays).float()) and torch.equal(out[:, 1], 2 * out[:, 0])


def test_window_size_matches_stripe_spans():
    import importlib
    ops_src = importlib.import_module("stnerf_amd.ops")
    for n_rays, stripe, world in [(64, 8, 2), (101, 8, 3), (2073600, 1920, 8), (50, 7, 4), (5, 8, 3)]:
        for r in range(world):
            want = sum(e - s for s, e in stripe_spans(n_rays, stripe, r, world))
            assert ops_src.window_size(n_rays, r * stripe, stripe, world * stripe) == want
            ids = _global_index(r * stripe, want, stripe, world * stripe)
            assert ids.tolist() == [i for s, e in stripe_spans(n_rays, stripe, r, world) for i in range(s, e)]
    assert ops_src.window_size(100, 30, 0, 0) == 70


def _worker(rank, world, port, n_rays, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        img = render_view_sharded(_fake_render, n_rays)
        tile = render_view_sharded(_fake_render, n_rays, gather=False)
        s, e = shard_range(n_rays, rank, world)
        ok = torch.equal(img, _fake_render(0, n_rays)) and torch.equal(tile, _fake_render(s, e - s))
        for stripe in (8, 16, 1000):   # interleaved stripes incl. a short last stripe and "more ranks than stripes"
            ok = ok and torch.equal(render_view_striped(_fake_render, n_rays, stripe), _fake_render(0, n_rays))
            one_call = _StripedFakeRender()      # the GPU renderer's interface: ONE call for all of a rank's stripes
            ok = ok and torch.equal(render_view_striped(one_call, n_rays, stripe), _fake_render(0, n_rays))
            ok = ok and len(one_call.calls) == 1 and one_call.calls[0][0] in ("striped", "rows")
        # wrong tile size is an error on every rank, not a hang
        try:
            gather_tiles(torch.zeros(3, 5), n_rays)
            ok = False
        except ValueError:
            pass
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world, n_rays", [(2, 64), (2, 101), (3, 101), (4, 203)])   # equal and ragged shards / stripes
def test_gloo_render_and_gather(world, n_rays):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rays, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True) for r in range(world)]


def test_single_process_path_needs_no_process_group():
    assert torch.equal(render_view_sharded(_fake_render, 33), _fake_render(0, 33))
    assert torch.equal(render_view_striped(_fake_render, 33, 8), _fake_render(0, 33))


# ---- the sharded CALL SURFACE: layered_batchify_ray / render_view / render_pose under a process group ------------------
def _surface_model(L=2, n1=8, n2=4):
    """The real LayeredRFRender host logic (boxes from row 0 of every reference chunk, launch pieces, ray windows); the
    ONE native call, _render_launch, is answered by a deterministic function of the rays, their index IN THE VIEW (from
    the ray window), the boxes and the thresholds -- what a bitwise sharded-vs-whole comparison needs to be sensitive to."""
    import types
    from stnerf_amd import synthetic as syn
    from stnerf_amd.modeling.layered_rfrender import LayeredRFRender
    m = types.SimpleNamespace(BOARDER_WEIGHT=1e10, SAMPLE_METHOD="BBOX", SAME_SPACENET=False, TKERNEL_INC_RAW=True,
                              POSE_REFINEMENT=False, USE_DIR=True, USE_DEFORM_VIEW=False, USE_DEFORM_TIME=True,
                              USE_SPACE_TIME=True, BKGD_USE_DEFORM_TIME=False, BKGD_USE_SPACE_TIME=False, DEEP_RGB=False,
                              COARSE_RAY_SAMPLING=n1, FINE_RAY_SAMPLING=n2)
    cfg = types.SimpleNamespace(MODEL=m, DATASETS=types.SimpleNamespace(LAYER_NUM=L))

    class HostLogic(LayeredRFRender):
        launches = 0

        def _render_launch(self, rays, boxes, pivot, retiming, only_coarse, thr, bthr, window, replay, force_fp32=False):
            type(self).launches += 1
            n, l = rays.shape[0], self.layer_num + 1
            first, stripe, period = window
            g = _global_index(first, n, stripe, period).float()
            bx = boxes if boxes.dim() == 4 else boxes.unsqueeze(0).expand(n, l, 8, 3)
            key = bx[:, :, 0, 0] + rays[:, 6:6 + l] * 0.25                        # (n, l): boxes + frame ids
            if replay is not None:
                key = key + replay["jitter"].sum(-1).transpose(0, 1) + 2 * replay["u"].sum(-1).transpose(0, 1)
            mix_f = torch.stack([g, torch.sin(g), rays[:, 0], rays[:, 4], key.sum(1) + thr + float(self.seed)], 1)
            mix_c = mix_f * 0.5 + bthr
            lo_f = torch.stack([key, key * g[:, None], key + 1, key + 2, key + 3], 2)        # (n, l, 5)
            lo_c = lo_f - 7.0
            mask = ((g[:, None] + torch.arange(l)) % 3 == 0).to(torch.uint8)
            return mix_f, mix_c, lo_f, lo_c, mask

        def render_rays_raw(self, rays, *a, **k):
            return super().render_rays_raw(_AsIfOnGpu(rays), *a, **k)

    model = HostLogic(cfg, camera_num=1).eval()
    bk, per = syn.scene_boxes(L)
    model.set_bkgd_bbox(bk)
    model.set_bboxes(per)
    for p in model.parameters():
        p.requires_grad_(False)
    return model


class _AsIfOnGpu(torch.Tensor):
    """A CPU tensor that answers ``is_cuda`` with True (the host logic refuses CPU rays: there is no CPU fallback)."""
    @staticmethod
    def __new__(cls, t):
        return torch.Tensor._make_subclass(cls, t)

    @property
    def is_cuda(self):
        return True


def _flatten5(out):
    flat = list(out[0]) + list(out[1])
    for trip in list(out[2]) + list(out[3]):
        flat += list(trip)
    return flat + list(out[4])


def _surface_rays(n, l, chunk, varying_frames):
    g = torch.Generator().manual_seed(n)
    rays = torch.rand(n, 6 + l, generator=g)
    rays[:, 6:] = 1.0
    if varying_frames:                       # a new frame id in every reference chunk: boxes differ from chunk to chunk
        rays[:, 7] = 1.0 + (torch.arange(n) // chunk % 3).float() * 0.5
    return rays


def _surface_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), str(rank)
    from stnerf_amd import ops, parallel
    from stnerf_amd.utils import layered_batchify_ray
    assert parallel.init_from_env(backend="gloo") == (rank, world)
    try:
        ok, chunk, l = True, 64, 3
        model = _surface_model()
        for n, varying in [(64 * 5 + 10, False), (64 * 6, True), (64 * 2 + 1, True), (64, False), (64 * 7 + 63, True)]:
            rays = _surface_rays(n, l, chunk, varying)
            for fresh, replay in [(False, False), (True, False), (False, True)]:
                model.fresh_draws_per_call, model.seed = fresh, 11
                model.replay = None
                if replay:
                    g = torch.Generator().manual_seed(5)
                    model.replay = {"jitter": torch.rand(l, n, 8, generator=g), "u": torch.rand(l, n, 4, generator=g)}
                model.shard_views = False
                whole = layered_batchify_ray(model, rays, None, None, chuncks=chunk, density_threshold=0.3, bkgd_density_threshold=0.1)
                seed_whole, model.seed = model.seed, 11
                model.shard_views = True
                type(model).launches = 0
                split = layered_batchify_ray(model, rays, None, None, chuncks=chunk, density_threshold=0.3, bkgd_density_threshold=0.1)
                ok = ok and model.seed == seed_whole and model.ray_window == (0, 0, 0)
                ok = ok and all(torch.equal(a, b) and a.dtype == b.dtype and a.shape == b.shape
                                for a, b in zip(_flatten5(whole), _flatten5(split)))
                owns = len(parallel.stripe_spans(n, chunk, rank, world)) > 0
                ok = ok and (type(model).launches > 0) == owns
        # fewer rays than one chunk: no sharding, the model is called like the reference calls it (defaults for the thresholds)
        model.replay, model.fresh_draws_per_call = None, False
        small = _AsIfOnGpu(_surface_rays(40, l, chunk, False))
        type(model).launches = 0
        layered_batchify_ray(model, small, None, None, chuncks=chunk, density_threshold=0.3)
        ok = ok and type(model).launches == 1
        # render_view / render_pose: device ray generation of the rank's row stripes only (here: a stand-in generator)
        seen = []

        def fake_generate_rays(K, T, h, w, frame_ids=None, first_ray=0, n=None, device="cpu", stripe=0, period=0):
            n = ops.window_size(h * w, first_ray, stripe, period) if n is None else n
            seen.append(n)
            g = _global_index(first_ray, n, stripe, period).float()
            cols = [g, g * 0.5, torch.cos(g), torch.sin(g), g % 7, g % 5] + [torch.full_like(g, f) for f in frame_ids]
            return _AsIfOnGpu(torch.stack(cols, 1))
        ops.generate_rays = fake_generate_rays
        from stnerf_amd.render.render_pose import render_pose
        K, T = torch.eye(3), torch.eye(4)
        for h, w, rows in [(9, 16, 1), (8, 16, 2), (5, 16, 1), (2, 40, 1)]:
            model.shard_views = False
            whole = parallel.render_view(model, K, T, h, w, [1.0, 2.5, 1.0], 0.2, 0.05, chuncks=chunk, stripe_rows=rows, device="cpu")
            pose_whole = render_pose(model, T, K, h, w, [(0, 1), (1, 2.5), (2, 1)], 20.0, 0.2, 0.05, device="cpu")
            model.shard_views = True
            seen.clear()
            split = parallel.render_view(model, K, T, h, w, [1.0, 2.5, 1.0], 0.2, 0.05, chuncks=chunk, stripe_rows=rows, device="cpu")
            ok = ok and all(torch.equal(a, b) for a, b in zip(_flatten5(whole), _flatten5(split)))
            if h * w >= chunk:   # only this rank's stripes were generated
                mine = sum(e - s for s, e in parallel.stripe_spans(h * w, w * rows, rank, world))
                ok = ok and seen == ([mine] if mine else [])      # (a rank without a stripe generates nothing)
            pose_split = render_pose(model, T, K, h, w, [(0, 1), (1, 2.5), (2, 1)], 20.0, 0.2, 0.05, device="cpu")
            ok = ok and torch.equal(pose_whole[0], pose_split[0]) and torch.equal(pose_whole[1], pose_split[1])
            ok = ok and all(torch.equal(a, b) for a, b in zip(pose_whole[2] + pose_whole[3], pose_split[2] + pose_split[3]))
        # what is gathered: "fine" (what render_pose consumes) and "final" return None for the rest, the same bits for what they carry
        rays = _surface_rays(64 * 5 + 10, l, chunk, True)
        model.shard_views, model.gather = False, "all"
        whole = layered_batchify_ray(model, rays, None, None, chuncks=chunk, density_threshold=0.3, bkgd_density_threshold=0.1)
        model.shard_views = True
        for mode, present in (("fine", (0, 2, 4)), ("final", (0, 1))):
            model.gather = mode
            got = layered_batchify_ray(model, rays, None, None, chuncks=chunk, density_threshold=0.3, bkgd_density_threshold=0.1)
            for j in range(5):
                if j not in present:
                    ok = ok and got[j] is None
                else:
                    flat = lambda x: [x] if torch.is_tensor(x) else [t for y in x for t in flat(y)]
                    ok = ok and all(torch.equal(a, b) and a.dtype == b.dtype for a, b in zip(flat(whole[j]), flat(got[j])))
        model.gather = "fine"            # render_pose reads only what "fine" carries
        pose_fine = render_pose(model, T, K, 9, 16, [(0, 1), (1, 2.5), (2, 1)], 20.0, 0.2, 0.05, device="cpu")
        model.gather, model.shard_views = "all", False
        pose_whole = render_pose(model, T, K, 9, 16, [(0, 1), (1, 2.5), (2, 1)], 20.0, 0.2, 0.05, device="cpu")
        ok = ok and all(torch.equal(a, b) for a, b in zip([pose_whole[0], pose_whole[1]] + pose_whole[2] + pose_whole[3],
                                                          [pose_fine[0], pose_fine[1]] + pose_fine[2] + pose_fine[3]))
        # a sharded call is collective: ranks that hand in DIFFERENT rays get an error on every rank, not a stitched picture
        model.shard_views = True
        bad = rays.clone()
        bad[0, 0] += float(rank)
        try:
            layered_batchify_ray(model, bad, None, None, chuncks=chunk)
            ok = False
        except RuntimeError as e:
            ok = ok and "different inputs" in str(e)
        try:
            parallel.render_view(model, K, T * (1.0 + rank), 9, 16, [1.0, 2.5, 1.0], chuncks=chunk, device="cpu")
            ok = False
        except RuntimeError as e:
            ok = ok and "different inputs" in str(e)
        ok = ok and model.ray_window == (0, 0, 0)
        # the switches: opt-in per model or by STNERF_SHARD=1, forbidden by STNERF_SHARD=0; a NEW model does not shard
        os.environ["STNERF_SHARD"] = "0"
        ok = ok and parallel.active_group(model) is None
        del os.environ["STNERF_SHARD"]
        ok = ok and parallel.active_group(model) == (rank, world, None)
        fresh = _surface_model()
        ok = ok and fresh.shard_views is False and parallel.active_group(fresh) is None
        os.environ["STNERF_SHARD"] = "1"
        ok = ok and parallel.active_group(fresh) == (rank, world, None)
        del os.environ["STNERF_SHARD"]
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_call_surface_shards_and_gathers_the_whole_tuple(world):
    """layered_batchify_ray / render_view / render_pose under gloo: every rank returns exactly what the single-process
    call returns -- all 2 + 2 l triples and the l masks, same dtypes and shapes -- with ragged last chunks, frame ids
    that change from chunk to chunk, replayed uniforms, fresh draws per call, and more ranks than stripes."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_surface_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True) for r in range(world)]


def test_take_stripes_and_pack_round_trip():
    from stnerf_amd.parallel import pack_outputs, packed_width, take_stripes, unpack_outputs
    x = torch.arange(103 * 2).reshape(103, 2)
    for stripe, world in [(8, 2), (8, 3), (5, 4), (200, 2), (103, 3)]:
        for r in range(world):
            want = [i for s, e in stripe_spans(103, stripe, r, world) for i in range(s, e)]
            assert take_stripes(x, stripe, r, world)[:, 0].tolist() == [2 * i for i in want]
    g = torch.Generator().manual_seed(0)
    l, n = 3, 17
    raw = (torch.rand(n, 5, generator=g), torch.rand(n, 5, generator=g), torch.rand(n, l, 5, generator=g),
           torch.rand(n, l, 5, generator=g), (torch.rand(n, l, generator=g) > 0.5).to(torch.uint8))
    packed = pack_outputs(raw)
    assert packed.shape == (n, packed_width(l)) == (n, 11 + 10 * l) and packed.dtype == torch.float32
    back = unpack_outputs(packed, l)
    assert all(torch.equal(a, b) and a.dtype == b.dtype and b.is_contiguous() for a, b in zip(raw, back))
    with pytest.raises(ValueError):
        unpack_outputs(packed, l + 1)
    # the hit hint of the C ABI (mask byte 2 = "missed, and the sampler knows") is not a hit; 24 layers still fit one column
    hinted = (raw[0], raw[1], raw[2], raw[3], raw[4] * 3)
    assert torch.equal(unpack_outputs(pack_outputs(hinted), l)[4], raw[4])
    fine = unpack_outputs(pack_outputs(raw, "fine"), l, "fine")
    assert fine[1] is None and fine[3] is None and torch.equal(fine[0], raw[0]) and torch.equal(fine[2], raw[2]) and torch.equal(fine[4], raw[4])
    final = unpack_outputs(pack_outputs(raw, "final"), l, "final")
    assert final[2:] == (None, None, None) and torch.equal(final[0], raw[0]) and torch.equal(final[1], raw[1])
    assert (packed_width(9, "all"), packed_width(9, "fine"), packed_width(9, "final")) == (101, 51, 10)
    big = (torch.rand(5, 24, generator=g) > 0.5).to(torch.uint8)
    z5 = torch.zeros(5, 5)
    assert torch.equal(unpack_outputs(pack_outputs((z5, z5, torch.zeros(5, 24, 5), torch.zeros(5, 24, 5), big)), 24)[4], big)
    with pytest.raises(ValueError):
        packed_width(25)
    with pytest.raises(ValueError):
        packed_width(3, "everything")
