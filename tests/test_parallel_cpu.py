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
