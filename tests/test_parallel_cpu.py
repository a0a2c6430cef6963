"""The N > 1 path on CPU: two gloo processes shard a view, 'render' their tiles and all-gather them.
(The renderer itself needs a GPU; here render_rows is a deterministic stand-in, so what is under test is
the sharding / collective logic that bench.py and stnerf_amd.parallel run on RCCL.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stnerf_amd.parallel import gather_tiles, render_view_sharded, render_view_striped, shard_range, stripe_spans


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


@pytest.mark.parametrize("n_rays", [64, 101])   # equal shards (single all_gather_into_tensor) and ragged shards
def test_two_rank_gloo_render_and_gather(n_rays):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_rays, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def test_single_process_path_needs_no_process_group():
    assert torch.equal(render_view_sharded(_fake_render, 33), _fake_render(0, 33))
    assert torch.equal(render_view_striped(_fake_render, 33, 8), _fake_render(0, 33))
