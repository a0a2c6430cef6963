"""Multi-GPU ray-tile sharding (one process per GPU, torch.distributed; backend "nccl" == RCCL on ROCm).

Rays are independent: GPU g renders a contiguous tile of the view's rays with
replicated weights, then ONE all-gather of the rendered tiles (20 B per ray per output image) rebuilds
the frame on every rank.  There is no other collective on the path.  The on-device RNG is keyed by the
global ray index, so a sharded render is bitwise identical to the single-GPU render.

Two partitions are offered: contiguous tiles (``render_view_sharded``: fewest launches, right when every
rank renders its own view or the view's cost is uniform) and interleaved stripes (``render_view_striped``:
rank r takes stripes r, r+G, r+2G, ... of the image; the performers cover only part of the picture, so
contiguous tiles of ONE view differ in cost by ~2x between its centre and its border, stripes do not).

The helpers are device-agnostic (they run under gloo on CPU tensors in tests/test_parallel_cpu.py);
only the ``render_rows`` callable passed in touches the GPU.
"""
from __future__ import annotations

from typing import Callable, List, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced partition of range(n): the first n % world shards get one extra item."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_tiles(tile: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All-gather per-rank tiles (rows of a (n_total, C) image split by ``shard_range``) -> (n_total, C)
    on every rank.  Equal shards use one all_gather_into_tensor (a single RCCL all-gather over xGMI);
    ragged shards are padded to the largest shard first."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    if tile.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank}: tile has {tile.shape[0]} rows, its shard has {sizes[rank]}")
    tile = tile.contiguous()
    if len(set(sizes)) == 1:
        out = torch.empty((n_total,) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
        dist.all_gather_into_tensor(out, tile, group=group)
        return out
    m = max(sizes)
    padded = torch.zeros((m,) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
    padded[: tile.shape[0]] = tile
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], 0)


def render_view_sharded(render_rows: Callable[[int, int], torch.Tensor], n_rays: int, group=None,
                        gather: bool = True) -> torch.Tensor:
    """``render_rows(first_ray, n) -> (n, C)`` renders rays [first_ray, first_ray+n) of the view.
    Each rank renders its shard; with ``gather`` every rank returns the whole (n_rays, C) image."""
    if not dist.is_available() or not dist.is_initialized():
        return render_rows(0, n_rays)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    s, e = shard_range(n_rays, rank, world)
    tile = render_rows(s, e - s)
    return gather_tiles(tile, n_rays, group) if gather else tile


def stripe_spans(n: int, stripe: int, rank: int, world: int) -> List[Tuple[int, int]]:
    """Ray ranges [start, end) of the stripes owned by ``rank``: stripe k = rays [k*stripe, (k+1)*stripe) and
    belongs to rank k % world (the last stripe may be short)."""
    if stripe < 1:
        raise ValueError("stripe must be >= 1 ray")
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    n_stripes = (n + stripe - 1) // stripe
    return [(k * stripe, min((k + 1) * stripe, n)) for k in range(rank, n_stripes, world)]


def render_view_striped(render_rows: Callable[[int, int], torch.Tensor], n_rays: int, stripe: int, group=None):
    """Like render_view_sharded with interleaved stripes of ``stripe`` rays (use a multiple of the image
    width): rank r renders stripes r, r+G, r+2G, ... -- in ONE call if ``render_rows`` offers
    ``render_rows.striped(first, n, stripe, period)`` (the GPU renderer does: one launch sequence over the rank's
    rays, ray window = include/stnerf.h), else one ``render_rows`` call per stripe -- then ONE all-gather of the
    concatenated stripes (padded to the largest rank) and the stripes are put back in image order."""
    if not dist.is_available() or not dist.is_initialized():
        return render_rows(0, n_rays)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    spans = [stripe_spans(n_rays, stripe, r, world) for r in range(world)]
    sizes = [sum(e - s for s, e in sp) for sp in spans]
    m = max(sizes)
    if spans[rank] and hasattr(render_rows, "striped"):
        local = render_rows.striped(rank * stripe, sizes[rank], stripe, world * stripe)
    elif spans[rank]:
        local = torch.cat([render_rows(s, e - s) for s, e in spans[rank]], 0)
    else:  # more ranks than stripes: this rank only takes part in the collective
        probe = render_rows(0, 0)
        local = probe.new_zeros((0,) + tuple(probe.shape[1:]))
    padded = local.new_zeros((m,) + tuple(local.shape[1:]))
    padded[: local.shape[0]] = local
    gathered = local.new_empty((world * m,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(gathered, padded.contiguous(), group=group)
    return unstripe(gathered, n_rays, stripe, world, m)


def unstripe(gathered: torch.Tensor, n_rays: int, stripe: int, world: int, per_rank: int) -> torch.Tensor:
    """(world * per_rank, C) = every rank's stripes back to back (padded to per_rank rows) -> (n_rays, C) in image
    order.  Full stripes move as ONE strided copy; only a short last stripe is handled on its own."""
    C = tuple(gathered.shape[1:])
    out = gathered.new_empty((n_rays,) + C)
    n_full = n_rays // stripe                       # stripes of full length
    rounds = n_full // world                        # rounds in which every rank owns a full stripe
    if rounds:
        src = gathered.reshape((world, per_rank) + C)[:, : rounds * stripe].reshape((world, rounds, stripe) + C)
        out[: rounds * world * stripe].reshape((rounds, world, stripe) + C).copy_(src.transpose(0, 1))
    for k in range(rounds * world, (n_rays + stripe - 1) // stripe):   # the last, incomplete round (< world stripes)
        r, j = k % world, k // world
        s, e = k * stripe, min((k + 1) * stripe, n_rays)
        out[s:e] = gathered[r * per_rank + j * stripe: r * per_rank + j * stripe + (e - s)]
    return out


def make_row_renderer(model, K, T, h: int, w: int, frame_ids, density_threshold: float = 0.0,
                      bkgd_density_threshold: float = 0.0, chuncks: int = 512 * 7, device="cuda"):
    """The GPU ``render_rows`` for a LayeredRFRender view: device ray generation for the row window,
    layered_batchify_ray semantics OF THE WHOLE VIEW (a view of at least ``chuncks`` rays is rendered with the
    thresholds whatever the size of the piece a rank renders; the reference drops them only when the whole
    call has fewer rays than one chunk, utils/batchify_rays.py:52-54), final image packed as (n, 5) =
    colour, depth, acc."""
    from stnerf_amd import ops
    from stnerf_amd.utils.batchify_rays import layered_batchify_ray

    def render_window(first: int, n: int, stripe: int, period: int) -> torch.Tensor:
        if n == 0:
            return torch.empty(0, 5, dtype=torch.float32, device=device)
        rays = ops.generate_rays(K, T, h, w, frame_ids=frame_ids, first_ray=first, n=n, device=device, stripe=stripe,
                                 period=period)
        model.ray_window = (first, stripe, period)
        try:
            with torch.no_grad():
                if h * w < chuncks:
                    fine = layered_batchify_ray(model, rays, None, None, chuncks=chuncks,
                                                density_threshold=density_threshold,
                                                bkgd_density_threshold=bkgd_density_threshold)[0]
                else:
                    out = model.render_rays(rays, False, density_threshold, bkgd_density_threshold, ref_chunk=chuncks)
                    fine = out[0]
                    render_rows.last_masks = out[4]
        finally:
            model.ray_window = (0, 0, 0)
        return torch.cat(list(fine), dim=1)

    def render_rows(first: int, n: int) -> torch.Tensor:
        return render_window(first, n, 0, 0)

    render_rows.striped = render_window     # (first, n, stripe, period): all of a rank's stripes in one launch sequence
    render_rows.last_masks = None           # per-layer hit masks of the latest call (bench.py counts evaluations)
    return render_rows
