"""Multi-GPU ray-tile sharding (one process per GPU, torch.distributed; backend "nccl" == RCCL on ROCm).

Rays are independent: GPU g renders a contiguous tile of the view's rays with
replicated weights, then ONE all-gather of the rendered tiles (20 B per ray per output image) rebuilds
the frame on every rank.  There is no other collective on the path.  The on-device RNG is keyed by the
global ray index, so a sharded render is bitwise identical to the single-GPU render.

The helpers are device-agnostic (they run under gloo on CPU tensors in tests/test_parallel_cpu.py);
only the ``render_rows`` callable passed in touches the GPU.
"""
from __future__ import annotations

from typing import Callable, Tuple

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


def make_row_renderer(model, K, T, h: int, w: int, frame_ids, density_threshold: float = 0.0,
                      bkgd_density_threshold: float = 0.0, chuncks: int = 512 * 7, device="cuda"):
    """The GPU ``render_rows`` for a LayeredRFRender view: device ray generation for the row window,
    layered_batchify_ray semantics, final image packed as (n, 5) = colour, depth, acc."""
    from stnerf_amd import ops
    from stnerf_amd.renderer import layered_batchify_ray

    def render_rows(first: int, n: int) -> torch.Tensor:
        rays = ops.generate_rays(K, T, h, w, frame_ids=frame_ids, first_ray=first, n=n, device=device)
        model.ray_index_base = first
        try:
            with torch.no_grad():
                fine = layered_batchify_ray(model, rays, None, None, chuncks=chuncks,
                                            density_threshold=density_threshold,
                                            bkgd_density_threshold=bkgd_density_threshold)[0]
        finally:
            model.ray_index_base = 0
        return torch.cat(list(fine), dim=1)

    return render_rows
