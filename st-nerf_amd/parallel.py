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

The call surface shards on request: with ``model.shard_views = True`` (the ``python -m
stnerf_amd.dropin`` launcher under torch.distributed.run and bench.py set it; ``STNERF_SHARD=1`` sets it for every model,
``STNERF_SHARD=0`` forbids it) and a torch.distributed group of more than one rank, ``utils.layered_batchify_ray`` (->
``render_rays_sharded``: the caller holds the whole ray tensor, as the reference's ``render_pose`` does,
render/layered_neural_renderer.py:364-391) and ``render.render_pose`` (-> ``render_view``: rays generated on the device for
the rank's stripes only) render interleaved stripes of the view and rebuild the 5-tuple on every rank with ONE all-gather.
The call is then COLLECTIVE: every rank must make it with the same rays.  It is therefore opt-in (the default is off: a
trainer's rank-0-only evaluation, or ranks rendering different views, must not meet a collective), and each sharded call first
all-reduces a fingerprint of its inputs and raises on every rank when they differ.  ``model.gather`` picks the payload
(``GATHER_MODES``): "all" = mixed + per-layer, fine + coarse, masks (11 + 10 l floats per ray), "fine" = what ``render_pose``
consumes (mixed fine + per-layer fine + masks: 6 + 5 l), "final" = the two mixed images (10); what is not gathered comes back
as None.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Tuple

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


# ---------------------------------------------------------------------------------------------------------------
# The sharded call surface: layered_batchify_ray / render_pose under an initialised process group
# ---------------------------------------------------------------------------------------------------------------
def active_group(model=None) -> Optional[Tuple[int, int, object]]:
    """(rank, world, group) when a view handed to the call surface is to be split over the ranks, else None.  Sharding is
    OPT-IN: ``model.shard_views`` is True (or ``STNERF_SHARD=1``), ``STNERF_SHARD`` is not "0", a process group of more than
    one rank is initialised, and no ray window is already set on the model (a caller that windows its own rays --
    ``make_row_renderer`` -- is doing the partition itself)."""
    env = os.environ.get("STNERF_SHARD", "")
    if env == "0":
        return None
    if model is not None and tuple(getattr(model, "ray_window", (0, 0, 0))) != (0, 0, 0):
        return None
    if not (env == "1" or (model is not None and getattr(model, "shard_views", False))):
        return None
    if not (dist.is_available() and dist.is_initialized()):
        return None
    group = getattr(model, "shard_group", None) if model is not None else None
    world = dist.get_world_size(group)
    if world < 2:
        return None
    return dist.get_rank(group), world, group


def check_collective(fingerprint, what: str, group=None, device=None) -> None:
    """A sharded call is collective: every rank must make it with the same inputs.  ``fingerprint`` (a short list of
    floats, exact in fp64) is all-reduced (MAX of [x, -x] = max and -min in one collective); a rank that sees max != min
    raises -- and so does every other rank, the reduced values being the same everywhere -- instead of stitching stripes of
    different views together.  (A rank that never makes the call cannot be detected from inside it: that hangs, like any
    unmatched collective; sharding is opt-in for that reason.)"""
    x = torch.tensor([float(v) for v in fingerprint], dtype=torch.float64)
    both = torch.cat([x, -x])
    if dist.get_backend(group) != "gloo" and device is not None:
        both = both.to(device)
    dist.all_reduce(both, op=dist.ReduceOp.MAX, group=group)
    both = both.cpu()
    n = x.numel()
    if not torch.equal(both[:n], -both[n:]):
        raise RuntimeError(f"{what}: the ranks of this process group called with different inputs (fingerprint max "
                           f"{both[:n].tolist()} vs min {(-both[n:]).tolist()}).  A sharded render is one view split over the "
                           "ranks: give every rank the same rays, or set model.shard_views = False (STNERF_SHARD=0) when the "
                           "ranks render different views")


def rays_fingerprint(rays: torch.Tensor):
    """N, the width, and three cheap checksums of a ray tensor (one tiny D2H)."""
    n = rays.shape[0]
    r = rays.detach()
    sums = torch.stack([r[0].double().sum(), r[n // 2].double().sum(), r[-1].double().sum(),
                        r[:: max(1, n // 4096)].double().sum()]).cpu().tolist()
    return [n, rays.shape[1]] + sums


def take_stripes(x: torch.Tensor, stripe: int, rank: int, world: int) -> torch.Tensor:
    """Rows of ``x`` (dim 0 = ray) that lie in the stripes of ``rank`` (``stripe_spans``), in order, as one dense tensor:
    the full rounds move as ONE strided copy."""
    n = x.shape[0]
    spans = stripe_spans(n, stripe, rank, world)
    rounds = (n // stripe) // world
    inner = tuple(x.shape[1:])
    parts = []
    if rounds:
        parts.append(x[: rounds * world * stripe].reshape((rounds, world, stripe) + inner)[:, rank].reshape((rounds * stripe,) + inner))
    parts += [x[s:e] for s, e in spans[rounds:]]
    if not parts:
        return x[:0].contiguous()
    return parts[0].contiguous() if len(parts) == 1 else torch.cat(parts, 0)


GATHER_MODES = ("all", "fine", "final")


def packed_width(l: int, mode: str = "all") -> int:
    """Floats per ray of the packed outputs: "all" = mixed fine + coarse (5 + 5), l layers fine + coarse (5 l + 5 l), the l hit
    masks as the bits of ONE float column (l <= 24: exact in fp32); "fine" = mixed fine, l layers fine, the mask column;
    "final" = mixed fine + mixed coarse."""
    if mode not in GATHER_MODES:
        raise ValueError(f"gather mode must be one of {GATHER_MODES}")
    if l > 24:
        raise ValueError("the hit masks of more than 24 layers do not fit one fp32 column")
    return {"all": 11 + 10 * l, "fine": 6 + 5 * l, "final": 10}[mode]


def pack_outputs(raw, mode: str = "all") -> torch.Tensor:
    """The five tensors of ``LayeredRFRender.render_rays_raw`` -> (n, packed_width(l, mode)) float32: the payload of the one
    all-gather."""
    mix_f, mix_c, lo_f, lo_c, mask = raw
    n, l = mix_f.shape[0], mask.shape[1]
    packed_width(l, mode)
    if mode == "final":
        return torch.cat([mix_f, mix_c], 1)
    bits = (mask != 0).to(torch.float32) @ (2.0 ** torch.arange(l, dtype=torch.float32, device=mask.device)).reshape(l, 1)
    if mode == "fine":
        return torch.cat([mix_f, lo_f.reshape(n, -1), bits], 1)
    return torch.cat([mix_f, mix_c, lo_f.reshape(n, -1), lo_c.reshape(n, -1), bits], 1)


def unpack_outputs(packed: torch.Tensor, l: int, mode: str = "all"):
    """Inverse of ``pack_outputs`` (dense tensors, the layout ``render_rays_raw`` returns; None for what the mode leaves out)."""
    n = packed.shape[0]
    if packed.shape[1] != packed_width(l, mode):
        raise ValueError(f"packed outputs ({mode}) of {l} layers are {packed_width(l, mode)} floats wide, got {packed.shape[1]}")
    if mode == "final":
        return packed[:, 0:5].contiguous(), packed[:, 5:10].contiguous(), None, None, None
    bits = packed[:, -1].to(torch.int32)
    mask = ((bits.unsqueeze(1) >> torch.arange(l, dtype=torch.int32, device=packed.device)) & 1).to(torch.uint8)
    if mode == "fine":
        return packed[:, 0:5].contiguous(), None, packed[:, 5:5 + 5 * l].reshape(n, l, 5).contiguous(), None, mask
    a, b, c = 10, 10 + 5 * l, 10 + 10 * l
    return (packed[:, 0:5].contiguous(), packed[:, 5:10].contiguous(), packed[:, a:b].reshape(n, l, 5).contiguous(),
            packed[:, b:c].reshape(n, l, 5).contiguous(), mask)


def gather_mode(model) -> str:
    mode = getattr(model, "gather", "all")
    if mode not in GATHER_MODES:
        raise ValueError(f"model.gather must be one of {GATHER_MODES}, got {mode!r}")
    return mode


def _all_gather_rows(local: torch.Tensor, per_rank: int, world: int, group=None) -> torch.Tensor:
    """(per_rank-padded) all-gather of every rank's rows -> (world * per_rank, C).  backend nccl (= RCCL over xGMI) moves
    device tensors directly; gloo (CPU tests, and the N-ranks-on-one-GPU debug mode) stages device tensors through the host."""
    padded = local
    if local.shape[0] != per_rank:
        padded = local.new_zeros((per_rank,) + tuple(local.shape[1:]))
        padded[: local.shape[0]] = local
    padded = padded.contiguous()
    if padded.is_cuda and dist.get_backend(group) == "gloo":
        host = padded.cpu()
        out = host.new_empty((world * per_rank,) + tuple(host.shape[1:]))
        dist.all_gather_into_tensor(out, host, group=group)
        return out.to(padded.device)
    out = padded.new_empty((world * per_rank,) + tuple(padded.shape[1:]))
    dist.all_gather_into_tensor(out, padded, group=group)
    return out


def gather_stripes(local: torch.Tensor, n_total: int, stripe: int, rank: int, world: int, group=None) -> torch.Tensor:
    """Every rank's rendered stripes (``take_stripes`` order) -> the (n_total, C) view on every rank: ONE all-gather."""
    sizes = [sum(e - s for s, e in stripe_spans(n_total, stripe, r, world)) for r in range(world)]
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank}: {local.shape[0]} rows rendered, its stripes hold {sizes[rank]}")
    m = max(sizes)
    return unstripe(_all_gather_rows(local, m, world, group), n_total, stripe, world, m)


def _render_local(model, local_rays, window, n_total, stripe, rank, world, group, only_coarse, thr, bthr, chuncks, replay_full):
    """This rank's stripes through the model (ray window set: the RNG stream is the view's), packed, gathered, unpacked."""
    l = model.layer_num + 1
    mode = gather_mode(model)
    saved_window, saved_replay = model.ray_window, model.replay
    try:
        model.ray_window = window
        if replay_full is not None:      # recorded uniforms (l, N, ns) follow their rays
            model.replay = {k: take_stripes(v.transpose(0, 1), stripe, rank, world).transpose(0, 1).contiguous()
                            for k, v in replay_full.items()}
        if local_rays.shape[0]:
            packed = pack_outputs(model.render_rays_raw(local_rays, only_coarse, thr, bthr, ref_chunk=chuncks), mode)
        else:                            # more ranks than stripes: only the collective (and the seed) on this rank
            packed = local_rays.new_zeros((0, packed_width(l, mode)))
            model.advance_seed()
    finally:
        model.ray_window, model.replay = saved_window, saved_replay
    whole = gather_stripes(packed, n_total, stripe, rank, world, group)
    return model.as_reference_tuple(unpack_outputs(whole, l, mode))


def render_rays_sharded(model, rays, chuncks: int, density_threshold=0.0, bkgd_density_threshold=0.0, only_coarse=False,
                        act=None):
    """``layered_batchify_ray`` of a view every rank holds the rays of (N >= chuncks), split over the ranks: rank r takes
    the reference's chunks r, r + G, r + 2G, ... (stripes of ``chuncks`` rays, so that row 0 of every reference chunk --
    where the reference reads the frame ids, layered_rfrender.py:200 -- stays row 0 of a chunk), renders them as one
    launch sequence and the whole 5-tuple is rebuilt on every rank by one all-gather.  Bitwise equal to the unsharded
    render: the device RNG is keyed by the ray's index in the view."""
    rank, world, group = act or active_group(model)
    check_collective(rays_fingerprint(rays) + [chuncks, float(density_threshold), float(bkgd_density_threshold), int(model.seed) % (1 << 52)],
                     "layered_batchify_ray (sharded)", group, rays.device)
    local = take_stripes(rays, chuncks, rank, world)
    return _render_local(model, local, (rank * chuncks, chuncks, world * chuncks), rays.shape[0], chuncks, rank, world, group,
                         only_coarse, density_threshold, bkgd_density_threshold, chuncks, model.replay)


def render_view(model, K, T, h: int, w: int, frame_ids, density_threshold=0.0, bkgd_density_threshold=0.0,
                chuncks: int = 512 * 7, stripe_rows: int = 1, device="cuda", gather: Optional[str] = None):
    """One view from its camera: rays generated on the device (no CPU ray tensor), ``layered_batchify_ray`` semantics, the
    reference's 5-tuple on every rank.  Without a process group: the whole view on this GPU.  With one: interleaved
    stripes of ``stripe_rows`` image rows over the ranks (rank r generates and renders rows r, r + G, ... only), one
    all-gather of ``gather`` (default: ``model.gather``).  The SAME function is bench.py's step at every N and what
    ``render.render_pose`` calls (with gather="fine": the images it returns)."""
    from stnerf_amd import ops
    from stnerf_amd.utils.batchify_rays import layered_batchify_ray
    act = active_group(model)
    n_total = h * w
    if act is None or n_total < chuncks:
        rays = ops.generate_rays(K, T, h, w, frame_ids=frame_ids, device=device)
        with torch.no_grad():
            return layered_batchify_ray(model, rays, None, None, chuncks=chuncks, density_threshold=density_threshold,
                                        bkgd_density_threshold=bkgd_density_threshold)
    rank, world, group = act
    stripe = w * max(1, int(stripe_rows))
    mode = gather_mode(model) if gather is None else gather
    check_collective([h, w, stripe_rows, chuncks, float(density_threshold), float(bkgd_density_threshold), int(model.seed) % (1 << 52)]
                     + torch.as_tensor(K, dtype=torch.float64).flatten().tolist() + torch.as_tensor(T, dtype=torch.float64).flatten().tolist()
                     + [float(f) for f in frame_ids], "render_view (sharded)", group, device)
    packed = render_view_share(model, K, T, h, w, frame_ids, rank, world, density_threshold, bkgd_density_threshold, chuncks,
                               stripe_rows, device, mode)
    whole = gather_stripes(packed, n_total, stripe, rank, world, group)
    return model.as_reference_tuple(unpack_outputs(whole, model.layer_num + 1, mode))


def render_view_share(model, K, T, h: int, w: int, frame_ids, rank: int, world: int, density_threshold=0.0,
                      bkgd_density_threshold=0.0, chuncks: int = 512 * 7, stripe_rows: int = 1, device="cuda",
                      mode: str = "all") -> torch.Tensor:
    """What rank ``rank`` of ``world`` computes for one view BEFORE the all-gather: its interleaved row stripes' rays generated
    on the device and rendered as one launch sequence, packed (``pack_outputs``).  No process group is touched:
    ``render_view`` calls this with the group's rank, ``bench.py --emulate-share`` with a made-up (0, N) to time a rank's
    share on one GPU."""
    from stnerf_amd import ops
    n_total = h * w
    stripe = w * max(1, int(stripe_rows))
    window = (rank * stripe, stripe, world * stripe)
    n_local = ops.window_size(n_total, *window)
    l = model.layer_num + 1
    if n_local == 0:                          # more ranks than stripes: only the collective (and the seed) on this rank
        model.advance_seed()
        return torch.zeros((0, packed_width(l, mode)), dtype=torch.float32, device=device)
    rays = ops.generate_rays(K, T, h, w, frame_ids=frame_ids, first_ray=window[0], n=n_local, device=device, stripe=stripe,
                             period=window[2])
    saved = model.ray_window
    try:
        model.ray_window = window
        with torch.no_grad():
            return pack_outputs(model.render_rays_raw(rays, False, density_threshold, bkgd_density_threshold, ref_chunk=chuncks), mode)
    finally:
        model.ray_window = saved


def init_from_env(backend: Optional[str] = None, single_device: bool = False):
    """One process per GPU from the variables torch.distributed.run sets (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*):
    picks cuda:LOCAL_RANK and initialises the group (backend "nccl" = RCCL over xGMI).  ``single_device`` puts every rank
    on cuda:0 over gloo (the multi-rank code path on a 1-GPU box).  No-op without WORLD_SIZE > 1.  -> (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world < 2:
        return rank, world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    backend = backend or os.environ.get("STNERF_DIST_BACKEND") or ("gloo" if single_device else "nccl")
    local = 0 if single_device else int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        if not single_device and torch.cuda.device_count() < world and backend == "nccl":
            raise RuntimeError(f"WORLD_SIZE={world} but this node exposes {torch.cuda.device_count()} GPU(s): one process per GPU "
                               "(check HIP_VISIBLE_DEVICES / the compute partition mode)")
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    return rank, world
