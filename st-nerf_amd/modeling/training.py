"""``LayeredRFRender.forward`` under autograd: what ``loss.backward()`` of
engine/layered_trainer.py:192-282 needs from modeling/layered_rfrender.py:141-734.

The same stages as the inference pipeline (csrc/pipeline.hip), launched op by op so that autograd can hold on to what lies
between them.  Everything numeric is a HIP kernel behind ``stnerf_amd.ops``; PyTorch owns the buffers, gathers / scatters the
hit rays of a layer (index_select / index_copy: data movement) and strings the ``torch.autograd.Function``s together:

    sampler (stnerf_sample_coarse; detached in the reference, :314-315)
      -> per performer: MotionNetFunction on the hit rays (:340-356), xyz += flow
      -> SpaceNetFunction per layer (:382-413)
      -> CompositeFunction: density edits + per-layer composites + depth merge + merged composite (:414-448)
    inverse-CDF resampling (stnerf_resample; detached, :460-461) -> the same three steps with the fine networks (:495-606)

Gradients reach every weight and bias of the SpaceNets and MotionNets; rays, boxes, depths and frame ids get none, as in the
reference (its POSE_REFINEMENT / USE_DEFORM_VIEW paths are outside SURVEY section 8).  The training forward runs the exact-f32
MFMA kernels whatever ``model.set_precision`` says for rendering: the backward walks back through the ReLU masks of the evaluation
that produced the loss (the forward kernels write them out as bit planes).
"""
from __future__ import annotations

from typing import Optional

import torch

from stnerf_amd import ops


class CompositeFunction(torch.autograd.Function):
    """(layer_out (n,l,5), mixed_out (n,5), weights (n,l,S)) = composite(t, raw, mask); d raw from d layer_out / d mixed_out
    (stnerf_composite / stnerf_composite_bwd).  ``weights`` is what the resampler reads: detached in the reference (:460)."""

    @staticmethod
    def forward(ctx, t, raw, mask, params):
        raw_c = raw.detach().contiguous()
        layer_out, mixed_out, weights, order = ops.composite(t, raw_c, mask, want_weights=True, want_order=True, two_pass=False,
                                                             params=params)
        ctx.save_for_backward(t, raw_c, mask, order)
        ctx.params = params
        ctx.mark_non_differentiable(weights)
        return layer_out, mixed_out, weights

    @staticmethod
    def backward(ctx, g_layer, g_mixed, _g_weights):
        t, raw, mask, order = ctx.saved_tensors
        gl = g_layer.contiguous().float() if g_layer is not None else None
        gm = g_mixed.contiguous().float() if g_mixed is not None else None
        if gl is None and gm is None:
            return None, None, None, None
        return None, ops.composite_bwd(t, raw, mask, order, ctx.params, gl, gm), None, None


def mixed_bkgd_ids(rays) -> bool:
    """More than one background frame id (column 6) in this call (a host synchronisation; only asked with BKGD_USE_SPACE_TIME)."""
    return bool((rays[:, 6] != rays[0, 6]).any())


def _stage(model, rays, xyz, hit, times_col, fine: bool, forced=None):
    """MotionNet + SpaceNet of every layer on a stage's points xyz (n,l,ns,3) -> raw (n,l,ns,4) with autograd history.
    modeling/layered_rfrender.py:340-418 (coarse) / :495-576 (fine).  ``forced[i]`` (teacher forcing, parity tests): the deformed
    points the REFERENCE handed performer i's SpaceNet, (hits, ns, 3); the SpaceNet is evaluated on exactly those, the deformation
    net keeps its place in the graph (its flow's VALUE is replaced, its gradient is the SpaceNet's d pos)."""
    n, l, ns = xyz.shape[0], xyz.shape[1], xyz.shape[2]
    bk, nets = model._nets(fine)
    dirs = rays[:, :6]
    raws = []
    # background: every ray (:382-392); its deformation net only with BKGD_USE_DEFORM_TIME (:358-367)
    x0 = xyz[:, 0]
    if model.bkgd_use_deform_time:
        tcol = rays[:, times_col(0)].reshape(n, 1, 1).expand(n, ns, 1)
        x0 = x0 + model.bkgd_time_deform_net(torch.cat([x0, tcol], -1))
    tm0 = rays[:, times_col(0)].reshape(n, 1) if model.bkgd_use_space_time else None
    if tm0 is not None and mixed_bkgd_ids(rays):
        # The reference hands the background net its frame ids as a 1-D tensor (modeling/layered_rfrender.py:380,385 and the fine
        # twins) and SpaceNet.forward TILES them over the samples (modeling/spacenet.py:117-118: unsqueeze(1).repeat(1, L, 1) of an
        # (n,) tensor, then reshape(-1, 1)): sample j of ray i is evaluated at the id of ray (i ns + j) mod n.  With one id per call
        # that is the identity; on a batch that mixes them (training rays) every sample has its own time, so every sample is
        # evaluated as a one-sample ray: its own point, its ray's direction, the tiled id.
        flat = torch.arange(n * ns, device=rays.device)
        rgb, sig = bk(x0.reshape(n * ns, 1, 3), dirs.index_select(0, flat // ns), tm0.reshape(n).index_select(0, flat % n).reshape(n * ns, 1))
        rgb, sig = rgb.reshape(n, ns, 3), sig.reshape(n, ns, 1)
    else:
        rgb, sig = bk(x0, dirs, tm0)
    raws.append(torch.cat([rgb, sig], -1))
    for i in range(1, l):
        zero = torch.zeros(n, ns, 4, dtype=torch.float32, device=xyz.device)      # the reference's zero tensors (:398-399)
        idx = hit[i]
        if idx.numel() == 0 or not model.is_shown_layer(i):
            raws.append(zero)
            continue
        pos = xyz[:, i].index_select(0, idx)
        r_i = rays.index_select(0, idx)
        tm = r_i[:, times_col(i)].reshape(-1, 1)
        if model.use_deform_time:
            flow = model.time_deform_nets[i - 1](torch.cat([pos, tm.reshape(-1, 1, 1).expand(-1, ns, 1)], -1))   # :340-356
            pos = pos + flow
            if forced is not None and forced[i] is not None:
                pos = forced[i] + (pos - pos.detach())       # the recorded points bit for bit (x - x = 0 exactly), d / d flow = identity
        rgb, sig = nets[i - 1](pos, r_i[:, :6], tm if model.use_space_time else None)
        raws.append(zero.index_copy(0, idx, torch.cat([rgb, sig], -1)))
    return torch.stack(raws, 1)


def _composite_params(model, fine: bool, retiming: bool, thr: float, bthr: float):
    """The stage's stnerf_composite_params, as csrc/pipeline.hip fills them (raw network outputs: sigmoid applied there)."""
    l = model.layer_num + 1
    if not fine:
        thresholds = [None] + [float(thr) if retiming else None] * (l - 1)                       # :416-418
        scale = None
    else:
        thresholds = [float(bthr) if retiming else None] + [float(thr) if retiming else None] * (l - 1)   # :538-547, :564-566
        scale = [1.0] * l
        if l > 2:
            scale[2] = float(model.alpha)                                                          # :575-576
    evaluated = [2] + [int(model.is_shown_layer(i)) for i in range(1, l)]
    return ops.composite_params(border=float(model.boarder_weight), near=float(model.near), fine=fine, cut_negative_t=not fine,
                                thresholds=thresholds, sigma_scale=scale, evaluated=evaluated, rgb_activated=False)


def render_rays_train(model, rays, boxes, pivot, retiming: bool, only_coarse: bool, thr: float, bthr: float, window,
                      replay: Optional[dict]):
    """One launch piece of ``LayeredRFRender.render_rays_raw`` with autograd history: the five raw tensors
    (mixed_fine (n,5), mixed_coarse (n,5), layer_fine (n,l,5), layer_coarse (n,l,5), mask (n,l) uint8)."""
    n, l = rays.shape[0], model.layer_num + 1
    n1, n2 = model.coarse_ray_sample, model.fine_ray_sample
    times_col = (lambda i: 6 + i) if retiming else (lambda i: 6)
    ec, ef = model._point_edits(l, False), model._point_edits(l, True)
    first, stripe, period = (int(x) for x in window)
    with torch.no_grad():
        t_c, xyz_c, mask = ops.sample_coarse(rays, boxes, n1, jitter=replay["jitter"] if replay else None, seed=int(model.seed),
                                             ray_index_base=first, edits=ec, pivot=pivot, ray_index_stripe=stripe,
                                             ray_index_period=period)
    # (the 0 / 1 mask, without the sampler's "missed" hints: with hints the resampler leaves a missed pair's fine depths
    # unwritten, and the backward walks the merged list through every source sample's depth)
    mask01 = mask
    # the hit rays of every performer, once for both stages (a nonzero is a host synchronisation: the output's size is data)
    hit = [None] + [mask01[:, i].nonzero(as_tuple=True)[0] for i in range(1, l)]
    forced = replay or {}
    raw_c = _stage(model, rays, xyz_c, hit, times_col, False, forced.get("xyz_c"))
    layer_c, mixed_c, w_c = CompositeFunction.apply(t_c, raw_c, mask, _composite_params(model, False, retiming, thr, bthr))
    if only_coarse:
        # layered_rfrender.py:704-723: the "fine" entries are the coarse ones
        return mixed_c, mixed_c, layer_c, layer_c, mask01
    with torch.no_grad():
        if "z" in forced:
            # teacher forcing: the reference's own new depths instead of a resampling of THESE coarse weights (whose last bits differ:
            # utils/sample_pdf.py:58-59 divides by cdf differences down to 1e-5).  The rest as modeling/layered_rfrender.py:462-465:
            # sort(cat[t, z]), z d + o -- separate IEEE multiplies and adds, the reference's values bit for bit
            if ef is not None:
                raise ValueError("replayed fine depths with box edits: not needed by any fixture")
            t_f = torch.sort(torch.cat([t_c, forced["z"].permute(1, 0, 2).to(t_c.dtype)], -1), -1)[0].contiguous()
            xyz_f = (t_f.unsqueeze(-1) * rays[:, None, None, 3:6] + rays[:, None, None, 0:3]).contiguous()
        else:
            t_f, xyz_f = ops.resample(t_c, w_c, n2, rays, u=replay.get("u") if replay else None, seed=int(model.seed),
                                      ray_index_base=first, edits=ef, pivot=pivot, ray_index_stripe=stripe, ray_index_period=period,
                                      mask=None)
    raw_f = _stage(model, rays, xyz_f, hit, times_col, True, forced.get("xyz_f"))
    layer_f, mixed_f, _ = CompositeFunction.apply(t_f, raw_f, mask, _composite_params(model, True, retiming, thr, bthr))
    return mixed_f, mixed_c, layer_f, layer_c, mask01
