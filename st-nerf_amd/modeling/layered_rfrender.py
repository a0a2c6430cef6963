"""The layered model with the reference's constructor, attributes and forward signature
(modeling/layered_rfrender.py:19-741).

Everything numeric runs in the HIP library through ``stnerf_amd.ops`` (ONE call into the C ABI per launch
sequence, ``stnerf_render_rays``); this file only does what the reference does on the host: config, per-frame box
interpolation/edit on l x 8 x 3 numbers, chunk bookkeeping and output packing.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from stnerf_amd import ops
from stnerf_amd.modeling.motion_net import MotionNet
from stnerf_amd.modeling.spacenet import SpaceNet

Tensor = torch.Tensor


class LayeredRFRender(nn.Module):
    """modeling/layered_rfrender.py:19-741 -- same constructor, attributes and forward signature."""

    def __init__(self, cfg, camera_num, scale=None, shift=None):
        super().__init__()
        M = cfg.MODEL
        if M.SAMPLE_METHOD != "BBOX":
            raise NotImplementedError("only SAMPLE_METHOD 'BBOX' is on the render path (both shipped ymls)")
        unsupported = dict(POSE_REFINEMENT=M.POSE_REFINEMENT, USE_DEFORM_VIEW=M.USE_DEFORM_VIEW)
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError(f"config flags outside the MI355X hot path (training-time features): {bad}")
        if M.BKGD_USE_SPACE_TIME and not M.USE_SPACE_TIME:
            raise ValueError("BKGD_USE_SPACE_TIME needs USE_SPACE_TIME: the reference hands the background SpaceNet "
                             "its frame id only then (layered_rfrender.py:382-390) and fails on the missing input")
        if not (M.USE_DEFORM_TIME or M.USE_SPACE_TIME):
            raise ValueError("one of USE_DEFORM_TIME / USE_SPACE_TIME must be on (the reference dereferences a "
                             "missing frame id otherwise, layered_rfrender.py:193)")
        layer_num = cfg.DATASETS.LAYER_NUM
        self.coarse_ray_sample, self.fine_ray_sample = M.COARSE_RAY_SAMPLING, M.FINE_RAY_SAMPLING
        self.sample_method = M.SAMPLE_METHOD
        self.boarder_weight = M.BOARDER_WEIGHT
        self.scale, self.shift = scale, shift
        self.near, self.alpha = 0, 1
        self.pose_refinement = False
        self.layer_num, self.camera_num = layer_num, camera_num
        self.use_deform_view, self.bkgd_use_deform_time = False, bool(M.BKGD_USE_DEFORM_TIME)
        self.use_deform_time, self.use_space_time = M.USE_DEFORM_TIME, M.USE_SPACE_TIME
        self.bkgd_use_space_time = bool(M.BKGD_USE_SPACE_TIME)

        self.deep_rgb = deep = bool(M.DEEP_RGB and M.USE_SPACE_TIME)      # :35
        inc, use_dir = bool(M.TKERNEL_INC_RAW), bool(M.USE_DIR)          # :27-29
        common = dict(include_input=inc, use_dir=use_dir, deep_rgb=deep)
        self.bkgd_spacenet = SpaceNet(use_time=self.bkgd_use_space_time, **common)
        self.bkgd_spacenet_fine = SpaceNet(use_time=self.bkgd_use_space_time, **common)
        self.spacenets, self.spacenets_fine = nn.ModuleList([]), nn.ModuleList([])
        for i in range(layer_num):
            self.spacenets.append(SpaceNet(use_time=self.use_space_time, **common))
            self.spacenets_fine.append(self.spacenets[i] if M.SAME_SPACENET
                                       else SpaceNet(use_time=self.use_space_time, **common))
        self.time_deform_nets = nn.ModuleList([])
        if self.use_deform_time:
            for i in range(layer_num):
                self.time_deform_nets.append(MotionNet(c_input=4, include_input=inc, input_time=True))
        if self.bkgd_use_deform_time:                                  # :92-93 (input_time stays False)
            self.bkgd_time_deform_net = MotionNet(c_input=4, include_input=inc)
        # the reference initialises the fine / other-layer nets as deep copies (:63-74); keep that for a
        # freshly built model (a loaded checkpoint overwrites everything anyway)
        self.bkgd_spacenet_fine.load_state_dict(self.bkgd_spacenet.state_dict())
        for i in range(layer_num):
            self.spacenets[i].load_state_dict(self.spacenets[0].state_dict())
            self.spacenets_fine[i].load_state_dict(self.spacenets[i].state_dict())
        self.maxs = self.mins = None
        self.display_layers = {i: 1 for i in range(layer_num + 1)}
        self.bkgd_bbox = None
        self.bboxes = None
        # MI355X-side knobs (not in the reference)
        self.seed = 0                      # Philox seed of the on-device jitter / resampling draws
        self.fresh_draws_per_call = type(self).FRESH_DRAWS_DEFAULT  # True: every forward() advances `seed` (the reference draws fresh torch.rand
                                           # numbers per call, layers/RaySamplePoint.py:98, utils/sample_pdf.py:31);
                                           # False: a call is a pure function of (rays, weights, seed) -- what the
                                           # sharded / chunked renders and the tests rely on.  dropin.patch_reference
                                           # switches it on for models built through the reference's own code
        self.max_rays_per_launch = 1 << 19 # rays per kernel sequence (workspace bound ~20 KB/ray, not a semantic chunk)
        self.replay = None                 # {"jitter": (l,N,N1), "u": (l,N,N2)} to replay recorded uniforms (parity tests); the training path
                                           # also takes "z" (l,N,N2) / "xyz_c" / "xyz_f": the reference's own fine depths and deformed
                                           # points (teacher forcing, stnerf_amd.modeling.training.render_rays_train)
        self.mlp_schedule = "stage"        # "stage": one persistent MLP launch per stage (stnerf_mlp_stage); "per_net":
                                           # one launch per (layer, network) as in round 1 (A/B measurements)
        self.ray_window = (0, 0, 0)        # (first, stripe, period): which rays of the view `rays` are (include/stnerf.h);
                                           # keeps the RNG stream of a view under multi-GPU sharding
        self.shard_views = type(self).SHARD_VIEWS_DEFAULT
                                           # OPT-IN (default False): with an initialised torch.distributed group of > 1 ranks,
                                           # layered_batchify_ray / render_pose cut the view into interleaved stripes over the ranks
                                           # and all-gather the outputs -- a COLLECTIVE call every rank must make with the same
                                           # rays (stnerf_amd.parallel).  The `python -m stnerf_amd.dropin` launcher under
                                           # torch.distributed.run and bench.py switch it on
        self.gather = "all"                # what a sharded call all-gathers: "all" (the whole 5-tuple), "fine" (what render_pose
                                           # consumes: mixed + per-layer fine images, masks), "final" (the two mixed images);
                                           # entries that are not gathered come back as None
        self.shard_group = None            # the process group to shard over (None: the default group)

    FRESH_DRAWS_DEFAULT = False   # what a new model's fresh_draws_per_call starts as (dropin.patch_reference: True)
    SHARD_VIEWS_DEFAULT = False   # what a new model's shard_views starts as (the dropin launcher under torch.distributed.run: True)

    def set_precision(self, precision: str):
        """"bf16x3" (the default: three bf16 pieces per fp32 operand, six MFMAs per product, two accumulators: fp32's
        significand and range, closer to an fp64 evaluation than an fp32 fma chain, 1.5 x the speed) or "fp32" (exact f32
        MFMA, v_mfma_f32_32x32x2_f32)."""
        if precision not in ops.PRECISIONS:
            raise ValueError(f"precision must be one of {ops.PRECISIONS}")
        for m in self.modules():
            if isinstance(m, (SpaceNet, MotionNet)):
                m.precision = precision
        return self

    # ---- reference API -----------------------------------------------------------------------
    def hide_layer(self, layer_id):
        self.display_layers[layer_id] = 0

    def show_layer(self, layer_id):
        self.display_layers[layer_id] = 1

    def is_shown_layer(self, layer_id):
        return self.display_layers[layer_id] == 1

    def set_bkgd_bbox(self, bbox):
        self.bkgd_bbox = bbox

    def set_bboxes(self, bboxes):
        self.bboxes = bboxes

    def set_bkgd_near_far(self, near_far):
        self.bkgd_near_torchfar = near_far

    def set_max_min(self, maxs, mins):
        self.maxs, self.mins = maxs, mins

    def bbox_interpolation(self, float_frame_id, layer_id):
        start = self.bboxes[math.floor(float_frame_id), layer_id]
        end = self.bboxes[math.ceil(float_frame_id), layer_id]
        return torch.lerp(start, end, float_frame_id - math.floor(float_frame_id))

    # ---- host-side scene maths (l x 8 x 3 numbers, same torch ops as the reference) -------------
    def _pivot(self):
        """Edit pivot, layered_rfrender.py:216-232: mean of the frame-0 centres of layers 1 and 2, with each
        centre's z replaced by corner-1's z."""
        first = torch.cat([self.bkgd_bbox.detach().cpu().float(), self.bboxes[0].detach().cpu().float()], 0)
        centre = torch.mean(first, 1)
        centre[:, 2] = first[:, 1, 2]
        return (centre[2] + centre[1]) / 2

    def _edit_boxes(self, boxes):
        """boxes (..., l, 8, 3) on any device; returns edited boxes and the pivot (:230-242)."""
        pivot = None
        if self.scale is not None:
            pivot = self._pivot()
            pv = pivot.to(boxes.device)
            for i in range(len(self.scale)):
                boxes[..., i, :, :] = (boxes[..., i, :, :] - pv) * self.scale[i] + pv
        if self.shift is not None:
            for i in range(len(self.shift)):
                if self.shift[i] is None:
                    continue
                boxes[..., i, :, :] += torch.tensor(self.shift[i], dtype=torch.float32, device=boxes.device)
        return boxes, pivot

    def _point_edits(self, l, fine):
        """Per-layer inverse edit for sample points: coarse :293-303; fine :467-475 (where a None shift
        skips that layer's scale step as well)."""
        if self.shift is None and self.scale is None:
            return None
        out = []
        for i in range(l):
            sh = sc = None
            if self.shift is not None:
                if fine:
                    if self.shift[i] is None:
                        out.append((None, None))
                        continue
                    sh = self.shift[i]
                elif i < len(self.shift):
                    sh = self.shift[i]
            if self.scale is not None and (fine or i < len(self.scale)):
                sc = self.scale[i]
            out.append((sh, sc))
        return out

    def _retimed_boxes(self, row0_frame_ids):
        """One box per layer for a (reference) chunk from ROW 0's frame ids (:195-208), edited.
        row0_frame_ids: CPU fp32 tensor (l,) = rays[0, 6:]."""
        L = self.layer_num
        bb = self.bboxes.detach().cpu().float()
        per = torch.zeros(L, 8, 3)
        for i in range(L):
            f = row0_frame_ids[i + 1] - 1                        # fp32 0-dim tensor, as in the reference
            per[i] = torch.lerp(bb[math.floor(f), i], bb[math.ceil(f), i], f - math.floor(f))
        boxes = torch.cat([self.bkgd_bbox.detach().cpu().float(), per], 0)
        return self._edit_boxes(boxes)

    # ---- the chunk pipeline ------------------------------------------------------------------------
    def _nets(self, fine):
        return (self.bkgd_spacenet_fine, self.spacenets_fine) if fine else (self.bkgd_spacenet, self.spacenets)

    def _stage(self, rays, xyz, raw, lst, cnt, times_col, fine):
        """Deform + evaluate every layer's network on its (masked) rays through the op-level entry points
        (:340-418 / :495-576).  The render path itself runs this inside stnerf_render_rays; this op-level
        composition is kept for stage-by-stage debugging (tools/debug_stages.py)."""
        l = self.layer_num + 1
        bk, nets = self._nets(fine)
        if self.use_deform_time:
            for i in range(1, l):
                if not self.is_shown_layer(i):
                    continue  # a hidden layer's points are never consumed
                ops.motionnet_fwd(self.time_deform_nets[i - 1]._packed("fp32"), xyz[:, i], rays[:, times_col(i)],
                                  add_to_xyz=True, ray_list=lst[i], ray_count=cnt[i:i + 1])
        ops.spacenet_fwd(bk._packed("fp32"), xyz[:, 0], rays[:, 3:6], None, raw[:, 0])
        for i in range(1, l):
            if not self.is_shown_layer(i):
                continue
            tm = rays[:, times_col(i)] if self.use_space_time else None
            ops.spacenet_fwd(nets[i - 1]._packed("fp32"), xyz[:, i], rays[:, 3:6], tm, raw[:, i], ray_list=lst[i],
                             ray_count=cnt[i:i + 1])

    def _render_launch(self, rays, boxes, pivot, retiming, only_coarse, thr, bthr, window, replay):
        """One kernel sequence over `rays` (n <= max_rays_per_launch) = ONE call into the C ABI
        (stnerf_render_rays, csrc/pipeline.hip).  boxes: (l,8,3) shared or (n,l,8,3)."""
        from stnerf_amd import hip
        n, l = rays.shape[0], self.layer_num + 1
        p = hip.RenderParams()
        p.l, p.n1, p.n2, p.ray_stride = l, self.coarse_ray_sample, self.fine_ray_sample, rays.shape[1]
        p.retiming, p.only_coarse = int(retiming), int(only_coarse)
        p.use_deform_time, p.use_space_time = int(self.use_deform_time), int(self.use_space_time)
        p.bkgd_use_deform_time, p.bkgd_use_space_time = int(self.bkgd_use_deform_time), int(self.bkgd_use_space_time)
        p.deep_rgb = int(self.deep_rgb)
        # 3: bf16x3 (the default); 0: exact f32, one persistent launch per network stage; 2: exact f32, one launch per network
        prec = self.bkgd_spacenet.precision
        p.precision = 3 if prec == "bf16x3" else (0 if self.mlp_schedule == "stage" else 2)
        for i in range(l):
            p.shown[i] = int(self.is_shown_layer(i))
        p.border, p.near, p.alpha = float(self.boarder_weight), float(self.near), float(self.alpha)
        p.density_threshold, p.bkgd_density_threshold = float(thr), float(bthr)
        p.seed = int(self.seed) & 0xFFFFFFFFFFFFFFFF
        p.ray_index_base, p.ray_index_stripe, p.ray_index_period = (int(x) for x in window)
        ec, ef = self._point_edits(l, False), self._point_edits(l, True)
        p.has_edits = int(ec is not None)
        ops.fill_edits(p.edits_coarse, ec, l)
        ops.fill_edits(p.edits_fine, ef, l)
        if pivot is not None:
            p.pivot[0], p.pivot[1], p.pivot[2] = pivot.tolist()
        nets = hip.Nets()
        keep = []                                                    # packed blobs must outlive the enqueue
        def ptr(module):
            pk = module._packed(prec)
            keep.append(pk)
            return pk.blob.data_ptr()
        nets.bkgd, nets.bkgd_fine = ptr(self.bkgd_spacenet), ptr(self.bkgd_spacenet_fine)
        if self.bkgd_use_deform_time:
            nets.motion[0] = ptr(self.bkgd_time_deform_net)
        for i in range(1, l):
            if not self.is_shown_layer(i):
                continue
            nets.space[i], nets.space_fine[i] = ptr(self.spacenets[i - 1]), ptr(self.spacenets_fine[i - 1])
            if self.use_deform_time:
                nets.motion[i] = ptr(self.time_deform_nets[i - 1])
        need = ops.render_workspace_bytes(n, l, p.n1, p.n2, only_coarse)
        ws = getattr(self, "_workspace", None)
        if ws is None or ws.numel() < need or ws.device != rays.device:
            self._workspace = ws = torch.empty(need, dtype=torch.uint8, device=rays.device)
        return ops.render_rays(rays, boxes, nets, p, ws, jitter=replay["jitter"] if replay else None,
                               u=(replay.get("u") if replay else None))

    def render_rays(self, rays, only_coarse=False, density_threshold=0.0001, bkgd_density_threshold=0.0,
                    ref_chunk: Optional[int] = None):
        """Render all `rays`; ``ref_chunk`` reproduces the reference's chunk semantics (boxes are taken
        from row 0 of every ``ref_chunk``-ray piece) while launching kernels over far larger pieces."""
        return self.as_reference_tuple(self.render_rays_raw(rays, only_coarse, density_threshold, bkgd_density_threshold,
                                                            ref_chunk))

    def as_reference_tuple(self, raw):
        """(mixed_fine (n,5), mixed_coarse (n,5), layer_fine (n,l,5), layer_coarse (n,l,5), mask (n,l)) -> the
        reference's 5-tuple of (color (n,3), depth (n,1), acc (n,1)) triples and bool masks (layered_rfrender.py:725-734)."""
        mix_f, mix_c, lo_f, lo_c, mask = raw
        l = self.layer_num + 1
        trip = lambda x: None if x is None else (x[:, 0:3], x[:, 3:4], x[:, 4:5])
        fine_layer = None if lo_f is None else [trip(lo_f[:, i]) for i in range(l)]
        coarse_layer = None if lo_c is None else [trip(lo_c[:, i]) for i in range(l)]
        ray_mask = None if mask is None else [mask[:, i].bool() for i in range(l)]
        return trip(mix_f), trip(mix_c), fine_layer, coarse_layer, ray_mask

    def _warn_if_eval_with_grad(self):
        """the reference builds an autograd graph in eval() mode too; here training is gated on train() (the inference
        kernels save nothing), so a fine-tuning script that never calls model.train() would get outputs without history.  Say so once."""
        if (torch.is_grad_enabled() and not self.training and not getattr(self, "_warned_eval_grad", False)
                and any(p.requires_grad for p in self.parameters())):
            import warnings
            self._warned_eval_grad = True
            warnings.warn("LayeredRFRender.forward in eval() mode with autograd enabled and trainable parameters: the MI355X inference "
                          "kernels run and the outputs carry NO autograd history (the reference would build a graph here).  Call "
                          "model.train() to train, or wrap rendering in torch.no_grad() to silence this.", RuntimeWarning, stacklevel=3)

    def advance_seed(self):
        """What a finished call does to ``seed`` (``fresh_draws_per_call``; a rank that owns no ray of a sharded view
        calls this too, so that every rank's stream stays the same)."""
        if self.fresh_draws_per_call and self.replay is None:
            self.seed = (int(self.seed) + 1) & 0xFFFFFFFFFFFFFFFF   # the next call draws new jitter / resampling numbers

    def render_rays_raw(self, rays, only_coarse=False, density_threshold=0.0001, bkgd_density_threshold=0.0,
                        ref_chunk: Optional[int] = None):
        """``render_rays`` before the outputs are cut into the reference's triples: the five tensors the library
        wrote (what stnerf_amd.parallel packs into its one all-gather)."""
        if not rays.is_cuda:
            raise RuntimeError("rays must live on the GPU: the MI355X render path has no CPU fallback")
        rays = rays.contiguous().float()
        N, L = rays.shape[0], self.layer_num
        width = rays.shape[1]
        if width == 7:
            retiming = False
        elif width == 7 + L:
            retiming = True
        else:
            raise ValueError(f"undefined ray format in LayeredRFRender, ray dimension is {width}")
        if self.bkgd_bbox is None or self.bboxes is None:
            raise RuntimeError("set_bkgd_bbox / set_bboxes must be called before rendering")
        if N == 0:  # the reference dereferences row 0 (rays_frame_id[0, i+1], layered_rfrender.py:200)
            raise IndexError("empty ray batch: LayeredRFRender needs at least one ray")
        # BKGD_USE_SPACE_TIME on a batch that MIXES background frame ids (training rays; off in both shipped ymls): the reference tiles
        # the ids over the samples (modeling/spacenet.py:117-118 with the 1-D tensor of layered_rfrender.py:380,385), so every sample of
        # layer 0 has its own time.  The fused pipeline keeps one time per ray; such a call takes the op-by-op path below, which
        # evaluates the background per sample (stnerf_amd.modeling.training._stage).  One id per call -- every rendered frame -- is the
        # identity and stays on the fused pipeline.
        per_sample_bkgd_time = False
        if self.bkgd_use_space_time and self.use_space_time:
            from stnerf_amd.modeling import training as _training
            per_sample_bkgd_time = _training.mixed_bkgd_ids(rays)
        # Training: model.train() + autograd enabled + trainable parameters = what engine/layered_trainer.py:186-194
        # sets up -> the same stages launched op by op with autograd history (stnerf_amd.modeling.training).  In eval() mode the
        # inference kernels run and the outputs carry no history, as under torch.no_grad() (render/layered_neural_renderer.py:377).
        train = torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters())
        self._warn_if_eval_with_grad()
        step = N if ref_chunk is None else ref_chunk
        groups = []  # (start, end, boxes, pivot)
        if retiming:
            row0 = rays[0::step, 6:].cpu()  # one D2H: row-0 frame ids of every reference chunk
            n_chunks, c0 = row0.shape[0], 0
            for c in range(1, n_chunks + 1):  # merge runs of reference chunks that share their boxes
                if c == n_chunks or not torch.equal(row0[c], row0[c0]):
                    boxes, pivot = self._retimed_boxes(row0[c0])
                    groups.append((c0 * step, min(c * step, N), boxes.to(rays.device), pivot))
                    c0 = c
        else:
            fid = rays[:, 6].to(torch.int64) - 1
            bb = self.bboxes.to(rays.device).float().index_select(0, fid)                          # :193
            bk = self.bkgd_bbox.to(rays.device).float().unsqueeze(0).expand(N, 1, 8, 3)
            boxes, pivot = self._edit_boxes(torch.cat([bk, bb], 1).contiguous())
            groups.append((0, N, boxes, pivot))
        outs = []
        cap = self.max_rays_per_launch
        first, stripe, period = self.ray_window
        if stripe > 0:                     # launch pieces must start on a stripe boundary
            cap = max(stripe, cap // stripe * stripe)
            if any(g0 % stripe for (g0, _, _, _) in groups):
                raise ValueError("a striped ray window needs chunk groups that start on a stripe boundary "
                                 "(frame ids that change inside the view: render it unstriped)")
        window_at = (lambda s: (first + s, 0, 0)) if stripe <= 0 else (lambda s: (first + s // stripe * period, stripe, period))
        for (g0, g1, boxes, pivot) in groups:
            for s in range(g0, g1, cap):
                e = min(s + cap, g1)
                bx = boxes if boxes.dim() == 3 else boxes[s:e].contiguous()
                rp = None
                if self.replay is not None:
                    # jitter / u / z: (l, N, *) per-ray draws, cut with the rays; xyz_c / xyz_f (teacher forcing, parity tests only): per
                    # performer the deformed points of its hit rays -- whole-batch lists, one launch piece only
                    rp = {k: (v[:, s:e].contiguous() if torch.is_tensor(v) else v) for k, v in self.replay.items()}
                    if any(not torch.is_tensor(v) for v in rp.values()) and (s, e) != (0, N):
                        raise ValueError("replayed deformed points (xyz_c / xyz_f) need the whole batch in one launch piece")
                if train or per_sample_bkgd_time:
                    from stnerf_amd.modeling.training import render_rays_train
                    outs.append(render_rays_train(self, rays[s:e], bx, pivot, retiming, only_coarse, density_threshold,
                                                  bkgd_density_threshold, window_at(s), rp))
                else:
                    outs.append(self._render_launch(rays[s:e], bx, pivot, retiming, only_coarse, density_threshold,
                                                    bkgd_density_threshold, window_at(s), rp))
        cat = (lambda j: outs[0][j]) if len(outs) == 1 else (lambda j: torch.cat([o[j] for o in outs], 0))
        raw = tuple(cat(j) for j in range(5))
        self.advance_seed()
        return raw

    def forward(self, rays, labels=None, bboxes=None, only_coarse=False, near_far=None, near_far_points=[],
                density_threshold=0.0001, bkgd_density_threshold=0):
        """(fine_mixed, coarse_mixed, fine_layer[l], coarse_layer[l], ray_mask[l]); every entry a
        (color (N,3), depth (N,1), acc (N,1)) triple.  layered_rfrender.py:141-734.
        labels / bboxes / near_far / near_far_points are accepted and ignored, as in the BBOX path."""
        return self.render_rays(rays, only_coarse, density_threshold, bkgd_density_threshold, ref_chunk=None)
