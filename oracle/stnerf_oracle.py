"""CPU oracle for the st-nerf layered ray-march hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain functional torch-on-CPU code, the algorithm of the reference
renderer's hot path.  It is the *checker* for the HIP
kernels: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  Nothing under ``st-nerf_amd/`` imports it and the product path never falls back
to it.

Parity pinning: the reference ships no golden vectors / KATs for this path, so
the oracle is pinned against *outputs of the reference itself*: ``tests/golden/make_golden.py``
imports ``/root/reference`` in the build container, runs the reference's own functions on
seeded inputs with its ``torch.rand`` draws recorded, and commits inputs + outputs as ``.npz``
fixtures.  ``tests/test_oracle_golden.py`` replays those fixtures through this file.

All arithmetic is the reference's: ATen ops in the dtype of the inputs (fp32 in production;
pass float64 tensors to get an fp64 evaluation of the same formulas for error analysis).
Every function cites the reference file:line it follows (paths relative to /root/reference).

One thing is NOT pinned because the reference does not define it: the order of two samples of DIFFERENT layers
at exactly the same depth in the merged lists (modeling/layered_rfrender.py:425,587 call torch.sort with its
default stable=False, and ATen's CPU sort is not stable -- a probe kept the lower layer first on 1162 of 2000
rows; CUDA's radix sort is).  It happens on about 1 ray in 1000 at 3 x 90 samples and moves that ray's merged
colour by up to 1e-3.  This file calls torch.sort as the reference does (so on such a ray it follows ATen's CPU
algorithm); the HIP compositor merges in the order of a stable sort; make_golden.py keeps such rays out of the
one fixture large enough to contain them (rays_with_depth_ties).

Teacher forcing (``render_chunk(forced=...)``, round 6): the fine depths and deformed points of a recorded
reference run replace the oracle's own, so that an fp64 evaluation sits on the reference's positions.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

FLOAT64_EPS = 2.220446049250313e-16  # np.finfo(float).eps, layers/RaySamplePoint.py:17-22


# --------------------------------------------------------------------------------------
# a1: pinhole ray generation            utils/render_helpers.py:42-128, utils/ray_sampling.py:22-72
# --------------------------------------------------------------------------------------
def generate_rays(K: Tensor, T: Tensor, h: int, w: int) -> Tensor:
    """Rays of one full view, row-major over (row, col): [origin(3), direction(3)].

    Pixel centres are the integer coordinates (u = col, v = row, no +0.5)
    (render_helpers.py:96-102); direction = normalise(K^-1 [u, v, 1]) rotated by T[:3,:3]
    (:105-114); origin = T[:3,3] (:116-117).  The bbox-crop branch (:44-82) is not on the render
    path (called with bbox=None from data/datasets/ray_dataset.py:263).
    """
    rows = torch.linspace(0, h - 1, steps=h, dtype=K.dtype)
    cols = torch.linspace(0, w - 1, steps=w, dtype=K.dtype)
    grid_r, grid_c = torch.meshgrid(rows, cols, indexing="ij")
    pix = torch.stack([grid_c, grid_r, torch.ones_like(grid_c)], dim=-1).unsqueeze(-1)  # (h,w,3,1)
    cam_dir = torch.matmul(torch.inverse(K), pix)  # (h,w,3,1)
    cam_dir = cam_dir / torch.norm(cam_dir, dim=2, keepdim=True)
    cam_dir = torch.cat([cam_dir, torch.zeros(h, w, 1, 1, dtype=K.dtype)], dim=2)  # homogeneous, w=0
    world_dir = torch.matmul(T, cam_dir)[:, :, 0:3, 0]  # (h,w,3)
    origin = T[0:3, 3].reshape(1, 1, 3).repeat(h, w, 1)
    return torch.cat([origin, world_dir], dim=2).reshape(-1, 6)


def append_frame_ids(rays: Tensor, layer_frame_pair: Sequence[Tuple[int, float]], layer_num: int) -> Tensor:
    """a2: data/datasets/ray_dataset.py:276-281 -- rays (N,6) -> (N, 6 + layer_num + 1)."""
    frame_ids = torch.zeros(rays.size(0), layer_num + 1, dtype=rays.dtype)
    for layer_id, frame_id in layer_frame_pair:
        frame_ids[:, layer_id] = frame_id
    return torch.cat([rays, frame_ids], dim=-1)


# --------------------------------------------------------------------------------------
# a5/a6: ray / box slab test and stratified coarse sampler       layers/RaySamplePoint.py:8-107
# --------------------------------------------------------------------------------------
def intersection(rays: Tensor, bbox: Tensor) -> Tensor:
    """(far, near) ray parameters of each ray against its 8-corner box; (-1000,-1000) on a miss.

    rays (n, >=6), bbox (n, 8, 3).  Six plane hits t = (face - o) / (d + eps) (:17-22), each kept
    only if its hit point lies inside the face rectangle, inclusive, using the corner pairs
    (0,7) (1,6) (0,5) (3,6) (0,2) (4,6) (:34-51); rejected faces count as -1000 (:53-59); the two
    largest survive, [:,0] = far, [:,1] = near (:60-62).
    """
    o, d = rays[:, 0:3], rays[:, 3:6]
    eps = FLOAT64_EPS
    # (face plane coordinate, axis) in the reference's column order: left,right,front,back,bottom,up
    planes = [(bbox[:, 0, 0], 0), (bbox[:, 6, 0], 0), (bbox[:, 0, 1], 1),
              (bbox[:, 6, 1], 1), (bbox[:, 0, 2], 2), (bbox[:, 6, 2], 2)]
    # (lo corner, hi corner, the two axes tested) per face, :34-51
    rect = [(0, 7, (1, 2)), (1, 6, (1, 2)), (0, 5, (0, 2)), (3, 6, (0, 2)), (0, 2, (0, 1)), (4, 6, (0, 1))]
    cols = []
    for (plane, axis), (lo, hi, axes) in zip(planes, rect):
        t = (plane - o[:, axis]) / (d[:, axis] + eps)
        hit = t.unsqueeze(1) * d + o
        inside = torch.ones_like(t, dtype=torch.bool)
        for a in axes:
            inside = inside & (hit[:, a] >= bbox[:, lo, a]) & (hit[:, a] <= bbox[:, hi, a])
        cols.append(torch.where(inside, t, torch.full_like(t, -1e3)))
    tl = torch.stack(cols, dim=1)
    return tl.topk(k=2, dim=-1)[0]


def sample_coarse(rays: Tensor, boxes: Tensor, n_coarse: int, jitter: Sequence[Tensor]):
    """RaySamplePoint.forward (:70-107).  boxes (n, l, 8, 3); jitter[i] (n, n_coarse) in [0,1).

    Returns lists over layers of t (n,N1,1), xyz (n,N1,3), mask (n,) bool.
    Layer 0 clamps a non-positive near hit to 0 (:93-95); performers keep it.  mask = |bin|>1e-5.
    """
    n, l = rays.shape[0], boxes.shape[1]
    k = torch.arange(0, n_coarse, dtype=rays.dtype).reshape(1, n_coarse)
    ts, pts, masks = [], [], []
    for i in range(l):
        far_near = intersection(rays, boxes[:, i])
        start = far_near[:, 1].reshape(n, 1).clone()
        if i == 0:
            start[start <= 0] = 0
        end = far_near[:, 0].reshape(n, 1)
        width = (end - start) / n_coarse
        t = ((k + jitter[i]) * width + start).unsqueeze(-1)
        ts.append(t)
        pts.append(t * rays[:, 3:6].unsqueeze(1) + rays[:, 0:3].unsqueeze(1))
        masks.append((torch.abs(width) > 1e-5).reshape(n))
    return ts, pts, masks


# --------------------------------------------------------------------------------------
# a7: positional encoding                                         utils/dimension_kernel.py:3-73
# --------------------------------------------------------------------------------------
def positional_encoding(x: Tensor, n_freq: int, include_input: bool = True) -> Tensor:
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)], each block d wide."""
    out = [x] if include_input else []
    for f in range(n_freq):
        freq = float(2.0 ** f)  # 2.**linspace(0, L-1, L), dimension_kernel.py:20
        out.append(torch.sin(x * freq))
        out.append(torch.cos(x * freq))
    return torch.cat(out, dim=-1)


# --------------------------------------------------------------------------------------
# a8: MotionNet                                                   modeling/motion_net.py:7-71
# --------------------------------------------------------------------------------------
def _linear(params: Dict[str, Tensor], key: str, x: Tensor) -> Tensor:
    return F.linear(x, params[key + ".weight"], params[key + ".bias"])


def motion_net(params: Dict[str, Tensor], prefix: str, xyzt: Tensor, input_time: bool = True) -> Tensor:
    """Scene flow of samples [x,y,z,t] (..., 4) -> (..., 3).  input_time=False (the flavour of
    bkgd_time_deform_net, layered_rfrender.py:92-93) encodes [xyz, t] as given (:61-62).

    PE_10 of the 4-vector (84 wide); for fractional t the encoding is the lerp of the encodings
    at floor(t) and floor(t)+1 (:49-60) -- taken for the WHOLE batch if any t is fractional
    (:53), which is value-identical to a per-row rule because the lerp weight is 0 on integer
    rows.  MLP 84-128-128-128-128-128-3, ReLU between (:20-32).
    """
    shape = xyzt.shape
    x = xyzt.reshape(-1, 4)
    xyz, t = x[:, :3], x[:, 3:]
    inc = params[f"{prefix}.motion_net.0.weight"].shape[1] == 84   # include_input (TKERNEL_INC_RAW), else 80
    lower = torch.floor(t)
    if input_time and not torch.all(torch.eq(lower, t)):
        w = t - lower
        enc = (1 - w) * positional_encoding(torch.cat([xyz, lower], -1), 10, inc) \
            + w * positional_encoding(torch.cat([xyz, lower + 1], -1), 10, inc)
    else:
        enc = positional_encoding(x, 10, inc)
    h = enc
    for j in (0, 2, 4, 6, 8):
        h = F.relu(_linear(params, f"{prefix}.motion_net.{j}", h))
    flow = _linear(params, f"{prefix}.motion_net.10", h)
    return flow.reshape(*shape[:-1], 3)


# --------------------------------------------------------------------------------------
# a9: SpaceNet                                                    modeling/spacenet.py:16-160
# --------------------------------------------------------------------------------------
def space_net(params: Dict[str, Tensor], prefix: str, pos: Tensor, dirs: Tensor,
              times: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """pos (n, S, 3), dirs (n, 3), times (n, 1) or None -> raw rgb (n,S,3), raw sigma (n,S,1).

    stage1: PE_10(pos)=63 -> 256 x4 (ReLU each) (:45-54); stage2 on [h, PE(pos)] = 319 -> 256 x3
    (ReLU each, :56-63); sigma = Linear(256,1) (:65-67); rgb_net = ReLU -> Linear(283|304,128) ->
    ReLU -> Linear(128,3) over [h, PE_4(dir), PE_10(t)] (:80-86) -- the leading ReLU also clamps
    the direction/time encodings.  Direction and time are repeated per sample (:115,:118).
    Whether the net takes time is decided by its first rgb layer's width (304 vs 283).
    """
    n, s = pos.shape[0], pos.shape[1]
    # the flavour (include_input / use_dir / use_time, :16-44) is read off the tensor shapes
    inc = params[f"{prefix}.stage1.0.weight"].shape[1] == 63
    dir_w, time_w = (27, 21) if inc else (24, 20)
    extra = params[f"{prefix}.rgb_net.1.weight"].shape[1] - 256
    use_dir, use_time = {dir_w + time_w: (True, True), dir_w: (True, False), time_w: (False, True),
                         0: (False, False)}[extra]
    p = positional_encoding(pos.reshape(-1, 3), 10, inc)
    h = p
    for j in (0, 2, 4, 6):
        h = F.relu(_linear(params, f"{prefix}.stage1.{j}", h))
    h = torch.cat([h, p], dim=1)
    for j in (0, 2, 4):
        h = F.relu(_linear(params, f"{prefix}.stage2.{j}", h))
    sigma = _linear(params, f"{prefix}.density_net.0", h)
    feat = [h]
    if use_dir:
        feat.append(positional_encoding(dirs.unsqueeze(1).repeat(1, s, 1).reshape(-1, 3), 4, inc))
    if use_time:
        # (n, 1): a ray's frame id on each of its samples (:117-118).  (n, S, 1): one time per SAMPLE -- only the background's call
        # site produces that, see render_chunk's run_nets
        per_sample = times if times.dim() == 3 else times.reshape(n, 1, 1).repeat(1, s, 1)
        feat.append(positional_encoding(per_sample.reshape(-1, 1), 10, inc))
    x = F.relu(torch.cat(feat, dim=1))
    x = F.relu(_linear(params, f"{prefix}.rgb_net.1", x))
    if f"{prefix}.rgb_net.7.weight" in params:      # deep_rgb (:68-79): 128 -> 128 -> 128 -> 3
        x = F.relu(_linear(params, f"{prefix}.rgb_net.3", x))
        x = F.relu(_linear(params, f"{prefix}.rgb_net.5", x))
        rgb = _linear(params, f"{prefix}.rgb_net.7", x)
    else:
        rgb = _linear(params, f"{prefix}.rgb_net.3", x)
    return rgb.reshape(n, s, 3), sigma.reshape(n, s, 1)


# --------------------------------------------------------------------------------------
# a12: alpha compositing                                          layers/render_layer.py:8-58
# --------------------------------------------------------------------------------------
def gen_weight(sigma: Tensor, delta: Tensor) -> Tensor:
    """alpha = 1-exp(-relu(sigma) delta); w = alpha * exclusive-cumprod(1-alpha+1e-10)  (:8-17)."""
    alpha = 1.0 - torch.exp(-F.relu(sigma) * delta)
    trans = 1.0 - alpha + 1e-10
    ones = torch.ones(alpha.shape[0], 1, dtype=alpha.dtype)
    return alpha * torch.cumprod(torch.cat([ones, trans], -1), -1)[:, :-1]


def composite(t: Tensor, rgb: Tensor, sigma: Tensor, border: float = 1e10):
    """VolumeRenderer.forward (:25-58): t (n,S,1), rgb (n,S,3) raw, sigma (n,S,1) raw ->
    color (n,3), depth (n,1), acc (n,1), weights (n,S,1).  Last interval = border (:37-40)."""
    delta = (t[:, 1:] - t[:, :-1]).squeeze(-1)
    delta = torch.cat([delta, border * torch.ones(delta.shape[0], 1, dtype=t.dtype)], dim=-1)
    w = gen_weight(sigma.squeeze(-1), delta).unsqueeze(-1)
    color = torch.sum(torch.sigmoid(rgb) * w, dim=1)
    depth = torch.sum(w * t, dim=1)
    acc = torch.sum(w, dim=1)
    return color, depth, acc, w


# --------------------------------------------------------------------------------------
# a13: inverse-CDF importance resampling                          utils/sample_pdf.py:18-63
# --------------------------------------------------------------------------------------
def sample_pdf(z_vals: Tensor, weights: Tensor, u: Tensor, return_aux: bool = False):
    """z_vals (n,N1), weights (n,N1-2) (interior weights), u (n,N2) in [0,1) -> z (n,N2).

    bins = midpoints (:20); pdf = (w+1e-5)/sum (:21-22); cdf = [0, cumsum] (:23-24);
    inds = searchsorted(cdf, u, right=True) (:47); below = max(inds-1,0), above = min(inds, N1-2)
    (:48-49); z = bins_b + (u-cdf_b)/den (bins_a-bins_b) with den<1e-5 -> 1 (:58-61).
    """
    bins = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bins_b, bins_a = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    den = cdf_a - cdf_b
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    z = bins_b + (u - cdf_b) / den * (bins_a - bins_b)
    if return_aux:
        return z, cdf, inds
    return z


def aten_cpu_row_sum(x) -> "numpy.float32":
    """What ``torch.sum(w, -1)`` (utils/sample_pdf.py:22) returns for one contiguous fp32 row on a CPU, restated in
    numpy: the order ATen's reduction kernel uses (aten/src/ATen/native/cpu/SumKernel.cpp -- dispatched with 8-float
    vectors at every x86 capability level).  Rows of >= 8 elements: ``vectorized_inner_sum`` = ``row_sum`` over the full
    8-float vectors (4 interleaved accumulators, folded ((a0+a1)+a2)+a3; a cascade level folds the accumulators away
    every 16 rows), then a scalar chain over the leftover elements followed by the 8 vector lanes in order.  Shorter
    rows: the scalar ``row_sum``.  The HIP resampler reproduces exactly this (csrc/render.hip); the oracle itself just
    calls torch.sum -- tests/test_host_cpu.py::test_aten_sum_order checks that the two agree on the running host."""
    import numpy as np
    f32 = np.float32
    x = np.asarray(x, dtype=f32)
    n = x.shape[0]

    def row_sum(get, size, zero):
        rows = size // 4
        a = [zero.copy() for _ in range(4)]
        b = [zero.copy() for _ in range(4)]
        i = 0
        while i + 16 <= rows:
            for _ in range(16):
                for k in range(4):
                    a[k] = (a[k] + get(4 * i + k)).astype(f32)
                i += 1
            for k in range(4):
                b[k] = (b[k] + a[k]).astype(f32)
                a[k] = zero.copy()
        while i < rows:
            for k in range(4):
                a[k] = (a[k] + get(4 * i + k)).astype(f32)
            i += 1
        for k in range(4):
            a[k] = (a[k] + b[k]).astype(f32)
        for j in range(rows * 4, size):
            a[0] = (a[0] + get(j)).astype(f32)
        for k in range(1, 4):
            a[0] = (a[0] + a[k]).astype(f32)
        return a[0]

    if n >= 8:
        nv = n // 8
        lanes = row_sum(lambda i: x[8 * i:8 * i + 8], nv, np.zeros(8, dtype=f32))
        fin = f32(0)
        for k in range(nv * 8, n):
            fin = f32(fin + x[k])
        for j in range(8):
            fin = f32(fin + lanes[j])
        return fin
    return f32(row_sum(lambda i: x[i:i + 1], n, np.zeros(1, dtype=f32))[0])


# --------------------------------------------------------------------------------------
# a4 + a10 + a11 + a14: one chunk                          modeling/layered_rfrender.py:141-734
# --------------------------------------------------------------------------------------
@dataclass
class OracleModel:
    """The state LayeredRFRender carries (layered_rfrender.py:21-127), as plain data."""
    layer_num: int
    n_coarse: int
    n_fine: int
    params: Dict[str, Tensor]                 # reference state_dict key names
    use_deform_time: bool = True
    use_space_time: bool = True
    bkgd_use_deform_time: bool = False        # :33, :92-93
    bkgd_use_space_time: bool = False         # :34, :62 (only meaningful together with use_space_time)
    border: float = 1e10
    bkgd_bbox: Optional[Tensor] = None        # (1,8,3)  set_bkgd_bbox :114
    bboxes: Optional[Tensor] = None           # (F,L,8,3) set_bboxes :117
    near: float = 0.0                         # :41
    alpha: float = 1.0                        # :42
    scale: Optional[list] = None              # :39
    shift: Optional[list] = None              # :40
    hidden: set = field(default_factory=set)  # display_layers :99-112

    def is_shown_layer(self, i: int) -> bool:
        return i not in self.hidden


RandFn = Callable[[Tuple[int, int]], Tensor]


def _default_rand(shape):
    return torch.rand(shape)


def layer_boxes(m: OracleModel, rays: Tensor):
    """Boxes (n,l,8,3) for a chunk + the edit pivot; layered_rfrender.py:151-242.

    Ray layouts (:151-181, deform-view off): 7 columns = per-ray frame id (non-retiming: boxes
    gathered per ray, :193); 7+L columns = one frame id per layer (retiming: one box per layer
    for the whole chunk, from ROW 0's frame id, lerped between floor/ceil frame, :195-200,
    :123-127).  Edit (:216-242): pivot = mean of the frame-0 centres of layers 1 and 2 with
    centre.z := corner-1 z; boxes[i] = (boxes[i]-pivot)*scale[i]+pivot; boxes[i] += shift[i].
    """
    n, L = rays.shape[0], m.layer_num
    width = rays.shape[1]
    if width == 7:
        retiming, frame_id = False, rays[:, -1]
    elif width == 7 + L:
        retiming, frame_id = True, rays[:, 6:]
    else:
        raise ValueError(f"undefined ray format, ray dimension is {width}")  # :161-163 (exit(-1))
    if not retiming:
        boxes = m.bboxes.index_select(0, frame_id.type(torch.int64) - 1)
    else:
        boxes = torch.zeros(n, L, 8, 3, dtype=rays.dtype)
        for i in range(L):
            f = frame_id[0, i + 1] - 1
            lo, hi = m.bboxes[math.floor(f), i], m.bboxes[math.ceil(f), i]
            boxes[:, i] = torch.lerp(lo, hi, f - math.floor(f))
    boxes = torch.cat([m.bkgd_bbox.unsqueeze(0).repeat(n, 1, 1, 1), boxes], 1)
    first = torch.cat([m.bkgd_bbox, m.bboxes[0, :]], 0)            # (l,8,3) frame-0 boxes
    centre = torch.mean(first, 1)                                   # (l,3)
    centre[:, 2] = first[:, 1, 2]                                   # :226
    pivot = None
    if m.scale is not None:
        pivot = (centre[2] + centre[1]) / 2                         # :232
        for i in range(len(m.scale)):
            boxes[:, i] = (boxes[:, i] - pivot) * m.scale[i] + pivot
    if m.shift is not None:
        for i in range(len(m.shift)):
            if m.shift[i] is None:
                continue
            boxes[:, i] += torch.tensor(m.shift[i], dtype=rays.dtype)
    return boxes, pivot, retiming, frame_id


def render_chunk(m: OracleModel, rays: Tensor, only_coarse: bool = False,
                 density_threshold: float = 0.0001, bkgd_density_threshold: float = 0.0,
                 rand: RandFn = _default_rand, trace: Optional[dict] = None, sample_dtype: Optional[torch.dtype] = None,
                 forced: Optional[dict] = None):
    """LayeredRFRender.forward for one chunk (BBOX sampling, no pose refinement / view deform /
    background deform, background net without time: the configuration of both shipped ymls).

    Returns (fine_mixed, coarse_mixed, fine_layer[l], coarse_layer[l], ray_mask[l]) with each
    entry a (color (n,3), depth (n,1), acc (n,1)) triple (:725-734).  ``rand(shape)`` supplies the
    uniform draws in the reference's order: l coarse-jitter tensors (RaySamplePoint.py:98) then l
    resampling tensors (sample_pdf.py:31).  ``trace`` (dict) receives every intermediate.
    ``sample_dtype`` (error analysis only): evaluate the DETACHED parts -- the coarse sampler and the inverse-CDF resampler with
    its sort and point generation (:314-315, :460-465) -- in that dtype and cast the depths / points back, so that an fp64
    evaluation of the differentiable graph sits on the sample positions of the fp32 one.
    ``forced`` (teacher forcing, error analysis only): {"z": (l, n, N2), "xyz_c" / "xyz_f": per performer the deformed points of its
    hit rays or None} -- the reference's own new fine depths (the value of sample_pdf at :460) and the points its performer SpaceNets
    were given (:355-356, :509-510), recorded by tests/golden/make_golden.py --grads --teacher: the networks are evaluated on those
    (the deformation nets keep their place in the graph: value replaced, gradient passed).
    """
    n, L = rays.shape[0], m.layer_num
    l, N1, N2 = L + 1, m.n_coarse, m.n_fine
    P = m.params
    boxes, pivot, retiming, frame_id = layer_boxes(m, rays)
    o, d = rays[:, 0:3], rays[:, 3:6]

    def fid(i):  # per-ray frame id of layer i
        return frame_id[:, i] if retiming else frame_id

    def unedit(x, i, fine):
        # coarse :293-303 (shift loop, then scale loop); fine :467-475 (a None shift skips scale)
        if m.shift is not None:
            if fine:
                if m.shift[i] is None:
                    return x
                x = x - torch.tensor(m.shift[i], dtype=x.dtype)
            elif i < len(m.shift) and m.shift[i] is not None:
                x = x - torch.tensor(m.shift[i], dtype=x.dtype)
        if m.scale is not None and (fine or i < len(m.scale)):
            x = (x - pivot) / m.scale[i] + pivot
        return x

    def deform(x, masks, ns, given=None):
        # :340-356 / :495-510: performers only, masked rays only, regardless of visibility
        if m.bkgd_use_deform_time:  # :358-367 / :512-523: every ray, MotionNet(input_time=False)
            tid = fid(0).view(-1, 1, 1).repeat(1, ns, 1)
            x[0] = x[0] + motion_net(P, "bkgd_time_deform_net", torch.cat([x[0], tid], -1), input_time=False)
        for i in range(1, l):
            if not m.use_deform_time:
                break
            idx = masks[i]
            if torch.sum(idx) == 0:
                continue
            tid = fid(i)[idx].view(-1, 1, 1).repeat(1, ns, 1)
            flow = motion_net(P, f"time_deform_nets.{i - 1}", torch.cat([x[i][idx], tid], -1))
            moved = x[i][idx] + flow
            if given is not None and given[i] is not None:
                moved = given[i].to(moved.dtype) + (moved - moved.detach())
            x[i][idx] = moved

    def run_nets(x, masks, ns, fine):
        sfx = "_fine" if fine else ""
        rgbs, sig = [], []
        # :382-394 / :531-549: the frame id is handed over whenever use_space_time is on; the network uses it
        # only if it was built with use_time (BKGD_USE_SPACE_TIME)
        # The BACKGROUND's call sites (:380, :385 / :531-537) hand the frame ids over as a 1-D tensor (rays[:, -1] or
        # rays_frame_id[:, 0]) where the performers' reshape theirs to (n, 1) (:403, :405): SpaceNet.forward's
        # `times.unsqueeze(1).repeat(1, L, 1).reshape(-1, 1)` (modeling/spacenet.py:117-118) then TILES the batch's ids -- sample j of
        # ray i gets the frame id of ray (i S + j) mod n.  Invisible when every ray of the call has the same background frame id (any
        # rendered frame); a training batch with BKGD_USE_SPACE_TIME mixes the ids across rays.  Restated as the reference does it.
        t0 = fid(0).reshape(-1).repeat(ns).reshape(n, ns, 1) if m.use_space_time else None
        c0, s0 = space_net(P, "bkgd_spacenet" + sfx, x[0], d, t0)
        if fine and retiming:
            s0[s0 < bkgd_density_threshold] = 0                           # :538-547
        rgbs.append(c0)
        sig.append(s0)
        for i in range(1, l):                                             # :397-418 / :552-576
            rgbs.append(torch.zeros(n, ns, 3, dtype=rays.dtype))
            sig.append(torch.zeros(n, ns, 1, dtype=rays.dtype))
            idx = masks[i]
            if torch.sum(idx) == 0 or not m.is_shown_layer(i):
                continue
            times = fid(i)[idx].reshape(-1, 1) if m.use_space_time else None
            ci, si = space_net(P, f"spacenets{sfx}.{i - 1}", x[i][idx], d[idx], times)
            rgbs[i][idx] = ci
            sig[i][idx] = si
            if not fine:
                sig[i][ts[i][:, :, 0] < 0, :] = 0.0                       # :414
            if retiming:
                sig[i][sig[i] < density_threshold] = 0                    # :416-418 / :564-566
            if fine and i == 2:
                sig[i] = sig[i] * m.alpha                                 # :575-576
        return rgbs, sig

    # ---- coarse
    jitter = [rand((n, N1)).to(rays.dtype) for _ in range(l)]
    if sample_dtype is None:
        ts, xyz, masks = sample_coarse(rays, boxes, N1, jitter)
    else:
        ts, xyz, masks = sample_coarse(rays.to(sample_dtype), boxes.to(sample_dtype), N1, [j.to(sample_dtype) for j in jitter])
        ts, xyz = [t_.to(rays.dtype) for t_ in ts], [x_.to(rays.dtype) for x_ in xyz]
    ts, xyz = [t_.detach() for t_ in ts], [x_.detach() for x_ in xyz]      # :314-315 (only matters under autograd)
    xyz = [unedit(xyz[i], i, False) for i in range(l)]
    deform(xyz, masks, N1, forced.get("xyz_c") if forced else None)
    rgbs, sig = run_nets(xyz, masks, N1, False)
    sig[0][ts[0][:, :, 0] < m.near, :] = 0.0                              # :422
    t_mix, order = torch.sort(torch.cat(ts, -2), -2)                      # :425
    rgb_mix = torch.cat(rgbs, -2).gather(1, order.repeat(1, 1, 3))
    sig_mix = torch.cat(sig, -2).gather(1, order)
    coarse_layer = [composite(ts[i], rgbs[i], sig[i], m.border) for i in range(l)]   # :435-444
    coarse_mixed = composite(t_mix, rgb_mix, sig_mix, m.border)                       # :448
    if trace is not None:
        trace.update(boxes=boxes[0], t_coarse=ts, mask=masks, xyz_coarse=xyz, rgb_coarse=rgbs,
                     sigma_coarse=sig, order_coarse=order, w_coarse=[c[3] for c in coarse_layer])
    pack = lambda c: (c[0], c[1], c[2])
    if only_coarse:                                                       # :684-722
        cl = [pack(c) for c in coarse_layer]
        return pack(coarse_mixed), pack(coarse_mixed), cl, cl, masks

    # ---- fine
    zf, xf, us, zs = [], [], [], []
    for i in range(l):                                                    # :459-475
        u = rand((n, N2)).to(rays.dtype)
        sd_ = rays.dtype if sample_dtype is None else sample_dtype
        z = sample_pdf(ts[i].squeeze(-1).to(sd_), coarse_layer[i][3].squeeze(-1)[..., 1:-1].detach().to(sd_), u.to(sd_)).detach()   # :460-461
        if forced and "z" in forced:
            z = forced["z"][i].to(sd_)
        zi, _ = torch.sort(torch.cat([ts[i].squeeze(-1).to(sd_), z], -1), -1)
        pts = zi.unsqueeze(-1) * d.to(sd_).unsqueeze(1) + o.to(sd_).unsqueeze(1)
        z, zi, pts = z.to(rays.dtype), zi.to(rays.dtype), pts.to(rays.dtype)
        us.append(u)
        zs.append(z)
        zf.append(zi)
        xf.append(unedit(pts, i, True))
    deform(xf, masks, N1 + N2, forced.get("xyz_f") if forced else None)
    rgbs, sig = run_nets(xf, masks, N1 + N2, True)
    z_mix, order = torch.sort(torch.cat(zf, -1), -1)                      # :587
    z_mix = z_mix.unsqueeze(-1)
    rgb_mix = torch.cat(rgbs, -2).gather(1, order.unsqueeze(-1).repeat(1, 1, 3))
    sig_mix = torch.cat(sig, -2).gather(1, order.unsqueeze(-1))
    fine_layer = [composite(zf[i].unsqueeze(-1), rgbs[i], sig[i], m.border) for i in range(l)]
    sig_mix[z_mix < m.near] = 0                                           # :605
    fine_mixed = composite(z_mix, rgb_mix, sig_mix, m.border)             # :606
    if trace is not None:
        trace.update(u=us, z_new=zs, t_fine=zf, xyz_fine=xf, rgb_fine=rgbs, sigma_fine=sig,
                     order_fine=order, jitter=jitter)
    return (pack(fine_mixed), pack(coarse_mixed), [pack(c) for c in fine_layer],
            [pack(c) for c in coarse_layer], masks)


# --------------------------------------------------------------------------------------
# a3: chunked invocation                                          utils/batchify_rays.py:51-140
# --------------------------------------------------------------------------------------
def layered_batchify_ray(m: OracleModel, rays: Tensor, chuncks: int = 512 * 7,
                         density_threshold: float = 0.0, bkgd_density_threshold: float = 0.0,
                         rand: RandFn = _default_rand):
    """Split into ``chuncks``-ray pieces, render each, concatenate.  A call with fewer rays than one
    chunk goes to the model WITHOUT the thresholds, i.e. with the model defaults 1e-4 / 0
    (:52-54 vs layered_rfrender.py:141)."""
    if rays.size(0) < chuncks:
        return render_chunk(m, rays, rand=rand)
    outs = [render_chunk(m, r, density_threshold=density_threshold,
                         bkgd_density_threshold=bkgd_density_threshold, rand=rand)
            for r in rays.split(chuncks, dim=0)]
    cat3 = lambda trips: tuple(torch.cat([t[j] for t in trips], 0) for j in range(3))
    l = len(outs[0][2])
    return (cat3([o_[0] for o_ in outs]), cat3([o_[1] for o_ in outs]),
            [cat3([o_[2][i] for o_ in outs]) for i in range(l)],
            [cat3([o_[3][i] for o_ in outs]) for i in range(l)],
            [torch.cat([o_[4][i] for o_ in outs], 0) for i in range(l)])


def psnr(pred: Tensor, gt: Tensor) -> float:
    """utils/metrics.py:4-17: -10 log10(mean squared error)."""
    return float(-10.0 * torch.log10(torch.mean((pred - gt) ** 2)))


# --------------------------------------------------------------------------------------
# Deterministic synthetic scene / weights shared by the golden script, the tests and bench.py
# --------------------------------------------------------------------------------------
def aabb_corners(lo: Sequence[float], hi: Sequence[float]) -> Tensor:
    """8 corners in the reference's order (data/datasets/frame_dataset.py:187-188):
    bottom face (z = lo) counter-clockwise from (lo,lo), then the top face."""
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    return torch.tensor([[x0, y0, z0], [x1, y0, z0], [x1, y1, z0], [x0, y1, z0],
                         [x0, y0, z1], [x1, y0, z1], [x1, y1, z1], [x0, y1, z1]], dtype=torch.float32)
