"""Torch-tensor wrappers over the C ABI (one function per entry point of include/stnerf.h).

PyTorch is plumbing here: device memory, streams.  Every function launches HIP kernels from
``libstnerf_hip.so`` on the current torch stream and returns device tensors; none of them has a
PyTorch implementation to fall back to.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch

from stnerf_amd import hip

Tensor = torch.Tensor


def _strided_view_ptr(t: Tensor, inner: Tuple[int, ...], name: str):
    """(pointer, ray stride) of a tensor whose dim 0 is the ray and whose inner dims are dense."""
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != torch.float32:
        raise ValueError(f"{name} must be float32, got {t.dtype}")
    if tuple(t.shape[1:]) != tuple(inner):
        raise ValueError(f"{name}: expected shape (n, {inner}), got {tuple(t.shape)}")
    expect = 1
    for d in range(t.dim() - 1, 0, -1):
        if t.shape[d] != 1 and t.stride(d) != expect:
            raise ValueError(f"{name}: inner dimensions must be dense, got strides {t.stride()}")
        expect *= t.shape[d]
    return C.c_void_p(t.data_ptr()), (t.stride(0) if t.shape[0] > 1 else expect)


def _edits(edits, pivot, l):
    """-> (LayerEdit array | None, float[3] | None).  edits: per layer (shift|None, scale|None)."""
    if edits is None:
        return None, None
    arr = (hip.LayerEdit * l)()
    for i in range(l):
        sh, sc = edits[i] if i < len(edits) else (None, None)
        if sh is not None:
            # the reference builds torch.tensor(shift[i]) (fp32), layered_rfrender.py:241,297
            v = torch.tensor(sh, dtype=torch.float32).tolist()
            arr[i].shift[0], arr[i].shift[1], arr[i].shift[2] = v
            arr[i].has_shift = 1
        if sc is not None:
            arr[i].scale = float(sc)
            arr[i].has_scale = 1
    pv = (C.c_float * 3)(*(float(x) for x in (pivot if pivot is not None else (0.0, 0.0, 0.0))))
    return arr, pv


def _boxes_arg(boxes: Tensor, n: int):
    if boxes.dim() == 3:  # (l,8,3) shared
        return hip.dptr(boxes, name="boxes"), 0, boxes.shape[0]
    if boxes.dim() == 4 and boxes.shape[0] == n:  # (n,l,8,3) per ray
        return hip.dptr(boxes, name="boxes"), boxes.shape[1] * 24, boxes.shape[1]
    raise ValueError(f"boxes must be (l,8,3) or (n,l,8,3), got {tuple(boxes.shape)}")


# ---------------------------------------------------------------------------------------- a1/a2
def window_size(total: int, first: int, stripe: int, period: int) -> int:
    """Number of rays of a `total`-ray view inside the window (first, stripe, period) of include/stnerf.h."""
    if stripe <= 0:
        return max(0, total - first)
    if first >= total:
        return 0
    full, rem = divmod(total - first, period)
    return full * stripe + min(rem, stripe)


def generate_rays(K: Tensor, T: Tensor, h: int, w: int, frame_ids: Optional[Sequence[float]] = None,
                  first_ray: int = 0, n: Optional[int] = None, device="cuda", stripe: int = 0, period: int = 0) -> Tensor:
    """Rays of a view, generated on the device: (n, 6 + len(frame_ids)).  (first_ray, stripe, period) = the ray
    window of include/stnerf.h: a contiguous piece of the view (stripe = 0) or interleaved stripes.
    utils/render_helpers.py:42-128 + data/datasets/ray_dataset.py:276-281."""
    n = window_size(h * w, first_ray, stripe, period) if n is None else n
    kinv = torch.inverse(K.detach().to("cpu", torch.float32)).contiguous()  # :103 (torch.inverse on the host)
    Tm = torch.as_tensor(T, dtype=torch.float32).detach().cpu().contiguous()
    fids = [float(x) for x in (frame_ids or [])]
    rays = torch.empty(n, 6 + len(fids), dtype=torch.float32, device=device)
    kin = (C.c_float * 9)(*kinv.reshape(-1).tolist())
    tin = (C.c_float * 16)(*Tm.reshape(-1).tolist())
    fin = (C.c_float * max(len(fids), 1))(*(fids or [0.0]))
    hip.check(hip.lib().stnerf_generate_rays(kin, tin, h, w, first_ray, stripe, period, n, fin if fids else None, len(fids),
                                             hip.dptr(rays), rays.shape[1], hip.stream_ptr()), "stnerf_generate_rays")
    return rays


# ---------------------------------------------------------------------------------------- a5/a6
def intersect(rays: Tensor, boxes: Tensor) -> Tensor:
    """(n,l,2) = (far, near); layers/RaySamplePoint.py:8-62."""
    n = rays.shape[0]
    bp, bstride, l = _boxes_arg(boxes, n)
    out = torch.empty(n, l, 2, dtype=torch.float32, device=rays.device)
    hip.check(hip.lib().stnerf_intersect(hip.dptr(rays, name="rays"), n, rays.shape[1], bp, bstride, l,
                                         hip.dptr(out), hip.stream_ptr()), "stnerf_intersect")
    return out


def sample_coarse(rays: Tensor, boxes: Tensor, n1: int, jitter: Optional[Tensor] = None, seed: int = 0,
                  ray_index_base: int = 0, edits=None, pivot=None, want_xyz: bool = True, ray_index_stripe: int = 0,
                  ray_index_period: int = 0, raw_mask: bool = False):
    """-> t (n,l,n1), xyz (n,l,n1,3) | None, mask (n,l) uint8 (0 / 1 = the reference's ray_mask; with ``raw_mask`` the
    library's byte: bit 1 = the sampler's "every depth of this pair is -1000" hint, which ``composite`` accepts).
    layers/RaySamplePoint.py:70-107."""
    n = rays.shape[0]
    bp, bstride, l = _boxes_arg(boxes, n)
    if jitter is not None and tuple(jitter.shape) != (l, n, n1):
        raise ValueError(f"jitter must be (l,n,n1) = {(l, n, n1)}, got {tuple(jitter.shape)}")
    t = torch.empty(n, l, n1, dtype=torch.float32, device=rays.device)
    xyz = torch.empty(n, l, n1, 3, dtype=torch.float32, device=rays.device) if want_xyz else None
    mask = torch.empty(n, l, dtype=torch.uint8, device=rays.device)
    ed, pv = _edits(edits, pivot, l)
    hip.check(hip.lib().stnerf_sample_coarse(hip.dptr(rays, name="rays"), n, rays.shape[1], bp, bstride, l, n1,
                                             hip.dptr(jitter, name="jitter"), seed, ray_index_base, ray_index_stripe,
                                             ray_index_period, ed, pv,
                                             hip.dptr(t), hip.dptr(xyz), hip.dptr(mask, torch.uint8),
                                             hip.stream_ptr()), "stnerf_sample_coarse")
    return t, xyz, (mask if raw_mask else mask.bitwise_and_(1))


def compact_rays(mask: Tensor):
    """-> ray_list (l,n) int32, ray_count (l,) int32 (device; no host sync)."""
    n, l = mask.shape
    ray_list = torch.empty(l, n, dtype=torch.int32, device=mask.device)
    ray_count = torch.zeros(l, dtype=torch.int32, device=mask.device)
    hip.check(hip.lib().stnerf_compact_rays(hip.dptr(mask, torch.uint8, "mask"), n, l,
                                            hip.dptr(ray_list, torch.int32), hip.dptr(ray_count, torch.int32),
                                            hip.stream_ptr()), "stnerf_compact_rays")
    return ray_list, ray_count


# ---------------------------------------------------------------------------------------- networks
SPACENET_KEYS = ["stage1.0", "stage1.2", "stage1.4", "stage1.6", "stage2.0", "stage2.2", "stage2.4",
                 "density_net.0", "rgb_net.1", "rgb_net.3"]
MOTIONNET_KEYS = [f"motion_net.{j}" for j in (0, 2, 4, 6, 8, 10)]


PRECISIONS = ("bf16x3", "fp32")   # the library default first


class PackedNet:
    """Kernel-layout weights of one network, resident on the device."""

    def __init__(self, kind: int, blob: Tensor, precision: str = "fp32"):
        self.kind, self.blob, self.precision = kind, blob, precision

    @property
    def use_time(self) -> bool:
        return self.kind in (hip.NET_SPACE_TIME, hip.NET_SPACE_TIME_DEEP)


_PACK_CHECK = [True]


class training_pack:
    """Context of the packs a training step makes (modeling/autograd.py): no finiteness check -- no synchronisation -- in the device bf16x3 packer."""

    def __enter__(self):
        self.prev, _PACK_CHECK[0] = _PACK_CHECK[0], False

    def __exit__(self, *exc):
        _PACK_CHECK[0] = self.prev


def pack_net(kind: int, weights: List[Tensor], biases: List[Tensor], device="cuda", precision: str = "fp32") -> PackedNet:
    """Repack reference-layout nn.Linear tensors (host copy) and upload.  precision: "fp32" (exact f32 MFMA) or "bf16x3"
    (three bf16 pieces per operand, six MFMAs, two accumulators: the full fp32 significand and exponent range; the stage
    kernel only)."""
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be one of {PRECISIONS}, got {precision!r}")
    lib = hip.lib()
    size_fn = {"fp32": lib.stnerf_packed_bytes, "bf16x3": lib.stnerf_packed_bytes_bf16x3}[precision]
    pack_fn = {"fp32": lib.stnerf_pack_net, "bf16x3": lib.stnerf_pack_net_bf16x3}[precision]
    nbytes = size_fn(kind)
    if nbytes < 0:
        hip.check(int(nbytes), "stnerf_packed_bytes")
    if all(t.is_cuda for t in list(weights) + list(biases)) and not (precision == "bf16x3" and os.environ.get("STNERF_PACK_BF16X3_HOST") == "1"):
        # tensors that live on the GPU are packed there (stnerf_pack_net_device / stnerf_pack_net_bf16x3_device: the same blobs bit for
        # bit): no D2H / CPU loop / H2D per network -- a training loop repacks after every optimizer.step().  (The bf16x3 device packer
        # does not refuse non-finite weights the way the host packer does: STNERF_PACK_BF16X3_HOST=1 keeps the host path.)
        dev = weights[0].device
        ws = [w.detach().to(torch.float32).contiguous() for w in weights]
        bs = [b.detach().to(torch.float32).contiguous() for b in biases]
        wp = (C.c_void_p * len(ws))(*(w.data_ptr() for w in ws))
        bp = (C.c_void_p * len(bs))(*(b.data_ptr() for b in bs))
        if precision == "fp32":
            blob = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
            fn, name = lib.stnerf_pack_net_device, "stnerf_pack_net_device"
        else:       # (the stage kernel streams it by 16-byte LDS-DMA from 1 KB-aligned sections)
            # The host packer refuses what cannot be split into three bf16 pieces (NaN, inf, |w| > 3.3895e38: include/stnerf.h).  Packing on
            # the device, the same refusal costs a host <- device synchronisation: paid where a model is packed to RENDER (once per weight
            # version), not inside a training step (``training_pack()``, modeling/autograd.py: the step would wait for the GPU after every
            # optimizer.step(); a non-finite weight there gives NaN outputs and a NaN loss).
            if _PACK_CHECK[0]:
                worst = float(torch.stack([t.abs().max() if t.numel() else t.new_zeros(()) for t in ws + bs]).nan_to_num(nan=float("inf")).max())
                if not worst <= 3.3895313892515355e38:
                    raise ValueError(f"stnerf_pack_net_bf16x3_device: a weight or bias is not finite or exceeds bf16's range (|w| <= 3.3895e38): "
                                     f"{worst:g} -- use the exact-f32 packing (precision='fp32') for such a network")
            raw = torch.empty(nbytes + 1024, dtype=torch.uint8, device=dev)
            off = (-raw.data_ptr()) % 1024
            blob = raw[off:off + nbytes].view(torch.float32)
            fn, name = lib.stnerf_pack_net_bf16x3_device, "stnerf_pack_net_bf16x3_device"
        with torch.cuda.device(dev):
            hip.check(fn(kind, wp, bp, len(ws), hip.dptr(blob), nbytes, hip.stream_ptr()), name)
        return PackedNet(kind, blob, precision)
    ws = [w.detach().to("cpu", torch.float32).contiguous() for w in weights]
    bs = [b.detach().to("cpu", torch.float32).contiguous() for b in biases]
    host = torch.empty(nbytes // 4, dtype=torch.float32)
    wp = (C.c_void_p * len(ws))(*(w.data_ptr() for w in ws))
    bp = (C.c_void_p * len(bs))(*(b.data_ptr() for b in bs))
    hip.check(pack_fn(kind, wp, bp, len(ws), C.c_void_p(host.data_ptr()), nbytes), "stnerf_pack_net")
    blob = host.to(device)
    if precision == "bf16x3" and blob.data_ptr() % 1024:   # (the stage kernel streams it by 16-byte LDS-DMA from 1 KB-aligned sections)
        raw = torch.empty(nbytes + 1024, dtype=torch.uint8, device=device)
        off = (-raw.data_ptr()) % 1024
        blob = raw[off:off + nbytes].view(torch.float32)
        blob.copy_(host)
    return PackedNet(kind, blob, precision)


def _pe_columns(w: Tensor, dim: int, n_freq: int, include_input: bool, present: bool = True) -> Tensor:
    """Columns of a Linear weight that multiply one positional-encoding block, widened to the kernels' layout
    [x, sin, cos, ...] (dim*(1+2L) columns).  The kernels always build that block; an encoding without the raw
    input (TKERNEL_INC_RAW=False, utils/dimension_kernel.py:12-14) or an absent direction encoding (USE_DIR=False,
    modeling/spacenet.py:22-31) gets ZERO weight columns there, which leaves every output bit-identical."""
    full = dim * (1 + 2 * n_freq)
    if not present:
        return w.new_zeros(w.shape[0], full)
    if include_input:
        assert w.shape[1] == full
        return w
    assert w.shape[1] == full - dim
    return torch.cat([w.new_zeros(w.shape[0], dim), w], 1)


def pack_spacenet(state: dict, prefix: str, device="cuda", precision: str = "fp32") -> PackedNet:
    """From reference state_dict keys ``{prefix}.stage1.0.weight`` ....  The SpaceNet flavour
    (use_time, use_dir, include_input, deep_rgb: modeling/spacenet.py:16-86) is read off the tensor shapes."""
    deep = f"{prefix}.rgb_net.7.weight" in state          # deep_rgb: rgb_net.{1,3,5,7} (modeling/spacenet.py:68-79)
    keys = SPACENET_KEYS + (["rgb_net.5", "rgb_net.7"] if deep else [])
    on_device = all(state[f"{prefix}.{k}.weight"].is_cuda for k in keys)     # (packed where they live, either arithmetic: pack_net)
    ws = [state[f"{prefix}.{k}.weight"].detach().float() if on_device else state[f"{prefix}.{k}.weight"].detach().float().cpu() for k in keys]
    bs = [state[f"{prefix}.{k}.bias"] for k in keys]
    pos_w = ws[0].shape[1]
    if pos_w not in (63, 60):
        raise ValueError(f"{prefix}.stage1.0 has in-width {pos_w}: only c_pos=3 with PE_10 is supported")
    inc = pos_w == 63
    dir_w, time_w = (27, 21) if inc else (24, 20)
    extra = ws[8].shape[1] - 256
    flavours = {dir_w + time_w: (True, True), dir_w: (True, False), time_w: (False, True), 0: (False, False)}
    if extra not in flavours:
        raise ValueError(f"{prefix}.rgb_net.1 has in-width {ws[8].shape[1]}: not a SpaceNet colour head")
    use_dir, use_time = flavours[extra]
    ws[0] = _pe_columns(ws[0], 3, 10, inc)
    ws[4] = torch.cat([ws[4][:, :256], _pe_columns(ws[4][:, 256:], 3, 10, inc)], 1)
    r = ws[8]
    parts = [r[:, :256], _pe_columns(r[:, 256:256 + (dir_w if use_dir else 0)], 3, 4, inc, use_dir)]
    if use_time:
        parts.append(_pe_columns(r[:, 256 + (dir_w if use_dir else 0):], 1, 10, inc))
    ws[8] = torch.cat(parts, 1)
    kind = ((hip.NET_SPACE_TIME_DEEP if deep else hip.NET_SPACE_TIME) if use_time
            else (hip.NET_SPACE_DEEP if deep else hip.NET_SPACE))
    return pack_net(kind, ws, bs, device, precision)


def pack_motionnet(state: dict, prefix: str, device="cuda", precision: str = "fp32") -> PackedNet:
    on_device = all(state[f"{prefix}.{k}.weight"].is_cuda for k in MOTIONNET_KEYS)
    ws = [state[f"{prefix}.{k}.weight"].detach().float() if on_device else state[f"{prefix}.{k}.weight"].detach().float().cpu() for k in MOTIONNET_KEYS]
    bs = [state[f"{prefix}.{k}.bias"] for k in MOTIONNET_KEYS]
    if ws[0].shape[1] not in (84, 80):
        raise ValueError(f"{prefix}.motion_net.0 has in-width {ws[0].shape[1]}: only c_input=4 with PE_10 is supported")
    ws[0] = _pe_columns(ws[0], 4, 10, ws[0].shape[1] == 84)
    return pack_net(hip.NET_MOTION, ws, bs, device, precision)


PROFILE_KERNELS = ("spacenet", "motionnet", "composite", "resample", "sample_coarse", "mlp_stage")


def profile_begin() -> None:
    """Start the library's launch profiler (HIP events on the launch stream around every kernel launch)."""
    hip.check(hip.lib().stnerf_profile_begin(), "stnerf_profile_begin")


def profile_end():
    """Stop profiling; -> list of dicts (kernel name, kind, n_rays, ns, tag, bytes_per_ray, ms) in launch order."""
    n = C.c_int(0)
    hip.check(hip.lib().stnerf_profile_end(None, 0, C.byref(n)), "stnerf_profile_end")
    if n.value == 0:
        return []
    buf = (hip.ProfileRecord * n.value)()
    hip.check(hip.lib().stnerf_profile_end(buf, n.value, C.byref(n)), "stnerf_profile_end")
    return [dict(kernel=PROFILE_KERNELS[r.kernel], kind=r.kind, n_rays=r.n_rays, ns=r.ns, tag=r.tag,
                 bytes_per_ray=r.bytes_per_ray, ms=r.ms) for r in buf]


def _worklist(ray_list, ray_count):
    return hip.dptr(ray_list, torch.int32, "ray_list"), hip.dptr(ray_count, torch.int32, "ray_count")


def spacenet_fwd(net: PackedNet, xyz: Tensor, dirs: Tensor, times: Optional[Tensor], raw: Tensor,
                 ray_list: Optional[Tensor] = None, ray_count: Optional[Tensor] = None) -> Tensor:
    """xyz (n,ns,3), dirs (n,3), times (n,) | None, raw (n,ns,4) out; all may be strided views whose
    dim 0 is the ray.  Writes raw {r,g,b,sigma} for the listed rays.  modeling/spacenet.py:101-160."""
    n, ns = xyz.shape[0], xyz.shape[1]
    if net.use_time and times is None:
        raise ValueError("this SpaceNet takes time: pass times (n,)")
    if net.precision == "bf16x3":   # the split-bf16 arithmetic exists as the stage kernel: a one-layer stage
        mlp_stage([dict(space=net, motion=None, xyz=xyz, raw=raw, times=times if net.use_time else None, ray_list=ray_list,
                        ray_count=ray_count)], dirs, ns, deep_rgb=net.kind in (hip.NET_SPACE_DEEP, hip.NET_SPACE_TIME_DEEP))
        return raw
    xp, xs = _strided_view_ptr(xyz, (ns, 3), "xyz")
    dp, ds = _strided_view_ptr(dirs, (3,), "dirs")
    rp, rs = _strided_view_ptr(raw, (ns, 4), "raw")
    if net.use_time:
        tp, ts = _strided_view_ptr(times.reshape(n), (), "times")
    else:
        tp, ts = C.c_void_p(0), 0
    lp, cp = _worklist(ray_list, ray_count)
    # workspace of the per-ray part of rgb_net.1 (stnerf_rgb_ray_bias): one row of 128 floats per ray
    ray_bias = torch.empty(n, 128, dtype=torch.float32, device=raw.device)
    hip.check(hip.lib().stnerf_spacenet_fwd(net.kind, hip.dptr(net.blob), n, ns, lp, cp, xp, xs, dp, ds, tp, ts, rp, rs,
                                            hip.dptr(ray_bias), hip.stream_ptr()), "stnerf_spacenet_fwd")
    return raw


def rgb_ray_bias(net: PackedNet, dirs: Tensor, times: Optional[Tensor], ray_list: Optional[Tensor] = None,
                 ray_count: Optional[Tensor] = None) -> Tensor:
    """The per-ray part of rgb_net.1 (stnerf_rgb_ray_bias): (n, 128) = bias + W[:, 256:] relu([PE_4(dir), PE_10(time)]) for
    the listed rays (other rows are zero here).  modeling/spacenet.py:80-86,141-151."""
    n = dirs.shape[0]
    if net.precision != "fp32":
        raise ValueError("stnerf_rgb_ray_bias reads the exact-f32 blob layout: pack the network 'fp32' for this call "
                         f"(got a {net.precision!r} blob)")
    dp, ds = _strided_view_ptr(dirs, (3,), "dirs")
    if net.use_time:
        if times is None:
            raise ValueError("this SpaceNet takes time: pass times (n,)")
        tp, ts = _strided_view_ptr(times.reshape(n), (), "times")
    else:
        tp, ts = C.c_void_p(0), 0
    lp, cp = _worklist(ray_list, ray_count)
    out = torch.zeros(n, 128, dtype=torch.float32, device=dirs.device)
    hip.check(hip.lib().stnerf_rgb_ray_bias(net.kind, hip.dptr(net.blob), n, lp, cp, dp, ds, tp, ts, hip.dptr(out),
                                            hip.stream_ptr()), "stnerf_rgb_ray_bias")
    return out


def motionnet_fwd(net: PackedNet, xyz: Tensor, times: Tensor, flow: Optional[Tensor] = None,
                  add_to_xyz: bool = True, ray_list: Optional[Tensor] = None, ray_count: Optional[Tensor] = None,
                  plain_time: bool = False):
    """xyz (n,ns,3) (updated in place if add_to_xyz), times (n,), flow (n,ns,3) out | None.
    modeling/motion_net.py:34-71 + layered_rfrender.py:355-356.  plain_time = MotionNet(input_time=False):
    the time column is encoded as given instead of lerping the encodings of floor(t) and floor(t)+1."""
    n, ns = xyz.shape[0], xyz.shape[1]
    if net.precision == "bf16x3":
        raise ValueError("a bf16x3 MotionNet runs fused in front of its SpaceNet (ops.mlp_stage); there is no stand-alone "
                         "launch -- pack it 'fp32' for the op-level call")
    xp, xs = _strided_view_ptr(xyz, (ns, 3), "xyz")
    tp, ts = _strided_view_ptr(times.reshape(n), (), "times")
    if flow is not None:
        fp, fs = _strided_view_ptr(flow, (ns, 3), "flow")
    else:
        fp, fs = C.c_void_p(0), 0
    lp, cp = _worklist(ray_list, ray_count)
    flags = (hip.MOTION_ADD_TO_XYZ if add_to_xyz else 0) | (hip.MOTION_PLAIN_TIME if plain_time else 0)
    hip.check(hip.lib().stnerf_motionnet_fwd(hip.dptr(net.blob), n, ns, lp, cp, xp, xs, tp, ts, fp, fs, flags,
                                             hip.stream_ptr()), "stnerf_motionnet_fwd")
    return flow


def mlp_stage(layers: Sequence[dict], dirs: Tensor, ns: int, deep_rgb: bool = False, sigmoid_rgb: bool = False) -> None:
    """One persistent launch over every listed layer (stnerf_mlp_stage).  Each dict: space (PackedNet), motion
    (PackedNet | None), xyz (n,ns,3), raw (n,ns,4) out, times (n,) | None, ray_list / ray_count | None,
    plain_time (bool).  Views may be strided as long as all layers share the ray strides.  The arithmetic follows the
    nets' packing: all "fp32" (exact f32 MFMA) or all "bf16x3" (split-bf16 MFMA)."""
    n = dirs.shape[0]
    arr = (hip.StageLayer * len(layers))()
    strides = None
    precs = set()
    dp, ds = _strided_view_ptr(dirs, (3,), "dirs")
    for i, ly in enumerate(layers):
        xp, xs = _strided_view_ptr(ly["xyz"], (ns, 3), "xyz")
        rp, rs = _strided_view_ptr(ly["raw"], (ns, 4), "raw")
        tp, ts = (_strided_view_ptr(ly["times"].reshape(n), (), "times") if ly.get("times") is not None else (C.c_void_p(0), 0))
        if strides is None:
            strides = [xs, rs, ts]
        if ts and not strides[2]:
            strides[2] = ts
        if (xs, rs) != tuple(strides[:2]) or (ts and ts != strides[2]):
            raise ValueError("mlp_stage: every layer must share the xyz / raw / times ray strides")
        precs.add(ly["space"].precision)
        if ly.get("motion") is not None:
            precs.add(ly["motion"].precision)
        lp, cp = _worklist(ly.get("ray_list"), ly.get("ray_count"))
        a = arr[i]
        a.space, a.motion = ly["space"].blob.data_ptr(), (ly["motion"].blob.data_ptr() if ly.get("motion") is not None else None)
        a.ray_list, a.ray_count, a.xyz, a.raw, a.times = lp.value, cp.value, xp.value, rp.value, tp.value
        a.use_time, a.motion_flags = int(ly["space"].use_time), (hip.MOTION_PLAIN_TIME if ly.get("plain_time") else 0)
    if precs not in ({"fp32"}, {"bf16x3"}):
        raise ValueError(f"mlp_stage: every network of a launch must be packed 'fp32' or every one 'bf16x3' (got {sorted(precs)})")
    bx = 4 if precs == {"bf16x3"} else 0
    queue = torch.zeros(1, dtype=torch.int32, device=dirs.device)
    ray_bias = torch.empty(len(layers), n, 128, dtype=torch.float32, device=dirs.device)   # rgb_net.1 per ray (stnerf_rgb_ray_bias)
    hip.check(hip.lib().stnerf_mlp_stage(arr, len(layers), n, ns, dp, ds, strides[2], strides[0], strides[1],
                                         (1 if deep_rgb else 0) | (2 if sigmoid_rgb else 0) | bx,
                                         C.c_void_p(queue.data_ptr()), hip.dptr(ray_bias), hip.stream_ptr()), "stnerf_mlp_stage")


def encode(x: Tensor, n_freq: int, include_input: bool = True) -> Tensor:
    """Positional encoding of x (..., dim) -> (..., dim*(include_input + 2*n_freq)); utils/dimension_kernel.py:3-73."""
    dim = x.shape[-1]
    flat = x.reshape(-1, dim).contiguous()
    y = torch.empty(flat.shape[0], dim * (int(include_input) + 2 * n_freq), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().stnerf_encode(hip.dptr(flat, name="x"), flat.shape[0], dim, n_freq, int(include_input),
                                      hip.dptr(y), hip.stream_ptr()), "stnerf_encode")
    return y.reshape(*x.shape[:-1], y.shape[-1])


def gen_weight(sigma: Tensor, delta: Tensor) -> Tensor:
    """sigma (n,S) raw, delta (n,S) -> weights (n,S); layers/render_layer.py:8-17."""
    n, S = delta.shape
    sg = sigma.reshape(n, S).contiguous()
    w = torch.empty(n, S, dtype=torch.float32, device=delta.device)
    hip.check(hip.lib().stnerf_gen_weight(hip.dptr(sg, name="sigma"), hip.dptr(delta.contiguous(), name="delta"), n, S,
                                          hip.dptr(w), hip.stream_ptr()), "stnerf_gen_weight")
    return w


# ---------------------------------------------------------------------------------------- a10-a13
def composite_params(border: float = 1e10, near: float = 0.0, fine: bool = False, cut_negative_t: bool = False,
                     thresholds: Optional[Sequence[Optional[float]]] = None, sigma_scale: Optional[Sequence[float]] = None,
                     evaluated: Optional[Sequence[int]] = None, rgb_activated: bool = False) -> "hip.CompositeParams":
    """The stnerf_composite_params of a stage (include/stnerf.h): shared by ``composite`` and ``composite_bwd``."""
    p = hip.CompositeParams()
    p.border, p.near, p.fine, p.cut_negative_t = border, near, int(fine), int(cut_negative_t)
    p.rgb_activated = int(rgb_activated)
    for i in range(hip.MAX_LAYERS):
        th = thresholds[i] if thresholds is not None and i < len(thresholds) else None
        p.threshold[i] = 0.0 if th is None else float(th)
        p.use_threshold[i] = 0 if th is None else 1
        p.sigma_scale[i] = float(sigma_scale[i]) if sigma_scale is not None and i < len(sigma_scale) else 1.0
        p.evaluated[i] = int(evaluated[i]) if evaluated is not None and i < len(evaluated) else 1
    return p


def composite(t: Tensor, raw: Tensor, mask: Optional[Tensor], border: float = 1e10, near: float = 0.0,
              fine: bool = False, cut_negative_t: bool = False, thresholds: Optional[Sequence[Optional[float]]] = None,
              sigma_scale: Optional[Sequence[float]] = None, evaluated: Optional[Sequence[int]] = None,
              want_weights: bool = False, want_order: bool = False, rgb_activated: bool = False, two_pass: bool = True,
              params: Optional["hip.CompositeParams"] = None):
    """t (n,l,S), raw (n,l,S,4), mask (n,l) uint8 | None ->
    layer_out (n,l,5), mixed_out (n,5), weights (n,l,S) | None, order (n,l*S) int32 | None.
    layers/render_layer.py:8-58 + modeling/layered_rfrender.py:414-448 / :538-606.
    evaluated[i]: 0 = hidden layer, 1 = evaluated where mask is set, 2 = evaluated on every ray (the background)."""
    n, l, S = t.shape
    p = params if params is not None else composite_params(border, near, fine, cut_negative_t, thresholds, sigma_scale, evaluated,
                                                           rgb_activated)
    layer_out = torch.empty(n, l, 5, dtype=torch.float32, device=t.device)
    mixed_out = torch.empty(n, 5, dtype=torch.float32, device=t.device)
    weights = torch.empty(n, l, S, dtype=torch.float32, device=t.device) if want_weights else None
    order = torch.empty(n, l * S, dtype=torch.int32, device=t.device) if want_order else None
    scratch = torch.empty(n, dtype=torch.uint8, device=t.device) if two_pass else None   # single-layer rays first (see the header)
    hip.check(hip.lib().stnerf_composite(hip.dptr(t, name="t"), hip.dptr(raw, name="raw"),
                                         hip.dptr(mask, torch.uint8, "mask"), n, l, S, C.byref(p), hip.dptr(layer_out),
                                         hip.dptr(mixed_out), hip.dptr(weights), hip.dptr(order, torch.int32),
                                         hip.dptr(scratch, torch.uint8), hip.stream_ptr()), "stnerf_composite")
    return layer_out, mixed_out, weights, order


def composite_bwd(t: Tensor, raw: Tensor, mask: Optional[Tensor], order: Optional[Tensor], params: "hip.CompositeParams",
                  g_layer: Optional[Tensor], g_mixed: Optional[Tensor]) -> Tensor:
    """dLoss/d raw (n,l,S,4) from dLoss/d layer_out (n,l,5) and dLoss/d mixed_out (n,5) (either may be None):
    stnerf_composite_bwd, the backward of ``composite`` called with the same t / raw / mask / params (order: its output)."""
    n, l, S = t.shape
    d_raw = torch.empty(n, l, S, 4, dtype=torch.float32, device=t.device)
    hip.check(hip.lib().stnerf_composite_bwd(hip.dptr(t, name="t"), hip.dptr(raw, name="raw"), hip.dptr(mask, torch.uint8, "mask"),
                                             hip.dptr(order, torch.int32, "order"), n, l, S, C.byref(params),
                                             hip.dptr(g_layer, name="g_layer"), hip.dptr(g_mixed, name="g_mixed"), hip.dptr(d_raw),
                                             hip.stream_ptr()), "stnerf_composite_bwd")
    return d_raw


def resample(t: Tensor, weights: Tensor, n2: int, rays: Tensor, u: Optional[Tensor] = None, seed: int = 0,
             ray_index_base: int = 0, edits=None, pivot=None, want_xyz: bool = True, debug: bool = False,
             ray_index_stripe: int = 0, ray_index_period: int = 0, mask: Optional[Tensor] = None):
    """t (n,l,n1), weights (n,l,n1) -> t_fine (n,l,n1+n2) ascending, xyz_fine (n,l,n1+n2,3) | None
    [, z_new (n,l,n2), inds (n,l,n2) int32, cdf (n,l,n1-1) if debug].
    utils/sample_pdf.py:18-63 + modeling/layered_rfrender.py:459-475.  ``mask`` (n,l) uint8 with the sampler's hints
    (``sample_coarse(raw_mask=True)``): pairs flagged "missed" are skipped, their output rows stay unwritten."""
    n, l, n1 = t.shape
    if u is not None and tuple(u.shape) != (l, n, n2):
        raise ValueError(f"u must be (l,n,n2) = {(l, n, n2)}, got {tuple(u.shape)}")
    dev = t.device
    t_fine = torch.empty(n, l, n1 + n2, dtype=torch.float32, device=dev)
    xyz = torch.empty(n, l, n1 + n2, 3, dtype=torch.float32, device=dev) if want_xyz else None
    z_new = torch.empty(n, l, n2, dtype=torch.float32, device=dev) if debug else None
    inds = torch.empty(n, l, n2, dtype=torch.int32, device=dev) if debug else None
    cdf = torch.empty(n, l, n1 - 1, dtype=torch.float32, device=dev) if debug else None
    ed, pv = _edits(edits, pivot, l)
    hip.check(hip.lib().stnerf_resample(hip.dptr(t, name="t"), hip.dptr(weights, name="weights"), n, l, n1, n2,
                                        hip.dptr(u, name="u"), seed, ray_index_base, ray_index_stripe, ray_index_period,
                                        hip.dptr(rays, name="rays"),
                                        rays.shape[1], ed, pv, hip.dptr(mask, torch.uint8, "mask"), hip.dptr(t_fine), hip.dptr(xyz), hip.dptr(z_new),
                                        hip.dptr(inds, torch.int32), hip.dptr(cdf), hip.stream_ptr()),
              "stnerf_resample")
    if debug:
        return t_fine, xyz, z_new, inds, cdf
    return t_fine, xyz


# ---------------------------------------------------------------------------------------- whole pipeline
def fill_edits(dst, edits, l):
    """Copy per-layer (shift|None, scale|None) pairs into a LayerEdit array field of RenderParams."""
    for i in range(l):
        sh, sc = edits[i] if (edits is not None and i < len(edits)) else (None, None)
        dst[i].has_shift, dst[i].has_scale, dst[i].scale = 0, 0, 1.0
        if sh is not None:
            v = torch.tensor(sh, dtype=torch.float32).tolist()   # fp32, as torch.tensor(shift[i]) in the reference
            dst[i].shift[0], dst[i].shift[1], dst[i].shift[2] = v
            dst[i].has_shift = 1
        if sc is not None:
            dst[i].scale, dst[i].has_scale = float(sc), 1


def render_workspace_bytes(n: int, l: int, n1: int, n2: int, only_coarse: bool) -> int:
    nb = hip.lib().stnerf_render_workspace_bytes(n, l, n1, n2, int(only_coarse))
    if nb < 0:
        hip.check(int(nb), "stnerf_render_workspace_bytes")
    return int(nb)


def render_rays(rays: Tensor, boxes: Tensor, nets: "hip.Nets", params: "hip.RenderParams", workspace: Tensor,
                jitter: Optional[Tensor] = None, u: Optional[Tensor] = None):
    """One call = the whole chunk pipeline (stnerf_render_rays).  Returns mixed_fine (n,5), mixed_coarse (n,5),
    layer_fine (n,l,5), layer_coarse (n,l,5), mask (n,l) uint8 (fine outputs alias the coarse ones if only_coarse)."""
    n, l = rays.shape[0], params.l
    bp, bstride, lb = _boxes_arg(boxes, n)
    if lb != l:
        raise ValueError(f"boxes carry {lb} layers, params.l = {l}")
    dev = rays.device
    mix_c = torch.empty(n, 5, dtype=torch.float32, device=dev)
    lo_c = torch.empty(n, l, 5, dtype=torch.float32, device=dev)
    mask = torch.empty(n, l, dtype=torch.uint8, device=dev)
    if params.only_coarse:
        mix_f, lo_f = None, None
    else:
        mix_f = torch.empty(n, 5, dtype=torch.float32, device=dev)
        lo_f = torch.empty(n, l, 5, dtype=torch.float32, device=dev)
    hip.check(hip.lib().stnerf_render_rays(hip.dptr(rays, name="rays"), n, bp, bstride, C.byref(nets), C.byref(params),
                                           hip.dptr(jitter, name="jitter"), hip.dptr(u, name="u"),
                                           hip.dptr(workspace, torch.uint8, "workspace"), workspace.numel(),
                                           hip.dptr(mix_f), hip.dptr(mix_c), hip.dptr(lo_f), hip.dptr(lo_c),
                                           hip.dptr(mask, torch.uint8), hip.stream_ptr()), "stnerf_render_rays")
    if params.only_coarse:
        return mix_c, mix_c, lo_c, lo_c, mask
    return mix_f, mix_c, lo_f, lo_c, mask


# ---------------------------------------------------------------------------------------- 8(f)4: training GEMMs
def _mat(t: Tensor, name: str, vec: bool = False):
    """(pointer, row stride) of a 2-D fp32 device view whose columns are dense; `vec`: read with 16-byte vectors."""
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != torch.float32 or t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f"{name}: expected a 2-D float32 view with dense columns, got {t.dtype} {tuple(t.shape)} strides {t.stride()}")
    if vec and (t.stride(0) % 4 or t.data_ptr() % 16):
        raise ValueError(f"{name}: rows must be 16-byte aligned (row stride {t.stride(0)} floats, pointer % 16 = {t.data_ptr() % 16})")
    return C.c_void_p(t.data_ptr()), t.stride(0)


def train_linear_fwd(x: Tensor, w: Tensor, bias: Optional[Tensor], y: Tensor, relu: bool) -> Tensor:
    """y (m,n) = act(x (m,k) @ w (n,k).T + bias): stnerf_train_linear_fwd (f32 MFMA).  x / w: views into padded storage."""
    (m, k), n = x.shape, w.shape[0]
    xp, ldx = _mat(x, "x", True)
    wp, ldw = _mat(w, "w", True)
    yp, ldy = _mat(y, "y")
    if w.shape[1] != k or tuple(y.shape) != (m, n):
        raise ValueError(f"train_linear_fwd: x {tuple(x.shape)}, w {tuple(w.shape)}, y {tuple(y.shape)}")
    hip.check(hip.lib().stnerf_train_linear_fwd(xp, ldx, wp, ldw, hip.dptr(bias, name="bias"), m, n, k, int(relu), yp, ldy,
                                                hip.stream_ptr()), "stnerf_train_linear_fwd")
    return y


def train_linear_dx(dy: Tensor, w: Tensor, dx: Tensor, mask: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """dx (m,k) (+)= (dy (m,n) @ w (n,k)) * (mask > 0): stnerf_train_linear_dx."""
    (m, n), k = dy.shape, w.shape[1]
    dp, lddy = _mat(dy, "dy", True)
    wp, ldw = _mat(w, "w", True)
    xp, lddx = _mat(dx, "dx")
    mp, ldm = _mat(mask, "mask") if mask is not None else (C.c_void_p(0), 0)
    if w.shape[0] != n or tuple(dx.shape) != (m, k) or (mask is not None and tuple(mask.shape) != (m, k)):
        raise ValueError(f"train_linear_dx: dy {tuple(dy.shape)}, w {tuple(w.shape)}, dx {tuple(dx.shape)}")
    hip.check(hip.lib().stnerf_train_linear_dx(dp, lddy, wp, ldw, m, n, k, mp, ldm, int(accumulate), xp, lddx, hip.stream_ptr()),
              "stnerf_train_linear_dx")
    return dx


_DW_WORKSPACE: dict = {}


def train_linear_dw(dy: Tensor, x: Tensor, dw: Tensor, db: Optional[Tensor], accumulate: bool) -> None:
    """dw (n,k) (+)= dy (m,n).T @ x (m,k); db (n,) (+)= dy.sum(0): stnerf_train_linear_dw (deterministic split reduction)."""
    (m, n), k = dy.shape, x.shape[1]
    dp, lddy = _mat(dy, "dy", True)
    xp, ldx = _mat(x, "x", True)
    wp, lddw = _mat(dw, "dw")
    if x.shape[0] != m or tuple(dw.shape) != (n, k) or (db is not None and tuple(db.shape) != (n,)):
        raise ValueError(f"train_linear_dw: dy {tuple(dy.shape)}, x {tuple(x.shape)}, dw {tuple(dw.shape)}")
    ws = _dw_workspace(int(hip.lib().stnerf_train_dw_workspace_bytes(m, n, k)), dy.device)
    hip.check(hip.lib().stnerf_train_linear_dw(dp, lddy, xp, ldx, m, n, k, wp, lddw, hip.dptr(db, name="db"), int(accumulate),
                                               hip.dptr(ws, torch.uint8, "workspace"), ws.numel(), hip.stream_ptr()), "stnerf_train_linear_dw")


def free_workspaces() -> None:
    """Drop the cached weight-gradient workspaces (tens of MB per (device, stream) that ran a backward); they are re-created on demand."""
    _DW_WORKSPACE.clear()


def _dw_workspace(need: int, device) -> Tensor:
    # one workspace per device and stream, grown on demand (a backward's calls run in order on one stream, so the partial tiles of
    # one may overwrite the previous call's -- ADVICE r04).  ASSUMES one host thread per stream: two threads that enqueue backwards on
    # the SAME stream would share the partial tiles (the trainer is single-threaded, one process per GPU).  The cache only
    # grows; free_workspaces() releases it.
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _DW_WORKSPACE.get(key)
    if ws is None or ws.numel() < need:
        ws = _DW_WORKSPACE[key] = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=device)
    return ws


DW_BATCH_MAX = 16


def train_dw_batch(layers: Sequence[Tuple[Tensor, Tensor, Tensor, Optional[Tensor]]], accumulate: bool) -> None:
    """For every (dy (m,n), x (m,k), dw (n,k), db (n,) | None) of `layers` (the same m): dw (+)= dy.T @ x, db (+)= dy.sum(0) --
    stnerf_train_dw_batch: all weight and bias gradients of a network in one launch and one deterministic reduction."""
    if not layers:
        return
    if len(layers) > DW_BATCH_MAX:
        raise ValueError(f"train_dw_batch: {len(layers)} layers, at most {DW_BATCH_MAX}")
    m = layers[0][0].shape[0]
    arr = (hip.DwProblem * len(layers))()
    for i, (dy, x, dw, db) in enumerate(layers):
        n, k = dy.shape[1], x.shape[1]
        dp, lddy = _mat(dy, f"dy[{i}]", True)
        xp, ldx = _mat(x, f"x[{i}]", True)
        wp, lddw = _mat(dw, f"dw[{i}]")
        if dy.shape[0] != m or x.shape[0] != m or tuple(dw.shape) != (n, k) or (db is not None and tuple(db.shape) != (n,)):
            raise ValueError(f"train_dw_batch: layer {i}: dy {tuple(dy.shape)}, x {tuple(x.shape)}, dw {tuple(dw.shape)}, m = {m}")
        bp = hip.dptr(db, name=f"db[{i}]")
        arr[i] = hip.DwProblem(dp.value, lddy, xp.value, ldx, wp.value, lddw, bp.value if bp is not None else None, n, k)
    need = int(hip.lib().stnerf_train_dw_batch_workspace_bytes(arr, len(layers), m))
    if need < 0:
        hip.check(need, "stnerf_train_dw_batch_workspace_bytes")
    ws = _dw_workspace(need, layers[0][0].device)
    hip.check(hip.lib().stnerf_train_dw_batch(arr, len(layers), m, int(accumulate), hip.dptr(ws, torch.uint8, "workspace"), ws.numel(),
                                              hip.stream_ptr()), "stnerf_train_dw_batch")


def train_encode(x: Tensor, y: Tensor, n_freq: int, include_input: bool = True, rows_per_src: int = 1, relu: bool = False,
                 lerp_col: int = -1) -> Tensor:
    """Positional encoding of x (src, dim) into the columns of the view y (rows, dim (include_input + 2 n_freq)):
    stnerf_train_encode (rows = src * rows_per_src)."""
    xp, ldx = _mat(x, "x")
    yp, ldy = _mat(y, "y")
    dim, rows = x.shape[1], y.shape[0]
    if y.shape[1] != dim * (int(include_input) + 2 * n_freq) or rows != x.shape[0] * rows_per_src:
        raise ValueError(f"train_encode: x {tuple(x.shape)} -> y {tuple(y.shape)}")
    hip.check(hip.lib().stnerf_train_encode(xp, ldx, dim, n_freq, int(include_input), rows, rows_per_src, int(relu), lerp_col, yp, ldy, 0,
                                            hip.stream_ptr()), "stnerf_train_encode")
    return y


def train_encode_bwd(x: Tensor, dy: Tensor, dx: Tensor, n_freq: int, include_input: bool = True, accumulate: bool = False) -> Tensor:
    """dx (rows, dim_out) (+)= d enc / d x . dy for the first dim_out = dx.shape[1] input columns: stnerf_train_encode_bwd."""
    xp, ldx = _mat(x, "x")
    dp, lddy = _mat(dy, "dy")
    op, lddx = _mat(dx, "dx")
    dim, rows = x.shape[1], x.shape[0]
    if dy.shape[1] != dim * (int(include_input) + 2 * n_freq) or dy.shape[0] != rows or dx.shape[0] != rows or dx.shape[1] > dim:
        raise ValueError(f"train_encode_bwd: x {tuple(x.shape)}, dy {tuple(dy.shape)}, dx {tuple(dx.shape)}")
    hip.check(hip.lib().stnerf_train_encode_bwd(xp, ldx, dim, n_freq, int(include_input), rows, dp, lddy, 0, dx.shape[1], int(accumulate),
                                                op, lddx, hip.stream_ptr()), "stnerf_train_encode_bwd")
    return dx


# ---------------------------------------------------------------------------------------- fused training launches (csrc/train_wave.hip)
def _matrix_list(mats: Sequence[Tensor], name: str):
    ptrs = (C.c_void_p * len(mats))()
    lds = (C.c_int32 * len(mats))()
    for i, m in enumerate(mats):
        p, ld = _mat(m, f"{name}[{i}]")
        ptrs[i], lds[i] = p.value, ld
    return ptrs, lds


def _bit_planes(bits: Tensor, rows: int):
    """(pointer, stage stride in words) of the ReLU bit planes: int32 (8, rows, 8), possibly a row range ``planes[:, a:b]`` of a larger
    launch's planes (rows and words dense, any stage stride)."""
    if not bits.is_cuda or bits.dtype != torch.int32 or tuple(bits.shape) != (8, rows, 8) or bits.stride(2) != 1 or bits.stride(1) != 8:
        raise ValueError(f"relu_bits must be a device int32 (8, {rows}, 8) with dense rows, got {bits.dtype} {tuple(bits.shape)} {bits.stride()}")
    return C.c_void_p(bits.data_ptr()), bits.stride(0)


def train_spacenet_fwd(net: PackedNet, xyz: Tensor, dirs: Tensor, times: Optional[Tensor], raw: Tensor, acts: Sequence[Tensor],
                       pe: Tensor, relu_bits: Optional[Tensor] = None) -> None:
    """The stage kernel of the network's arithmetic (exact f32: stnerf_train_spacenet_fwd; split bf16: ..._bf16x3) on one SpaceNet,
    every ray, writing each layer's input as it goes: acts[0..6] (rows, 256), acts[7] (rows, 128), pe (rows, 64), rows = n * ns in
    (ray, sample) order -- views into padded row-major storage -- and relu_bits (8, rows, 8) int32: the ReLU masks as bit planes.
    xyz (n,ns,3), dirs (n,3), times (n,) | None, raw (n,ns,4) out."""
    entry = "stnerf_train_spacenet_fwd" if net.precision == "fp32" else "stnerf_train_spacenet_fwd_bf16x3"
    n, ns = xyz.shape[0], xyz.shape[1]
    xp, xs = _strided_view_ptr(xyz, (ns, 3), "xyz")
    rp, rs = _strided_view_ptr(raw, (ns, 4), "raw")
    dp, ds = _strided_view_ptr(dirs, (3,), "dirs")
    tp, ts = _strided_view_ptr(times.reshape(n), (), "times") if times is not None else (C.c_void_p(0), 0)
    ptrs, lds = _matrix_list(acts, "acts")
    pp, ldp = _mat(pe, "pe")
    bp, bstride = _bit_planes(relu_bits, n * ns) if relu_bits is not None else (C.c_void_p(0), 0)
    queue = torch.zeros(1, dtype=torch.int32, device=xyz.device)
    ray_bias = torch.empty(n, 128, dtype=torch.float32, device=xyz.device)
    hip.check(getattr(hip.lib(), entry)(net.kind, hip.dptr(net.blob), n, ns, xp, xs, dp, ds, tp, ts, rp, rs, ptrs, lds, pp, ldp,
                                        bp, bstride, hip.dptr(queue, torch.int32), hip.dptr(ray_bias), hip.stream_ptr()), entry)


def train_spacenet_dx(wt: Tensor, offsets: Sequence[int], d_raw: Tensor, relu_bits: Tensor, dys: Sequence[Tensor], dpe: Optional[Tensor]) -> None:
    """The backward chain through one SpaceNet's layers (stnerf_train_spacenet_dx): d_raw (rows,4) -> dys[0..7] (the layers'
    pre-activation gradients, widths 256 x 7, 128) and, if given, dpe (rows,64) = dLoss / d PE(pos).  relu_bits (8, rows, 8) int32:
    the masks ``train_spacenet_fwd`` wrote for these rows.  wt / offsets: the transposed weight sections
    (stnerf_amd.modeling.autograd.transposed_spacenet)."""
    rows = d_raw.shape[0]
    bp, bstride = _bit_planes(relu_bits, rows)
    yp, yld = _matrix_list(dys, "dys")
    off = (C.c_uint32 * 10)(*[int(o) for o in offsets])
    pp, ldp = _mat(dpe, "dpe") if dpe is not None else (C.c_void_p(0), 0)
    hip.check(hip.lib().stnerf_train_spacenet_dx(hip.dptr(wt, name="wt"), off, hip.dptr(d_raw, name="d_raw"), rows,
                                                 bp, bstride, yp, yld, pp, ldp, hip.stream_ptr()),
              "stnerf_train_spacenet_dx")


def pack_dx_bf16x3(kind: int, weights: Sequence[Tensor], with_dpos: bool) -> Tensor:
    """The split-bf16 backward chain's weights in one blob (stnerf_pack_dx_bf16x3_device): the network's 10 weight tensors in reference
    layout (device, fp32) -> a 1 KB-aligned uint8 device tensor.  with_dpos: with the two half passes towards PE(pos)."""
    lib = hip.lib()
    nbytes = lib.stnerf_packed_bytes_dx_bf16x3(kind, int(with_dpos))
    if nbytes < 0:
        hip.check(int(nbytes), "stnerf_packed_bytes_dx_bf16x3")
    ws = [w.detach().to(torch.float32).contiguous() for w in weights]
    if not all(w.is_cuda for w in ws):
        raise ValueError("pack_dx_bf16x3 packs on the device: the weights must be CUDA tensors")
    dev = ws[0].device
    raw = torch.empty(nbytes + 1024, dtype=torch.uint8, device=dev)
    off = (-raw.data_ptr()) % 1024
    blob = raw[off:off + nbytes]
    wp = (C.c_void_p * len(ws))(*(w.data_ptr() for w in ws))
    with torch.cuda.device(dev):
        hip.check(lib.stnerf_pack_dx_bf16x3_device(kind, wp, len(ws), int(with_dpos), C.c_void_p(blob.data_ptr()), nbytes, hip.stream_ptr()),
                  "stnerf_pack_dx_bf16x3_device")
    return blob


def train_spacenet_dx_bf16x3(blob: Tensor, d_raw: Tensor, relu_bits: Tensor, dys: Sequence[Tensor], dpe: Optional[Tensor],
                             dpe_skip: Optional[Tensor]) -> None:
    """``train_spacenet_dx`` in split bf16 (stnerf_train_spacenet_dx_bf16x3): blob = ``pack_dx_bf16x3``; dLoss / d PE(pos) leaves as
    dpe + dpe_skip (both (rows,64)) when the blob was packed with_dpos, else both None."""
    rows = d_raw.shape[0]
    bp, bstride = _bit_planes(relu_bits, rows)
    yp, yld = _matrix_list(dys, "dys")
    pp, ldp = _mat(dpe, "dpe") if dpe is not None else (C.c_void_p(0), 0)
    sp, lds_ = _mat(dpe_skip, "dpe_skip") if dpe_skip is not None else (C.c_void_p(0), 0)
    hip.check(hip.lib().stnerf_train_spacenet_dx_bf16x3(C.c_void_p(blob.data_ptr()), int(dpe is not None), hip.dptr(d_raw, name="d_raw"), rows,
                                                        bp, bstride, yp, yld, pp, ldp, sp, lds_, hip.stream_ptr()),
              "stnerf_train_spacenet_dx_bf16x3")


def _motion_bit_planes(bits: Tensor, rows: int):
    """(pointer, stage stride in words) of a MotionNet's ReLU bit planes: int32 (5, rows, 4), possibly a row range of a larger launch's."""
    if not bits.is_cuda or bits.dtype != torch.int32 or tuple(bits.shape) != (5, rows, 4) or bits.stride(2) != 1 or bits.stride(1) != 4:
        raise ValueError(f"relu_bits must be a device int32 (5, {rows}, 4) with dense rows, got {bits.dtype} {tuple(bits.shape)} {bits.stride()}")
    return C.c_void_p(bits.data_ptr()), bits.stride(0)


def train_motionnet_fwd(net: PackedNet, xt: Tensor, flow: Tensor, enc: Tensor, acts: Sequence[Tensor], relu_bits: Tensor,
                        plain_time: bool = False) -> None:
    """flow (rows,3) = MotionNet(xt (rows,4) = [xyz | frame id]) with the inference kernels' exact-f32 arithmetic, writing the staged
    encoding enc (rows, >= 88), the five post-ReLU outputs acts[0..4] (rows,128) -- views into padded row-major storage -- and their
    masks as bit planes relu_bits (5, rows, 4) int32 (stnerf_train_motionnet_fwd)."""
    if net.kind != hip.NET_MOTION or net.precision != "fp32":
        raise ValueError("train_motionnet_fwd wants an exact-f32 packed MotionNet")
    rows = xt.shape[0]
    if tuple(xt.shape) != (rows, 4) or not xt.is_contiguous() or tuple(flow.shape) != (rows, 3) or not flow.is_contiguous() or len(acts) != 5:
        raise ValueError(f"train_motionnet_fwd: xt {tuple(xt.shape)}, flow {tuple(flow.shape)}, {len(acts)} activation matrices")
    ap, ald = _matrix_list(acts, "acts")
    ep, lde = _mat(enc, "enc", True)
    if enc.shape != (rows, enc.shape[1]) or enc.shape[1] < 88 or any(tuple(m.shape) != (rows, 128) for m in acts):
        raise ValueError("train_motionnet_fwd: enc needs >= 88 columns, the activation matrices 128, one row per sample")
    bp, bstride = _motion_bit_planes(relu_bits, rows)
    hip.check(hip.lib().stnerf_train_motionnet_fwd(hip.dptr(net.blob), hip.dptr(xt, name="xt"), rows, hip.MOTION_PLAIN_TIME if plain_time else 0,
                                                   hip.dptr(flow, name="flow"), ep, lde, ap, ald, bp, bstride, hip.stream_ptr()),
              "stnerf_train_motionnet_fwd")


def train_motionnet_dx(wt: Tensor, offsets: Sequence[int], d_flow: Tensor, relu_bits: Tensor, dys: Sequence[Tensor], denc: Optional[Tensor]) -> None:
    """The backward chain through one MotionNet (stnerf_train_motionnet_dx): d_flow (rows,3) as a view of (rows,4) storage -> dys[0..4]
    (rows,128), the layers' pre-activation gradients, and, if given, denc (rows, >= 96) = dLoss / d encoding.  wt / offsets:
    stnerf_amd.modeling.autograd.transposed_motionnet."""
    rows = d_flow.shape[0]
    dp, ldf = _mat(d_flow, "d_flow", True)
    if d_flow.shape[1] != 3 or ldf != 4 or len(dys) != 5 or any(tuple(m.shape) != (rows, 128) for m in dys):
        raise ValueError(f"train_motionnet_dx: d_flow {tuple(d_flow.shape)} (row stride {ldf}), {len(dys)} gradient matrices")
    bp, bstride = _motion_bit_planes(relu_bits, rows)
    yp, yld = _matrix_list(dys, "dys")
    off = (C.c_uint32 * 6)(*[int(o) for o in offsets])
    pp, ldp = _mat(denc, "denc", True) if denc is not None else (C.c_void_p(0), 0)
    if denc is not None and (denc.shape[0] != rows or denc.shape[1] < 96):
        raise ValueError(f"train_motionnet_dx: denc {tuple(denc.shape)} needs {rows} rows of >= 96 columns")
    hip.check(hip.lib().stnerf_train_motionnet_dx(hip.dptr(wt, name="wt"), off, dp, rows, bp, bstride, yp, yld, pp, ldp, hip.stream_ptr()),
              "stnerf_train_motionnet_dx")


def pack_transposed(sections: Sequence[Tuple[Tensor, int]], dst: Tensor) -> List[int]:
    """The A operands of the fused backward chains in one launch (stnerf_pack_transposed): for every (W (out, in) fp32 device view with
    dense columns, n_pad) a section [out / 4][n_pad][4] of `dst` (n_pad = 0: W copied as it is), packed back to back from offset 0.
    -> the sections' float offsets."""
    arr = (hip.TransposeSection * len(sections))()
    offsets, off = [], 0
    for i, (w, n_pad) in enumerate(sections):
        wp, ldw = _mat(w, f"w[{i}]")
        out, k = w.shape
        arr[i] = hip.TransposeSection(wp.value, ldw, off, out, k, n_pad)
        offsets.append(off)
        off += out * (n_pad if n_pad else k)
    if dst.dtype != torch.float32 or not dst.is_contiguous() or dst.numel() < off:
        raise ValueError(f"pack_transposed: the destination needs {off} contiguous fp32 values")
    hip.check(hip.lib().stnerf_pack_transposed(arr, len(sections), hip.dptr(dst, name="dst"), dst.numel(), hip.stream_ptr()),
              "stnerf_pack_transposed")
    return offsets
