"""Deterministic synthetic scenes and weights.

There is no dataset and no checkpoint for the reference (``.MISSING_LARGE_BLOBS``), so every
parity test, golden fixture and benchmark runs on the scene defined here: a pinhole camera looking
down +z at side-by-side performer slabs inside a (-3,3)^3 background box, and weights drawn from a
frozen ``numpy.random.RandomState`` stream (stable across numpy versions, independent of torch's
RNG) under the reference's state_dict key names (checkpoint row).
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import numpy as np
import torch

# (key suffix, out features, in features) per network; in-widths follow the reference's
# encodings: PE_10(xyz)=63, PE_4(dir)=27, PE_10(t)=21, PE_10(xyzt)=84.
_SPACENET_LAYERS = [("stage1.0", 256, 63), ("stage1.2", 256, 256), ("stage1.4", 256, 256),
                    ("stage1.6", 256, 256), ("stage2.0", 256, 319), ("stage2.2", 256, 256),
                    ("stage2.4", 256, 256), ("density_net.0", 1, 256)]
_MOTION_LAYERS = [("motion_net.0", 128, 84), ("motion_net.2", 128, 128), ("motion_net.4", 128, 128),
                  ("motion_net.6", 128, 128), ("motion_net.8", 128, 128), ("motion_net.10", 3, 128)]


def aabb_corners(lo: Sequence[float], hi: Sequence[float]) -> torch.Tensor:
    """8 corners in the reference's order (data/datasets/frame_dataset.py:187-188)."""
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    return torch.tensor([[x0, y0, z0], [x1, y0, z0], [x1, y1, z0], [x0, y1, z0],
                         [x0, y0, z1], [x1, y0, z1], [x1, y1, z1], [x0, y1, z1]], dtype=torch.float32)


def _linear(rs: np.random.RandomState, out_f: int, in_f: int, gain: float = 1.0, bias_shift: float = 0.0):
    bound = 1.0 / np.sqrt(in_f)  # nn.Linear default: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    w = rs.uniform(-bound, bound, size=(out_f, in_f)).astype(np.float32) * np.float32(gain)
    b = rs.uniform(-bound, bound, size=(out_f,)).astype(np.float32) * np.float32(gain) + np.float32(bias_shift)
    return torch.from_numpy(w), torch.from_numpy(b)


def spacenet_state(prefix: str, rs: np.random.RandomState, use_time: bool,
                   sigma_gain: float = 60.0, sigma_bias: float = 0.5, deep_rgb: bool = False,
                   include_input: bool = True, use_dir: bool = True) -> Dict[str, torch.Tensor]:
    """One SpaceNet (modeling/spacenet.py:45-86).  ``sigma_gain``/``sigma_bias`` make the density
    head 'trained-like' (random init gives sigma ~ 0 and a numerically trivial composite)."""
    sd = {}
    drop = 0 if include_input else 3     # TKERNEL_INC_RAW=False: the encodings lose their raw-input block
    for name, o, i in _SPACENET_LAYERS:
        g, bs = (sigma_gain, sigma_bias) if name == "density_net.0" else (1.0, 0.0)
        i = i - drop if name in ("stage1.0", "stage2.0") else i
        sd[f"{prefix}.{name}.weight"], sd[f"{prefix}.{name}.bias"] = _linear(rs, o, i, g, bs)
    rgb_in = 256 + ((27 - drop) if use_dir else 0) + ((21 - (0 if include_input else 1)) if use_time else 0)
    sd[f"{prefix}.rgb_net.1.weight"], sd[f"{prefix}.rgb_net.1.bias"] = _linear(rs, 128, rgb_in)
    if deep_rgb:  # modeling/spacenet.py:68-79: two more 128-wide hidden layers
        sd[f"{prefix}.rgb_net.3.weight"], sd[f"{prefix}.rgb_net.3.bias"] = _linear(rs, 128, 128)
        sd[f"{prefix}.rgb_net.5.weight"], sd[f"{prefix}.rgb_net.5.bias"] = _linear(rs, 128, 128)
        sd[f"{prefix}.rgb_net.7.weight"], sd[f"{prefix}.rgb_net.7.bias"] = _linear(rs, 3, 128, 4.0)
    else:
        sd[f"{prefix}.rgb_net.3.weight"], sd[f"{prefix}.rgb_net.3.bias"] = _linear(rs, 3, 128, 4.0)
    return sd


def motionnet_state(prefix: str, rs: np.random.RandomState, flow_gain: float = 0.25,
                    include_input: bool = True) -> Dict[str, torch.Tensor]:
    """One MotionNet (modeling/motion_net.py:20-32)."""
    sd = {}
    for name, o, i in _MOTION_LAYERS:
        g = flow_gain if name == "motion_net.10" else 1.0
        i = i - 4 if (name == "motion_net.0" and not include_input) else i
        sd[f"{prefix}.{name}.weight"], sd[f"{prefix}.{name}.bias"] = _linear(rs, o, i, g)
    return sd


def make_state_dict(layer_num: int, use_space_time: bool, use_deform_time: bool, seed: int = 0,
                    sigma_gain: float = 60.0, sigma_bias: float = 0.5, bkgd_use_space_time: bool = False,
                    bkgd_use_deform_time: bool = False, same_spacenet: bool = False,
                    deep_rgb: bool = False, include_input: bool = True, use_dir: bool = True) -> Dict[str, torch.Tensor]:
    """Full LayeredRFRender state_dict (key names of modeling/layered_rfrender.py:59-93).  With
    ``same_spacenet`` the fine performer nets ARE the coarse ones (:70-71): both key sets, same tensors."""
    rs = np.random.RandomState(seed)
    sd: Dict[str, torch.Tensor] = {}
    deep_rgb = deep_rgb and use_space_time                              # layered_rfrender.py:35
    kw = dict(deep_rgb=deep_rgb, include_input=include_input, use_dir=use_dir)
    sd.update(spacenet_state("bkgd_spacenet", rs, bkgd_use_space_time, sigma_gain, sigma_bias, **kw))
    sd.update(spacenet_state("bkgd_spacenet_fine", rs, bkgd_use_space_time, sigma_gain, sigma_bias, **kw))
    for i in range(layer_num):
        sd.update(spacenet_state(f"spacenets.{i}", rs, use_space_time, sigma_gain, sigma_bias, **kw))
        if same_spacenet:
            sd.update({k.replace(f"spacenets.{i}.", f"spacenets_fine.{i}."): v for k, v in sd.items()
                       if k.startswith(f"spacenets.{i}.")})
        else:
            sd.update(spacenet_state(f"spacenets_fine.{i}", rs, use_space_time, sigma_gain, sigma_bias, **kw))
    if use_deform_time:
        for i in range(layer_num):
            sd.update(motionnet_state(f"time_deform_nets.{i}", rs, include_input=include_input))
    if bkgd_use_deform_time:
        sd.update(motionnet_state("bkgd_time_deform_net", rs, include_input=include_input))
    return sd


def state_dict_for_flags(layer_num: int, use_space_time: bool, use_deform_time: bool, seed: int, flags=None):
    """make_state_dict with the optional cfg.MODEL flags of a test case ({"DEEP_RGB": True, ...})."""
    f = flags or {}
    return make_state_dict(layer_num, use_space_time, use_deform_time, seed,
                           bkgd_use_space_time=f.get("BKGD_USE_SPACE_TIME", False),
                           bkgd_use_deform_time=f.get("BKGD_USE_DEFORM_TIME", False),
                           same_spacenet=f.get("SAME_SPACENET", False), deep_rgb=f.get("DEEP_RGB", False),
                           include_input=f.get("TKERNEL_INC_RAW", True), use_dir=f.get("USE_DIR", True))


def camera(h: int, w: int, orbit_deg: float = 0.0, dist: float = 4.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """K = [[f,0,w/2],[0,f,h/2],[0,0,1]] with f = w; T = camera-to-world looking at the origin from
    distance ``dist`` on a circle in the x-z plane (orbit 0 = at (0,0,-dist) looking down +z)."""
    f = float(w)
    K = torch.tensor([[f, 0.0, w / 2.0], [0.0, f, h / 2.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
    a = np.deg2rad(orbit_deg)
    c, s = float(np.cos(a)), float(np.sin(a))
    T = torch.tensor([[c, 0.0, s, -dist * s],
                      [0.0, 1.0, 0.0, 0.0],
                      [-s, 0.0, c, -dist * c],
                      [0.0, 0.0, 0.0, 1.0]], dtype=torch.float32)
    return K, T


def scene_boxes(layer_num: int, frames: int = 3) -> Tuple[torch.Tensor, torch.Tensor]:
    """Background box (1,8,3) = AABB(-3,3)^3; performer boxes (frames, L, 8, 3): disjoint slabs of
    width 2.4/L tiling x in [-1.2,1.2], y,z in [-1,1]; frame f shifts every performer by 0.05 f in x."""
    bk = aabb_corners((-3.0, -3.0, -3.0), (3.0, 3.0, 3.0)).reshape(1, 8, 3)
    wdt = 2.4 / max(layer_num, 1)
    per = torch.zeros(frames, layer_num, 8, 3)
    for f in range(frames):
        for i in range(layer_num):
            x0 = -1.2 + i * wdt + 0.05 * f
            per[f, i] = aabb_corners((x0, -1.0, -1.0), (x0 + wdt * 0.9, 1.0, 1.0))
    return bk, per


def frame_id_columns(n: int, layer_num: int, frame: float = 2.5, bkgd_frame: float = 1.0) -> torch.Tensor:
    """Per-layer frame-id columns (data/datasets/ray_dataset.py:276-281); layer 0 gets frame 1."""
    cols = torch.full((n, layer_num + 1), float(frame), dtype=torch.float32)
    cols[:, 0] = float(bkgd_frame)
    return cols


def digest_positions(name: str, numel: int, k: int) -> torch.Tensor:
    """k distinct flat positions of a tensor of ``numel`` entries, a fixed function of (name, numel): the gradient fixtures of
    tests/golden/make_golden.py keep a parameter's gradient at these positions (whole networks are too large to commit)."""
    import zlib
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff)
    return torch.from_numpy(np.sort(rs.choice(numel, size=min(k, numel), replace=False)).astype(np.int64))


def tensor_digest(name: str, g: torch.Tensor, k: int = 512) -> torch.Tensor:
    """One float32 vector standing for a (gradient / parameter) tensor in a fixture: the tensor itself when it has at most 4096
    entries, else [absmax, L2 norm, row sums, column sums, k entries at ``digest_positions(name, ...)``] (sums in fp64)."""
    g = g.detach().double().cpu()
    if g.numel() <= 4096:
        return g.reshape(-1).float()
    g2 = g.reshape(g.shape[0], -1)
    return torch.cat([g.abs().max().reshape(1), g.norm().reshape(1), g2.sum(1), g2.sum(0),
                      g.reshape(-1)[digest_positions(name, g.numel(), k)]]).float()
