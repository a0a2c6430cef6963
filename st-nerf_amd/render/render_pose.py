"""Per-pose rendering with the reference's post-processing.

Mirrors ``LayeredNeuralRenderer.render_pose`` (render/layered_neural_renderer.py:364-391) without the
dataset object: rays are generated on the device from (K, pose) -- no CPU ray tensor, no 75 MB H2D per
frame -- and the images stay on the device until the caller moves them.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import torch

from stnerf_amd.parallel import render_view


def render_pose(model, pose, K, height: int, width: int, layer_frame_pair: Sequence[Tuple[int, float]], far: float,
                density_threshold: float = 0, bkgd_density_threshold: float = 0, device="cuda"):
    """-> color (H,W,3), depth (H,W,1), color_layer [l x (H,W,3)], depth_layer [l x (H,W,1)].

    ``layer_frame_pair``: (layer_id, frame_id) pairs as in data/datasets/ray_dataset.py:276-281.
    Depth post-processing as in the reference: negative mixed depth -> 0, then / far (:382-383); the
    per-layer depths are divided by far; their ``depth_1[depth < 0] = 0`` (:388) tests the already
    clamped mixed depth and therefore never fires -- reproduced as a no-op.

    One process per GPU under an initialised torch.distributed group: every rank generates and renders its interleaved
    row stripes of the view only (``model.shard_views = True``: opt-in), one all-gather rebuilds the images this function
    returns on every rank -- gather mode "fine": 6 + 5 l floats per ray instead of the whole 5-tuple's 11 + 10 l
    (stnerf_amd.parallel.render_view); the return value is the single-GPU one, bit for bit."""
    L = model.layer_num
    frame_ids = [0.0] * (L + 1)
    for layer_id, frame_id in layer_frame_pair:
        frame_ids[layer_id] = float(frame_id)
    stage2, _, stage2_layer, _, _ = render_view(model, torch.as_tensor(K, dtype=torch.float32),
                                                torch.as_tensor(pose, dtype=torch.float32), height, width, frame_ids,
                                                density_threshold, bkgd_density_threshold, device=device, gather="fine")
    color = stage2[0].reshape(height, width, 3)
    depth = stage2[1].reshape(height, width, 1).clamp_min(0) / far
    color_layer = [t[0].reshape(height, width, 3) for t in stage2_layer]
    depth_layer = [t[1].reshape(height, width, 1) / far for t in stage2_layer]
    return color, depth, color_layer, depth_layer


def to_uint8(image: torch.Tensor) -> torch.Tensor:
    """[0,1] float image -> uint8 on the device (one small D2H afterwards instead of fp32 planes)."""
    return (image.clamp(0, 1) * 255.0 + 0.5).to(torch.uint8)
