"""``gen_weight`` / ``VolumeRenderer`` with the reference's signatures (layers/render_layer.py:8-58), computed by
the wave-scan compositor (csrc/render.hip: ``stnerf_gen_weight`` / ``stnerf_composite``)."""
from __future__ import annotations

import torch
from torch import nn

from stnerf_amd import ops


def gen_weight(sigma, delta, act_fn=None):
    """layers/render_layer.py:8-17 (act_fn is relu, as everywhere in the reference)."""
    if act_fn is not None and act_fn is not torch.nn.functional.relu:
        raise NotImplementedError("gen_weight: only the relu activation of the reference is implemented")
    return ops.gen_weight(sigma.squeeze(-1) if sigma.dim() == delta.dim() + 1 else sigma, delta)


class VolumeRenderer(nn.Module):
    """layers/render_layer.py:19-58."""

    def __init__(self, use_mask=False, boarder_weight=1e10):
        super().__init__()
        if use_mask:
            raise NotImplementedError("use_mask is False everywhere in the reference")
        self.boarder_weight, self.use_mask = boarder_weight, use_mask

    def forward(self, depth, rgb, sigma, noise=0):
        if noise > 0.:
            raise NotImplementedError("density noise is a training-time feature")
        n, s = depth.shape[0], depth.shape[1]
        raw = torch.cat([rgb, sigma], -1).reshape(n, 1, s, 4).contiguous()
        lo, _, w, _ = ops.composite(depth.reshape(n, 1, s).contiguous(), raw, None, border=self.boarder_weight,
                                    want_weights=True)
        return lo[:, 0, 0:3], lo[:, 0, 3:4], lo[:, 0, 4:5], w[:, 0].unsqueeze(-1)
