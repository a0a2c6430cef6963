"""Slab intersection + stratified coarse sampling with the reference's signatures
(layers/RaySamplePoint.py:8-107), computed by csrc/sampler.hip."""
from __future__ import annotations

from torch import nn

from stnerf_amd import ops


def intersection(rays, bbox):
    """layers/RaySamplePoint.py:8-62: rays (n,>=6), bbox (n,8,3) -> (n,2) = (far, near)."""
    return ops.intersect(rays.contiguous(), bbox.unsqueeze(1).contiguous())[:, 0]


class RaySamplePoint(nn.Module):
    """layers/RaySamplePoint.py:64-107.  ``jitter`` (l,n,N) replays given uniform draws; otherwise the device Philox
    stream keyed by ``seed`` is used, and -- as the reference draws fresh torch.rand numbers per call (:98) -- the seed
    advances with every call unless ``deterministic`` is set."""

    def __init__(self, coarse_num=64):
        super().__init__()
        self.coarse_num = coarse_num
        self.seed = 0
        self.deterministic = False

    def forward(self, rays, bbox, pdf=None, method="coarse", jitter=None):
        t, xyz, mask = ops.sample_coarse(rays.contiguous(), bbox.contiguous(), self.coarse_num, jitter=jitter,
                                         seed=self.seed)
        if jitter is None and not self.deterministic:
            self.seed += 1
        l = t.shape[1]
        return ([t[:, i].unsqueeze(-1) for i in range(l)], [xyz[:, i] for i in range(l)],
                [mask[:, i].bool() for i in range(l)])
