"""Radiance MLP with the reference's constructor, attribute names and forward (modeling/spacenet.py:13-160).

The parameters are ordinary ``nn.Linear`` modules under the reference's attribute names, so
``state_dict()`` / ``load_state_dict()`` speak the reference's checkpoint keys; the forward pass is the
fused positional-encoding + 9-layer MFMA kernel behind ``stnerf_spacenet_fwd`` (csrc/mlp.hip).
"""
from __future__ import annotations

import torch
from torch import nn

from stnerf_amd import ops
from stnerf_amd.modeling._packed import _PackedMixin


class SpaceNet(nn.Module, _PackedMixin):
    """Radiance MLP, modeling/spacenet.py:13-160 (same constructor, attribute names and forward)."""

    def __init__(self, c_pos=3, include_input=True, use_dir=True, use_time=False, deep_rgb=False):
        super().__init__()
        if c_pos != 3:
            raise NotImplementedError("HIP SpaceNet supports c_pos=3")
        self.c_pos, self.use_dir, self.use_time, self.deep_rgb = c_pos, use_dir, use_time, deep_rgb
        self.include_input = include_input
        raw = int(include_input)             # encodings without the raw input lose d columns (dimension_kernel.py:12-14)
        self.pos_dim = 3 * (raw + 20)
        self.dir_dim = 3 * (raw + 8) if use_dir else 0
        self.time_dim = (raw + 20) if use_time else 0
        bd, hd = 256, 128
        self.stage1 = nn.Sequential(nn.Linear(self.pos_dim, bd), nn.ReLU(inplace=True), nn.Linear(bd, bd),
                                    nn.ReLU(inplace=True), nn.Linear(bd, bd), nn.ReLU(inplace=True),
                                    nn.Linear(bd, bd), nn.ReLU(inplace=True))
        self.stage2 = nn.Sequential(nn.Linear(bd + self.pos_dim, bd), nn.ReLU(inplace=True), nn.Linear(bd, bd),
                                    nn.ReLU(inplace=True), nn.Linear(bd, bd), nn.ReLU(inplace=True))
        self.density_net = nn.Sequential(nn.Linear(bd, 1))
        if deep_rgb:                                                        # modeling/spacenet.py:68-79
            self.rgb_net = nn.Sequential(nn.ReLU(inplace=True), nn.Linear(bd + self.dir_dim + self.time_dim, hd),
                                         nn.ReLU(inplace=True), nn.Linear(hd, hd), nn.ReLU(inplace=True),
                                         nn.Linear(hd, hd), nn.ReLU(inplace=True), nn.Linear(hd, 3))
        else:
            self.rgb_net = nn.Sequential(nn.ReLU(inplace=True), nn.Linear(bd + self.dir_dim + self.time_dim, hd),
                                         nn.ReLU(inplace=True), nn.Linear(hd, 3))

    def training_parameters(self):
        """weight, bias of every nn.Linear in evaluation order (ops.SPACENET_KEYS [+ rgb_net.5, rgb_net.7 with deep_rgb])."""
        named = dict(self.named_parameters())
        keys = ops.SPACENET_KEYS + (["rgb_net.5", "rgb_net.7"] if self.deep_rgb else [])
        return [named[f"{k}.{what}"] for k in keys for what in ("weight", "bias")]

    def _pack(self, sd, dev):
        return ops.pack_spacenet({"net." + k: v for k, v in sd.items()}, "net", dev, self.precision)

    def forward(self, pos, rays, times=None, maxs=None, mins=None):
        """pos (N,L,3) or (N,3); rays (N,>=6); times (N,1) -> rgbs (N,L,3)|(N,3), density (N,L,1)|(N,1)."""
        if maxs is not None:
            raise NotImplementedError("maxs/mins normalisation is unused by the reference (always None)")
        bins = pos.dim() > 2
        x = pos if bins else pos.unsqueeze(1)
        n, s = x.shape[0], x.shape[1]
        x = x.contiguous()
        raw = torch.empty(n, s, 4, dtype=torch.float32, device=x.device)
        tm = times.reshape(n).contiguous() if (self.use_time and times is not None) else None
        if torch.is_grad_enabled() and (x.requires_grad or (self.training and any(p.requires_grad for p in self.parameters()))):
            # training (model.train(), or an input that asks for its gradient): fused exact-f32 forward,
            # recompute-and-walk-back backward on the f32 MFMA GEMMs of csrc/train.hip.  A plain call in eval() mode stays on the
            # inference path below even when autograd happens to be enabled: nothing is saved, the result carries no history
            from stnerf_amd.modeling.autograd import SpaceNetFunction
            if self.use_time and tm is None:
                raise ValueError("this SpaceNet takes time: pass times")
            rgb, sig = SpaceNetFunction.apply(self, x, rays[:, 3:6], tm, *self.training_parameters())
            return (rgb, sig) if bins else (rgb[:, 0], sig[:, 0])
        ops.spacenet_fwd(self._packed(), x, rays[:, 3:6], tm, raw)
        rgb, sig = raw[..., :3], raw[..., 3:]
        return (rgb, sig) if bins else (rgb[:, 0], sig[:, 0])
