"""Deformation MLP with the reference's constructor and forward (modeling/motion_net.py:5-71); the forward pass
is ``stnerf_motionnet_fwd`` (csrc/mlp.hip), including the fractional-time lerp of the encodings (:49-60)."""
from __future__ import annotations

import torch
from torch import nn

from stnerf_amd import ops
from stnerf_amd.modeling._packed import _PackedMixin


class MotionNet(nn.Module, _PackedMixin):
    """Deformation MLP, modeling/motion_net.py:5-71."""

    def __init__(self, c_input=5, include_input=True, input_time=False):
        super().__init__()
        if c_input != 4:
            raise NotImplementedError("HIP MotionNet supports c_input=4 (the time-deformation nets of the layered "
                                      "model; input_time selects the fractional-time lerp)")
        self.c_input, self.input_time, self.pos_dim = c_input, input_time, 4 * (int(include_input) + 20)
        d = 128
        self.motion_net = nn.Sequential(nn.Linear(self.pos_dim, d), nn.ReLU(inplace=False), nn.Linear(d, d),
                                        nn.ReLU(inplace=True), nn.Linear(d, d), nn.ReLU(inplace=True),
                                        nn.Linear(d, d), nn.ReLU(inplace=True), nn.Linear(d, d),
                                        nn.ReLU(inplace=True), nn.Linear(d, 3))

    def _pack(self, sd, dev):
        return ops.pack_motionnet({"net." + k: v for k, v in sd.items()}, "net", dev, self.precision)

    def forward(self, input_0):
        """input_0 (N,L,4) or (N,4) = [x,y,z,t] -> flow (N,L,3) or (N,3).  The time may differ per sample."""
        bins = input_0.dim() > 2
        if torch.is_grad_enabled() and (input_0.requires_grad or (self.training and any(p.requires_grad for p in self.parameters()))):
            # training (model.train(), or an input that asks for its gradient): see stnerf_amd.modeling.autograd
            from stnerf_amd.modeling.autograd import MotionNetFunction
            named = dict(self.named_parameters())
            params = [named[f"{k}.{what}"] for k in ops.MOTIONNET_KEYS for what in ("weight", "bias")]
            flow = MotionNetFunction.apply(self, input_0.reshape(-1, 4).float(), *params)
            return flow.reshape(*input_0.shape[:-1], 3)
        x = input_0.reshape(-1, 1, 4)
        xyz = x[..., :3].contiguous()
        flow = torch.empty_like(xyz)
        # (bf16x3 has no stand-alone MotionNet launch -- it runs fused in front of a SpaceNet: the op-level call is exact f32)
        ops.motionnet_fwd(self._packed("fp32" if self.precision == "bf16x3" else None), xyz, x[:, 0, 3].contiguous(), flow=flow, add_to_xyz=False,
                          plain_time=not self.input_time)
        return flow.reshape(*input_0.shape[:-1], 3) if bins else flow.reshape(-1, 3)
