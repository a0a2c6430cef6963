"""Lazy repacking of ``nn.Linear`` parameters into the kernel layout (shared by SpaceNet and MotionNet)."""
from __future__ import annotations

from torch import nn


def _params_fingerprint(module: nn.Module):
    return tuple((p.data_ptr(), p._version, str(p.device)) for p in module.parameters())


class _PackedMixin:
    """Lazily (re)packs a module's nn.Linear weights into the kernel layout."""

    precision = "fp32"   # "fp32": exact f32 MFMA;  "fp16x3": fp32-accurate split-fp16 MFMA (ops.PRECISIONS)

    def _packed(self):
        fp = _params_fingerprint(self) + (self.precision,)
        if getattr(self, "_pack_fp", None) != fp:
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("this network lives on %s: call .cuda() first -- the render path runs on the "
                                   "MI355X only (no CPU fallback)" % dev)
            sd = {k: v for k, v in self.state_dict().items()}
            self._pack_net = self._pack(sd, dev)
            self._pack_fp = fp
        return self._pack_net
