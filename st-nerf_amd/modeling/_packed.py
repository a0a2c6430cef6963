"""Lazy repacking of ``nn.Linear`` parameters into the kernel layout (shared by SpaceNet and MotionNet)."""
from __future__ import annotations

from torch import nn


def _params_fingerprint(module: nn.Module):
    return tuple((p.data_ptr(), p._version, str(p.device)) for p in module.parameters())


class _PackedMixin:
    """Lazily (re)packs a module's nn.Linear weights into the kernel layout."""

    precision = "bf16x3"   # the library default: split-bf16 MFMA (fp32 operands as three bf16 pieces: fp32's significand and
                           # range, closer to an fp64 evaluation than an fp32 fma chain); "fp32": exact f32 MFMA (ops.PRECISIONS)

    def _packed(self, precision=None):
        """The kernel-layout blob for `precision` (default: the module's own).  One blob per precision is kept as long
        as the parameters do not change (the op-level MotionNet call is exact f32 whatever the module's arithmetic, and A/B
        runs switch back and forth: neither repacks on every call)."""
        precision = precision or self.precision
        fp = _params_fingerprint(self)
        cache = getattr(self, "_pack_cache", None)
        if cache is None or cache[0] != fp:
            cache = self._pack_cache = (fp, {})
        if precision not in cache[1]:
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("this network lives on %s: call .cuda() first -- the render path runs on the "
                                   "MI355X only (no CPU fallback)" % dev)
            sd = {k: v for k, v in self.state_dict().items()}
            keep, self.precision = self.precision, precision
            try:
                cache[1][precision] = self._pack(sd, dev)
            finally:
                self.precision = keep
        return cache[1][precision]
