"""Positional encoding with the reference's class surface (utils/dimension_kernel.py:54-73)."""
from stnerf_amd import ops


class Trigonometric_kernel:
    """Positional encoding, utils/dimension_kernel.py:54-73 (same constructor, __call__ and calc_dim)."""

    def __init__(self, L=10, input_dim=3, include_input=True):
        self.L, self.input_dim, self.include_input = L, input_dim, include_input
        self.out_ch = input_dim * (int(include_input) + 2 * L)

    def __call__(self, x):
        return ops.encode(x, self.L, self.include_input)

    def calc_dim(self, dims=0):
        return self.out_ch
