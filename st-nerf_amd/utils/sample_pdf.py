"""Inverse-CDF importance resampling with the reference's signature (utils/sample_pdf.py:18-63), computed by the
wave-scan resampler (csrc/render.hip: ``stnerf_resample``)."""
import torch

from stnerf_amd import ops


def sample_pdf(z_vals, weights, N_samples, det=False, pytest=False, u=None, seed=0):
    """utils/sample_pdf.py:18-63: z_vals (n,N1), weights (n,N1-2) -> new samples (n,N_samples).
    ``u`` (n,N_samples) replays uniform draws; default is the device Philox stream."""
    if det or pytest:
        n = z_vals.shape[0]
        u = torch.linspace(0., 1., steps=N_samples, device=z_vals.device).expand(n, N_samples).contiguous()
    n, n1 = z_vals.shape
    pad = torch.zeros(n, 1, device=z_vals.device)
    wfull = torch.cat([pad, weights, pad], -1).reshape(n, 1, n1).contiguous()
    rays = torch.zeros(n, 6, device=z_vals.device)
    out = ops.resample(z_vals.reshape(n, 1, n1).contiguous(), wfull, N_samples, rays,
                       u=None if u is None else u.reshape(1, n, N_samples).contiguous(), seed=seed,
                       want_xyz=False, debug=True)
    return out[2][:, 0]
