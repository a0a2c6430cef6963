"""Inverse-CDF importance resampling with the reference's signature (utils/sample_pdf.py:18-63), computed by the
wave-scan resampler (csrc/render.hip: ``stnerf_resample``)."""
import itertools

import numpy as np
import torch

from stnerf_amd import ops

_calls = itertools.count()   # seed of the device Philox stream when the caller names none: fresh draws per call (:31)


def sample_pdf(z_vals, weights, N_samples, det=False, pytest=False, u=None, seed=None):
    """utils/sample_pdf.py:18-63: z_vals (n,N1), weights (n,N1-2) -> new samples (n,N_samples).
    ``u`` (n,N_samples) replays uniform draws.  Otherwise, as the reference: ``det`` -> u = linspace(0, 1, N_samples) (:28-29);
    ``pytest`` -> numpy's seed-0 stream, np.random.seed(0) + np.random.rand(n, N_samples) (:33-42; with ``det`` the
    same linspace, broadcast); else fresh uniform draws -- the device Philox stream, keyed by ``seed`` if given and by
    a per-process call counter if not."""
    n = z_vals.shape[0]
    if pytest:
        np.random.seed(0)                       # (:36: the reference reseeds numpy's GLOBAL stream; so does this)
        if det:
            un = np.broadcast_to(np.linspace(0., 1., N_samples), (n, N_samples))
        else:
            un = np.random.rand(n, N_samples)
        u = torch.as_tensor(np.ascontiguousarray(un), dtype=torch.float32, device=z_vals.device)   # torch.Tensor(u) (:41)
    elif det:
        u = torch.linspace(0., 1., steps=N_samples, device=z_vals.device).expand(n, N_samples).contiguous()
    if seed is None:
        # only a call that really draws advances the per-process counter: det / pytest / an explicit u must not shift the
        # random stream of later calls
        seed = next(_calls) if u is None else 0
    n, n1 = z_vals.shape
    pad = torch.zeros(n, 1, device=z_vals.device)
    wfull = torch.cat([pad, weights, pad], -1).reshape(n, 1, n1).contiguous()
    rays = torch.zeros(n, 6, device=z_vals.device)
    out = ops.resample(z_vals.reshape(n, 1, n1).contiguous(), wfull, N_samples, rays,
                       u=None if u is None else u.reshape(1, n, N_samples).contiguous(), seed=seed,
                       want_xyz=False, debug=True)
    return out[2][:, 0]
