"""Ray generation with the reference's signature (utils/render_helpers.py:42-128), on the device
(``stnerf_generate_rays``, csrc/sampler.hip)."""
import torch

from stnerf_amd import ops


def generate_rays(K, T, bbox, h, w, device="cuda"):
    """-> (rays (h*w, 6) on the device, ray_mask (h,w,1)).  Only the full-view call (bbox=None) is on the
    render path (data/datasets/ray_dataset.py:263)."""
    if bbox is not None:
        raise NotImplementedError("bbox-cropped ray generation is a training-data helper (out of scope)")
    return ops.generate_rays(torch.as_tensor(K, dtype=torch.float32), torch.as_tensor(T, dtype=torch.float32), h, w,
                             device=device), torch.ones(h, w, 1)
