"""``ray_sampling`` with the reference's signature (utils/ray_sampling.py:22-72), on the device."""
import torch

from stnerf_amd import ops


def ray_sampling(Ks, Ts, image_size, masks=None, mask_threshold=0.5, images=None, outlier_map=None, device="cuda"):
    """Rays of M views, (M*h*w, 6); utils/ray_sampling.py:22-72 without masks/images (training inputs)."""
    if masks is not None or images is not None or outlier_map is not None:
        raise NotImplementedError("mask / image / outlier sampling is training-data preparation (out of scope)")
    h, w = image_size
    return torch.cat([ops.generate_rays(Ks[m], Ts[m], h, w, device=device) for m in range(Ks.shape[0])], 0), None
