"""Ray generation with the reference's signatures (utils/render_helpers.py:42, utils/ray_sampling.py:22)."""
import torch

from stnerf_amd import ops


def generate_rays(K, T, bbox, h, w, device="cuda"):
    """-> (rays (h*w, 6) on the device, ray_mask (h,w,1)).  Only the full-view call (bbox=None) is on the
    render path (data/datasets/ray_dataset.py:263)."""
    if bbox is not None:
        raise NotImplementedError("bbox-cropped ray generation is a training-data helper (out of scope)")
    return ops.generate_rays(torch.as_tensor(K, dtype=torch.float32), torch.as_tensor(T, dtype=torch.float32), h, w,
                             device=device), torch.ones(h, w, 1)


def ray_sampling(Ks, Ts, image_size, masks=None, mask_threshold=0.5, images=None, outlier_map=None, device="cuda"):
    """Rays of M views, (M*h*w, 6); utils/ray_sampling.py:22-72 without masks/images (training inputs)."""
    if masks is not None or images is not None or outlier_map is not None:
        raise NotImplementedError("mask / image / outlier sampling is training-data preparation (out of scope)")
    h, w = image_size
    return torch.cat([ops.generate_rays(Ks[m], Ts[m], h, w, device=device) for m in range(Ks.shape[0])], 0), None
