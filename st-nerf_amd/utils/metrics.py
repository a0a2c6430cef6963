"""Image metrics with the reference's signatures (utils/metrics.py:4-24)."""
import torch
import torch.nn.functional as F


def mse(image_pred, image_gt, valid_mask=None, reduction="mean"):
    """utils/metrics.py:4-10."""
    value = (image_pred - image_gt) ** 2
    if valid_mask is not None:
        value = value[valid_mask]
    return torch.mean(value) if reduction == "mean" else value


def mae(image_pred, image_gt):
    """utils/metrics.py:12-14."""
    return torch.mean(torch.abs(image_pred - image_gt))


def psnr(image_pred, image_gt, valid_mask=None, reduction="mean"):
    """utils/metrics.py:16-17."""
    return -10 * torch.log10(mse(image_pred, image_gt, valid_mask, reduction))


def _dssim(img1, img2, window_size=3, reduction="mean", max_val=1.0):
    """``kornia.losses.ssim`` (the dependency utils/metrics.py:2 imports; kornia is neither vendored nor pinned by the
    reference -- README.md:30 ``pip install kornia`` -- so its published 0.4-era algorithm is restated here):
    Gaussian window (sigma 1.5, normalised) filtered per channel with reflect borders, C1 = (0.01 max)^2,
    C2 = (0.03 max)^2, loss = clamp(1 - ssim_map, 0, 1) / 2."""
    x = torch.arange(window_size, dtype=img1.dtype, device=img1.device) - window_size // 2
    if window_size % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2) / (2 * 1.5 ** 2))
    g = g / g.sum()
    kernel = torch.outer(g, g)
    c = img1.shape[1]
    weight = kernel.expand(c, 1, window_size, window_size).contiguous()
    pad = window_size // 2

    def blur(t):
        return F.conv2d(F.pad(t, (pad, pad, pad, pad), mode="reflect"), weight, groups=c)

    c1, c2 = (0.01 * max_val) ** 2, (0.03 * max_val) ** 2
    mu1, mu2 = blur(img1), blur(img2)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1, s2, s12 = blur(img1 * img1) - mu1_sq, blur(img2 * img2) - mu2_sq, blur(img1 * img2) - mu12
    ssim_map = ((2.0 * mu12 + c1) * (2.0 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))
    loss = torch.clamp(1.0 - ssim_map, min=0, max=1) / 2.0
    if reduction == "mean":
        return torch.mean(loss)
    if reduction == "sum":
        return torch.sum(loss)
    return loss


def ssim(image_pred, image_gt, reduction="mean"):
    """utils/metrics.py:19-24: image_pred, image_gt (3,H,W) -> structural similarity in [-1, 1].

    PARITY UNPINNED: the reference delegates to ``kornia.losses.ssim`` (an unpinned, un-vendored dependency that is absent from the
    build container), so there is no reference output to compare with -- ``_dssim`` restates kornia's published window-3 algorithm and
    is checked against its own definition only.  ``mse`` / ``mae`` / ``psnr`` are pinned to the reference's own functions
    (tests/test_host_cpu.py)."""
    return 1 - 2 * _dssim(image_pred.unsqueeze(0), image_gt.unsqueeze(0), 3, reduction)
