"""Mirror of the reference's ``utils`` package for the render path (utils/__init__.py:7-13)."""
from stnerf_amd.renderer import Trigonometric_kernel, layered_batchify_ray, mae, mse, psnr, sample_pdf  # noqa: F401
from stnerf_amd.raygen import generate_rays, ray_sampling  # noqa: F401
