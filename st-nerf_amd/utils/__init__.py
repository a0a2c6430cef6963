"""The reference's ``utils`` package for the render path (utils/__init__.py:7-13): same module and symbol names."""
from .dimension_kernel import Trigonometric_kernel  # noqa: F401
from .ray_sampling import ray_sampling  # noqa: F401
from .batchify_rays import layered_batchify_ray  # noqa: F401
from .sample_pdf import sample_pdf  # noqa: F401
from .render_helpers import generate_rays  # noqa: F401
from .metrics import mae, mse, psnr, ssim  # noqa: F401
