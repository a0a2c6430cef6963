"""Drop the MI355X render path into an unmodified checkout of the reference (DarlingHang/st-nerf).

The reference has no plugin interface: its callers bind the render-path symbols by name at import time
(``from modeling import build_layered_model``, ``from utils import layered_batchify_ray, add_two_dim_dict``,
render/layered_neural_renderer.py:7-8; ``from layers.RaySamplePoint import RaySamplePoint``,
demo/taekwondo_demo.py:21).  ``patch_reference`` therefore imports the reference's OWN ``modeling``, ``utils`` and
``layers`` packages -- so everything this framework does not replace (``add_two_dim_dict``, ``make_loss``,
``utils.logger``, ``vis_density``, the datasets ...) stays exactly what it was -- and rebinds only the
render-path symbols to the HIP implementation, in those packages and in every already-imported reference module
that holds one of the originals.

**Rendering and training.**  The replacements (LayeredRFRender, SpaceNet, MotionNet, RaySamplePoint, sample_pdf,
Trigonometric_kernel, VolumeRenderer) run on the MI355X only -- CPU tensors are refused.  Under ``torch.no_grad()``
(render/layered_neural_renderer.py:377) or in ``eval()`` mode the fused inference pipeline runs; in ``train()`` mode under autograd
-- what ``engine.layered_trainer.do_train`` sets up (:186-202) -- ``LayeredRFRender.forward`` runs the same stages with autograd
history and ``loss.backward()`` reaches every network parameter through hand-written HIP backward kernels
(stnerf_amd.modeling.training; tests/test_dropin.py runs the reference's own ``do_train`` on the patched model).  NEAR_FAR
sampling, USE_DEFORM_VIEW and POSE_REFINEMENT configurations are refused by the constructor (``patch.undo()`` restores the
reference's classes).  Models built after patching draw fresh jitter / resampling numbers on every forward
call, as the reference's torch.rand does (``model.fresh_draws_per_call``; set ``model.seed`` and switch it off for
reproducible frames).  Two lines at the top of a reference script::

    import stnerf_amd.dropin
    stnerf_amd.dropin.patch_reference("/path/to/st-nerf")

or no edit at all, from the reference's root directory::

    python -m stnerf_amd.dropin demo/taekwondo_demo.py -c configs/config_taekwondo.yml

**N GPUs, still no edit.**  Under ``python -m torch.distributed.run --nproc-per-node N -m stnerf_amd.dropin demo/...`` the
launcher picks cuda:LOCAL_RANK and initialises the process group (RCCL) before the script starts; from then on every
``layered_batchify_ray`` call of at least one chunk -- the reference's own ``render_pose`` makes one per frame,
render/layered_neural_renderer.py:378 -- deals its chunks out to the ranks in turn and all-gathers the whole 5-tuple
(stnerf_amd.parallel.render_rays_sharded), so every rank holds every image and the frame takes 1 / N of the time.  Ranks
other than 0 have ``imageio``'s writers silenced (if imageio is installed) so that each file is written once.
Sharding is opt-in -- this launcher opts in (``LayeredRFRender.SHARD_VIEWS_DEFAULT = True``: it runs the same script on every
rank, the contract of a collective call); a process group set up any other way (a DDP trainer whose ranks evaluate different
views, or rank 0 alone) leaves the call surface local unless ``model.shard_views = True`` / ``STNERF_SHARD=1`` says otherwise.
``STNERF_SHARD=0`` or ``model.shard_views = False`` turns the sharding off.  Every sharded call checks a fingerprint of its
rays across the ranks first and raises if they differ.

Exercised against the real reference tree by tests/test_dropin.py.
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import runpy
import sys
import types
from typing import Dict, List, Optional, Tuple

# reference module -> attribute -> (our module, our attribute).  Every row is a symbol of SURVEY.md section 8(b).
_REPLACEMENTS: Dict[str, Dict[str, Tuple[str, str]]] = {
    "modeling.layered_rfrender": {"LayeredRFRender": ("stnerf_amd.modeling.layered_rfrender", "LayeredRFRender")},
    "modeling.spacenet": {"SpaceNet": ("stnerf_amd.modeling.spacenet", "SpaceNet")},
    "modeling.motion_net": {"MotionNet": ("stnerf_amd.modeling.motion_net", "MotionNet")},
    "modeling": {"build_layered_model": ("stnerf_amd.modeling", "build_layered_model")},
    "utils.batchify_rays": {"layered_batchify_ray": ("stnerf_amd.utils.batchify_rays", "layered_batchify_ray")},
    "utils.sample_pdf": {"sample_pdf": ("stnerf_amd.utils.sample_pdf", "sample_pdf")},
    "utils.dimension_kernel": {"Trigonometric_kernel": ("stnerf_amd.utils.dimension_kernel", "Trigonometric_kernel")},
    "layers.RaySamplePoint": {"RaySamplePoint": ("stnerf_amd.layers.RaySamplePoint", "RaySamplePoint"),
                              "intersection": ("stnerf_amd.layers.RaySamplePoint", "intersection")},
    "layers.render_layer": {"VolumeRenderer": ("stnerf_amd.layers.render_layer", "VolumeRenderer"),
                            "gen_weight": ("stnerf_amd.layers.render_layer", "gen_weight")},
}
# Only with device_ray_generation=True: the reference's callers concatenate CPU frame-id columns to these rays
# (data/datasets/ray_dataset.py:276-281), so device rays need the dataset method replaced as well.
_RAYGEN_REPLACEMENTS = {
    "utils.render_helpers": {"generate_rays": ("stnerf_amd.utils.render_helpers", "generate_rays")},
    "utils.ray_sampling": {"ray_sampling": ("stnerf_amd.utils.ray_sampling", "ray_sampling")},
}

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))


class _Patch:
    """What one patch_reference() call changed: (module, attribute, original) triples, newest last."""

    def __init__(self):
        self.rebound: List[Tuple[types.ModuleType, str, object]] = []
        self.installed_modules: List[str] = []
        self.class_attrs: List[Tuple[type, str, object]] = []
        self.root: Optional[str] = None
        self.added_path = False
        self.originals: Dict[int, object] = {}      # id(replacement) -> the reference's object

    def undo(self):
        for mod, name, orig in reversed(self.rebound):
            setattr(mod, name, orig)
        # reference modules imported after the patch bound the replacements by name: give them the originals too
        for mod in list(sys.modules.values()):
            if mod is None or self.root is None or not _is_reference_module(mod, self.root):
                continue
            for name, val in list(vars(mod).items()):
                orig = self.originals.get(id(val))
                if orig is not None:
                    setattr(mod, name, orig)
        for cls, name, orig in reversed(self.class_attrs):
            setattr(cls, name, orig)
        for name in self.installed_modules:
            sys.modules.pop(name, None)
        if self.added_path and self.root in sys.path:
            sys.path.remove(self.root)
        self.rebound, self.class_attrs, self.installed_modules = [], [], []
        global _active
        if _active is self:
            _active = None


_active: Optional[_Patch] = None


def install_yacs_shim() -> bool:
    """``from yacs.config import CfgNode as CN`` (config/defaults.py:1) without yacs: register this package's
    ``CfgNode`` (same attribute-tree / merge_from_file / freeze surface) under that name.  No-op if yacs exists."""
    try:
        import yacs.config  # noqa: F401
        return False
    except ImportError:
        pass
    from stnerf_amd.config.defaults import CfgNode
    yacs = types.ModuleType("yacs")
    yacs_config = types.ModuleType("yacs.config")
    yacs_config.CfgNode = CfgNode
    yacs.config = yacs_config
    sys.modules["yacs"], sys.modules["yacs.config"] = yacs, yacs_config
    return True


def _is_reference_module(mod, root: str) -> bool:
    f = getattr(mod, "__file__", None)
    if not f:
        return False
    f = os.path.abspath(f)
    return f.startswith(root + os.sep) and not f.startswith(_PKG_DIR + os.sep)


def _device_rays_by_pose_and_K(self, T, K, layer_frame_pair):
    """Replacement for ``Ray_Dataset_Render.get_rays_by_pose_and_K`` (data/datasets/ray_dataset.py:260-283): the
    (H*W, 6 + l) ray tensor is generated on the device (no 75 MB H2D per 1080p frame); labels / bboxes / near_fars
    are unused by the BBOX path (:265) and come back as zero-stride views instead of N x 8 x 3 zeros."""
    import torch
    from stnerf_amd import ops
    frame_ids = None
    if self.use_deform_time or self.use_space_time:
        frame_ids = [0.0] * (self.layer_num + 1)
        for layer_id, frame_id in layer_frame_pair:
            frame_ids[layer_id] = float(frame_id)
    rays = ops.generate_rays(torch.as_tensor(K, dtype=torch.float32), torch.as_tensor(T, dtype=torch.float32),
                             self.height, self.width, frame_ids=frame_ids)
    n, dev = rays.shape[0], rays.device
    near_fars = self.near_far.to(dev).expand(n, 2)
    return rays, torch.zeros(1, device=dev).expand(n), torch.zeros(1, 8, 3, device=dev).expand(n, 8, 3), near_fars


def patch_reference(reference_root: Optional[str] = None, device_ray_generation: bool = False,
                    yacs_shim: bool = True) -> _Patch:
    """Rebind the render-path symbols of the reference checkout at ``reference_root`` (default: the directory the
    already-importable ``modeling`` package lives in, else the current directory) to the HIP implementation.
    Returns a handle whose ``undo()`` restores the originals.  Idempotent."""
    global _active
    if _active is not None:
        return _active
    patch = _Patch()
    if reference_root is None:
        spec = importlib.util.find_spec("modeling")
        origin = getattr(spec, "origin", None) if spec else None
        reference_root = os.path.dirname(os.path.dirname(origin)) if origin else os.getcwd()
    root = os.path.abspath(reference_root)
    if not os.path.isfile(os.path.join(root, "modeling", "layered_rfrender.py")):
        raise FileNotFoundError(f"{root} is not a checkout of the reference (modeling/layered_rfrender.py is missing)")
    patch.root = root
    if root not in [os.path.abspath(p) for p in sys.path]:
        sys.path.insert(0, root)
        patch.added_path = True
    if yacs_shim and install_yacs_shim():
        patch.installed_modules += ["yacs", "yacs.config"]

    table = dict(_REPLACEMENTS)
    if device_ray_generation:
        table.update(_RAYGEN_REPLACEMENTS)
    # originals by identity -> replacement
    swap: Dict[int, object] = {}
    for ref_name, attrs in table.items():
        ref_mod = importlib.import_module(ref_name)
        if not _is_reference_module(ref_mod, root):
            raise ImportError(f"`{ref_name}` resolves to {getattr(ref_mod, '__file__', None)}, not to the reference "
                              f"checkout at {root}: another package of that name is ahead of it on sys.path")
        for attr, (our_mod, our_attr) in attrs.items():
            orig = getattr(ref_mod, attr)
            new = getattr(importlib.import_module(our_mod), our_attr)
            swap[id(orig)] = new
            patch.originals[id(new)] = orig
    # rebind in every imported reference module that holds an original (the defining module, the package
    # __init__ that re-exports it, and callers that did `from utils import layered_batchify_ray`)
    for mod in list(sys.modules.values()):
        if mod is None or not _is_reference_module(mod, root):
            continue
        for name, val in list(vars(mod).items()):
            new = swap.get(id(val))
            if new is not None and val is not new:
                patch.rebound.append((mod, name, val))
                setattr(mod, name, new)
    # models the reference's code builds from here on draw fresh jitter / resampling numbers per forward call, as its
    # torch.rand does (layers/RaySamplePoint.py:98, utils/sample_pdf.py:31)
    from stnerf_amd.modeling.layered_rfrender import LayeredRFRender as _LRF
    patch.class_attrs.append((_LRF, "FRESH_DRAWS_DEFAULT", _LRF.FRESH_DRAWS_DEFAULT))
    _LRF.FRESH_DRAWS_DEFAULT = True
    if device_ray_generation:
        try:
            ray_dataset = importlib.import_module("data.datasets.ray_dataset")
            cls = ray_dataset.Ray_Dataset_Render
            patch.class_attrs.append((cls, "get_rays_by_pose_and_K", cls.get_rays_by_pose_and_K))
            cls.get_rays_by_pose_and_K = _device_rays_by_pose_and_K
        except ImportError:
            pass  # the dataset stack (torchvision, PIL ...) is not installed: nothing to patch
    _active = patch
    return patch


def unpatch_reference() -> None:
    if _active is not None:
        _active.undo()


def join_process_group() -> Tuple[int, int]:
    """Launched by torch.distributed.run (WORLD_SIZE > 1 in the environment): this process takes cuda:LOCAL_RANK and
    joins the group, which is all the render path needs to split every view over the ranks (stnerf_amd.parallel).  On
    ranks other than 0 the image / video writers of imageio (render/layered_neural_renderer.py:467-485,624-637) become
    no-ops, if imageio is installed.  -> (rank, world); (0, 1) and nothing done for a plain ``python`` launch."""
    from stnerf_amd import parallel
    rank, world = parallel.init_from_env(single_device=os.environ.get("STNERF_SINGLE_DEVICE", "0") == "1")
    if world > 1:
        # sharding is opt-in (a collective every rank must join with the same rays): this launcher runs the SAME render script
        # on every rank, which is exactly that contract, so models built from here on shard their views
        from stnerf_amd.modeling.layered_rfrender import LayeredRFRender
        LayeredRFRender.SHARD_VIEWS_DEFAULT = True
    if world > 1 and rank != 0:
        try:
            imageio = importlib.import_module("imageio")
            for name in ("imwrite", "imsave", "mimwrite", "mimsave"):
                if hasattr(imageio, name):
                    setattr(imageio, name, lambda *a, **k: None)
        except ImportError:
            pass
    return rank, world


def main(argv=None) -> None:
    """``python -m stnerf_amd.dropin [--reference ROOT] [--device-rays] script.py [script args]``"""
    argv = list(sys.argv[1:] if argv is None else argv)
    root, device_rays = None, False
    while argv and argv[0].startswith("--"):
        flag = argv.pop(0)
        if flag == "--reference":
            root = argv.pop(0)
        elif flag == "--device-rays":
            device_rays = True
        else:
            raise SystemExit(f"unknown option {flag}\n{main.__doc__}")
    if not argv:
        raise SystemExit(main.__doc__)
    patch_reference(root if root is not None else os.getcwd(), device_ray_generation=device_rays)
    join_process_group()
    sys.argv = argv
    runpy.run_path(argv[0], run_name="__main__")


if __name__ == "__main__":
    main()
