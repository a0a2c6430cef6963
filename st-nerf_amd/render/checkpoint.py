"""Reference checkpoint discovery and loading (data/datasets/utils.py:42-60,
render/layered_neural_renderer.py:96-121)."""
import glob
import os

import torch


def get_iteration_path(root_dir, fix_iter=-1):
    """Path of the newest ``layered_rfnr_checkpoint_<iter>.pt`` under ``root_dir`` (None if there is none or the
    directory does not exist); ``fix_iter`` names ``<root>/frame/layered_rfnr_checkpoint_<fix_iter>.pt`` without
    looking.  data/datasets/utils.py:42-60 (only file names that split into exactly four ``_`` pieces count)."""
    if fix_iter != -1:
        return os.path.join(root_dir, "frame", "layered_rfnr_checkpoint_%d.pt" % fix_iter)
    if not os.path.exists(root_dir):
        return None
    newest = -1
    for name in glob.glob(os.path.join(root_dir, "layered_rfnr_checkpoint_*.pt")):
        pieces = name.split("/")[-1].split("_")
        if len(pieces) != 4:
            continue
        newest = max(newest, int(pieces[-1].split(".")[0]))
    path = os.path.join(root_dir, "layered_rfnr_checkpoint_%d.pt" % newest)
    return path if os.path.exists(path) else None


def load_reference_checkpoint(model, path, map_location="cuda"):
    """Load a reference ``layered_rfnr_checkpoint_N.pt`` ({'model': state_dict, ...}); keys the checkpoint
    lacks keep the model's current values, as render/layered_neural_renderer.py:110-117 does."""
    ckpt = torch.load(path, map_location=map_location)
    sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
    own = model.state_dict()
    for k, v in own.items():
        if k not in sd:
            sd[k] = v
    model.load_state_dict({k: v for k, v in sd.items() if k in own})
    return model
