#!/usr/bin/env python3
"""Per-parameter agreement of a HIP training step with the reference fixture (tests/golden/train_*.npz): prints every gradient's
error relative to the tensor's largest entry, worst first.  `python tools/debug_training_step.py [train_c3]` on the GPU box."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import test_gpu_training as T                                   # noqa: E402
from stnerf_amd import synthetic as syn                         # noqa: E402
from train_step_common import compare_digest                    # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "train_c3"
model, out, loss, parts, z, meta = T.training_step(name)
print(name, "loss", float(loss), "reference", float(z["loss"][0]), {k: float(v) for k, v in parts.items()})
for key in ("coarse_mixed_color", "fine_mixed_color", "coarse_layer1_acc", "fine_layer1_acc"):
    trip = {"coarse_mixed": out[1], "fine_mixed": out[0]}.get(key.rsplit("_", 1)[0])
    if trip is None:
        lst = out[3] if key.startswith("coarse") else out[2]
        trip = lst[int(key.split("layer")[1][0])]
    got = trip[{"color": 0, "depth": 1, "acc": 2}[key.rsplit("_", 1)[1]]].detach().cpu()
    print(f"  {key}: max |d| {float((got - torch.from_numpy(z[key])).abs().max()):.3e}")
named = dict(model.named_parameters())
rows = []
for k in z.files:
    if k.startswith("grad|"):
        p = k.split("|", 1)[1]
        rows.append((compare_digest(p, syn.tensor_digest(p, named[p].grad, meta["grad_samples"]), z[k], rel=1.0), p))
for r, p in sorted(rows, reverse=True):
    print(f"  {r:.3e}  {p}")
