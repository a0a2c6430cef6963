import json
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need an MI355X and the built library: skip (not fail) them on a box without either, so that a plain
    `pytest tests` is green on a CPU machine.  On a GPU box nothing is skipped -- a missing library fails loudly."""
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a GPU (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    """-> (meta dict, {key: torch tensor})."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    arrs = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    return meta, arrs


@pytest.fixture(scope="session")
def golden():
    return load_golden
