"""Host-side camera-path / retiming / edit-schedule logic of the user-facing renderer mirror against a
fixture produced by the reference's own LayeredNeuralRenderer methods (tests/golden/make_golden.py:g_path).
CPU only: no kernel is launched (rendering itself is covered by the GPU tests)."""
import types

import numpy as np
import torch
import pytest

from conftest import load_golden


class _FakeModel:
    """Only what the path logic touches on the model."""
    def __init__(self):
        self.scale = self.shift = None
        self.alpha, self.near = 1, 0
        self.hidden = set()

    def hide_layer(self, i):
        self.hidden.add(i)

    def show_layer(self, i):
        self.hidden.discard(i)


def _renderer(a, meta, **kw):
    from stnerf_amd.render import LayeredNeuralRenderer
    cfg = types.SimpleNamespace(DATASETS=types.SimpleNamespace(LAYER_NUM=meta["L"], FRAME_NUM=meta["frame_num"], FRAME_OFFSET=0),
                                INPUT=types.SimpleNamespace(SIZE_TEST=[96, 54]), OUTPUT_DIR="")
    return LayeredNeuralRenderer(cfg, model=_FakeModel(), gt_poses=a["gt_poses"], gt_Ks=list(a["gt_Ks"]), **kw)


def _pairs(r):
    return np.array([[list(p) for p in row] for row in r.layer_frame_pairs], dtype=np.float64)


def _stack(xs):
    return np.stack([np.asarray(x, dtype=np.float64) for x in xs], 0)


def test_smooth_path_around_with_smooth_time_and_edit_schedule():
    meta, a = load_golden("path")
    r = _renderer(a, meta, s_shift=[[[0.0, 0.0, 0.0], [0.1, 0.0, 0.0], [0.0, 0.1, 0.0]],
                                    [[0.0, 0.0, 0.0], [0.3, 0.0, 0.1], [0.0, -0.1, 0.0]]],
                  s_scale=[[1.0, 1.0, 1.0], [1.0, 1.5, 0.5]], s_alpha=[1.0, 0.2])
    assert r.shift == r.s_shift[0] and r.scale == r.s_scale[0] and r.alpha == 1.0 and r.model.scale == r.s_scale[0]
    r.set_smooth_path_poses(9, around=True, smooth_time=True)
    np.testing.assert_allclose(_stack(r.poses), a["around_smooth_poses"].numpy(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(_stack(r.Ks), a["around_smooth_Ks"].numpy(), rtol=0, atol=1e-5)
    np.testing.assert_array_equal(_pairs(r), a["around_smooth_pairs"].numpy())
    np.testing.assert_allclose(np.array(r.s_shift_frame), a["around_smooth_shift"].numpy(), atol=1e-15)
    np.testing.assert_allclose(np.array(r.s_scale_frame), a["around_smooth_scale"].numpy(), atol=1e-15)
    np.testing.assert_allclose(np.array(r.s_alpha_frame), a["around_smooth_alpha"].numpy(), atol=1e-15)
    assert len(r.layer_frame_pairs) == len(r.poses) + 1            # the reference's off-by-one is kept


def test_end_to_end_rotation_path_and_key_frame_retiming():
    meta, a = load_golden("path")
    r = _renderer(a, meta)
    r.set_smooth_path_poses(7, around=False, smooth_time=False)
    np.testing.assert_allclose(_stack(r.poses), a["ends_int_poses"].numpy(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(_stack(r.Ks), a["ends_int_Ks"].numpy(), rtol=0, atol=1e-5)
    np.testing.assert_array_equal(_pairs(r), a["ends_int_pairs"].numpy())
    r.retime_by_key_frames(1, [5, 18, 20], [7, 11, 16])
    r.retime_by_key_frames(2, [3], [12])
    np.testing.assert_array_equal(_pairs(r), a["ends_int_retimed_pairs"].numpy())


def test_gt_and_fixed_camera_paths_with_a_hidden_layer():
    meta, a = load_golden("path")
    r = _renderer(a, meta)
    r.hide_layer(1)
    assert 1 in r.model.hidden and not r.is_shown_layer(1)
    r.set_path_gt_poses()
    r.set_path_fixed_gt_poses(2, num=4)
    np.testing.assert_allclose(_stack(r.poses), a["gt_fixed_poses"].numpy(), atol=1e-7)
    np.testing.assert_allclose(_stack(r.Ks), a["gt_fixed_Ks"].numpy(), atol=1e-5)
    flat = np.array([v for row in r.layer_frame_pairs for p in row for v in p], dtype=np.float64)
    np.testing.assert_array_equal(flat, a["gt_fixed_pairs_flat"].numpy())
    np.testing.assert_array_equal(np.array([len(row) for row in r.layer_frame_pairs]), a["gt_fixed_pairs_len"].numpy())


def test_constructor_needs_explicit_scene():
    from stnerf_amd.render import LayeredNeuralRenderer
    with pytest.raises(NotImplementedError, match="pass model="):
        LayeredNeuralRenderer(types.SimpleNamespace())


def test_metrics_mirror():
    """utils/metrics.py:4-17 (mse / mae / psnr with the optional mask and reduction)."""
    from stnerf_amd.utils import mae, mse, psnr
    torch.manual_seed(0)
    a, b = torch.rand(3, 8, 8), torch.rand(3, 8, 8)
    m = torch.rand(3, 8, 8) > 0.5
    assert torch.allclose(mse(a, b), ((a - b) ** 2).mean())
    assert torch.allclose(mse(a, b, m), ((a - b) ** 2)[m].mean())
    assert mse(a, b, reduction="none").shape == a.shape
    assert torch.allclose(mae(a, b), (a - b).abs().mean())
    assert torch.allclose(psnr(a, b, m), -10 * torch.log10(((a - b) ** 2)[m].mean()))
