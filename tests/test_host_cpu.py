"""Host-side pieces of the drop-in that need no GPU: the yacs-free config against both shipped
ymls, checkpoint key names against the reference model's, checkpoint discovery, ssim, and the summation order the
resampler reproduces."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import stnerf_oracle as O
from stnerf_amd import synthetic as syn

REFERENCE = "/root/reference"
MODEL_KEYS_READ = ["BOARDER_WEIGHT", "SAMPLE_METHOD", "SAME_SPACENET", "TKERNEL_INC_RAW", "POSE_REFINEMENT", "USE_DIR",
                   "USE_DEFORM_VIEW", "USE_DEFORM_TIME", "USE_SPACE_TIME", "BKGD_USE_DEFORM_TIME", "BKGD_USE_SPACE_TIME",
                   "DEEP_RGB", "COARSE_RAY_SAMPLING", "FINE_RAY_SAMPLING"]      # modeling/layered_rfrender.py:23-37 (+ LAYER_NUM)

# hand-written from configs/config_taekwondo.yml and configs/config_walking.yml + config/defaults.py for absent keys
EXPECT = {
    "config_taekwondo.yml": dict(BOARDER_WEIGHT=1e10, SAMPLE_METHOD="BBOX", SAME_SPACENET=False, TKERNEL_INC_RAW=True,
                                 POSE_REFINEMENT=False, USE_DIR=True, USE_DEFORM_VIEW=False, USE_DEFORM_TIME=True,
                                 USE_SPACE_TIME=True, BKGD_USE_DEFORM_TIME=False, BKGD_USE_SPACE_TIME=False,
                                 DEEP_RGB=False, COARSE_RAY_SAMPLING=90, FINE_RAY_SAMPLING=30, LAYER_NUM=2,
                                 FRAME_NUM=101, FRAME_OFFSET=0, SIZE_TEST=[1920, 1080], OUTPUT_DIR="outputs/taekwondo"),
    "config_walking.yml": dict(BOARDER_WEIGHT=1e10, SAMPLE_METHOD="BBOX", SAME_SPACENET=False, TKERNEL_INC_RAW=True,
                               POSE_REFINEMENT=False, USE_DIR=True, USE_DEFORM_VIEW=False, USE_DEFORM_TIME=True,
                               USE_SPACE_TIME=False, BKGD_USE_DEFORM_TIME=False, BKGD_USE_SPACE_TIME=False,
                               DEEP_RGB=True,           # not in the yml: config/defaults.py:39 default (inert: USE_SPACE_TIME off)
                               COARSE_RAY_SAMPLING=90, FINE_RAY_SAMPLING=30, LAYER_NUM=2, FRAME_NUM=50, FRAME_OFFSET=25,
                               SIZE_TEST=[1920, 1080], OUTPUT_DIR="outputs/walking"),
}


def _check_cfg(cfg, want):
    for k in MODEL_KEYS_READ:
        got = getattr(cfg.MODEL, k)
        assert got == want[k] and type(got) is type(want[k]), (k, got, want[k])
    assert cfg.DATASETS.LAYER_NUM == want["LAYER_NUM"] and cfg.DATASETS.FRAME_NUM == want["FRAME_NUM"]
    assert cfg.DATASETS.FRAME_OFFSET == want["FRAME_OFFSET"] and list(cfg.INPUT.SIZE_TEST) == want["SIZE_TEST"]
    assert cfg.OUTPUT_DIR == want["OUTPUT_DIR"]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "configs")), reason="the shipped ymls live in the reference checkout")
@pytest.mark.parametrize("yml", sorted(EXPECT))
def test_shipped_ymls_load_with_the_values_the_model_reads(yml):
    from stnerf_amd.config import get_cfg_defaults
    cfg = get_cfg_defaults()
    cfg.merge_from_file(os.path.join(REFERENCE, "configs", yml))
    cfg.freeze()
    _check_cfg(cfg, EXPECT[yml])
    with pytest.raises(AttributeError):
        cfg.MODEL.USE_DIR = False                        # frozen, as yacs
    # the model builds from it (on the CPU: construction needs no GPU) with the reference's checkpoint keys
    from stnerf_amd.modeling import build_layered_model
    model = build_layered_model(cfg, camera_num=1)
    keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    tag = "taekwondo" if "taekwondo" in yml else "walking"
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == keys[tag]["state_dict"]


def test_config_from_yml_text_and_defaults(tmp_path):
    """The same keys from a yml written here (runs on the GPU box too): overrides, type coercion of '1e10', defaults."""
    from stnerf_amd.config import cfg as global_cfg, get_cfg_defaults
    p = tmp_path / "c.yml"
    p.write_text("MODEL:\n  COARSE_RAY_SAMPLING: 90\n  FINE_RAY_SAMPLING: 30\n  SAMPLE_METHOD: \"BBOX\"\n  BOARDER_WEIGHT: 1e10\n"
                 "  POSE_REFINEMENT: False\n  USE_DEFORM_TIME: True\n  USE_SPACE_TIME: True\n  DEEP_RGB: False\n"
                 "DATASETS:\n  LAYER_NUM: 2\n  FRAME_NUM: 101\nINPUT:\n  SIZE_TEST: [1920,1080]\nOUTPUT_DIR: \"outputs/taekwondo\"\n")
    cfg = get_cfg_defaults()
    cfg.merge_from_file(str(p))
    _check_cfg(cfg, EXPECT["config_taekwondo.yml"])
    assert isinstance(cfg.MODEL.BOARDER_WEIGHT, float)
    d = get_cfg_defaults()                                # config/defaults.py:17-153
    assert d.MODEL.SAMPLE_METHOD == "NEAR_FAR" and d.MODEL.POSE_REFINEMENT is True and d.MODEL.DEEP_RGB is True
    assert d.MODEL.COARSE_RAY_SAMPLING == 64 and d.MODEL.FINE_RAY_SAMPLING == 80 and d.DATASETS.LAYER_NUM == 0
    assert global_cfg.MODEL.COARSE_RAY_SAMPLING == 64    # clones do not touch the global node


@pytest.mark.parametrize("tag", ["taekwondo", "walking", "deep", "bkgd_time", "same"])
def test_state_dict_keys_and_shapes_are_the_reference_models(tag):
    """Checkpoint compatibility (render/layered_neural_renderer.py:110-117): for every model flavour the state_dict of
    this framework's LayeredRFRender has exactly the reference model's keys and shapes (state_dict_keys.json was
    written by make_golden.py from the reference's own build_layered_model)."""
    import test_gpu_render as R
    from stnerf_amd.modeling import build_layered_model
    spec = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))[tag]
    cfg = R.make_cfg(spec["L"], 8, 4, spec["space_time"], spec["deform_time"], spec["flags"])
    model = build_layered_model(cfg, camera_num=1)
    got = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert got == spec["state_dict"]
    # and the synthetic weights used everywhere load strictly
    model.load_state_dict(syn.state_dict_for_flags(spec["L"], spec["space_time"], spec["deform_time"], 1, spec["flags"]))


def test_get_iteration_path(tmp_path):
    from stnerf_amd.data import get_iteration_path
    assert get_iteration_path(str(tmp_path / "missing")) is None          # data/datasets/utils.py:46-47
    assert get_iteration_path(str(tmp_path)) is None                      # no checkpoint: checkpoint_-1 does not exist
    for it in (3000, 45000, 6000):
        (tmp_path / f"layered_rfnr_checkpoint_{it}.pt").write_bytes(b"")
    (tmp_path / "layered_rfnr_checkpoint_99999_old.pt").write_bytes(b"")  # five '_' pieces: skipped (:52-53)
    (tmp_path / "rfnr_checkpoint_70000.pt").write_bytes(b"")              # another prefix: not globbed
    assert get_iteration_path(str(tmp_path)) == os.path.join(str(tmp_path), "layered_rfnr_checkpoint_45000.pt")
    assert get_iteration_path(str(tmp_path), 7) == os.path.join(str(tmp_path), "frame", "layered_rfnr_checkpoint_7.pt")


def test_load_reference_checkpoint_keeps_missing_keys(tmp_path):
    import test_gpu_render as R
    from stnerf_amd.modeling import build_layered_model
    from stnerf_amd.render import load_reference_checkpoint
    model = build_layered_model(R.make_cfg(1, 8, 4, True, True), camera_num=1)
    sd = syn.state_dict_for_flags(1, True, True, 3, {})
    gone = "time_deform_nets.0.motion_net.10.weight"
    before = model.state_dict()[gone].clone()
    torch.save({"model": {k: v for k, v in sd.items() if k != gone}, "optimizer": {}, "scheduler": {}},
               tmp_path / "layered_rfnr_checkpoint_1.pt")
    load_reference_checkpoint(model, str(tmp_path / "layered_rfnr_checkpoint_1.pt"), map_location="cpu")
    now = model.state_dict()
    assert torch.equal(now[gone], before)
    assert all(torch.equal(now[k], v) for k, v in sd.items() if k != gone)


def test_ssim_restatement():
    """utils/metrics.py:19-24 = 1 - 2 * kornia.losses.ssim(window 3): identical images -> 1; the map formula checked
    against a direct per-pixel evaluation of the published definition."""
    from stnerf_amd.utils import mae, mse, psnr, ssim
    torch.manual_seed(0)
    a, b = torch.rand(3, 12, 14), torch.rand(3, 12, 14)
    assert float(ssim(a, a)) == pytest.approx(1.0, abs=1e-6)
    s_ab = float(ssim(a, b))
    assert -1.0 <= s_ab < 0.5 and float(ssim(b, a)) == pytest.approx(s_ab, abs=1e-6)
    # direct evaluation at one interior pixel, channel 1
    g = torch.exp(-torch.tensor([-1.0, 0.0, 1.0]) ** 2 / (2 * 1.5 ** 2))
    g = g / g.sum()
    k = torch.outer(g, g)
    pa, pb = a[1, 4:7, 5:8], b[1, 4:7, 5:8]
    mu1, mu2 = (k * pa).sum(), (k * pb).sum()
    s1, s2, s12 = (k * pa * pa).sum() - mu1 ** 2, (k * pb * pb).sum() - mu2 ** 2, (k * pa * pb).sum() - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    want = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 ** 2 + mu2 ** 2 + c1) * (s1 + s2 + c2))
    from stnerf_amd.utils.metrics import _dssim
    loss_map = _dssim(a.unsqueeze(0), b.unsqueeze(0), 3, "none")
    assert float(loss_map[0, 1, 5, 6]) == pytest.approx(float(torch.clamp(1 - want, 0, 1) / 2), abs=1e-6)
    assert float(psnr(a, b)) == pytest.approx(float(-10 * torch.log10(mse(a, b))), abs=1e-6) and float(mae(a, a)) == 0.0


@pytest.mark.parametrize("n", [1, 3, 6, 7, 8, 10, 14, 15, 16, 30, 62, 88, 126, 190, 254, 600, 1100])
def test_aten_sum_order(n):
    """torch.sum over the last dim of a contiguous fp32 (rows, n) tensor == the order the resampler reproduces on the
    device (csrc/render.hip: aten_cpu_row_sum; restated in numpy as oracle.aten_cpu_row_sum): 8-float vectors x 4
    interleaved accumulators, whatever the host's SIMD level.  If a torch upgrade changes ATen's reduction order this
    fails here, on the CPU, before any GPU parity test does."""
    torch.manual_seed(n)
    w = torch.rand(200, n) ** 6 + 1e-5
    want = torch.sum(w, -1, keepdim=True).numpy()[:, 0]
    got = np.array([O.aten_cpu_row_sum(r) for r in w.numpy()])
    assert np.array_equal(want, got), f"{int((want != got).sum())} of 200 rows differ at n={n}"


def test_aten_cumsum_accumulates_in_double():
    torch.manual_seed(1)
    p = torch.rand(300, 126) ** 8
    p = p / p.sum(-1, keepdim=True)
    assert torch.equal(torch.cumsum(p, -1), torch.cumsum(p.double(), -1).float())


def test_sample_pdf_host_flags_follow_the_reference(monkeypatch):
    """utils/sample_pdf.py:26-42: det -> linspace; pytest -> numpy's seed-0 stream (np.random.seed(0); np.random.rand(n, N)),
    or the linspace broadcast when det is set too; otherwise fresh draws per call (:31) -- here: the device stream under a
    seed that changes from call to call unless the caller names one.  Host logic only: the native call is intercepted."""
    from stnerf_amd import ops
    import importlib
    mod = importlib.import_module("stnerf_amd.utils.sample_pdf")
    seen = []

    def fake_resample(t, w, n2, rays, u=None, seed=0, **kw):
        seen.append((None if u is None else u.clone(), seed))
        n = t.shape[0]
        return None, None, torch.zeros(n, 1, n2)
    monkeypatch.setattr(ops, "resample", fake_resample)
    n, n1, n2 = 5, 12, 7
    z = torch.sort(torch.rand(n, n1), -1)[0]
    w = torch.rand(n, n1 - 2)
    mod.sample_pdf(z, w, n2, det=True)
    assert torch.equal(seen[-1][0].reshape(n, n2), torch.linspace(0., 1., n2).expand(n, n2))
    mod.sample_pdf(z, w, n2, pytest=True)
    np.random.seed(0)
    want = torch.as_tensor(np.random.rand(n, n2), dtype=torch.float32)
    assert torch.equal(seen[-1][0].reshape(n, n2), want)                      # the reference's numbers, bit for bit
    mod.sample_pdf(z, w, n2, det=True, pytest=True)                            # (:37-39: numpy's float64 linspace, then torch.Tensor(u))
    assert torch.equal(seen[-1][0].reshape(n, n2), torch.as_tensor(np.linspace(0., 1., n2), dtype=torch.float32).expand(n, n2))
    mod.sample_pdf(z, w, n2)
    mod.sample_pdf(z, w, n2)
    assert seen[-1][0] is None and seen[-2][0] is None and seen[-1][1] != seen[-2][1]     # fresh draws per call
    mod.sample_pdf(z, w, n2, seed=9)
    mod.sample_pdf(z, w, n2, seed=9)
    assert seen[-1][1] == seen[-2][1] == 9                                     # reproducible on request
