"""Teacher-forced fine stage: the HIP resampler, networks and compositor fed the REFERENCE'S OWN intermediates
(tests/golden/tf_*.npz, recorded from inside modeling/layered_rfrender.py:459-611 by tests/golden/make_golden.py
--teacher), every ray held to the fp32 tolerance.

The free-running fine image can only be compared through a relative bar (tests/test_gpu_render.py fine_stage_bar): a
last-ulp difference in the coarse weights moves a fine sample across the ``den < 1e-5`` switch of utils/sample_pdf.py:59
and, times 2^9 in the positional encoding, moves a pixel by more than any fp32 tolerance -- in the reference's own fp32
evaluation just as much.  With the reference's depths / points as the input that sensitivity is gone and the comparison is
direct:

  1. ops.resample(reference coarse t, reference coarse weights, reference u) == the recorded sorted z_vals_fine, BIT FOR
     BIT, and its points == the recorded pre-deformation points;
  2. ops.mlp_stage(recorded pre-deformation points) -> (rgb, sigma) of every fine sample within NET_TOL of what the
     reference's fine SpaceNets returned (MotionNet fused in front, frame ids fractional);
  3. ops.composite(recorded z_vals_fine, those network outputs, the call's thresholds / alpha / near) -> EVERY ray of the
     fine mixed image and of every layer's image within 5e-5 (colour, acc) / 5e-4 (depth) of the reference's outputs.

Both arithmetics (bf16x3 = the library default, exact f32).  Needs an MI355X: `pytest -m gpu`."""
import pytest
import torch

from conftest import load_golden
from stnerf_amd import synthetic as syn

pytestmark = pytest.mark.gpu

CASES = ["tf_c3_64_64", "tf_c3_90_30", "tf_c4", "tf_c5", "tf_edit"]
COLOR_ATOL, DEPTH_ATOL = 5e-5, 5e-4            # the stated fp32 tolerance of the path (tests/test_gpu_render.py)
# network outputs against the reference's fp32 evaluation (both sides round: twice the vs-fp64 bar of tests/test_gpu_stage.py)
NET_RTOL, NET_ATOL_RGB, NET_ATOL_SIGMA = 4e-5, 4e-5 * 4.0, 4e-5 * 60.0


@pytest.fixture(scope="module")
def ops():
    from stnerf_amd import ops as _ops
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return _ops


def dev(x):
    return x.cuda().contiguous()


def _load(name):
    meta, a = load_golden(name)
    l = meta["L"] + 1
    stack = lambda key: torch.stack([a[f"{key}{i}"] for i in range(l)], 1)
    mask = torch.stack([a[f"mask{i}"] for i in range(l)], 1)
    return meta, a, l, stack, mask


def _call_params(meta):
    """thresholds / sigma scale / near of the fine stage as modeling/layered_rfrender.py:538-576,605 applies them for a
    direct model(...) call (retiming ray layout)."""
    l = meta["L"] + 1
    kw = meta["call_kwargs"]
    thr, bthr = kw.get("density_threshold", 0.0001), kw.get("bkgd_density_threshold", 0)
    alpha = meta["edit"].get("alpha", 1)
    near = meta["edit"].get("near", 0)
    return [bthr] + [thr] * (l - 1), [alpha if i == 2 else 1.0 for i in range(l)], near


def _edits(meta, l, fine=True):
    """Per-layer inverse point edits of the fine stage (:467-475: a None shift skips that layer's scale step too)."""
    shift, scale = meta["edit"].get("shift"), meta["edit"].get("scale")
    if shift is None and scale is None:
        return None
    out = []
    for i in range(l):
        if shift is not None and shift[i] is None:
            out.append((None, None))
            continue
        out.append((shift[i] if shift is not None else None, scale[i] if scale is not None else None))
    return out


def _pivot(L):
    bk, per = syn.scene_boxes(L)
    first = torch.cat([bk.float(), per[0].float()], 0)
    centre = first.mean(1)
    centre[:, 2] = first[:, 1, 2]
    return (centre[2] + centre[1]) / 2


@pytest.mark.parametrize("name", CASES)
def test_resampler_on_reference_coarse_outputs_is_bit_equal(ops, name):
    meta, a, l, stack, mask = _load(name)
    n, n1, n2 = a["rays"].shape[0], meta["n1"], meta["n2"]
    t, w = stack("pdf_t"), stack("coarse_weights")
    u = torch.stack([a[f"u{i}"] for i in range(l)], 0)
    edits = _edits(meta, l)
    tf, xyz, z, inds, cdf = ops.resample(dev(t), dev(w), n2, dev(a["rays"]), u=dev(u), edits=edits,
                                         pivot=_pivot(meta["L"]) if edits else None, debug=True)
    tf, xyz, z = tf.cpu(), xyz.cpu(), z.cpu()
    live = mask.bool().clone()
    live[:, 0] = True                                   # the background is resampled on every ray
    for i in range(l):
        rows = live[:, i]
        assert int(rows.sum()) > 0
        assert torch.equal(z[rows, i], a[f"pdf_z{i}"][rows]), f"layer {i}: new depths differ from the reference's sample_pdf"
        assert torch.equal(tf[rows, i], a[f"z_vals_fine{i}"][rows]), f"layer {i}: sorted fine depths differ"
        hit = mask[:, i].bool() if i else rows
        assert torch.equal(xyz[hit, i], a[f"xyz_fine_pre{i}"][hit]), f"layer {i}: fine sample points differ"


@pytest.mark.parametrize("precision", ["bf16x3", "fp32"])
@pytest.mark.parametrize("name", CASES)
def test_fine_stage_on_reference_samples_every_ray(ops, name, precision):
    meta, a, l, stack, mask = _load(name)
    L, n, S = meta["L"], a["rays"].shape[0], meta["n1"] + meta["n2"]
    st, dt = meta["space_time"], meta["deform_time"]
    sd = syn.make_state_dict(L, st, dt, seed=meta["weight_seed"])
    rays, dm = dev(a["rays"]), dev(mask.to(torch.uint8))
    lst, cnt = ops.compact_rays(dm)
    xyz = dev(stack("xyz_fine_pre"))                                       # (n, l, S, 3): the reference's fine points
    raw = torch.full((n, l, S, 4), 7.0, device="cuda")
    layers = []
    for i in range(1, l):
        layers.append(dict(space=ops.pack_spacenet(sd, f"spacenets_fine.{i - 1}", precision=precision),
                           motion=ops.pack_motionnet(sd, f"time_deform_nets.{i - 1}", precision=precision) if dt else None,
                           xyz=xyz[:, i], raw=raw[:, i], times=rays[:, 6 + i], ray_list=lst[i], ray_count=cnt[i:i + 1]))
    layers.append(dict(space=ops.pack_spacenet(sd, "bkgd_spacenet_fine", precision=precision), motion=None, xyz=xyz[:, 0],
                       raw=raw[:, 0], times=None, plain_time=True))
    ops.mlp_stage(layers, rays[:, 3:6], S)
    got = raw.cpu()
    hit = mask.bool().clone()
    hit[:, 0] = True
    assert bool((got[~hit] == 7.0).all())
    # 2. every fine sample's network output against the reference's own fine SpaceNet outputs
    worst = {}
    for i in range(l):
        rows = hit[:, i]
        for what, g, ref, atol in (("rgb", got[rows, i, :, :3], a[f"raw_rgb_fine{i}"][rows], NET_ATOL_RGB),
                                   ("sigma", got[rows, i, :, 3:], a[f"raw_sigma_fine{i}"][rows], NET_ATOL_SIGMA)):
            err = (g - ref).abs()
            bound = NET_RTOL * ref.abs() + atol
            worst[(i, what)] = float((err / bound).max())
            assert bool((err <= bound).all()), f"{name} {precision} layer {i} {what}: max err {float(err.max()):.3e} = {worst[(i, what)]:.2f} x the bound"
    # 3. composite on the reference's depths: every ray, every layer, mixed and per-layer
    thresholds, sigma_scale, near = _call_params(meta)
    raw[~hit.cuda()] = 0.0                                                 # rows of rays a layer does not list are never read ...
    lo, mo, _, _ = ops.composite(dev(stack("z_vals_fine")), raw, dm, near=near, fine=True, thresholds=thresholds,
                                 sigma_scale=sigma_scale, evaluated=[2] + [1] * (l - 1))
    lo, mo = lo.cpu(), mo.cpu()
    checks = [("fine_mixed", mo)] + [(f"fine_layer{i}", lo[:, i]) for i in range(l)]
    for tag, g in checks:
        for key, cols, tol in (("color", slice(0, 3), COLOR_ATOL), ("depth", slice(3, 4), DEPTH_ATOL), ("acc", slice(4, 5), COLOR_ATOL)):
            ref = a[f"{tag}_{key}"]
            err = (g[:, cols] - ref).abs()
            assert bool((err <= tol).all()), (f"{name} {precision} {tag} {key}: {int((err > tol).sum())} of {err.numel()} values beyond {tol:g}, "
                                              f"max {float(err.max()):.3e}")
    assert float(a["fine_mixed_color"].std()) > 0.01
    print(f"{name} {precision}: worst network error / bound {max(worst.values()):.2f}; "
          f"fine mixed colour max |d| {float((mo[:, :3] - a['fine_mixed_color']).abs().max()):.2e}, "
          f"depth {float((mo[:, 3:4] - a['fine_mixed_depth']).abs().max()):.2e}")


@pytest.mark.parametrize("name", CASES)
def test_compositor_alone_on_reference_network_outputs(ops, name):
    """The compositor with BOTH inputs from the reference (its fine depths and its fine network outputs): what is left is the
    compositor's own arithmetic -- exp, sigmoid, the transmittance product, the merge."""
    meta, a, l, stack, mask = _load(name)
    raw = torch.cat([stack("raw_rgb_fine"), stack("raw_sigma_fine")], -1)
    thresholds, sigma_scale, near = _call_params(meta)
    lo, mo, _, _ = ops.composite(dev(stack("z_vals_fine")), dev(raw), dev(mask.to(torch.uint8)), near=near, fine=True,
                                 thresholds=thresholds, sigma_scale=sigma_scale, evaluated=[2] + [1] * (l - 1))
    lo, mo = lo.cpu(), mo.cpu()
    for tag, g in [("fine_mixed", mo)] + [(f"fine_layer{i}", lo[:, i]) for i in range(l)]:
        torch.testing.assert_close(g[:, 0:3], a[f"{tag}_color"], rtol=1e-5, atol=4e-6)
        torch.testing.assert_close(g[:, 3:4], a[f"{tag}_depth"], rtol=1e-5, atol=4e-5)
        torch.testing.assert_close(g[:, 4:5], a[f"{tag}_acc"], rtol=1e-5, atol=4e-6)
