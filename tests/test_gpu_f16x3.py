"""The "fp16x3" SpaceNet kernel (fp32-accurate split-fp16 MFMA) against the same oracle, with the SAME stated
tolerances as the exact-fp32 kernel.  Needs an MI355X: `pytest -m gpu`."""
import numpy as np
import pytest
import torch

from oracle import stnerf_oracle as O
from stnerf_amd import synthetic as syn
from test_gpu_ops import _net_close, dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from stnerf_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("use_time, deep", [(False, False), (True, False), (False, True), (True, True)])
def test_spacenet_f16x3_vs_fp64_oracle(ops, use_time, deep):
    torch.manual_seed(11)
    rs = np.random.RandomState(5)
    sd = syn.spacenet_state("net", rs, use_time, deep_rgb=deep)
    n, s = 700, 13
    pos = (torch.rand(n, s, 3) - 0.5) * 6.0
    dirs = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1)
    times = torch.rand(n) * 100 + 1
    net = ops.pack_spacenet(sd, "net", precision="fp16x3")
    raw = torch.full((n, s, 4), float("nan"), device="cuda")
    ops.spacenet_fwd(net, dev(pos), dev(dirs), dev(times) if use_time else None, raw)
    sd64 = {k: v.double() for k, v in sd.items()}
    rgb64, sig64 = O.space_net(sd64, "net", pos.double(), dirs.double(), times.double().reshape(-1, 1) if use_time else None)
    _net_close(raw[..., :3].cpu(), rgb64, 4.0, "rgb")
    _net_close(raw[..., 3:].cpu(), sig64, 60.0, "sigma")
    # accuracy class: no worse than 4x the fp32 CPU chain's own error vs fp64 (same bar as the fp32 kernel)
    rgb32, sig32 = O.space_net(sd, "net", pos, dirs, times.reshape(-1, 1) if use_time else None)
    e_gpu = float((raw[..., 3:].cpu().double() - sig64).abs().max())
    e_cpu = float((sig32.double() - sig64).abs().max())
    print(f"sigma max err: fp16x3 {e_gpu:.3e}  fp32 CPU chain {e_cpu:.3e}")
    assert e_gpu <= 4 * e_cpu + 1e-6, (e_gpu, e_cpu)
    # the exact-fp32 kernel on the same inputs agrees to fp32 rounding
    net32 = ops.pack_spacenet(sd, "net")
    raw32 = torch.empty_like(raw)
    ops.spacenet_fwd(net32, dev(pos), dev(dirs), dev(times) if use_time else None, raw32)
    torch.testing.assert_close(raw, raw32, rtol=2e-5, atol=2e-4)


def test_spacenet_f16x3_worklist_determinism_and_untouched_rows(ops):
    torch.manual_seed(12)
    rs = np.random.RandomState(6)
    sd = syn.spacenet_state("net", rs, True)
    n, l, s = 500, 3, 10
    xyz = (torch.rand(n, l, s, 3) - 0.5) * 4.0
    rays = torch.cat([torch.zeros(n, 3), torch.nn.functional.normalize(torch.randn(n, 3), dim=-1), torch.rand(n, l) * 5], -1)
    mask = (torch.rand(n, l) < 0.4).to(torch.uint8)
    net = ops.pack_spacenet(sd, "net", precision="fp16x3")
    dx, dr, dm = dev(xyz), dev(rays), dev(mask)
    lst, cnt = ops.compact_rays(dm)
    layer = 2
    outs = []
    for _ in range(2):
        raw = torch.full((n, l, s, 4), 7.0, device="cuda")
        ops.spacenet_fwd(net, dx[:, layer], dr[:, 3:6], dr[:, 6 + layer], raw[:, layer], ray_list=lst[layer],
                         ray_count=cnt[layer:layer + 1])
        outs.append(raw)
    assert torch.equal(outs[0], outs[1])
    idx = mask[:, layer].bool()
    rgb, sig = O.space_net(sd, "net", xyz[idx, layer], rays[idx, 3:6], rays[idx, 6 + layer].reshape(-1, 1))
    rc = outs[0].cpu()
    torch.testing.assert_close(rc[idx][:, layer, :, :3], rgb, rtol=2e-5, atol=1e-4)
    torch.testing.assert_close(rc[idx][:, layer, :, 3:], sig, rtol=2e-5, atol=2e-3)
    assert bool((rc[~idx] == 7.0).all()) and bool((rc[:, :layer] == 7.0).all())


def test_f16x3_rejects_weights_outside_the_split_range(ops):
    sd = syn.spacenet_state("net", np.random.RandomState(1), False)
    sd["net.stage1.2.weight"] = sd["net.stage1.2.weight"].clone()
    sd["net.stage1.2.weight"][3, 5] = 400.0
    with pytest.raises(ValueError, match="does not fit the fp16 split"):
        ops.pack_spacenet(sd, "net", precision="fp16x3")


def test_motionnet_f16x3_vs_fp64_oracle(ops):
    torch.manual_seed(13)
    rs = np.random.RandomState(7)
    sd = syn.motionnet_state("net", rs)
    n, s = 333, 9
    pos = (torch.rand(n, s, 3) - 0.5) * 4.0
    times = torch.where(torch.rand(n) < 0.5, torch.floor(torch.rand(n) * 50), torch.rand(n) * 50)
    net = ops.pack_motionnet(sd, "net", precision="fp16x3")
    flow = torch.empty(n, s, 3, device="cuda")
    x = dev(pos)
    ops.motionnet_fwd(net, x, dev(times), flow=flow, add_to_xyz=True)
    sd64 = {k: v.double() for k, v in sd.items()}
    xt = torch.cat([pos, times.view(n, 1, 1).repeat(1, s, 1)], -1)
    ref64 = O.motion_net(sd64, "net", xt.double())
    _net_close(flow.cpu(), ref64, 1.0, "flow")
    torch.testing.assert_close(x.cpu(), pos + flow.cpu(), rtol=0, atol=1e-6)


@pytest.mark.parametrize("name", ["fwd_c1", "fwd_c3", "fwd_edit", "fwd_hide", "fwd_nonretime", "fwd_only_coarse",
                                  "batchify_chunked", "batchify_small", "fwd_bkgd_time", "fwd_same_spacenet", "fwd_deep_rgb", "fwd_no_raw_no_dir", "fwd_c4", "fwd_c5"])
def test_whole_path_fp16x3_matches_reference_fixtures(name):
    """The drop-in boundary in fp16x3 precision against the reference's own outputs: same tolerances as fp32."""
    import test_gpu_render as R
    R.run_forward_case(name, precision="fp16x3")


def test_c2_full_view_fp16x3_agrees_with_fp32():
    """C2 (512x512, 64+64) rendered in both arithmetic modes with the same device RNG stream."""
    import test_gpu_render as R
    from stnerf_amd.render.render_pose import render_pose
    meta = dict(L=1, n1=64, n2=64, space_time=True, deform_time=False, weight_seed=40, edit={})
    model = R.build_model(meta)
    K, T = syn.camera(512, 512, 8.0)
    model.seed = 3
    imgs = {}
    for prec in ("fp32", "fp16x3"):
        model.set_precision(prec)
        imgs[prec] = render_pose(model, T, K, 512, 512, [(0, 1), (1, 2.5)], far=20.0)[0]
    a, b = imgs["fp32"], imgs["fp16x3"]
    per_pix = (a - b).abs().max(-1)[0]
    frac = float((per_pix <= R.COLOR_ATOL).float().mean())
    quality = float(-10 * torch.log10(torch.mean((a - b) ** 2)))
    print(f"fp16x3 vs fp32 at C2: {100 * frac:.3f} % of pixels within {R.COLOR_ATOL}, PSNR {quality:.1f} dB, max {float(per_pix.max()):.2e}")
    assert frac >= 0.99 and quality >= 70.0     # two fp32-accurate evaluations of one view (mostly background rays)


def test_f16x3_range_guard_flags_overflow_and_the_model_falls_back_to_f32(ops):
    """An activation beyond the fp16 range (>= 65520) becomes inf in the hi/lo split.  The kernels track the largest
    activation they split and raise a device flag; LayeredRFRender re-runs the launch in exact f32."""
    import test_gpu_render as R
    rs = np.random.RandomState(3)
    sd = syn.spacenet_state("net", rs, False)
    for k in ("net.stage1.2.weight", "net.stage1.4.weight"):           # |W| stays < 234, activations reach ~1e5 .. 1e6
        sd[k] = sd[k] * 1500.0
    n, s = 300, 8
    pos = (torch.rand(n, s, 3) - 0.5) * 4.0
    dirs = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1)
    ref_rgb, ref_sig = O.space_net(sd, "net", pos, dirs)
    assert bool(torch.isfinite(ref_sig).all()), "the f32 network itself must be fine"
    net = ops.pack_spacenet(sd, "net", precision="fp16x3")
    raw = torch.empty(n, s, 4, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.spacenet_fwd(net, dev(pos), dev(dirs), None, raw, overflow=flag)
    # the damage is SILENT without the flag: inf - inf = NaN in the next layer's products, which that layer's integer ReLU
    # flushes to 0 -- finite, wrong outputs
    assert int(flag.item()) == 1
    assert float((raw[..., 3:].cpu() - ref_sig).abs().max()) > 0.5 * float(ref_sig.abs().max()) and bool(torch.isfinite(raw).all())
    # a well-scaled network leaves the flag alone
    ok = ops.pack_spacenet(syn.spacenet_state("net", np.random.RandomState(3), False), "net", precision="fp16x3")
    flag.zero_()
    ops.spacenet_fwd(ok, dev(pos), dev(dirs), None, raw, overflow=flag)
    assert int(flag.item()) == 0 and bool(torch.isfinite(raw).all())
    # whole path: a model whose background net overflows in fp16x3 renders exactly what the f32 mode renders
    meta = dict(L=1, n1=12, n2=6, space_time=True, deform_time=False, weight_seed=77, edit={})
    model = R.build_model(meta)
    with torch.no_grad():
        for name in ("stage1", ):
            getattr(model.bkgd_spacenet, name)[2].weight.mul_(1500.0)
            getattr(model.bkgd_spacenet, name)[4].weight.mul_(1500.0)
    K, T = syn.camera(20, 32, 9.0)
    rays = ops.generate_rays(K, T, 20, 32, frame_ids=[1.0, 2.5])
    model.seed = 4
    with torch.no_grad():
        want = model(rays, None, None)
        model.set_precision("fp16x3")
        model.f16_fallbacks = 0
        got = model(rays, None, None)
    assert model.f16_fallbacks == 1 and model.bkgd_spacenet.precision == "fp16x3"
    assert torch.equal(got[0][0], want[0][0]) and torch.equal(got[1][0], want[1][0]) and bool(torch.isfinite(got[0][0]).all())
