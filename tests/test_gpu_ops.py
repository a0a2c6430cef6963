"""Op-level parity of the HIP kernels (through the C ABI) against the CPU oracle and the golden
fixtures produced by the reference.  Needs an MI355X: `pytest -m gpu`.

Exactness classes:
  * bit-exact   : hit masks, coarse depths/points (same jitter), searchsorted indices and merge
                  order (same inputs);
  * fp32 tol.   : network outputs, composited colours (accumulation order differs from ATen's).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import stnerf_oracle as O
from stnerf_amd import synthetic as syn

pytestmark = pytest.mark.gpu

# Stated fp32 tolerances.  Network outputs are compared against an fp64 evaluation of the same
# formulas: |err| <= NET_RTOL * |ref| + NET_ATOL * (scale of the output head).
NET_RTOL = 2e-5
NET_ATOL = 2e-5


@pytest.fixture(scope="module")
def ops():
    from stnerf_amd import ops as _ops
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return _ops


def dev(x):
    return x.cuda().contiguous()


def test_device_is_gfx950():
    from stnerf_amd import hip
    info = hip.device_info()
    assert info["arch"].startswith("gfx950"), info
    assert info["cu_count"] == 256 and info["lds_bytes_per_cu"] >= 160 * 1024, info


def test_generate_rays(ops):
    meta, a = load_golden("generate_rays")
    h, w = meta["h"], meta["w"]
    rays = ops.generate_rays(a["K"], a["T"], h, w, frame_ids=[1.0, 2.5, 3.0])
    assert rays.shape == (h * w, 9)
    torch.testing.assert_close(rays[:, :6].cpu(), a["rays"], rtol=0, atol=1e-6)
    assert torch.equal(rays[:, 6:].cpu(), torch.tensor([1.0, 2.5, 3.0]).repeat(h * w, 1))
    # a window of a large view == the same rows of the oracle's full view
    K, T = syn.camera(270, 480, 17.0)
    full = O.generate_rays(K, T, 270, 480)
    part = ops.generate_rays(K, T, 270, 480, first_ray=1000, n=5000)
    torch.testing.assert_close(part.cpu(), full[1000:6000], rtol=0, atol=1e-6)


def test_sampler_golden_bit_exact(ops):
    meta, a = load_golden("sampler")
    rays, boxes = dev(a["rays"]), dev(a["boxes"])
    fn = ops.intersect(rays, boxes)
    assert torch.equal(fn.cpu(), a["far_near"].permute(1, 0, 2))
    t, xyz, mask = ops.sample_coarse(rays, boxes, meta["n1"], jitter=dev(a["jitter"]))
    assert torch.equal(t.cpu(), a["t"].squeeze(-1).permute(1, 0, 2))
    assert torch.equal(xyz.cpu(), a["xyz"].permute(1, 0, 2, 3))
    assert torch.equal(mask.cpu().bool(), a["mask"].permute(1, 0))


@pytest.mark.parametrize("n1", [24, 90, 13])   # 4, 2 and 1 samples per thread (sampler.hip)
@pytest.mark.parametrize("edit", [False, True])
def test_sampler_random_bit_exact(ops, edit, n1):
    torch.manual_seed(7)
    n, L = 6000, 3
    K, T = syn.camera(60, 100, -25.0)
    rays = torch.cat([O.generate_rays(K, T, 60, 100), syn.frame_id_columns(n, L)], -1)
    bk, per = syn.scene_boxes(L)
    boxes = torch.cat([bk, per[2]], 0)
    jitter = torch.rand(L + 1, n, n1)
    edits, pivot = None, None
    if edit:
        edits = [(None, 1.0), ([0.1, -0.05, 0.2], 1.3), (None, None), ([0.0, 0.3, 0.0], 0.7)]
        pivot = torch.tensor([0.1, 0.0, -1.0])
    # per-ray boxes path == shared boxes path
    t, xyz, mask = ops.sample_coarse(dev(rays), dev(boxes), n1, jitter=dev(jitter), edits=edits, pivot=pivot)
    t2, xyz2, mask2 = ops.sample_coarse(dev(rays), dev(boxes.unsqueeze(0).repeat(n, 1, 1, 1)), n1,
                                        jitter=dev(jitter), edits=edits, pivot=pivot)
    assert torch.equal(t, t2) and torch.equal(xyz, xyz2) and torch.equal(mask, mask2)
    ts, pts, ms = O.sample_coarse(rays, boxes.unsqueeze(0).repeat(n, 1, 1, 1), n1, list(jitter))
    for i in range(L + 1):
        assert torch.equal(t[:, i].cpu(), ts[i].squeeze(-1)), i
        assert torch.equal(mask[:, i].cpu().bool(), ms[i]), i
        p = pts[i]
        if edit:
            sh, sc = edits[i]
            if sh is not None:
                p = p - torch.tensor(sh)
            if sc is not None:
                p = (p - pivot) / sc + pivot
        assert torch.equal(xyz[:, i].cpu(), p), i
    assert 0.05 < ms[1].float().mean() < 0.95


def test_device_rng_statistics_and_chunk_invariance(ops):
    n, L, n1 = 4096, 1, 30   # 30: the two-samples-per-thread variant of the sampler
    K, T = syn.camera(64, 64, 5.0)
    rays = dev(torch.cat([O.generate_rays(K, T, 64, 64), syn.frame_id_columns(n, L)], -1))
    bk, per = syn.scene_boxes(L)
    boxes = dev(torch.cat([bk, per[0]], 0))
    t, _, _ = ops.sample_coarse(rays, boxes, n1, seed=1234)
    fn = ops.intersect(rays, boxes)
    start = fn[:, 0, 1].clamp(min=0)
    width = (fn[:, 0, 0] - start) / n1
    xi = (t[:, 0] - start[:, None]) / width[:, None] - torch.arange(n1, device="cuda")[None]
    assert xi.min() > -1e-3 and xi.max() < 1 + 1e-3
    assert abs(float(xi.mean()) - 0.5) < 5e-3 and abs(float(xi.var()) - 1 / 12) < 5e-3
    # the draw of (ray, layer, sample) does not depend on chunking
    ta, _, _ = ops.sample_coarse(rays[1000:3000].contiguous(), boxes, n1, seed=1234, ray_index_base=1000)
    assert torch.equal(ta, t[1000:3000])
    tb, _, _ = ops.sample_coarse(rays, boxes, n1, seed=1235)
    assert not torch.equal(tb, t)


def test_compact_rays(ops):
    torch.manual_seed(3)
    mask = (torch.rand(10000, 4) < torch.tensor([1.0, 0.3, 0.0, 0.7])).to(torch.uint8)
    lst, cnt = ops.compact_rays(dev(mask))
    for i in range(4):
        c = int(cnt[i])
        assert c == int(mask[:, i].sum())
        got = lst[i, :c].cpu().long().sort()[0]
        assert torch.equal(got, torch.nonzero(mask[:, i])[:, 0])


def _net_close(got, ref64, scale, what):
    err = (got.double() - ref64).abs()
    bound = NET_RTOL * ref64.abs() + NET_ATOL * scale
    assert bool((err <= bound).all()), f"{what}: max err {float(err.max()):.3e}, worst excess {float((err - bound).max()):.3e}"


def test_nets_golden(ops):
    meta, a = load_golden("nets")
    rs = np.random.RandomState(meta["weight_seed"])
    sd_t = syn.spacenet_state("net", rs, True)
    sd_n = syn.spacenet_state("net", rs, False)
    sd_m = syn.motionnet_state("net", rs)
    pos, dirs, times = dev(a["pos"]), dev(a["dirs"]), dev(a["times"].reshape(-1))
    n, s = pos.shape[:2]
    for sd, rgb_k, sig_k, tm in ((sd_t, "rgb_t", "sigma_t", times), (sd_n, "rgb_n", "sigma_n", None)):
        net = ops.pack_spacenet(sd, "net")
        raw = torch.full((n, s, 4), float("nan"), device="cuda")
        ops.spacenet_fwd(net, pos, dirs, tm, raw)
        torch.testing.assert_close(raw[..., :3].cpu(), a[rgb_k], rtol=2e-5, atol=1e-4)
        torch.testing.assert_close(raw[..., 3:].cpu(), a[sig_k], rtol=2e-5, atol=2e-3)
    mot = ops.pack_motionnet(sd_m, "net")
    for tv, key in ((times, "flow_frac"), (torch.floor(times), "flow_int")):
        flow = torch.empty(n, s, 3, device="cuda")
        x = pos.clone()
        ops.motionnet_fwd(mot, x, tv, flow=flow, add_to_xyz=True)
        torch.testing.assert_close(flow.cpu(), a[key], rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(x.cpu(), a["pos"] + a[key], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("use_time, deep", [(False, False), (True, False), (False, True), (True, True)])
def test_spacenet_vs_fp64_oracle(ops, use_time, deep):
    torch.manual_seed(11)
    rs = np.random.RandomState(5)
    sd = syn.spacenet_state("net", rs, use_time, deep_rgb=deep)
    n, s = 700, 13                       # 9100 rows: many tiles, ragged tail, rays straddling tiles
    pos = (torch.rand(n, s, 3) - 0.5) * 6.0
    dirs = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1)
    times = torch.rand(n) * 100 + 1
    net = ops.pack_spacenet(sd, "net")
    raw = torch.full((n, s, 4), float("nan"), device="cuda")
    ops.spacenet_fwd(net, dev(pos), dev(dirs), dev(times) if use_time else None, raw)
    sd64 = {k: v.double() for k, v in sd.items()}
    rgb64, sig64 = O.space_net(sd64, "net", pos.double(), dirs.double(), times.double().reshape(-1, 1) if use_time else None)
    _net_close(raw[..., :3].cpu(), rgb64, 4.0, "rgb")
    _net_close(raw[..., 3:].cpu(), sig64, 60.0, "sigma")
    # and it is at least as close to fp64 as the fp32 CPU oracle is (x4 slack)
    rgb32, sig32 = O.space_net(sd, "net", pos, dirs, times.reshape(-1, 1) if use_time else None)
    e_gpu = float((raw[..., 3:].cpu().double() - sig64).abs().max())
    e_cpu = float((sig32.double() - sig64).abs().max())
    assert e_gpu <= 4 * e_cpu + 1e-6, (e_gpu, e_cpu)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("include_input, use_dir, use_time", [(False, True, True), (True, False, True), (False, False, False)])
def test_spacenet_flavours_without_raw_input_or_direction(ops, precision, include_input, use_dir, use_time):
    """TKERNEL_INC_RAW=False / USE_DIR=False are packed as zero weight columns (ops._pe_columns): same kernels."""
    torch.manual_seed(12)
    sd = syn.spacenet_state("net", np.random.RandomState(6), use_time, include_input=include_input, use_dir=use_dir)
    n, s = 300, 9
    pos = (torch.rand(n, s, 3) - 0.5) * 6.0
    dirs = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1)
    times = torch.rand(n) * 50 + 1
    net = ops.pack_spacenet(sd, "net", precision=precision)
    assert net.use_time == use_time
    raw = torch.full((n, s, 4), float("nan"), device="cuda")
    ops.spacenet_fwd(net, dev(pos), dev(dirs), dev(times) if use_time else None, raw)
    sd64 = {k: v.double() for k, v in sd.items()}
    rgb64, sig64 = O.space_net(sd64, "net", pos.double(), dirs.double(), times.double().reshape(-1, 1) if use_time else None)
    _net_close(raw[..., :3].cpu(), rgb64, 4.0, "rgb")
    _net_close(raw[..., 3:].cpu(), sig64, 60.0, "sigma")


def test_motionnet_without_raw_input(ops):
    sd = syn.motionnet_state("net", np.random.RandomState(8), include_input=False)
    torch.manual_seed(13)
    n, s = 200, 7
    xyz = (torch.rand(n, s, 3) - 0.5) * 4.0
    times = torch.rand(n) * 20 + 1
    flow = torch.empty(n, s, 3, device="cuda")
    ops.motionnet_fwd(ops.pack_motionnet(sd, "net"), dev(xyz), dev(times), flow=flow, add_to_xyz=False)
    sd64 = {k: v.double() for k, v in sd.items()}
    ref = O.motion_net(sd64, "net", torch.cat([xyz.double(), times.double().view(n, 1, 1).repeat(1, s, 1)], -1))
    _net_close(flow.cpu(), ref, 1.0, "flow")


def test_spacenet_worklist_and_strided_views(ops):
    """Masked evaluation through (ray_list, ray_count) into ray-major strided buffers."""
    torch.manual_seed(12)
    rs = np.random.RandomState(6)
    sd = syn.spacenet_state("net", rs, True)
    n, l, s = 500, 3, 10
    xyz = (torch.rand(n, l, s, 3) - 0.5) * 4.0
    rays = torch.cat([torch.zeros(n, 3), torch.nn.functional.normalize(torch.randn(n, 3), dim=-1),
                      torch.rand(n, l) * 5], -1)
    mask = (torch.rand(n, l) < 0.4).to(torch.uint8)
    net = ops.pack_spacenet(sd, "net")
    dx, dr, dm = dev(xyz), dev(rays), dev(mask)
    lst, cnt = ops.compact_rays(dm)
    raw = torch.full((n, l, s, 4), 7.0, device="cuda")
    layer = 2
    ops.spacenet_fwd(net, dx[:, layer], dr[:, 3:6], dr[:, 6 + layer], raw[:, layer], ray_list=lst[layer], ray_count=cnt[layer:layer + 1])
    idx = mask[:, layer].bool()
    rgb, sig = O.space_net(sd, "net", xyz[idx, layer], rays[idx, 3:6], rays[idx, 6 + layer].reshape(-1, 1))
    rc = raw.cpu()
    torch.testing.assert_close(rc[idx][:, layer, :, :3], rgb, rtol=2e-5, atol=1e-4)
    torch.testing.assert_close(rc[idx][:, layer, :, 3:], sig, rtol=2e-5, atol=2e-3)
    assert bool((rc[~idx] == 7.0).all()) and bool((rc[:, :layer] == 7.0).all())  # untouched elsewhere
    # determinism: same inputs twice -> bitwise equal
    raw2 = torch.full((n, l, s, 4), 7.0, device="cuda")
    ops.spacenet_fwd(net, dx[:, layer], dr[:, 3:6], dr[:, 6 + layer], raw2[:, layer], ray_list=lst[layer], ray_count=cnt[layer:layer + 1])
    assert torch.equal(raw, raw2)


@pytest.mark.parametrize("use_time", [False, True])
def test_rgb_ray_bias_vs_fp64(ops, use_time):
    """stnerf_rgb_ray_bias: the direction / time columns of rgb_net.1 evaluated once per ray,
    out[j] = bias + W[:, 256:] relu([PE_4(dir_j), PE_10(time_j)]) (modeling/spacenet.py:80-86,141-151), against an fp64
    evaluation; rays a list leaves out are not written."""
    torch.manual_seed(3)
    rs = np.random.RandomState(11)
    sd = syn.spacenet_state("net", rs, use_time)
    n = 777
    dirs = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1)
    times = torch.where(torch.rand(n) < 0.5, torch.floor(torch.rand(n) * 30), torch.rand(n) * 30) + 1
    net = ops.pack_spacenet(sd, "net")
    enc = [O.positional_encoding(dirs.double(), 4)]
    if use_time:
        enc.append(O.positional_encoding(times.double().reshape(n, 1), 10))
    e = torch.relu(torch.cat(enc, -1))
    W, b = sd["net.rgb_net.1.weight"].double(), sd["net.rgb_net.1.bias"].double()
    want = b + e @ W[:, 256:].T
    got = ops.rgb_ray_bias(net, dev(dirs), dev(times) if use_time else None).cpu().double()
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 4e-6 * scale, float((got - want).abs().max())
    mask = (torch.rand(n, 2) < 0.4).to(torch.uint8)
    lst, cnt = ops.compact_rays(dev(mask))
    part = ops.rgb_ray_bias(net, dev(dirs), dev(times) if use_time else None, ray_list=lst[1], ray_count=cnt[1:2]).cpu().double()
    hit = mask[:, 1].bool()
    assert torch.equal(part[hit], got[hit]) and bool((part[~hit] == 0).all())


@pytest.mark.parametrize("deep, bkgd_deform, ns", [(False, False, 13), (False, False, 64), (True, True, 9), (False, True, 128), (False, True, 90)])
def test_mlp_stage_is_bit_identical_to_the_per_network_launches(ops, deep, bkgd_deform, ns):
    """stnerf_mlp_stage (one persistent launch: work queue over every layer, MotionNet fused in front of its SpaceNet)
    == stnerf_motionnet_fwd(ADD_TO_XYZ) + stnerf_spacenet_fwd per layer, bit for bit; rows of rays a layer does not list
    stay untouched (csrc/mlp_wave.hip: a wave owns 32 samples, the activations never leave its registers; the stage
    kernel is compared with the fp64 oracle directly in tests/test_gpu_stage.py)."""
    torch.manual_seed(41 + ns)
    rs = np.random.RandomState(9)
    n, l = 1100, 3
    sd_b = syn.spacenet_state("net", rs, bkgd_deform, deep_rgb=deep)        # (a timed background only together with its deform net)
    sd_p = [syn.spacenet_state("net", rs, True, deep_rgb=deep) for _ in range(l - 1)]
    sd_m = [syn.motionnet_state("net", rs) for _ in range(l)]
    xyz = (torch.rand(n, l, ns, 3) - 0.5) * 5.0
    dirs = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1)
    times = torch.where(torch.rand(n, l) < 0.5, torch.floor(torch.rand(n, l) * 30), torch.rand(n, l) * 30) + 1
    rays = torch.cat([torch.zeros(n, 3), dirs, times], -1)
    mask = (torch.rand(n, l) < 0.45).to(torch.uint8)
    mask[:, 0] = 1
    dr, dm = dev(rays), dev(mask)
    lst, cnt = ops.compact_rays(dm)
    bk = ops.pack_spacenet(sd_b, "net")
    sp = [ops.pack_spacenet(s_, "net") for s_ in sd_p]
    mo = [ops.pack_motionnet(s_, "net") for s_ in sd_m]
    # ---- per-network launches (round-1 scheduling)
    x1 = dev(xyz)
    raw1 = torch.full((n, l, ns, 4), 7.0, device="cuda")
    if bkgd_deform:
        ops.motionnet_fwd(mo[0], x1[:, 0], dr[:, 6], add_to_xyz=True, plain_time=True)
    ops.spacenet_fwd(bk, x1[:, 0], dr[:, 3:6], dr[:, 6] if bkgd_deform else None, raw1[:, 0])
    for i in range(1, l):
        ops.motionnet_fwd(mo[i], x1[:, i], dr[:, 6 + i], add_to_xyz=True, ray_list=lst[i], ray_count=cnt[i:i + 1])
        ops.spacenet_fwd(sp[i - 1], x1[:, i], dr[:, 3:6], dr[:, 6 + i], raw1[:, i], ray_list=lst[i], ray_count=cnt[i:i + 1])
    # ---- one persistent launch
    x2 = dev(xyz)
    raw2 = torch.full((n, l, ns, 4), 7.0, device="cuda")
    layers = [dict(space=sp[i - 1], motion=mo[i], xyz=x2[:, i], raw=raw2[:, i], times=dr[:, 6 + i], ray_list=lst[i],
                   ray_count=cnt[i:i + 1]) for i in range(1, l)]
    layers.append(dict(space=bk, motion=mo[0] if bkgd_deform else None, xyz=x2[:, 0], raw=raw2[:, 0],
                       times=dr[:, 6] if bkgd_deform else None, plain_time=True))
    ops.mlp_stage(layers, dr[:, 3:6], ns, deep_rgb=deep)
    for i in range(l):
        bad = (raw2[:, i] != raw1[:, i]).reshape(n, -1).any(-1)
        assert not bool(bad.any()), f"layer {i}: {int(bad.sum())} rays differ, max |d| {float((raw2[:, i] - raw1[:, i]).abs().max()):.3e}"
    assert torch.equal(x2.cpu(), xyz)                              # the deformed points are not written back
    hit = mask.bool()
    assert bool((raw2.cpu()[~hit] == 7.0).all()) and bool(torch.isfinite(raw2).all())
    # STNERF_STAGE_SIGMOID_RGB: the colour comes out as torch.sigmoid(rgb), sigma untouched; the compositor then skips its
    # own sigmoid (rgb_activated) and produces bit-identical images
    raw4 = torch.full((n, l, ns, 4), 7.0, device="cuda")
    for ly, i in zip(layers, list(range(1, l)) + [0]):
        ly["raw"] = raw4[:, i]
    ops.mlp_stage(layers, dr[:, 3:6], ns, deep_rgb=deep, sigmoid_rgb=True)
    hit_d = dm.bool()
    assert torch.equal(raw4[..., 3][hit_d], raw2[..., 3][hit_d])
    torch.testing.assert_close(raw4[..., :3][hit_d], torch.sigmoid(raw2[..., :3][hit_d]), rtol=3e-7, atol=1e-7)
    tt = torch.sort(torch.rand(n, l, ns, device="cuda") * 4.0, -1)[0]
    img_a = ops.composite(tt, raw2.contiguous(), dm, evaluated=[1] * l)
    img_b = ops.composite(tt, raw4.contiguous(), dm, evaluated=[1] * l, rgb_activated=True)
    assert torch.equal(img_a[0], img_b[0]) and torch.equal(img_a[1], img_b[1])
    # and once more: same queue, same bits (dynamic scheduling does not touch the arithmetic)
    raw3 = torch.full((n, l, ns, 4), 7.0, device="cuda")
    for ly, i in zip(layers, list(range(1, l)) + [0]):
        ly["raw"] = raw3[:, i]
    ops.mlp_stage(layers, dr[:, 3:6], ns, deep_rgb=deep)
    assert torch.equal(raw3, raw2)


def test_motionnet_vs_fp64_oracle(ops):
    torch.manual_seed(13)
    rs = np.random.RandomState(7)
    sd = syn.motionnet_state("net", rs)
    n, s = 333, 9
    pos = (torch.rand(n, s, 3) - 0.5) * 4.0
    times = torch.where(torch.rand(n) < 0.5, torch.floor(torch.rand(n) * 50), torch.rand(n) * 50)
    net = ops.pack_motionnet(sd, "net")
    flow = torch.empty(n, s, 3, device="cuda")
    x = dev(pos)
    ops.motionnet_fwd(net, x, dev(times), flow=flow, add_to_xyz=False)
    assert torch.equal(x.cpu(), pos)
    sd64 = {k: v.double() for k, v in sd.items()}
    xt = torch.cat([pos, times.view(n, 1, 1).repeat(1, s, 1)], -1)
    ref64 = O.motion_net(sd64, "net", xt.double())
    _net_close(flow.cpu(), ref64, 1.0, "flow")


def test_composite_golden(ops):
    meta, a = load_golden("composite")
    t = dev(a["t"].squeeze(-1).unsqueeze(1))                      # (n,1,S)
    raw = dev(torch.cat([a["rgb"], a["sigma"]], -1).unsqueeze(1))  # (n,1,S,4)
    layer_out, mixed, w, order = ops.composite(t, raw, None, border=meta["border"], want_weights=True, want_order=True)
    torch.testing.assert_close(w[:, 0].cpu(), a["weights"].squeeze(-1), rtol=1e-5, atol=1e-7)
    for out in (layer_out[:, 0].cpu(), mixed.cpu()):  # with one layer the merged stream == the layer
        torch.testing.assert_close(out[:, :3], a["color"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(out[:, 3:4], a["depth"], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(out[:, 4:5], a["acc"], rtol=1e-5, atol=1e-6)
    S = t.shape[-1]
    assert torch.equal(order.cpu(), torch.arange(S, dtype=torch.int32).repeat(t.shape[0], 1))


@pytest.mark.parametrize("fine", [False, True])
def test_composite_merge_vs_oracle(ops, fine):
    torch.manual_seed(21 + fine)
    n, l, S = 300, 3, 40
    t = torch.sort(torch.rand(n, l, S) * 6.0 - 0.5, -1)[0]
    t[:, 2][torch.rand(n) < 0.3] = -1000.0                       # missed performers: all samples at -1000
    raw = torch.randn(n, l, S, 4) * torch.tensor([2.0, 2.0, 2.0, 4.0])
    mask = torch.ones(n, l, dtype=torch.uint8)
    mask[:, 2] = (t[:, 2, 0] > -999).to(torch.uint8)
    mask[:, 1] = (torch.rand(n) < 0.6).to(torch.uint8)
    near, alpha, thr, bthr = 0.7, 0.4, 0.5, 0.3
    # oracle: the same edits as render_chunk applies (layered_rfrender.py:414-422 / :538-576, :605)
    sig = [raw[:, i, :, 3:].clone() for i in range(l)]
    rgb = [raw[:, i, :, :3].clone() for i in range(l)]
    for i in range(l):
        dead = mask[:, i] == 0
        sig[i][dead] = 0
        rgb[i][dead] = 0
    if fine:
        sig[0][sig[0] < bthr] = 0
    for i in range(1, l):
        if not fine:
            sig[i][t[:, i] < 0] = 0
        sig[i][sig[i] < thr] = 0
        if fine and i == 2:
            sig[i] = sig[i] * alpha
    if not fine:
        sig[0][t[:, 0] < near] = 0
    ts = [t[:, i].unsqueeze(-1) for i in range(l)]
    t_mix, order = torch.sort(torch.cat(ts, -2), dim=-2, stable=True)
    rgb_mix = torch.cat(rgb, -2).gather(1, order.repeat(1, 1, 3))
    sig_mix = torch.cat(sig, -2).gather(1, order)
    per = [O.composite(ts[i], rgb[i], sig[i]) for i in range(l)]
    if fine:
        sig_mix[t_mix < near] = 0
    mix = O.composite(t_mix, rgb_mix, sig_mix)
    lo, mo, w, od = ops.composite(dev(t), dev(raw), dev(mask), near=near, fine=fine, cut_negative_t=not fine,
                                  thresholds=[bthr if fine else None, thr, thr],
                                  sigma_scale=[1.0, 1.0, alpha if fine else 1.0], want_weights=True, want_order=True)
    assert torch.equal(od.cpu().long(), order.squeeze(-1))        # merge order: bit-exact
    for i in range(l):
        torch.testing.assert_close(w[:, i].cpu(), per[i][3].squeeze(-1), rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(lo[:, i, :3].cpu(), per[i][0], rtol=1e-5, atol=2e-6)
        torch.testing.assert_close(lo[:, i, 3:4].cpu(), per[i][1], rtol=1e-5, atol=2e-3)   # depth carries -1000 terms
        torch.testing.assert_close(lo[:, i, 4:5].cpu(), per[i][2], rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(mo[:, :3].cpu(), mix[0], rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(mo[:, 3:4].cpu(), mix[1], rtol=1e-5, atol=2e-3)
    torch.testing.assert_close(mo[:, 4:5].cpu(), mix[2], rtol=1e-5, atol=2e-6)


def test_composite_descending_and_unsorted_layers(ops):
    """A strictly descending layer (negative bin width: an edited box, or a ray that misses the background box) is merged
    through a reversed view; a layer that is neither ascending nor strictly descending takes the general rank.  Both give
    the order of a stable sort of the concatenation, and the composites follow."""
    torch.manual_seed(23)
    n, l, S = 60, 3, 17
    t = torch.sort(torch.rand(n, l, S) * 4.0, -1)[0]
    t[:20, 0] = t[:20, 0].flip(-1)                                # strictly descending background
    t[10:30, 2] = t[10:30, 2].flip(-1)                            # ... and / or a descending performer
    t[40:50, 1] = t[40:50, 1][:, torch.randperm(S)]               # unsorted
    t[50:, 1] = t[50:, 1].flip(-1)
    t[50:, 1, 3] = t[50:, 1, 2]                                   # a tie inside a descending list: general rank
    raw = torch.randn(n, l, S, 4)
    lo, mo, _, od = ops.composite(dev(t), dev(raw), None, want_order=True)
    _, order = torch.sort(t.reshape(n, l * S), dim=-1, stable=True)
    assert torch.equal(od.cpu().long(), order)
    ts = [t[:, i].unsqueeze(-1) for i in range(l)]
    t_mix = torch.cat(ts, -2).gather(1, order.unsqueeze(-1))
    rgb_mix = raw[..., :3].reshape(n, l * S, 3).gather(1, order.unsqueeze(-1).repeat(1, 1, 3))
    sig_mix = raw[..., 3:].reshape(n, l * S, 1).gather(1, order.unsqueeze(-1))
    mix = O.composite(t_mix, rgb_mix, sig_mix)
    torch.testing.assert_close(mo[:, :3].cpu(), mix[0], rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(mo[:, 4:5].cpu(), mix[2], rtol=1e-5, atol=2e-6)
    # the production call (no order output) takes the same branches: bitwise the same images
    lo2, mo2, _, _ = ops.composite(dev(t), dev(raw), None)
    assert torch.equal(lo, lo2) and torch.equal(mo, mo2)


@pytest.mark.parametrize("fine", [False, True])
def test_composite_production_shortcuts_are_bitwise_neutral(ops, fine):
    """Without the `order` output the kernel drops layers a ray misses and reuses a single live layer's composite as the
    mix (render.hip); with `order` it merges everything.  Same inputs -> bit-identical images and weights."""
    torch.manual_seed(29 + fine)
    n, l, S = 4000, 3, 96
    t = torch.sort(torch.rand(n, l, S) * 6.0 - 0.3, -1)[0]
    miss1, miss2 = torch.rand(n) < 0.5, torch.rand(n) < 0.6
    t[:, 1][miss1] = -1000.0                                      # missed performers (most rays: one live layer)
    t[:, 2][miss2] = -1000.0
    bk_miss = torch.rand(n) < 0.1                                 # rays that miss the background box: 0 .. -1000, descending
    t[bk_miss, 0] = -(torch.arange(S).float() + torch.rand(int(bk_miss.sum()), S)) * (1000.0 / S)
    raw = torch.randn(n, l, S, 4) * torch.tensor([2.0, 2.0, 2.0, 4.0])
    mask = torch.stack([torch.rand(n) < 0.97, ~miss1, ~miss2], 1).to(torch.uint8)   # a few grazing background rays
    kw = dict(near=0.4, fine=fine, cut_negative_t=not fine, thresholds=[0.3 if fine else None, 0.5, 0.5],
              sigma_scale=[1.0, 1.0, 0.4 if fine else 1.0], evaluated=[2, 1, 1], want_weights=True)
    lo_a, mo_a, w_a, _ = ops.composite(dev(t), dev(raw), dev(mask), want_order=False, **kw)
    lo_b, mo_b, w_b, od = ops.composite(dev(t), dev(raw), dev(mask), want_order=True, **kw)
    bits = lambda x: x.contiguous().view(torch.int32)      # bit patterns: a descending fine-stage background has negative
    assert torch.equal(bits(lo_a), bits(lo_b))             # deltas, i.e. inf / NaN composites in the reference as well
    assert torch.equal(bits(mo_a), bits(mo_b)) and torch.equal(bits(w_a), bits(w_b))
    assert float(torch.nan_to_num(mo_a[:, 4], nan=0.0, posinf=0.0, neginf=0.0).max()) > 0.5
    # ... and the latency-pipelined kernel for single-layer rays (two_pass, the default) == the general kernel alone
    lo_c, mo_c, w_c, _ = ops.composite(dev(t), dev(raw), dev(mask), want_order=False, two_pass=False, **kw)
    assert torch.equal(bits(lo_a), bits(lo_c)) and torch.equal(bits(mo_a), bits(mo_c)) and torch.equal(bits(w_a), bits(w_c)) and int(((~miss1) & (~miss2)).sum()) > 100 and int((miss1 & miss2).sum()) > 100
    _, order = torch.sort(t.reshape(n, l * S), dim=-1, stable=True)
    assert torch.equal(od.cpu().long(), order)


def test_composite_background_is_composited_where_its_mask_is_clear(ops):
    """evaluated = 2: the layer's network output is used on every ray (bkgd_spacenet runs on all rays and is composited even
    where ray_mask[0] is False, layered_rfrender.py:382-392); evaluated = 1 zeroes it where the mask is clear."""
    torch.manual_seed(33)
    n, S = 64, 20
    t = torch.sort(torch.rand(n, 1, S) * 3.0, -1)[0]
    t[:8] = t[:8, :, :1]                                          # grazing rays: all samples at the same depth
    raw = torch.randn(n, 1, S, 4) + torch.tensor([0.0, 0.0, 0.0, 2.0])
    mask = torch.ones(n, 1, dtype=torch.uint8)
    mask[:8] = 0
    lo2, mo2, _, _ = ops.composite(dev(t), dev(raw), dev(mask), evaluated=[2])
    lo1, _, _, _ = ops.composite(dev(t), dev(raw), dev(mask), evaluated=[1])
    ref = O.composite(t[:, 0].unsqueeze(-1), raw[:, 0, :, :3], raw[:, 0, :, 3:])
    torch.testing.assert_close(lo2[:, 0, :3].cpu(), ref[0], rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(mo2[:, 4:5].cpu(), ref[2], rtol=1e-5, atol=2e-6)
    assert float(lo2[:8, 0, 4].min()) > 0.0 and float(lo1[:8, 0, 4].abs().max()) == 0.0
    assert torch.equal(lo1[8:], lo2[8:])


def test_resample_golden(ops):
    meta, a = load_golden("sample_pdf")
    n, n1 = a["t"].shape
    n2 = meta["n2"]
    # the reference passes interior weights w[..., 1:-1]; the kernel takes the full per-sample weights
    wfull = torch.cat([torch.full((n, 1), 9.0), a["w"], torch.full((n, 1), 9.0)], -1)
    rays = torch.cat([torch.rand(n, 3), torch.nn.functional.normalize(torch.randn(n, 3), dim=-1)], -1)
    tf, xyz, z, inds, cdf = ops.resample(dev(a["t"].unsqueeze(1)), dev(wfull.unsqueeze(1)), n2, dev(rays),
                                         u=dev(a["u"].unsqueeze(0)), debug=True)
    z_ref, cdf_ref, inds_ref = O.sample_pdf(a["t"], a["w"], a["u"], return_aux=True)
    assert torch.equal(cdf[:, 0].cpu(), cdf_ref)                  # ATen-order sum + fp64 cumsum: bit-equal
    assert torch.equal(inds[:, 0].cpu().long(), inds_ref)
    assert torch.equal(z[:, 0].cpu(), a["z"]) and torch.equal(z_ref, a["z"])    # ... down to the reference's own z
    ref_sorted = torch.sort(torch.cat([a["t"], z[:, 0].cpu()], -1), -1)[0]
    assert torch.equal(tf[:, 0].cpu(), ref_sorted)                # sort/merge: bit-exact
    assert torch.equal(xyz[:, 0].cpu(), ref_sorted.unsqueeze(-1) * rays[:, None, 3:6] + rays[:, None, 0:3])
    # n2 = 0 (C1: no fine samples): the fine list is the coarse list
    tf0, _ = ops.resample(dev(a["t"].unsqueeze(1)), dev(wfull.unsqueeze(1)), 0, dev(rays))
    assert torch.equal(tf0[:, 0].cpu(), a["t"])


def test_resample_random_layers_edits_and_device_rng(ops):
    torch.manual_seed(31)
    n, l, n1, n2 = 400, 3, 64, 48
    t = torch.sort(torch.rand(n, l, n1) * 5.0, -1)[0]
    w = torch.rand(n, l, n1) ** 6
    u = torch.rand(l, n, n2)
    rays = torch.cat([torch.rand(n, 3), torch.nn.functional.normalize(torch.randn(n, 3), dim=-1)], -1)
    edits = [(None, None), ([0.2, 0.0, -0.1], 1.25), (None, 0.8)]
    pivot = torch.tensor([0.3, -0.2, 0.1])
    tf, xyz, z, inds, cdf = ops.resample(dev(t), dev(w), n2, dev(rays), u=dev(u), edits=edits, pivot=pivot, debug=True)
    for i in range(l):
        z_ref, cdf_ref, inds_ref = O.sample_pdf(t[:, i], w[:, i, 1:-1], u[i], return_aux=True)
        assert torch.equal(cdf[:, i].cpu(), cdf_ref)              # bit-exact: cdf, searchsorted indices, new depths
        assert torch.equal(inds[:, i].cpu().long(), inds_ref)
        assert torch.equal(z[:, i].cpu(), z_ref)
        srt = torch.sort(torch.cat([t[:, i], z[:, i].cpu()], -1), -1)[0]
        assert torch.equal(tf[:, i].cpu(), srt)
        p = srt.unsqueeze(-1) * rays[:, None, 3:6] + rays[:, None, 0:3]
        sh, sc = edits[i]
        if sh is not None:
            p = p - torch.tensor(sh)
        if sc is not None:
            p = (p - pivot) / sc + pivot
        assert torch.equal(xyz[:, i].cpu(), p)
    # device RNG: deterministic per (seed, ray index), sorted output, samples follow the pdf
    ta, _ = ops.resample(dev(t), dev(w), n2, dev(rays), seed=99)
    tb, _ = ops.resample(dev(t[100:300]), dev(w[100:300]), n2, dev(rays[100:300]), seed=99, ray_index_base=100)
    assert torch.equal(ta[100:300], tb)
    assert bool((ta[..., 1:] >= ta[..., :-1]).all())


@pytest.mark.parametrize("n1, n2", [(3, 2), (5, 4), (8, 4), (9, 5), (10, 64), (12, 6), (16, 8), (18, 7), (64, 64),
                                    (90, 30), (128, 64), (200, 40), (300, 20), (512, 512), (640, 128)])   # (the last two: > 64 KB of LDS)
def test_resample_bit_exact_vs_oracle_on_random_inputs(ops, n1, n2):
    """utils/sample_pdf.py on the CPU = torch.sum (ATen's fp32 reduction order) + torch.cumsum (fp64 accumulator);
    the kernel reproduces both orders, so cdf, inds and z are bit-equal for every row shape: rows shorter than one
    8-float vector (scalar path), with and without leftover elements, one, two and four 64-lane blocks (the software-pipelined
    kernel) and more than 256 samples (the general one)."""
    torch.manual_seed(1000 + n1)
    n, l = 700, 2
    t = torch.sort(torch.rand(n, l, n1) * 5.0, -1)[0]
    w = torch.rand(n, l, n1) ** 10                                # peaky, as a trained density gives
    w = w / w.sum(-1, keepdim=True) * torch.rand(n, l, 1)
    w[:5] = 0.0                                                   # flat pdf rows
    w[5:10] = 0.0
    w[5:10, :, n1 // 2] = 0.9                                     # one spike: den < 1e-5 bins all around it
    u = torch.rand(l, n, n2)
    u[:, :, 0] = 0.0
    rays = torch.cat([torch.rand(n, 3), torch.nn.functional.normalize(torch.randn(n, 3), dim=-1)], -1)
    tf, _, z, inds, cdf = ops.resample(dev(t), dev(w), n2, dev(rays), u=dev(u), debug=True)
    for i in range(l):
        z_ref, cdf_ref, inds_ref = O.sample_pdf(t[:, i], w[:, i, 1:-1], u[i], return_aux=True)
        assert torch.equal(cdf[:, i].cpu(), cdf_ref), f"cdf differs in {int((cdf[:, i].cpu() != cdf_ref).sum())} places"
        assert torch.equal(inds[:, i].cpu().long(), inds_ref)
        assert torch.equal(z[:, i].cpu(), z_ref)
        assert torch.equal(tf[:, i].cpu(), torch.sort(torch.cat([t[:, i], z_ref], -1), -1)[0])


@pytest.mark.parametrize("n1, n2", [(64, 64), (128, 64), (90, 30), (200, 40), (300, 20), (9, 5)])
def test_resample_production_flavour_is_bit_identical_to_the_checked_one(ops, n1, n2):
    """The production call (device draws, no debug outputs, no edits) runs the kernel's lean flavour and takes the
    shortcut for layers a ray misses; with the debug outputs requested the same draws go through the flavour the oracle
    tests above pin (and no shortcut).  Same seed -> the same depths and points, bit for bit."""
    torch.manual_seed(77 + n1)
    n, l = 1500, 3
    t = torch.sort(torch.rand(n, l, n1) * 5.0 + 0.2, -1)[0]
    w = torch.rand(n, l, n1) ** 8
    miss = torch.rand(n, l) < 0.4
    miss[:, 0] = False
    t[miss] = -1000.0
    w[miss] = 0.0
    t[:40, 0] = t[:40, 0].flip(-1)                                # descending lists (a ray that misses the background box)
    rays = torch.cat([torch.rand(n, 3), torch.nn.functional.normalize(torch.randn(n, 3), dim=-1)], -1)
    tf_a, xyz_a = ops.resample(dev(t), dev(w), n2, dev(rays), seed=1234, ray_index_base=17)
    tf_b, xyz_b, _, _, _ = ops.resample(dev(t), dev(w), n2, dev(rays), seed=1234, ray_index_base=17, debug=True)
    assert torch.equal(tf_a, tf_b) and torch.equal(xyz_a, xyz_b)
    assert bool((tf_a[..., 1:] >= tf_a[..., :-1]).all()) and bool((tf_a[miss.cuda()] == -1000.0).all())


@pytest.mark.parametrize("name", ["sample_pdf_90_30", "sample_pdf_128_64"])
def test_resample_reference_fixtures_at_production_sample_counts(ops, name):
    meta, a = load_golden(name)
    n, n1 = a["t"].shape
    wfull = torch.cat([torch.zeros(n, 1), a["w"], torch.zeros(n, 1)], -1)
    rays = torch.zeros(n, 6)
    _, _, z, _, _ = ops.resample(dev(a["t"].unsqueeze(1)), dev(wfull.unsqueeze(1)), meta["n2"], dev(rays),
                                 u=dev(a["u"].unsqueeze(0)), debug=True)
    assert torch.equal(z[:, 0].cpu(), a["z"])                     # the reference's own output, bit for bit


def test_resample_descending_coarse_list(ops):
    """A ray that misses the background box has start = 0, far = -1000: the coarse depths DEscend.  Only the sorted values
    leave the kernel; they must equal torch.sort(cat[t, z])."""
    torch.manual_seed(37)
    n, n1, n2 = 300, 64, 64
    t = -(torch.arange(n1).float() + torch.rand(n, n1)) * (1000.0 / n1)
    w = torch.zeros(n, n1)
    u = torch.rand(1, n, n2)
    rays = torch.cat([torch.rand(n, 3), torch.nn.functional.normalize(torch.randn(n, 3), dim=-1)], -1)
    tf, _, z, _, _ = ops.resample(dev(t.unsqueeze(1)), dev(w.unsqueeze(1)), n2, dev(rays), u=dev(u), debug=True)
    z_ref = O.sample_pdf(t, w[:, 1:-1], u[0])
    assert torch.equal(z[:, 0].cpu(), z_ref)
    assert torch.equal(tf[:, 0].cpu(), torch.sort(torch.cat([t, z_ref], -1), -1)[0])
    tf2, _ = ops.resample(dev(t.unsqueeze(1)), dev(w.unsqueeze(1)), n2, dev(rays), u=dev(u))   # production path
    assert torch.equal(tf2, tf)


def test_encode_and_gen_weight_op_level(ops):
    _, a = load_golden("encoding")
    for tag, nf in (("pos", 10), ("dir", 4), ("time", 10), ("motion", 10)):
        y = ops.encode(dev(a[f"x_{tag}"]), nf)
        torch.testing.assert_close(y.cpu(), a[f"y_{tag}"], rtol=0, atol=2.5e-7)   # < 1 ulp-of-1 sin/cos
        assert torch.equal(y[:, : a[f"x_{tag}"].shape[1]].cpu(), a[f"x_{tag}"])
    meta, c = load_golden("composite")
    t = c["t"]
    delta = torch.cat([(t[:, 1:] - t[:, :-1]).squeeze(-1), 1e10 * torch.ones(t.shape[0], 1)], -1)
    w = ops.gen_weight(dev(c["sigma"].squeeze(-1)), dev(delta))
    torch.testing.assert_close(w.cpu(), c["gen_weight"], rtol=1e-5, atol=1e-7)
    from stnerf_amd.utils import Trigonometric_kernel
    tk = Trigonometric_kernel(L=10, input_dim=3)
    assert tk.calc_dim(3) == 63 and tk(dev(a["x_pos"])).shape == (16, 63)
