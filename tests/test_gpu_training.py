"""the training path around the networks' backward -- the compositor's backward
(csrc/render_bwd.hip: layers/render_layer.py:8-58, the merge gather of modeling/layered_rfrender.py:425-429 / :587-592) and
``LayeredRFRender.forward`` under autograd (stnerf_amd.modeling.training) -- i.e. one iteration of the reference trainer's inner
loop (engine/layered_trainer.py:186-283) on the MI355X.  Needs a GPU: `pytest -m gpu`.

Pins:
  * the compositor backward against torch.autograd (fp64) through the oracle's composite on random streams with every density edit;
  * a whole training step against tests/golden/train_*.npz: loss, outputs, every parameter's gradient and the parameters after
    one Adam step as the REFERENCE's own model / loss / optimiser produced them (make_golden.py --grads).
"""
import os
import sys
import types

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import stnerf_oracle as O                                               # noqa: E402
from stnerf_amd import synthetic as syn                                             # noqa: E402
from train_step_common import compare_digest, load_fixture, oracle_step, replay_of, trainer_loss  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from stnerf_amd import ops as _ops
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return _ops


# ---------------------------------------------------------------------------------------------------------------------
def _edited_sigma(t, sig, layer, fine, cut_neg, thresholds, scale, near):
    """The reference's in-place density edits on one layer's (n, S) densities (functional form: masked writes -> where)."""
    s = sig
    if (not fine) and cut_neg and layer > 0:
        s = torch.where(t < 0, torch.zeros_like(s), s)                 # :414
    if thresholds[layer] is not None:
        s = torch.where(s < thresholds[layer], torch.zeros_like(s), s)  # :416-418 / :538-547 / :564-566
    s = s * scale[layer]                                               # :575-576
    if (not fine) and layer == 0:
        s = torch.where(t < near, torch.zeros_like(s), s)              # :422
    return s


def _reference_composite(t, raw, mask, fine, cut_neg, thresholds, scale, near, evaluated, border=1e10):
    """fp64 autograd graph of a stage's composites: per-layer (render_layer.py:25-58 on every evaluated layer) and the depth
    merge (:425-448 / :587-606), from raw (n,l,S,4) requiring grad."""
    n, l, S = t.shape
    rgbs, sigs = [], []
    for i in range(l):
        have = (torch.ones(n, dtype=torch.bool) if evaluated[i] == 2 else
                (mask[:, i] != 0) if evaluated[i] == 1 else torch.zeros(n, dtype=torch.bool))
        hv = have.reshape(n, 1, 1).to(raw.dtype)
        rgbs.append(raw[:, i, :, :3] * hv + 0.0)
        s = _edited_sigma(t[:, i], raw[:, i, :, 3], i, fine, cut_neg, thresholds, scale, near) * have.reshape(n, 1).to(raw.dtype)
        sigs.append(s.unsqueeze(-1))
    layer = [O.composite(t[:, i].unsqueeze(-1), rgbs[i], sigs[i], border) for i in range(l)]
    # rgb of a layer without network output is a ZERO tensor in the reference: sigmoid(0) = 0.5, but its sigma = 0 makes it moot
    t_mix, order = torch.sort(t.reshape(n, l * S), dim=-1, stable=True)
    rgb_mix = torch.cat(rgbs, 1).gather(1, order.unsqueeze(-1).repeat(1, 1, 3))
    sig_mix = torch.cat(sigs, 1).gather(1, order.unsqueeze(-1))
    if fine:
        sig_mix = torch.where(t_mix.unsqueeze(-1) < near, torch.zeros_like(sig_mix), sig_mix)     # :605
    mixed = O.composite(t_mix.unsqueeze(-1), rgb_mix, sig_mix, border)
    layer_out = torch.stack([torch.cat([c[0], c[1], c[2]], -1) for c in layer], 1)                 # (n,l,5)
    mixed_out = torch.cat([mixed[0], mixed[1], mixed[2]], -1)
    return layer_out, mixed_out


@pytest.mark.parametrize("l, S, fine", [(3, 80, False), (3, 80, True), (1, 64, False), (2, 33, True), (5, 130, True), (9, 192, False)])
def test_composite_backward_matches_fp64_autograd(ops, l, S, fine):
    g = torch.Generator().manual_seed(7 * l + S + int(fine))
    n = 97
    # ascending depths per layer (some layers start below zero / below `near`), a missed layer on some rays (all -1000)
    t = (torch.rand(n, l, S, generator=g) * 0.9 + 0.05).cumsum(-1) * (6.0 / S) - 0.4
    mask = (torch.rand(n, l, generator=g) < 0.7).to(torch.uint8)
    mask[:, 0] = 1
    t = torch.where((mask == 0).unsqueeze(-1) & (torch.rand(n, l, 1, generator=g) < 0.5), torch.full_like(t, -1000.0), t)
    raw = torch.randn(n, l, S, 4, generator=g)
    raw[..., 3] = raw[..., 3] * 1.5 + 0.3
    near = 0.35
    thresholds = [0.05 if fine else None] + [0.1] * (l - 1)
    scale = [1.0] * l
    if fine and l > 2:
        scale[2] = 0.6
    evaluated = [2] + [1] * (l - 1)
    if l > 3:
        evaluated[3] = 0                                               # a hidden layer
    g_layer, g_mixed = torch.randn(n, l, 5, generator=g), torch.randn(n, 5, generator=g)
    p = ops.composite_params(border=1e10, near=near, fine=fine, cut_negative_t=not fine, thresholds=thresholds, sigma_scale=scale,
                             evaluated=evaluated)
    td, rd, md = t.cuda(), raw.cuda(), mask.cuda()
    layer_out, mixed_out, _, order = ops.composite(td, rd, md, want_weights=True, want_order=True, two_pass=False, params=p)
    r64 = raw.double().requires_grad_(True)
    lo64, mo64 = _reference_composite(t.double(), r64, mask, fine, True, thresholds, scale, near, evaluated)
    assert torch.allclose(layer_out.cpu().double(), lo64.detach(), rtol=1e-4, atol=2e-5)
    assert torch.allclose(mixed_out.cpu().double(), mo64.detach(), rtol=1e-4, atol=2e-5)
    for gl, gm in ((g_layer, g_mixed), (g_layer, None), (None, g_mixed)):
        loss = (0 if gl is None else (lo64 * gl.double()).sum()) + (0 if gm is None else (mo64 * gm.double()).sum())
        (want,) = torch.autograd.grad(loss, r64, retain_graph=True)
        got = ops.composite_bwd(td, rd, md, order if gm is not None else None, p, None if gl is None else gl.cuda(),
                                None if gm is None else gm.cuda()).cpu().double()
        scale_c, scale_s = float(want[..., :3].abs().max()), float(want[..., 3].abs().max())
        assert float((got[..., :3] - want[..., :3]).abs().max()) <= 2e-5 * scale_c
        assert float((got[..., 3] - want[..., 3]).abs().max()) <= 2e-5 * scale_s
        # what the reference's zero tensors / overwritten densities cannot receive
        dead = (mask == 0) & torch.tensor([e != 2 for e in evaluated]).reshape(1, l)
        assert float(got[dead].abs().max() if dead.any() else 0.0) == 0.0


def test_composite_function_is_an_autograd_node(ops):
    from stnerf_amd.modeling.training import CompositeFunction
    g = torch.Generator().manual_seed(3)
    n, l, S = 40, 2, 70
    t = (torch.rand(n, l, S, generator=g).cumsum(-1) * 0.05).cuda()
    raw = torch.randn(n, l, S, 4, generator=g).cuda().requires_grad_(True)
    mask = torch.ones(n, l, dtype=torch.uint8, device="cuda")
    p = ops.composite_params(evaluated=[2, 1])
    layer_out, mixed_out, w = CompositeFunction.apply(t, raw, mask, p)
    assert not w.requires_grad and layer_out.requires_grad and mixed_out.requires_grad
    (mixed_out[:, :3].square().mean() + layer_out[..., 4].abs().sum() * 1e-3).backward()
    assert raw.grad is not None and raw.grad.shape == raw.shape and bool(torch.isfinite(raw.grad).all()) and float(raw.grad.abs().max()) > 0


# ---------------------------------------------------------------------------------------------------------------------
def _model_for(meta, device="cuda"):
    from stnerf_amd.modeling import build_layered_model
    m = types.SimpleNamespace(BOARDER_WEIGHT=1e10, SAMPLE_METHOD="BBOX", SAME_SPACENET=False, TKERNEL_INC_RAW=True,
                              POSE_REFINEMENT=False, USE_DIR=True, USE_DEFORM_VIEW=False, USE_DEFORM_TIME=meta["deform_time"],
                              USE_SPACE_TIME=meta["space_time"], BKGD_USE_DEFORM_TIME=False, BKGD_USE_SPACE_TIME=False, DEEP_RGB=False,
                              COARSE_RAY_SAMPLING=meta["n1"], FINE_RAY_SAMPLING=meta["n2"])
    for k, v in meta.get("flags", {}).items():
        setattr(m, k, v)
    model = build_layered_model(types.SimpleNamespace(MODEL=m, DATASETS=types.SimpleNamespace(LAYER_NUM=meta["L"])), camera_num=1)
    model.load_state_dict(syn.state_dict_for_flags(meta["L"], meta["space_time"], meta["deform_time"], meta["weight_seed"], meta.get("flags", {})))
    bk, per = syn.scene_boxes(meta["L"])
    model.set_bkgd_bbox(bk)
    model.set_bboxes(per)
    return model.to(device)


def training_step(name, device="cuda"):
    """The trainer's iteration (engine/layered_trainer.py:186-283) with this framework's model on the fixture's batch and draws.
    -> (model, out, loss, parts, z, meta) after backward (before the optimiser step)."""
    z, meta = load_fixture(name)
    model = _model_for(meta, device)
    model.train()                                                            # :186
    _, model.replay = replay_of(z, meta, device=device)
    rays = torch.from_numpy(z["rays"]).to(device)
    n = rays.shape[0]
    out = model(rays, torch.zeros(n, device=device), torch.zeros(n, 8, 3, device=device), meta["only_coarse"],
                near_far=torch.zeros(n, 2, device=device))                    # :199-202
    loss, parts = trainer_loss(out, torch.from_numpy(z["rgbs"]).to(device), torch.from_numpy(z["labels"]).to(device),
                               meta["only_coarse"], meta["remove_outliers"])
    loss.backward()                                                          # :281
    return model, out, loss, parts, z, meta


TF_MOTION_FACTOR = 2.5
TRAINER_CLEAN_FACTOR = 2.0   # train_tf_trainer only: a clean network's tensor may be 4e-5 from the fp64 evaluation (measured 3.0e-5)
EVENT_RTOL = 5e-3   # a ReLU'(0) event's reach in a network whose gradient a few samples dominate (train_tf_trainer's docstring)
GRAD_RTOL = 2e-5   # of each gradient tensor's largest entry (the bar of the network-level tests, tests/test_gpu_backward.py)
FP32_NOISE_FACTOR = 3.0
CONDITIONED_RTOL = 1e-3   # networks behind a MotionNet or the resampler: see the test's docstring (the conditioning of sin(2^9 x), not of the kernels)


@pytest.mark.parametrize("name, train_fwd", [("train_tf_c3", ""), ("train_tf_c3", "fp32"), ("train_tf_trainer", "")])
def test_teacher_forced_training_step_meets_the_piecewise_bar_end_to_end(ops, monkeypatch, name, train_fwd):
    """The whole step -- forward, loss, loss.backward() through both stages -- with every network evaluated ON WHAT THE REFERENCE'S WAS:
    make_golden.py --grads --teacher recorded, inside the reference's own do_train iteration, every layer's new fine depths
    (modeling/layered_rfrender.py:460) and the deformed points handed to the performer SpaceNets (:355-356, :509-510); model.replay feeds
    them to the training path (stnerf_amd.modeling.training: "z", "xyz_c", "xyz_f").  What is left to differ is arithmetic, and the bar is
    the piecewise one with NO fp64 escape: every SpaceNet gradient -- background, performers, coarse and fine -- within GRAD_RTOL = 2e-5 of
    its tensor's largest entry against the reference's autograd; the deformation nets, whose cotangent is the performer SpaceNets' d pos
    at those identical points, likewise.  train_tf_trainer is the trainer's own batch (2000 rays, 90 + 30 samples, two performers:
    configs/config_taekwondo.yml:6,52-53).  Gradients are compared as digests (tensors of <= 4096 entries whole; larger ones as absmax,
    norm, row / column sums and 512 seeded entries: stnerf_amd.synthetic.tensor_digest)."""
    # (train_fwd: "" = the fused SpaceNet forward and gradient chain in the model's own arithmetic, split bf16; "fp32": the exact-f32 pair --
    # STNERF_TRAIN_FWD -- at the same bars)
    from stnerf_amd.modeling import autograd as A
    monkeypatch.setattr(A, "TRAIN_FWD", train_fwd)
    model, out, loss, parts, z, meta = training_step(name)
    assert meta["teacher"]
    # forward: every output image of both stages against the reference's, every ray (the fine stage sits on the reference's samples)
    worst = {"color": 0.0, "depth": 0.0, "acc": 0.0}
    images = [("coarse_mixed", out[1]), ("fine_mixed", out[0])]
    for i in range(meta["L"] + 1):
        assert torch.equal(out[4][i].cpu(), torch.from_numpy(z[f"mask{i}"]))
        images += [(f"coarse_layer{i}", out[3][i]), (f"fine_layer{i}", out[2][i])]
    for tag, o in images:
        for j, what in enumerate(("color", "depth", "acc")):
            err = (o[j].detach().cpu() - torch.from_numpy(z[f"{tag}_{what}"])).abs()
            if float(err.max()) > (2e-4 if what == "depth" else 2e-5):
                print(f"  {tag}.{what}: max {float(err.max()):.2e}, {int((err.reshape(err.shape[0], -1).amax(-1) > 2e-5).sum())} rays above 2e-5, first rows {(err.reshape(err.shape[0], -1).amax(-1) > 2e-5).nonzero().flatten()[:6].tolist()}")
            worst[what] = max(worst[what], float(err.max()))
    print(f"{name}: worst |output - reference| over {len(images)} images x {meta['n_rays']} rays: {worst}")
    assert worst["color"] <= 2e-5 and worst["acc"] <= 2e-5 and worst["depth"] <= 2e-4, worst
    assert float(loss.detach()) == pytest.approx(float(z["loss"][0]), rel=2e-6)
    named = dict(model.named_parameters())
    recorded = [k.split("|", 1)[1] for k in z.files if k.startswith("grad|")]
    assert set(recorded) == set(named) and not meta["without_grad"]
    ratios = {}
    for pname in recorded:
        g = named[pname].grad
        assert g is not None and bool(torch.isfinite(g).all()), pname
        ratios[pname] = compare_digest(pname, syn.tensor_digest(pname, g, meta["grad_samples"]), z["grad|" + pname], rel=GRAD_RTOL)
    off = {p: round(r, 2) for p, r in ratios.items() if r > 1.0}
    worst_space = max(r for p, r in ratios.items() if "spacenet" in p)
    worst_motion = max([r for p, r in ratios.items() if "deform" in p] or [0.0])
    print(f"{name}: worst SpaceNet gradient {worst_space * GRAD_RTOL:.2e}, worst deformation-net gradient {worst_motion * GRAD_RTOL:.2e} of the tensor's largest entry")
    # (the deformation nets: TF_MOTION_FACTOR x the bar -- their cotangent is the sum of two performer SpaceNets' d pos, each itself at
    # ~1e-5, through a sin(2^9 x) chain rule; measured 2.8e-5 on train_tf_c3)
    off = {p: r for p, r in off.items() if "spacenet" in p or r > TF_MOTION_FACTOR}
    if name == "train_tf_c3":
        assert off == {}, sorted(off.items(), key=lambda kv: -kv[1])       # no escape of any kind on the small fixture
        return
    # The trainer's batch: 240,000 background samples, ~96,000 rows per deformation net.  At that size an fp32 evaluation carries
    # ReLU'(0) events (a hidden unit within rounding of zero passes its cotangent in one evaluation and not in another) and the noise of
    # fp32 sums over 10^5 terms -- the REFERENCE's included: against an fp64 evaluation of the same graph ON THE SAME POSITIONS (the
    # oracle, teacher forced like the HIP step) the fixture's own bkgd_spacenet.stage1.4 is 1.2e-4 of its largest entry off and its
    # time_deform_nets.0 3.4e-4; every other network is within 2e-5.  A tensor over the bar is therefore re-judged against that fp64
    # evaluation: within TRAINER_CLEAN_FACTOR x the bar of fp64 passes; otherwise its NETWORK must be one where the reference itself is over the
    # bar there (else the HIP gradient is simply wrong);
    # the HIP gradient may then be FP32_NOISE_FACTOR x as far from fp64 as the reference is, or -- every fp32 evaluation has its own
    # events: the HIP run's deformation net 0 has one the reference's has not -- within EVENT_RTOL.  Networks the reference evaluates
    # cleanly get no allowance at all, and there is no "conditioned" 1e-3: the positions are forced.
    sd64, _, loss64, _ = oracle_step(z, meta, torch.float64, sample_dtype=torch.float32, teacher=True)
    assert float(loss64) == pytest.approx(float(z["loss"][0]), rel=1e-6)
    network = lambda pname: pname.rsplit(".", 3)[0]          # "time_deform_nets.0.motion_net.8.weight" -> "time_deform_nets.0"
    d64 = {pname: syn.tensor_digest(pname, sd64[pname].grad.float(), meta["grad_samples"]) for pname in recorded}
    ref_vs_64 = {pname: compare_digest(pname, torch.from_numpy(z["grad|" + pname]), d64[pname], rel=GRAD_RTOL) for pname in recorded}
    noisy = {network(pname) for pname, r in ref_vs_64.items() if r > 1.0}
    print(f"    networks where the reference's own fp32 gradients are over the bar against fp64: {sorted(noisy)}")
    cleared = {}
    for pname, r in sorted(off.items(), key=lambda kv: -kv[1]):
        hip_vs_64 = cleared[pname] = compare_digest(pname, syn.tensor_digest(pname, named[pname].grad, meta["grad_samples"]), d64[pname], rel=GRAD_RTOL)
        print(f"    {pname}: HIP vs reference {r:.1f} x the bar; against fp64 on the same positions: reference {ref_vs_64[pname]:.1f} x, HIP {hip_vs_64:.1f} x")
        if hip_vs_64 <= TRAINER_CLEAN_FACTOR:
            # within (TRAINER_CLEAN_FACTOR x) the bar of the exact evaluation.  Measured: the head layers of spacenets.0 (stage2.4,
            # rgb_net.1: column sums and products over 96,000 rows) 3.0e-5 from fp64 where the reference is 1.5e-7 -- the HIP run's own
            # fp32 noise at this batch size; everything else of the clean networks within 2e-5
            continue
        assert network(pname) in noisy, (pname, r, ref_vs_64[pname], hip_vs_64)
        assert hip_vs_64 <= max(1.0, FP32_NOISE_FACTOR * ref_vs_64[pname], EVENT_RTOL / GRAD_RTOL), (pname, r, ref_vs_64[pname], hip_vs_64)
    assert len(noisy) <= 2, noisy
    # the fine networks, the second performer and its deformation net: over the bar against the reference only where the REFERENCE is that
    # far from fp64 and the HIP gradient is not (round 6: with the split-bf16 forward the HIP step sits on the fp64 evaluation -- measured
    # 0.0 x the bar -- where the reference's spacenets_fine.0.stage2.4.weight is 1.0 x off)
    clean = ("bkgd_spacenet_fine", "spacenets_fine", "spacenets.1", "time_deform_nets.1")
    assert all(cleared[p] <= 1.0 and ref_vs_64[p] >= 0.9 * off[p] for p in off if p.startswith(clean)), \
        sorted((p, off[p], cleared[p], ref_vs_64[p]) for p in off if p.startswith(clean))


@pytest.mark.parametrize("name", ["train_c3", "train_coarse_only", "train_c4", "train_flags", "train_same_spacenet", "train_bkgd_time"])
def test_training_step_matches_the_reference_fixture(ops, name):
    """Bar: every parameter's gradient within GRAD_RTOL of the REFERENCE's fp32 autograd (the fixture).  Where the reference's own
    autograd departs from an fp64 evaluation of the same graph by more than that -- a ReLU'(0) event: train_c3 has ONE hidden unit
    of bkgd_spacenet.stage2.0 whose pre-activation is 4e-9 in fp64 and <= 0 in ATen's fp32 sgemm, which moves that network's
    gradients by up to 3.7 % (tests/test_train_step_cpu.py) -- either side of the event is right: the HIP gradient must then be no
    further from the fp64 evaluation than FP32_NOISE_FACTOR x the reference's own distance from it.  The second kind of fp32 noise in
    this fixture: a performer's points are xyz + MotionNet(xyz, t), and every fine-stage point is o + z d with z resampled from the
    coarse weights -- both in front of a 2^9 positional-encoding frequency, so the last bit of a flow or of a weight (dot products
    summed in another order than ATen's sgemm) moves sin(512 x) by ~5e-5.  The reference's fp32 gradients of those networks are
    2 - 4e-4 (of a tensor's largest entry) away from the fp64 evaluation, and so are the HIP ones -- from it and from each other
    (measured 1e-4 .. 6e-4, moving with every change of a kernel's summation order): for every network but the coarse background one
    (whose points are the sampler's, bit for bit) the bar is CONDITIONED_RTOL.  train_coarse_only has neither effect: there the 2e-5
    bar holds for every tensor (measured 3e-7)."""
    model, out, loss, parts, z, meta = training_step(name)
    # forward: the reference's outputs on the same rays, weights and draws
    assert torch.allclose(out[1][0].detach().cpu(), torch.from_numpy(z["coarse_mixed_color"]), atol=2e-6)
    assert torch.allclose(out[0][0].detach().cpu(), torch.from_numpy(z["fine_mixed_color"]), atol=2e-6)
    for i in range(meta["L"] + 1):
        assert torch.equal(out[4][i].cpu(), torch.from_numpy(z[f"mask{i}"]))
        for j, what in enumerate(("color", "depth", "acc")):
            assert torch.allclose(out[3][i][j].detach().cpu(), torch.from_numpy(z[f"coarse_layer{i}_{what}"]), atol=5e-6), (i, what)
    assert float(loss.detach()) == pytest.approx(float(z["loss"][0]), rel=2e-6)
    for k, v in parts.items():
        assert float(v.detach()) == pytest.approx(float(z[k][0]), rel=1e-5, abs=1e-9), k
    named = dict(model.named_parameters())
    recorded = [k.split("|", 1)[1] for k in z.files if k.startswith("grad|")]
    assert set(recorded) | set(meta["without_grad"]) == set(named)
    digest = lambda p, t: syn.tensor_digest(p, t, meta["grad_samples"])
    vs_ref = {}
    for pname in recorded:
        g = named[pname].grad
        assert g is not None and bool(torch.isfinite(g).all()), pname
        vs_ref[pname] = compare_digest(pname, digest(pname, g), z["grad|" + pname], rel=GRAD_RTOL)
    off = [p for p, r in vs_ref.items() if r > 1.0]
    if off:
        sd64, _, _, _ = oracle_step(z, meta, torch.float64, sample_dtype=torch.float32)
        for pname in off:
            d64 = digest(pname, sd64[pname].grad)
            ref_vs_64 = compare_digest(pname, torch.from_numpy(z["grad|" + pname]), d64, rel=GRAD_RTOL)
            hip_vs_64 = compare_digest(pname, digest(pname, named[pname].grad), d64, rel=GRAD_RTOL)
            # behind a MotionNet and / or the resampler (with BKGD_USE_DEFORM_TIME the coarse background net is, too)
            conditioned = not pname.startswith("bkgd_spacenet.") or bool(meta.get("flags", {}).get("BKGD_USE_DEFORM_TIME"))
            bar = max(1.0, FP32_NOISE_FACTOR * ref_vs_64, CONDITIONED_RTOL / GRAD_RTOL if conditioned else 0.0)
            # (ref_vs_64 is measured ON the reference run's sample positions: for a conditioned network it does not contain the
            # position noise the HIP run has against both -- train_c4's fine networks: reference 6e-6 from fp64, HIP 8e-5 from either)
            print(f"    {name} {pname}: HIP vs reference {vs_ref[pname]:.2f} x the bar; vs fp64: reference {ref_vs_64:.2f} x, HIP {hip_vs_64:.2f} x (bar {bar:.1f})")
            assert (ref_vs_64 > 1.0 or conditioned) and hip_vs_64 <= bar, (pname, vs_ref[pname], hip_vs_64, ref_vs_64)
        assert name in ("train_c3", "train_c4", "train_flags", "train_same_spacenet", "train_bkgd_time"), off      # (train_c4: deformation nets and a fine stage, no ReLU'(0) event)
    for pname in meta["without_grad"]:       # the fine networks of a coarse-only epoch
        assert named[pname].grad is None or float(named[pname].grad.abs().max()) == 0.0, pname
    # optimizer.step() (:283; solver/build.py:18).  Adam's first step moves every entry by lr * sign(g) whatever |g| is: the stepped
    # parameters agree wherever the gradients' signs do -- an entry whose gradient is within rounding of zero may go the other way
    # (2 lr = 8e-4 apart), and entries with |g| <~ 1e-6 move by lr g / (|g| + 1e-8), i.e. with the gradient's last bits: the bar is on
    # the fraction of digest entries that agree to 1e-6 (measured 99.3 %), not on the worst one
    opt = torch.optim.Adam([named[p] for p in recorded], lr=meta["lr"], betas=(0.9, 0.999), weight_decay=0.0)
    opt.step()
    agree = total = 0
    for pname in recorded:
        if named[pname].numel() > 4096:
            got, want = named[pname].detach().reshape(-1)[syn.digest_positions(pname, named[pname].numel(), meta["grad_samples"]).cuda()].cpu(), \
                torch.from_numpy(z["stepped|" + pname])[-min(meta["grad_samples"], named[pname].numel()):]
        else:
            got, want = named[pname].detach().reshape(-1).cpu(), torch.from_numpy(z["stepped|" + pname])
        agree += int(((got - want).abs() <= 1e-6).sum())
        total += got.numel()
    assert agree >= 0.98 * total, (agree, total)


def test_eval_mode_and_no_grad_stay_on_the_inference_path(ops):
    z, meta = load_fixture("train_coarse_only")
    model = _model_for(meta)
    _, model.replay = replay_of(z, meta, device="cuda")
    rays = torch.from_numpy(z["rays"]).cuda()
    model.eval()
    out = model(rays, None, None)                       # autograd enabled, trainable parameters, eval(): no history, nothing saved
    assert not out[0][0].requires_grad
    model.train()
    with torch.no_grad():
        out2 = model(rays, None, None)
    assert not out2[0][0].requires_grad
    out3 = model(rays, None, None)                      # train() + autograd: the training path
    assert out3[0][0].requires_grad and out3[1][0].grad_fn is not None
    # the training forward is the exact-f32 arithmetic; the inference default is bf16x3: same picture to fp32 rounding
    assert torch.allclose(out3[1][0].detach(), out2[1][0], atol=5e-5)
