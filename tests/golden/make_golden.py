#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own code (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py        # the base set
    ... make_golden.py --round2 | --model-flags | --more-layers | --teacher | --psnr-view | --grads | --grads --teacher     # one group each;
    every file regenerates byte for byte (--grads: train_c3 / train_coarse_only / train_c4 / train_flags / train_same_spacenet / train_bkgd_time,
    one iteration of the reference's do_train inner loop each, with the thread count pinned; --grads --teacher: train_tf_c3 and
    train_tf_trainer -- 2000 rays, 90 + 30 --, the same iteration with the fine depths and deformed points it ran on recorded)

Imports ``/root/reference`` (a pure-Python/PyTorch repo) on CPU with the three shims of
SURVEY.md section 8(c): a SimpleNamespace cfg (yacs is absent), ``Tensor.cuda`` = identity (no
GPU here; the reference hard-codes ``.cuda()``), and frame-id ray columns.  Every uniform draw
the reference makes (``torch.rand`` at layers/RaySamplePoint.py:98 and utils/sample_pdf.py:31) is
recorded so that the oracle and the HIP kernels can replay it.  Weights come from
``stnerf_amd.synthetic.make_state_dict`` (frozen numpy RandomState stream) and are NOT stored.

The fixtures are the pin for ``oracle/stnerf_oracle.py`` (the reference has no golden vectors of
its own for this path).  ``/root/reference`` does not exist on the GPU box; nothing at test time
reads it -- only these committed ``.npz`` files.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

torch.Tensor.cuda = lambda self, *a, **k: self  # shim 2

import modeling as ref_modeling                      # noqa: E402
import utils as ref_utils                            # noqa: E402
from layers.RaySamplePoint import RaySamplePoint, intersection  # noqa: E402
from layers.render_layer import VolumeRenderer, gen_weight       # noqa: E402
from modeling.spacenet import SpaceNet               # noqa: E402
from modeling.motion_net import MotionNet            # noqa: E402

from stnerf_amd import synthetic as syn              # noqa: E402


class RandRecorder:
    """Wraps torch.rand: records every draw (in call order)."""

    def __init__(self):
        self.draws = []
        self._orig = torch.rand

    def __enter__(self):
        def rec(*a, **k):
            x = self._orig(*a, **k)
            self.draws.append(x.clone())
            return x
        torch.rand = rec
        return self

    def __exit__(self, *exc):
        torch.rand = self._orig


def make_cfg(layer_num, n1, n2, space_time, deform_time, flags=None):
    m = types.SimpleNamespace(BOARDER_WEIGHT=1e10, SAMPLE_METHOD="BBOX", SAME_SPACENET=False,
                              TKERNEL_INC_RAW=True, POSE_REFINEMENT=False, USE_DIR=True,
                              USE_DEFORM_VIEW=False, USE_DEFORM_TIME=deform_time, USE_SPACE_TIME=space_time,
                              BKGD_USE_DEFORM_TIME=False, BKGD_USE_SPACE_TIME=False, DEEP_RGB=False,
                              COARSE_RAY_SAMPLING=n1, FINE_RAY_SAMPLING=n2)
    for k, v in (flags or {}).items():   # SAME_SPACENET / BKGD_USE_DEFORM_TIME / BKGD_USE_SPACE_TIME
        assert hasattr(m, k)
        setattr(m, k, v)
    return types.SimpleNamespace(MODEL=m, DATASETS=types.SimpleNamespace(LAYER_NUM=layer_num))


def save(name, meta, **arrays):
    out = {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"wrote {name}.npz  ({sum(a.nbytes for a in out.values())} B raw)")


def view_rays(h, w, layer_num, orbit=20.0, frame=2.5, per_ray_frames=False, bkgd_frame=1.0):
    K, T = syn.camera(h, w, orbit)
    rays, _ = ref_utils.generate_rays(K, T, None, h, w)
    if per_ray_frames:
        fid = (torch.arange(rays.shape[0]) % 3 + 1).float().reshape(-1, 1)
        return torch.cat([rays, fid], -1)
    return torch.cat([rays, syn.frame_id_columns(rays.shape[0], layer_num, frame, bkgd_frame)], -1)


def flatten_out(out, prefix=""):
    fm, cm, fl, cl, masks = out
    d = {}
    for tag, trip in (("fine_mixed", fm), ("coarse_mixed", cm)):
        for nm, x in zip(("color", "depth", "acc"), trip):
            d[f"{prefix}{tag}_{nm}"] = x
    for tag, lst in (("fine_layer", fl), ("coarse_layer", cl)):
        for i, trip in enumerate(lst):
            for nm, x in zip(("color", "depth", "acc"), trip):
                d[f"{prefix}{tag}{i}_{nm}"] = x
    for i, mk in enumerate(masks):
        d[f"{prefix}mask{i}"] = mk
    return d


# ----------------------------------------------------------------------------- op-level fixtures
def g_generate_rays():
    h, w = 6, 8
    K, T = syn.camera(h, w, 33.0)
    rays, _ = ref_utils.generate_rays(K, T, None, h, w)
    rays2, _ = ref_utils.ray_sampling(K.unsqueeze(0), T.unsqueeze(0), (h, w))
    assert torch.equal(rays, rays2)
    save("generate_rays", dict(h=h, w=w), K=K, T=T, rays=rays)


def g_sampler():
    L, n1 = 2, 8
    rays = view_rays(8, 8, L)
    # edge cases: an axis-parallel ray (zero direction components), a ray starting inside a
    # performer box, a ray pointing away from everything
    extra = torch.tensor([[0.0, 0.0, -4.0, 0.0, 0.0, 1.0], [-0.5, 0.1, 0.0, 0.6, 0.0, 0.8],
                          [0.0, 0.0, -4.0, 0.0, 0.0, -1.0], [5.0, 5.0, 5.0, 1.0, 0.0, 0.0]])
    extra = torch.cat([extra, syn.frame_id_columns(4, L)], -1)
    rays = torch.cat([rays, extra], 0)
    bk, per = syn.scene_boxes(L)
    boxes = torch.cat([bk, per[1]], 0).unsqueeze(0).repeat(rays.shape[0], 1, 1, 1)
    fn = torch.stack([intersection(rays, boxes[:, i]) for i in range(L + 1)], 0)
    torch.manual_seed(1)
    with RandRecorder() as rr:
        t, xyz, mask = RaySamplePoint(n1).forward(rays, boxes)
    save("sampler", dict(L=L, n1=n1), rays=rays, boxes=boxes[0], far_near=fn,
         jitter=torch.stack(rr.draws, 0), t=torch.stack(t, 0), xyz=torch.stack(xyz, 0),
         mask=torch.stack(mask, 0))


def g_encoding():
    torch.manual_seed(2)
    arrs = {}
    for tag, nf, dim in (("pos", 10, 3), ("dir", 4, 3), ("time", 10, 1), ("motion", 10, 4)):
        x = (torch.rand(16, dim) - 0.5) * 6.0
        arrs[f"x_{tag}"] = x
        arrs[f"y_{tag}"] = ref_utils.Trigonometric_kernel(L=nf, input_dim=dim)(x)
    save("encoding", {}, **arrs)


def _strip(sd, prefix):
    return {k[len(prefix) + 1:]: v for k, v in sd.items() if k.startswith(prefix + ".")}


def g_nets():
    torch.manual_seed(3)
    n, s = 6, 5
    pos = (torch.rand(n, s, 3) - 0.5) * 5.0
    d = torch.nn.functional.normalize(torch.rand(n, 3) - 0.5, dim=-1)
    rays = torch.cat([torch.zeros(n, 3), d], -1)
    times = torch.tensor([[1.0], [2.0], [2.5], [3.0], [1.25], [7.0]])
    rs = np.random.RandomState(11)
    sd_t = syn.spacenet_state("net", rs, True)
    sd_n = syn.spacenet_state("net", rs, False)
    sd_m = syn.motionnet_state("net", rs)
    net_t = SpaceNet(use_time=True)
    net_t.load_state_dict(_strip(sd_t, "net"))
    net_n = SpaceNet(use_time=False)
    net_n.load_state_dict(_strip(sd_n, "net"))
    mot = MotionNet(c_input=4, input_time=True)
    mot.load_state_dict(_strip(sd_m, "net"))
    with torch.no_grad():
        rgb_t, sig_t = net_t(pos.clone(), rays, times)
        rgb_n, sig_n = net_n(pos.clone(), rays)
        xt_frac = torch.cat([pos, times.view(n, 1, 1).repeat(1, s, 1)], -1)
        flow_frac = mot(xt_frac)
        xt_int = torch.cat([pos, torch.floor(times).view(n, 1, 1).repeat(1, s, 1)], -1)
        flow_int = mot(xt_int)
    save("nets", dict(weight_seed=11), pos=pos, dirs=d, times=times, rgb_t=rgb_t, sigma_t=sig_t,
         rgb_n=rgb_n, sigma_n=sig_n, flow_frac=flow_frac, flow_int=flow_int)


def g_composite():
    torch.manual_seed(4)
    n, s = 10, 12
    t = torch.sort(torch.rand(n, s) * 6.0, -1)[0].unsqueeze(-1)
    rgb = torch.randn(n, s, 3) * 2.0
    sigma = torch.randn(n, s, 1) * 3.0
    sigma[0] = -1.0                      # all-empty ray
    sigma[1] = 50.0                      # opaque at the first sample
    with torch.no_grad():
        color, depth, acc, w = VolumeRenderer(boarder_weight=1e10)(t, rgb, sigma)
        delta = torch.cat([(t[:, 1:] - t[:, :-1]).squeeze(-1), 1e10 * torch.ones(n, 1)], -1)
        w2 = gen_weight(sigma, delta)
    save("composite", dict(border=1e10), t=t, rgb=rgb, sigma=sigma, color=color, depth=depth, acc=acc,
         weights=w, gen_weight=w2)


def g_sample_pdf():
    torch.manual_seed(5)
    n, n1, n2 = 10, 12, 6
    t = torch.sort(torch.rand(n, n1) * 6.0, -1)[0]
    w = torch.rand(n, n1 - 2) ** 4
    w[0] = 0.0                           # flat pdf (all 1e-5)
    w[1] = 0.0
    w[1, 4] = 1.0                        # one spike: den<1e-5 branches around it
    with RandRecorder() as rr:
        z = ref_utils.sample_pdf(t, w, n2)
        z0 = ref_utils.sample_pdf(t, w, 0)
    u = rr.draws[0]
    save("sample_pdf", dict(n2=n2), t=t, w=w, u=u, z=z, z_empty=z0)


# ----------------------------------------------------------------------------- host path logic
def _import_reference_renderer():
    """render/layered_neural_renderer.py imports yacs, imageio, robopy, the dataset package ... none of which
    exist here and none of which the path/retiming methods touch: stub them and load the module file."""
    import importlib.util
    for name in ("imageio", "robopy", "data"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["data"].make_ray_data_loader_render = None
    sys.modules["data"].get_iteration_path = None
    cfgmod = types.ModuleType("config")
    cfgmod.cfg = None
    sys.modules["config"] = cfgmod
    rpkg = types.ModuleType("render")
    rpkg.__path__ = [os.path.join(REF, "render")]
    sys.modules["render"] = rpkg
    rf = types.ModuleType("render.render_functions")
    rf.__all__ = []
    sys.modules["render.render_functions"] = rf
    ref_utils.add_two_dim_dict = getattr(ref_utils, "add_two_dim_dict", None)
    spec = importlib.util.spec_from_file_location("render.layered_neural_renderer",
                                                  os.path.join(REF, "render", "layered_neural_renderer.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.LayeredNeuralRenderer


def path_scene(C=5):
    """C cameras on an arc looking at the origin, slightly different intrinsics per camera."""
    poses, Ks = [], []
    for i in range(C):
        K, T = syn.camera(54, 96, orbit_deg=-30.0 + 15.0 * i, dist=4.0 + 0.2 * i)
        T[1, 3] = 0.1 * i
        poses.append(T)
        K[0, 0] += 3.0 * i
        K[1, 1] += 3.0 * i
        Ks.append(K)
    return torch.stack(poses, 0), Ks


def g_path():
    LNR = _import_reference_renderer()
    gt_poses, gt_Ks = path_scene()

    def fresh(L=2, frame_num=21, s_shift=None, s_scale=None, s_alpha=None):
        r = object.__new__(LNR)                       # skip __init__ (it loads a dataset + checkpoint)
        r.layer_num, r.frame_num = L, frame_num
        r.display_layers = {i: 1 for i in range(L + 1)}
        r.gt_poses, r.gt_Ks = gt_poses.clone(), [k.clone() for k in gt_Ks]
        r.min_frame = [1 for _ in range(L + 1)]
        r.max_frame = [frame_num for _ in range(L + 1)]
        r.camera_num = gt_poses.shape[0]
        r.min_camera_id, r.max_camera_id = 0, r.camera_num - 1
        r.poses, r.Ks, r.layer_frame_pairs = [], [], []
        r.s_shift, r.s_scale, r.s_alpha = s_shift, s_scale, s_alpha
        r.dataset = types.SimpleNamespace(poses=r.gt_poses, Ks=r.gt_Ks)   # what set_path_gt_poses reads (:173,:198)
        return r

    arrays = {}

    def dump(tag, r):
        arrays[f"{tag}_poses"] = np.stack([np.asarray(p, dtype=np.float64) for p in r.poses], 0)
        arrays[f"{tag}_Ks"] = np.stack([np.asarray(k, dtype=np.float64) for k in r.Ks], 0)
        arrays[f"{tag}_pairs"] = np.array([[list(p) for p in row] for row in r.layer_frame_pairs], dtype=np.float64)

    a = fresh(s_shift=[[[0.0, 0.0, 0.0], [0.1, 0.0, 0.0], [0.0, 0.1, 0.0]],
                       [[0.0, 0.0, 0.0], [0.3, 0.0, 0.1], [0.0, -0.1, 0.0]]],
              s_scale=[[1.0, 1.0, 1.0], [1.0, 1.5, 0.5]], s_alpha=[1.0, 0.2])
    a.set_smooth_path_poses(9, around=True, smooth_time=True)
    dump("around_smooth", a)
    arrays["around_smooth_shift"] = np.array(a.s_shift_frame)
    arrays["around_smooth_scale"] = np.array(a.s_scale_frame)
    arrays["around_smooth_alpha"] = np.array(a.s_alpha_frame)
    b = fresh()
    b.set_smooth_path_poses(7, around=False, smooth_time=False)
    dump("ends_int", b)
    b.retime_by_key_frames(1, [5, 18, 20], [7, 11, 16])          # demo-style retiming of layer 1
    b.retime_by_key_frames(2, [3], [12])
    arrays["ends_int_retimed_pairs"] = np.array([[list(p) for p in row] for row in b.layer_frame_pairs], dtype=np.float64)
    c = fresh()
    c.display_layers[1] = 0                                        # hidden layer: absent from the pairs
    c.set_path_gt_poses()
    c.set_path_fixed_gt_poses(2, num=4)
    arrays["gt_fixed_poses"] = np.stack([np.asarray(p, dtype=np.float64) for p in c.poses], 0)
    arrays["gt_fixed_Ks"] = np.stack([np.asarray(k, dtype=np.float64) for k in c.Ks], 0)
    arrays["gt_fixed_pairs_flat"] = np.array([v for row in c.layer_frame_pairs for p in row for v in p], dtype=np.float64)
    arrays["gt_fixed_pairs_len"] = np.array([len(row) for row in c.layer_frame_pairs])
    save("path", dict(C=int(gt_poses.shape[0]), L=2, frame_num=21), gt_poses=gt_poses, gt_Ks=torch.stack(gt_Ks, 0), **arrays)


# ----------------------------------------------------------------------------- whole-path fixtures
def build_ref_model(L, n1, n2, st, dt, seed, flags=None):
    flags = flags or {}
    model = ref_modeling.build_layered_model(make_cfg(L, n1, n2, st, dt, flags), camera_num=1).eval()
    model.load_state_dict(syn.state_dict_for_flags(L, st, dt, seed, flags))
    bk, per = syn.scene_boxes(L)
    model.set_bkgd_bbox(bk)
    model.set_bboxes(per)
    return model


def g_forward(name, L, n1, n2, st, dt, seed, h, w, frame=2.5, per_ray_frames=False, edit=None,
              call_kwargs=None, chunk=None, only_coarse=False, flags=None, bkgd_frame=1.0, extra_rays=None):
    model = build_ref_model(L, n1, n2, st, dt, seed, flags)
    edit = edit or {}
    for k in ("scale", "shift", "alpha", "near"):
        if k in edit:
            setattr(model, k, edit[k])
    for i in edit.get("hide", []):
        model.hide_layer(i)
    rays = view_rays(h, w, L, frame=frame, per_ray_frames=per_ray_frames, bkgd_frame=bkgd_frame)
    if extra_rays is not None:      # hand-made (origin, direction) rows appended to the view
        ex = torch.tensor(extra_rays, dtype=torch.float32)
        rays = torch.cat([rays, torch.cat([ex, syn.frame_id_columns(ex.shape[0], L, frame, bkgd_frame)], -1)], 0)
    n = rays.shape[0]
    labels, bb, nf = torch.zeros(n), torch.zeros(n, 8, 3), torch.zeros(n, 2)
    kw = dict(call_kwargs or {})
    torch.manual_seed(100 + seed)
    with RandRecorder() as rr, torch.no_grad():
        if chunk is None:
            out = model(rays, labels, bb, only_coarse=only_coarse, near_far=nf, **kw)
        else:
            out = ref_utils.layered_batchify_ray(model, rays, labels, bb, chuncks=chunk, near_far=nf, **kw)
    meta = dict(L=L, n1=n1, n2=n2, space_time=st, deform_time=dt, weight_seed=seed, h=h, w=w,
                edit={k: v for k, v in edit.items()}, call_kwargs=kw, chunk=chunk, only_coarse=only_coarse,
                n_draws=len(rr.draws), flags=dict(flags or {}))
    arrays = flatten_out(out)
    for i, dr in enumerate(rr.draws):
        arrays[f"draw{i}"] = dr
    save(name, meta, rays=rays, **arrays)


# ----------------------------------------------------------------------------- teacher-forced fine stage
def g_teacher(name, L, n1, n2, st, dt, seed, h, w, frame=2.5, call_kwargs=None, edit=None):
    """The reference's OWN intermediates of one forward call (one chunk), so that the fine stage of the HIP path can be fed
    the reference's fine samples instead of its own (a last-ulp cdf difference moves a fine sample through the
    ``den < 1e-5`` switch of utils/sample_pdf.py:59 and, times 2^9 in the encoding, moves a pixel: with the reference's
    samples as the input every ray can be held to the fp32 tolerance).  Recorded, per layer i:

    * ``pdf_t{i}`` / ``pdf_w{i}`` / ``pdf_z{i}``: what modeling/layered_rfrender.py:460 hands to sample_pdf (coarse depths,
      ``weights[..., 1:-1]``) and what it gets back; ``u{i}`` the draw it made;
    * ``z_vals_fine{i}``: the sorted union, the value of ``torch.sort`` at :462;
    * ``xyz_fine_pre{i}``: the fine sample points before deformation -- what the time-deformation net is given at :505
      (masked rays only; the background has no deformation here, so its SpaceNet input at :526-540 is recorded);
    * ``raw_rgb_fine{i}`` / ``raw_sigma_fine{i}``: the fine SpaceNet's outputs as returned (before thresholds / alpha);
    * ``xyz_fine_post{i}`` / ``density_fine_post{i}``: the stashes of :610-611 (after deformation / after the density edits);
    * ``coarse_weights{i}``: the per-layer coarse weights (:437-444), whole rows;
    plus the rays, every draw and all final outputs (same keys as the fwd_* fixtures)."""
    import modeling.layered_rfrender as lr
    model = build_ref_model(L, n1, n2, st, dt, seed)
    edit = edit or {}
    for k in ("scale", "shift", "alpha", "near"):
        if k in edit:
            setattr(model, k, edit[k])
    rays = view_rays(h, w, L, frame=frame)
    n, l, S = rays.shape[0], L + 1, n1 + n2
    labels, bb, nf = torch.zeros(n), torch.zeros(n, 8, 3), torch.zeros(n, 2)
    kw = dict(call_kwargs or {})
    rec = dict(pdf=[], sorts=[], weights=[])
    orig_pdf, orig_sort, orig_vr = lr.sample_pdf, torch.sort, model.volume_render.forward

    def pdf(bins, weights, N_samples, det=False, pytest=False):
        z = orig_pdf(bins, weights, N_samples=N_samples, det=det, pytest=pytest)
        rec["pdf"].append((bins.clone(), weights.clone(), z.clone()))
        return z

    def sort(*a, **k):
        out = orig_sort(*a, **k)
        rec["sorts"].append(out[0].clone())
        return out

    def vr(depth, rgb, sigma, *a, **k):
        out = orig_vr(depth, rgb, sigma, *a, **k)
        rec["weights"].append(out[3].clone())
        return out

    calls = {}

    def hook(tag):
        def fn(module, inputs, output):
            calls.setdefault(tag, []).append(([x.clone() if isinstance(x, torch.Tensor) else x for x in inputs],
                                              tuple(o.clone() for o in output) if isinstance(output, tuple) else output.clone()))
        return fn
    handles = [model.bkgd_spacenet_fine.register_forward_hook(hook("space0"))]
    for i in range(L):
        handles.append(model.spacenets_fine[i].register_forward_hook(hook(f"space{i + 1}")))
        if dt:
            handles.append(model.time_deform_nets[i].register_forward_hook(hook(f"motion{i + 1}")))
    torch.manual_seed(100 + seed)
    lr.sample_pdf, torch.sort, model.volume_render.forward = pdf, sort, vr
    try:
        with RandRecorder() as rr, torch.no_grad():
            out = model(rays, labels, bb, near_far=nf, **kw)
    finally:
        lr.sample_pdf, torch.sort, model.volume_render.forward = orig_pdf, orig_sort, orig_vr
        for hd in handles:
            hd.remove()
    masks = out[4]
    arrays = flatten_out(out)
    for i, dr in enumerate(rr.draws):
        arrays[f"draw{i}"] = dr
    assert len(rec["pdf"]) == l and len(rr.draws) == 2 * l
    fine_sorts = [x for x in rec["sorts"] if x.dim() == 2 and x.shape == (n, S)]
    assert len(fine_sorts) == l, [tuple(x.shape) for x in rec["sorts"]]
    assert len(rec["weights"]) == 2 * (l + 1)            # l per-layer + 1 merged composite per stage
    for i in range(l):
        t_c, w_c, z = rec["pdf"][i]
        arrays[f"pdf_t{i}"], arrays[f"pdf_w{i}"], arrays[f"pdf_z{i}"] = t_c, w_c, z
        arrays[f"u{i}"] = rr.draws[l + i]
        arrays[f"z_vals_fine{i}"] = fine_sorts[i]
        arrays[f"coarse_weights{i}"] = rec["weights"][i].reshape(n, n1)
        assert torch.equal(arrays[f"coarse_weights{i}"][:, 1:-1], w_c)
        arrays[f"xyz_fine_post{i}"] = model.xyz_fine_temp[i]
        arrays[f"density_fine_post{i}"] = model.density_fine_temp[i]
        idx = masks[i] if i > 0 else torch.ones(n, dtype=torch.bool)
        pre = torch.zeros(n, S, 3)
        rgb, sig = torch.zeros(n, S, 3), torch.zeros(n, S, 1)
        if i == 0:
            (inp, outp), = calls["space0"]
            pre, (rgb, sig) = inp[0], outp
        elif bool(idx.any()):
            (inp, outp), = calls[f"space{i}"]
            rgb[idx], sig[idx] = outp
            if dt:
                m_in = calls[f"motion{i}"][-1][0][0]                   # the fine-stage call: (hits, S, 4) = [xyz, frame id]
                assert m_in.shape[1] == S
                pre[idx] = m_in[..., :3]
            else:
                pre[idx] = inp[0]
        arrays[f"xyz_fine_pre{i}"], arrays[f"raw_rgb_fine{i}"], arrays[f"raw_sigma_fine{i}"] = pre, rgb, sig
    meta = dict(L=L, n1=n1, n2=n2, space_time=st, deform_time=dt, weight_seed=seed, h=h, w=w, edit=dict(edit), call_kwargs=kw,
                chunk=None, only_coarse=False, n_draws=len(rr.draws), flags={}, frame=frame)
    save(name, meta, rays=rays, **arrays)


def g_teacher_cases():
    """Same scenes, seeds and sizes as fwd_c3_64_64 / fwd_c3_90_30 / fwd_c4 / fwd_c5 (their final outputs are therefore
    the same numbers), plus the edit case (shift / scale / alpha / near and both thresholds, as fwd_edit)."""
    g_teacher("tf_c3_64_64", 2, 64, 64, True, True, 43, 6, 8)
    g_teacher("tf_c3_90_30", 2, 90, 30, True, True, 41, 8, 8, call_kwargs=dict(density_threshold=0.05, bkgd_density_threshold=0.02))
    g_teacher("tf_c4", 4, 10, 6, False, True, 35, 6, 12)
    g_teacher("tf_c5", 8, 12, 4, False, True, 36, 4, 16, call_kwargs=dict(density_threshold=0.05))
    g_teacher("tf_edit", 2, 12, 6, True, True, 23, 8, 8, frame=2.0,
              edit=dict(scale=[1.0, 1.2, 0.8], shift=[None, None, [-0.1, 0.02, 0.05]], alpha=0.5, near=1.5),
              call_kwargs=dict(density_threshold=0.2, bkgd_density_threshold=0.1))



def g_more_layers():
    """BASELINE C4 / C5 shapes at fixture size: 4 performers with deformation only (configs/config_walking.yml has
    USE_SPACE_TIME off), 8 performers; rays hit 0-2 of the side-by-side performer slabs."""
    g_forward("fwd_c4", 4, 10, 6, False, True, 35, 6, 12)
    g_forward("fwd_c5", 8, 12, 4, False, True, 36, 4, 16, call_kwargs=dict(density_threshold=0.05))


def g_model_flag_cases():
    """The model flags both shipped ymls leave off."""
    # background deformation net (MotionNet(input_time=False), with a fractional background frame id so that the
    # plain-time encoding differs from the lerp) + background space-time
    g_forward("fwd_bkgd_time", 2, 12, 6, True, True, 29, 8, 8, bkgd_frame=1.25,
              flags=dict(BKGD_USE_DEFORM_TIME=True, BKGD_USE_SPACE_TIME=True))
    # ... and the same flags on training-style rays (one integer frame id per RAY, mixed over the batch): the background call sites
    # pass the ids 1-D, SpaceNet.forward tiles them over the samples (modeling/spacenet.py:117-118) -- pins the oracle's restatement
    g_forward("fwd_bkgd_time_mixed_ids", 1, 8, 4, True, True, 36, 6, 8, per_ray_frames=True,
              flags=dict(BKGD_USE_SPACE_TIME=True))
    # fine performer nets shared with the coarse ones
    g_forward("fwd_same_spacenet", 2, 12, 6, True, True, 32, 6, 8, flags=dict(SAME_SPACENET=True))
    # config/defaults.py:39 has DEEP_RGB = True: any USE_SPACE_TIME config that does not switch it off gets the
    # 4-layer colour head in every SpaceNet, background included (layered_rfrender.py:35,62)
    g_forward("fwd_deep_rgb", 2, 12, 6, True, True, 33, 6, 8, flags=dict(DEEP_RGB=True))
    # encodings without their raw-input block and no view dependence (TKERNEL_INC_RAW / USE_DIR off)
    g_forward("fwd_no_raw_no_dir", 2, 12, 6, True, True, 34, 6, 8, flags=dict(TKERNEL_INC_RAW=False, USE_DIR=False))


def g_round2():
    """Round-2 additions (VERDICT r01 items 2a / ADVICE): the shipped yml's sample counts, full 64-lane blocks,
    rays that graze / miss the background box, a wider sample_pdf case, the reference's checkpoint key names."""
    # configs/config_taekwondo.yml: 90 coarse + 30 fine = a ragged second 64-lane block in every scan, and n2 = 30
    # (padded bitonic sort); thresholds as the demos pass them (demo/taekwondo_demo.py: density_threshold, bkgd 0.8 ...)
    g_forward("fwd_c3_90_30", 2, 90, 30, True, True, 41, 8, 8, chunk=3584,
              call_kwargs=dict(density_threshold=0.05, bkgd_density_threshold=0.02))
    g_forward("fwd_c3_90_30_chunked", 2, 90, 30, True, True, 42, 8, 9, chunk=40,
              call_kwargs=dict(density_threshold=0.05, bkgd_density_threshold=0.02))
    # the metric's 64 + 64 (two full 64-lane blocks)
    g_forward("fwd_c3_64_64", 2, 64, 64, True, True, 43, 6, 8)
    # background-box corner cases (layers/RaySamplePoint.py:8-107 with layer 0): a ray through the box EDGE
    # x=-3,z=3 (both slab hits at t=1 exactly: start == end, bin width 0, ray_mask[0] False, yet the reference still
    # evaluates and composites the background, layered_rfrender.py:382-392); the same through a CORNER; a ray that
    # misses the background box altogether (far = -1000, start clamped to 0: descending depths, mask True); a ray
    # starting inside a performer box; an axis-parallel ray along a face of the background box
    g_forward("fwd_grazing", 2, 12, 6, True, True, 44, 4, 6,
              extra_rays=[[-4.0, 0.0, 2.0, 1.0, 0.0, 1.0], [-4.0, -4.0, 2.0, 1.0, 1.0, 1.0],
                          [5.0, 5.0, 5.0, 1.0, 0.0, 0.0], [0.0, 9.0, 0.0, 0.0, 1.0, 0.0],
                          [-0.5, 0.1, 0.0, 0.6, 0.0, 0.8], [-5.0, 0.0, 3.0, 1.0, 0.0, 0.0]])
    # sample_pdf at the yml's counts on peaky weights (what a trained density gives): pins the cdf bit pattern
    torch.manual_seed(6)
    n, n1, n2 = 96, 90, 30
    t = torch.sort(torch.rand(n, n1) * 6.0, -1)[0]
    w = torch.rand(n, n1 - 2) ** 12
    w = w / w.sum(-1, keepdim=True) * torch.rand(n, 1)
    w[0] = 0.0
    w[1] = 0.0
    w[1, 40] = 0.97
    with RandRecorder() as rr:
        z = ref_utils.sample_pdf(t, w, n2)
    save("sample_pdf_90_30", dict(n2=n2), t=t, w=w, u=rr.draws[0], z=z)
    n, n1, n2 = 64, 128, 64
    t = torch.sort(torch.rand(n, n1) * 6.0, -1)[0]
    w = torch.rand(n, n1 - 2) ** 12
    w = w / w.sum(-1, keepdim=True) * torch.rand(n, 1)
    with RandRecorder() as rr:
        z = ref_utils.sample_pdf(t, w, n2)
    save("sample_pdf_128_64", dict(n2=n2), t=t, w=w, u=rr.draws[0], z=z)
    # checkpoint key names and shapes of the reference model for the shipped configurations
    # (render/layered_neural_renderer.py:110-117 loads dict_0['model'] into exactly these)
    keys = {}
    for tag, (L, st, dt, flags) in dict(taekwondo=(2, True, True, {}), walking=(2, False, True, {}),
                                        deep=(2, True, True, dict(DEEP_RGB=True)),
                                        bkgd_time=(1, True, True, dict(BKGD_USE_DEFORM_TIME=True, BKGD_USE_SPACE_TIME=True)),
                                        same=(2, True, True, dict(SAME_SPACENET=True))).items():
        m = ref_modeling.build_layered_model(make_cfg(L, 8, 4, st, dt, flags), camera_num=1)
        keys[tag] = dict(L=L, space_time=st, deform_time=dt, flags=flags,
                         state_dict={k: list(v.shape) for k, v in m.state_dict().items()})
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)
    print("wrote state_dict_keys.json")


def g_psnr_view():
    """PSNR parity of a renderer that draws its own random numbers (last row): the reference rendered twice
    with two torch seeds on a 128 x 128 view of the benchmark scene (bench.py: L=2, 64+64, space-time + deform,
    make_state_dict seed 0, batchify defaults).  PSNR(B, A) is the reference's own run-to-run spread; a device-RNG
    render must land within 1 dB of it against A."""
    L, n1, n2, H, W = 2, 64, 64, 128, 128
    model = build_ref_model(L, n1, n2, True, True, 0)
    rays = view_rays(H, W, L, orbit=10.0)
    n = rays.shape[0]
    imgs = {}
    for tag, seed in (("a", 1), ("b", 2)):
        torch.manual_seed(seed)
        with torch.no_grad():
            out = ref_utils.layered_batchify_ray(model, rays, torch.zeros(n), torch.zeros(n, 8, 3), chuncks=3584,
                                                 near_far=torch.zeros(n, 2))
        imgs[f"color_{tag}"] = out[0][0].half()       # fp16 storage: 6e-4 abs, far below the ~35 dB sampling noise
        imgs[f"acc_{tag}"] = out[0][2].half()
    save("psnr_view", dict(L=L, n1=n1, n2=n2, h=H, w=W, orbit=10.0, weight_seed=0, frame=2.5, seeds=[1, 2]), **imgs)


# ----------------------------------------------------------------------------- training step
GRAD_SAMPLES = 512


def grad_digest(name, g):
    """What a fixture keeps of one parameter's gradient (whole SpaceNets are 463 k floats each), as ONE float vector:
    stnerf_amd.synthetic.tensor_digest (the tensor itself when small, else absmax, L2 norm, row sums, column sums and
    GRAD_SAMPLES entries at seeded positions); the test digests its own gradient with the same function."""
    return {name: syn.tensor_digest(name.split("|", 1)[1], g.detach(), GRAD_SAMPLES)}


class SeededDraws:
    """Replaces torch.rand by draws a test can regenerate: call k returns torch.rand(shape, generator=manual_seed(base + k)) -- for the
    fixture at the trainer's batch size, whose 720,000 uniforms would otherwise be most of the file.  Records the shapes."""

    def __init__(self, base):
        self.base, self.shapes, self.draws, self._orig = base, [], [], torch.rand

    def __enter__(self):
        def rec(*a, **k):
            shape = tuple(a[0]) if len(a) == 1 and not isinstance(a[0], int) else tuple(a)
            x = self._orig(shape, generator=torch.Generator().manual_seed(self.base + len(self.shapes)))
            self.shapes.append(list(shape))
            self.draws.append(x)
            return x
        torch.rand = rec
        return self

    def __exit__(self, *exc):
        torch.rand = self._orig


def rays_with_depth_ties(model, rays, seed_base):
    """Rows of `rays` on which two samples of the merged lists have EXACTLY the same depth (about 1 ray in 1000 at 3 x 90 samples).
    modeling/layered_rfrender.py:425,587 merge the layers with torch.sort's default stable=False, and ATen's CPU sort is not stable
    (a probe: 1162 of 2000 rows kept the lower layer first), so on such a ray the reference's own result depends on its sort
    algorithm -- and differs between its CPU and CUDA runs.  The framework merges in the order of a stable sort (what CUDA's radix
    sort gives); a fixture that is held to 2e-5 leaves such rays out."""
    found = []
    orig = torch.sort

    def sort(*a, **k):
        out = orig(*a, **k)
        v = out[0]
        if v.dim() == 3 and v.shape[-1] == 1:
            v = v[..., 0]
        if v.dim() == 2 and v.shape[0] == rays.shape[0] and v.shape[1] > model.coarse_ray_sample + model.fine_ray_sample:   # a merged list
            tie = ((v[:, 1:] == v[:, :-1]) & (v[:, 1:] > -999.0)).any(-1)
            found.append(tie)
        return out
    n = rays.shape[0]
    torch.sort = sort
    try:
        with SeededDraws(seed_base), torch.no_grad():
            model(rays, torch.zeros(n), torch.zeros(n, 8, 3), False, near_far=torch.zeros(n, 2))
    finally:
        torch.sort = orig
    assert len(found) == 2, len(found)
    return (found[0] | found[1]).nonzero().flatten().tolist()


def g_train_step(name, L, n1, n2, st, dt, seed, n_rays=96, only_coarse=False, remove_outliers=True, h=40, w=64, flags=None, teacher=False,
                 seeded_draws=False):
    """One iteration of do_train's inner loop (engine/layered_trainer.py:178-282) with the reference's OWN model, loss
    (layers/loss.py:4) and optimiser (solver/build.py:10-27, Adam as configs/config_taekwondo.yml:3-5), on a batch of
    training-style rays (7 columns: one integer frame id per ray, data/datasets/ray_dataset.py): the loss, every parameter's
    gradient (digested) and the parameters after optimizer.step().  The trainer's surroundings (loaders, tensorboard, checkpoints,
    val_vis) are not run: the lines between `optimizer.zero_grad()` (:187) and `optimizer.step()` (:279) are restated here with
    the same expressions."""
    from layers import make_loss
    from solver import make_optimizer
    # ATen's CPU weight-gradient reductions depend on the thread count in their last bits (2e-8 of a tensor's largest entry between
    # 2 threads and the container's default): pinned, so that the fixture regenerates byte for byte on any host
    torch.set_num_threads(4)
    model = build_ref_model(L, n1, n2, st, dt, seed, flags)
    cfg = types.SimpleNamespace(SOLVER=types.SimpleNamespace(OPTIMIZER_NAME="Adam", BASE_LR=0.0004, WEIGHT_DECAY=0.0))
    loss_fn = make_loss(cfg)
    optimizer = make_optimizer(cfg, model)
    all_rays = view_rays(h, w, L, per_ray_frames=True)
    g = torch.Generator().manual_seed(500 + seed)
    perm = torch.randperm(all_rays.shape[0], generator=g)
    pick = perm[:n_rays].clone()
    rays = all_rays[pick].contiguous()
    replaced = []
    if seeded_draws:
        # the trainer-sized batch: rays with an exact depth tie in a merged list are swapped for the next unused rays of the view
        # (rays_with_depth_ties; the draws are seeded per call, so every probe sees the draws of the recorded run)
        spare = n_rays
        for _ in range(20):
            tied = rays_with_depth_ties(model, rays, 9000 + 100 * seed)
            if not tied:
                break
            for r in tied:
                pick[r] = perm[spare]
                replaced.append((r, int(perm[spare])))
                spare += 1
            rays = all_rays[pick].contiguous()
        else:
            raise RuntimeError("depth ties keep turning up")
        print(f"{name}: {len(replaced)} rays with exact depth ties replaced: {replaced}")
    rgbs = torch.rand(n_rays, 3, generator=g)
    bbox_labels, bboxes, near_far = torch.zeros(n_rays), torch.zeros(n_rays, 8, 3), torch.zeros(n_rays, 2)
    epoch, coarse_stage = (1, 10) if only_coarse else (1, 0)
    torch.manual_seed(600 + seed)
    model.train()                                                       # :186
    optimizer.zero_grad()                                               # :187
    # teacher forcing (round 6): what the fine stage and the performer networks were evaluated ON -- every layer's new fine depths
    # (the value of sample_pdf at modeling/layered_rfrender.py:460) and the deformed points handed to the performer SpaceNets
    # (:355-356 / :509-510 -> :404-411 / :559-566, hit rays only, in ray order)
    import modeling.layered_rfrender as lr
    tf = dict(z=[], xyz={})
    orig_pdf = lr.sample_pdf

    def pdf(bins, weights, N_samples, det=False, pytest=False):
        zz = orig_pdf(bins, weights, N_samples=N_samples, det=det, pytest=pytest)
        tf["z"].append(zz.detach().clone())
        return zz
    handles = []
    if teacher:
        lr.sample_pdf = pdf
        seen = set()
        for i in range(L):
            for net in (model.spacenets[i], model.spacenets_fine[i]):
                if id(net) in seen:
                    continue
                seen.add(id(net))
                handles.append(net.register_forward_pre_hook(
                    lambda mod, inputs, i=i: tf["xyz"].setdefault(i + 1, []).append(inputs[0].detach().clone())))
    try:
        with (SeededDraws(9000 + 100 * seed) if seeded_draws else RandRecorder()) as rr:
            if epoch < coarse_stage:                                        # :199-202
                stage2, stage1, stage2_layer, stage1_layer, ray_mask = model(rays, bbox_labels, bboxes, True, near_far=near_far)
            else:
                stage2, stage1, stage2_layer, stage1_layer, ray_mask = model(rays, bbox_labels, bboxes, False, near_far=near_far)
    finally:
        lr.sample_pdf = orig_pdf
        for hd in handles:
            hd.remove()
    # labels (N,1): 0 = outlier, i = the ray belongs to layer i (the dataset's masks); here: the first performer the ray hits
    labels = torch.zeros(n_rays, 1)
    for i in range(L, 0, -1):
        labels[ray_mask[i]] = float(i)
    labels[torch.rand(n_rays, generator=g) < 0.25] = 0.0
    predict_rgb_0, predict_rgb_1 = stage1[0], stage2[0]                 # :211-212
    loss1 = loss_fn(predict_rgb_0, rgbs)                                # :224-225
    loss2 = loss_fn(predict_rgb_1, rgbs)
    if epoch < 3 and remove_outliers:                                   # :226-272
        outliers_1, outliers_2, inliers_1, inliers_2 = [], [], [], []
        for i in range(len(stage1_layer)):
            if i != 0:
                outliers_1.append(stage1_layer[i][2][labels == 0])
                outliers_2.append(stage2_layer[i][2][labels == 0])
            inliers_1.append(stage1_layer[i][2][labels == i])
            inliers_2.append(stage2_layer[i][2][labels == i])
        if outliers_1 != []:
            outliers_1 = torch.cat(outliers_1, 0)
            outliers_2 = torch.cat(outliers_2, 0)
        inliers_1 = torch.cat(inliers_1, 0)
        inliers_2 = torch.cat(inliers_2, 0)
        scalar, penalty = 100000, 1
        if outliers_1 != []:
            loss_mask_0 = torch.sum(torch.abs(outliers_1)) * penalty + torch.sum(torch.abs(1 - inliers_1))
            loss_mask_1 = torch.sum(torch.abs(outliers_2)) * penalty + torch.sum(torch.abs(1 - inliers_2))
        else:
            loss_mask_0 = torch.sum(torch.abs(1 - inliers_1))
            loss_mask_1 = torch.sum(torch.abs(1 - inliers_2))
        loss_mask_0 = loss_mask_0 / scalar if loss_mask_0 > rays.shape[0] * 0.0005 and remove_outliers else torch.Tensor([0])
        loss_mask_1 = loss_mask_1 / scalar if loss_mask_1 > rays.shape[0] * 0.0005 and remove_outliers else torch.Tensor([0])
    else:
        loss_mask_0, loss_mask_1 = torch.Tensor([0]), torch.Tensor([0])
    if epoch < coarse_stage:                                            # :274-277
        loss = loss1 + loss_mask_0
    else:
        loss = loss1 + loss2 + loss_mask_0 + loss_mask_1
    loss.backward()                                                     # :281
    arrays = dict(rays=rays, rgbs=rgbs, labels=labels, loss=loss.detach().reshape(1), loss1=loss1.detach().reshape(1),
                  loss2=loss2.detach().reshape(1), loss_mask_0=torch.as_tensor(loss_mask_0).detach().reshape(1),
                  loss_mask_1=torch.as_tensor(loss_mask_1).detach().reshape(1))
    arrays.update(flatten_out(tuple(tuple(x.detach() for x in t) if isinstance(t, tuple) else
                                    [tuple(x.detach() for x in tr) if isinstance(tr, tuple) else tr for tr in t]
                                    for t in (stage2, stage1, stage2_layer, stage1_layer, ray_mask))))
    without_grad = []
    for pname, prm in model.named_parameters():
        if prm.grad is None:
            without_grad.append(pname)
            continue
        arrays.update(grad_digest("grad|" + pname, prm.grad))
    optimizer.step()                                                    # :283
    for pname, prm in model.named_parameters():
        if prm.grad is not None:
            arrays.update(grad_digest("stepped|" + pname, prm.detach()))
    if not seeded_draws:
        for i, dr in enumerate(rr.draws):
            arrays[f"draw{i}"] = dr
    extra = {}
    if seeded_draws:
        extra.update(draw_seed_base=rr.base, draw_shapes=rr.shapes)
    if teacher:
        assert len(tf["z"]) == L + 1 and not only_coarse
        for i in range(L + 1):
            arrays[f"tf_z{i}"] = tf["z"][i]                              # (n, n2): layer i's new fine depths
        for i in range(1, L + 1):
            calls = tf["xyz"].get(i, [])
            hits = int(ray_mask[i].sum())
            assert len(calls) == (2 if hits else 0), (i, len(calls), hits)
            if hits:                                                     # (hits, n1, 3) and (hits, n1 + n2, 3)
                assert calls[0].shape == (hits, n1, 3) and calls[1].shape == (hits, n1 + n2, 3)
                arrays[f"tf_xyz_c{i}"], arrays[f"tf_xyz_f{i}"] = calls[0], calls[1]
        extra.update(teacher=True)
    meta = dict(L=L, n1=n1, n2=n2, space_time=st, deform_time=dt, weight_seed=seed, n_rays=n_rays, only_coarse=only_coarse,
                remove_outliers=remove_outliers, n_draws=len(rr.draws), without_grad=without_grad, lr=0.0004,
                grad_samples=GRAD_SAMPLES, scalar=100000, penalty=1, **({"flags": dict(flags)} if flags else {}),
                hit_fraction=[float(mk.float().mean()) for mk in ray_mask], **extra)
    save(name, meta, **arrays)


def g_train_cases():
    # C3-shaped (two performers, space-time + deform), both stages, with the outlier / inlier mask losses on the layers' acc maps;
    # S = 80 > 64 samples per layer, 240 in the merged list: the compositor backward's scans cross wave blocks
    g_train_step("train_c3", 2, 40, 40, True, True, 41)
    # the coarse-only epochs (:199-200, :274-275) of a deform-only single performer
    g_train_step("train_coarse_only", 1, 24, 8, False, True, 42, n_rays=64, only_coarse=True)
    # C4-shaped (four performers, deformation nets, SpaceNets WITHOUT the time input: configs/config_walking.yml), both stages: five
    # layers in the merged list, the fine stage's alpha on layer 2, rays that hit 0 .. 3 performers
    g_train_step("train_c4", 4, 24, 16, False, True, 43, n_rays=96)
    # the model flags both shipped ymls leave off, under autograd: the 4-layer colour head (config/defaults.py:39 DEEP_RGB) and a
    # background with its own deformation net (MotionNet(input_time=False)).  (Not BKGD_USE_SPACE_TIME: on a batch with mixed frame
    # ids the reference tiles the background's ids over the samples, see fwd_bkgd_time_mixed_ids)
    g_train_step("train_flags", 1, 16, 8, True, True, 44, n_rays=64, flags=dict(DEEP_RGB=True, BKGD_USE_DEFORM_TIME=True))
    # SAME_SPACENET: the fine stage runs the COARSE performer networks again (layered_rfrender.py:70-71) -- one nn.Module called twice
    # per iteration, its gradients the sum of both calls
    # (weight seed 48: with seed 45 the background density head's gradient -- the SUM of the signed density cotangents over all
    # samples -- cancelled 76-fold, so that fp32 rounding of the terms was 1e-4 of the sum, in the reference's autograd as in any
    # other; the fixtures' bars are relative to a tensor's largest entry and want sums that are not differences: 1 - 8 x here)
    g_train_step("train_same_spacenet", 2, 16, 8, True, True, 48, n_rays=64, flags=dict(SAME_SPACENET=True))
    # BKGD_USE_SPACE_TIME on training rays (one integer frame id per ray, mixed over the batch): the background's ids are tiled over
    # the samples (modeling/spacenet.py:117-118 with the 1-D tensor of layered_rfrender.py:380,385): every sample has its own time
    g_train_step("train_bkgd_time", 1, 16, 8, True, True, 55, n_rays=64, flags=dict(BKGD_USE_SPACE_TIME=True))


def g_train_teacher_cases():
    """--grads --teacher (round 6): the same iteration with what the networks were evaluated ON recorded, so that the HIP step can be
    held to the piecewise bar (2e-5 of a gradient tensor's largest entry) END TO END: with the reference's own fine depths and
    deformed points every SpaceNet sees the reference's inputs bit for bit, and what is compared is arithmetic, not the conditioning of
    sin(2^9 x) behind a MotionNet or the resampler."""
    g_train_step("train_tf_c3", 2, 40, 40, True, True, 53, teacher=True)
    # the trainer's own batch: 2000 rays (IMS_PER_BATCH), 90 + 30 samples, two performers (configs/config_taekwondo.yml:6,52-53)
    g_train_step("train_tf_trainer", 2, 90, 30, True, True, 52, n_rays=2000, teacher=True, seeded_draws=True)



def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--psnr-view":
        g_psnr_view()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--round2":
        g_round2()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--model-flags":   # (re)generate these cases without touching the rest
        g_model_flag_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--more-layers":
        g_more_layers()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--teacher":
        g_teacher_cases()
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--grads" and sys.argv[2] == "--teacher":
        g_train_teacher_cases()
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--grads" and sys.argv[2] == "--bkgd-time":    # (one case, without touching the rest)
        g_train_step("train_bkgd_time", 1, 16, 8, True, True, 55, n_rays=64, flags=dict(BKGD_USE_SPACE_TIME=True))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--grads":
        g_train_cases()
        return
    g_generate_rays()
    g_sampler()
    g_encoding()
    g_nets()
    g_composite()
    g_sample_pdf()
    g_path()
    # C1-shaped: single performer, space-time only, no fine stage
    g_forward("fwd_c1", 1, 8, 0, True, False, 21, 6, 8)
    # C3-shaped: two performers, space-time + deform, fractional (retimed) frame id
    g_forward("fwd_c3", 2, 12, 6, True, True, 22, 8, 8)
    # every per-frame edit knob the renderer pokes (layered_neural_renderer.py:435-440) + thresholds
    g_forward("fwd_edit", 2, 12, 6, True, True, 23, 8, 8, frame=2.0,
              edit=dict(scale=[1.0, 1.2, 0.8], shift=[None, None, [-0.1, 0.02, 0.05]], alpha=0.5, near=1.5),
              call_kwargs=dict(density_threshold=0.2, bkgd_density_threshold=0.1))
    g_forward("fwd_hide", 2, 12, 6, True, True, 24, 8, 8, edit=dict(hide=[1]))
    # non-retiming layout (training-style rays: one frame id per ray), deform only
    g_forward("fwd_nonretime", 1, 10, 5, False, True, 25, 6, 8, per_ray_frames=True)
    g_forward("fwd_only_coarse", 2, 12, 6, True, True, 26, 6, 8, only_coarse=True)
    # chunked call surface (batchify_rays.py:51-140): >1 chunk incl. a ragged tail, and the N<chunk path
    g_forward("batchify_chunked", 2, 12, 6, True, True, 27, 8, 8, chunk=24,
              call_kwargs=dict(density_threshold=0.05, bkgd_density_threshold=0.02))
    g_forward("batchify_small", 2, 12, 6, True, True, 28, 6, 8, chunk=3584,
              call_kwargs=dict(density_threshold=0.05, bkgd_density_threshold=0.02))
    g_model_flag_cases()
    g_more_layers()
    g_round2()
    g_psnr_view()
    g_teacher_cases()
    g_train_cases()
    g_train_teacher_cases()


if __name__ == "__main__":
    main()
