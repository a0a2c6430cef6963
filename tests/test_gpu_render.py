"""End-to-end parity of the drop-in boundary (LayeredRFRender / layered_batchify_ray mirrors running on
the HIP kernels) against fixtures produced by the reference's own forward, with the reference's
torch.rand draws replayed.  Needs an MI355X: `pytest -m gpu`.

Stated fp32 tolerance for composited outputs: |color|, |acc| <= 5e-5 abs, depth <= 5e-4 abs (depths reach
~10 and hit-less layers carry t = -1000 samples), ray masks bit-exact.  Coarse-stage outputs must meet it
on EVERY ray.  Fine-stage outputs must meet it on >= 99 % of the rays (all but 2 in the tiny fixtures) and stay within FINE_CAP on all of
them with PSNR >= 70 dB: the reference's inverse-CDF divides by cdf differences down to 1e-5 (and has a hard
`den < 1e-5 -> 1` switch, utils/sample_pdf.py:59), so a last-ulp difference in a coarse weight can move a
fine sample by ~1e-4, which the 2^9 positional-encoding frequency and a sharp density turn into a visible
per-ray difference.  Any two fp32 evaluations of the reference (e.g. its CPU and GPU ATen backends) differ
the same way; stage-level parity with identical stage inputs is covered in test_gpu_ops.py.
"""
import types

import pytest
import torch

from conftest import load_golden
from stnerf_amd import synthetic as syn

pytestmark = pytest.mark.gpu

FWD_CASES = ["fwd_c1", "fwd_c3", "fwd_edit", "fwd_hide", "fwd_nonretime", "fwd_only_coarse",
             "batchify_chunked", "batchify_small"]
COLOR_ATOL, DEPTH_ATOL = 5e-5, 5e-4
FINE_CAP, FINE_FRACTION, FINE_PSNR = 2e-3, 0.99, 70.0


def make_cfg(layer_num, n1, n2, space_time, deform_time):
    m = types.SimpleNamespace(BOARDER_WEIGHT=1e10, SAMPLE_METHOD="BBOX", SAME_SPACENET=False, TKERNEL_INC_RAW=True,
                              POSE_REFINEMENT=False, USE_DIR=True, USE_DEFORM_VIEW=False, USE_DEFORM_TIME=deform_time,
                              USE_SPACE_TIME=space_time, BKGD_USE_DEFORM_TIME=False, BKGD_USE_SPACE_TIME=False,
                              DEEP_RGB=False, COARSE_RAY_SAMPLING=n1, FINE_RAY_SAMPLING=n2)
    return types.SimpleNamespace(MODEL=m, DATASETS=types.SimpleNamespace(LAYER_NUM=layer_num))


def build_model(meta):
    from stnerf_amd.modeling import build_layered_model
    L = meta["L"]
    model = build_layered_model(make_cfg(L, meta["n1"], meta["n2"], meta["space_time"], meta["deform_time"]), camera_num=1)
    model.load_state_dict(syn.make_state_dict(L, meta["space_time"], meta["deform_time"], meta["weight_seed"]))
    bk, per = syn.scene_boxes(L)
    model.set_bkgd_bbox(bk)
    model.set_bboxes(per)
    e = meta["edit"]
    for k in ("scale", "shift", "alpha", "near"):
        if k in e:
            setattr(model, k, e[k])
    for i in e.get("hide", []):
        model.hide_layer(i)
    return model.cuda().eval()


def assemble_replay(meta, a, n_rays):
    """Recorded draws (per reference chunk: l jitter tensors, then l resampling tensors) -> (l,N,*)."""
    l = meta["L"] + 1
    draws = [a[f"draw{i}"] for i in range(meta["n_draws"])]
    per_chunk = l if meta["only_coarse"] else 2 * l
    chunks = [draws[i:i + per_chunk] for i in range(0, len(draws), per_chunk)]
    jitter = torch.cat([torch.stack(c[:l], 0) for c in chunks], 1)
    rp = {"jitter": jitter.cuda()}
    if not meta["only_coarse"]:
        rp["u"] = torch.cat([torch.stack(c[l:], 0) for c in chunks], 1).cuda()
    assert jitter.shape[1] == n_rays
    return rp


def flatten(out):
    fm, cm, fl, cl, masks = out
    d = {}
    for tag, trip in (("fine_mixed", fm), ("coarse_mixed", cm)):
        for nm, x in zip(("color", "depth", "acc"), trip):
            d[f"{tag}_{nm}"] = x
    for tag, lst in (("fine_layer", fl), ("coarse_layer", cl)):
        for i, trip in enumerate(lst):
            for nm, x in zip(("color", "depth", "acc"), trip):
                d[f"{tag}{i}_{nm}"] = x
    for i, mk in enumerate(masks):
        d[f"mask{i}"] = mk
    return d


@pytest.mark.parametrize("name", FWD_CASES)
def test_forward_matches_reference(name):
    from stnerf_amd.utils import layered_batchify_ray
    meta, a = load_golden(name)
    model = build_model(meta)
    rays = a["rays"].cuda()
    n = rays.shape[0]
    model.replay = assemble_replay(meta, a, n)
    labels, bb, nf = torch.zeros(n).cuda(), torch.zeros(n, 8, 3).cuda(), torch.zeros(n, 2).cuda()
    kw = meta["call_kwargs"]
    with torch.no_grad():
        if meta["chunk"] is None:
            out = model(rays, labels, bb, only_coarse=meta["only_coarse"], near_far=nf, **kw)
        else:
            out = layered_batchify_ray(model, rays, labels, bb, chuncks=meta["chunk"], near_far=nf, **kw)
    got = flatten(out)
    keys = [k for k in a if k != "rays" and not k.startswith("draw")]
    assert set(keys) == set(got)
    worst = {}
    for k in keys:
        g = got[k].cpu()
        if k.startswith("mask"):
            assert g.dtype == torch.bool and torch.equal(g, a[k]), k
            continue
        assert g.shape == a[k].shape, (k, g.shape, a[k].shape)
        per_ray = (g - a[k]).abs().max(-1)[0]
        err = float(per_ray.max())
        worst[k.split("_")[-1]] = max(worst.get(k.split("_")[-1], 0.0), err)
        tol = DEPTH_ATOL if k.endswith("depth") else COLOR_ATOL
        if k.startswith("coarse") or meta["only_coarse"]:
            assert err <= tol, f"{name}/{k}: max abs err {err:.3e} > {tol}"
        else:
            n_out = int((per_ray > tol).sum())
            frac = 1.0 if n_out <= 2 else 1.0 - n_out / per_ray.numel()   # tiny fixtures: allow 2 rays
            mse = float(((g - a[k]) ** 2).mean())
            quality = 200.0 if mse == 0 else -10.0 * torch.log10(torch.tensor(mse)).item()
            assert frac >= FINE_FRACTION and err <= FINE_CAP * (10 if k.endswith("depth") else 1), \
                f"{name}/{k}: {100 * frac:.2f} % of rays within {tol}, max abs err {err:.3e}"
            if not k.endswith("depth"):
                assert quality >= FINE_PSNR, f"{name}/{k}: PSNR {quality:.1f} dB"
    print(f"{name}: max abs err " + ", ".join(f"{k}={v:.2e}" for k, v in worst.items()))


def test_chunking_and_launch_size_do_not_change_the_image():
    """Device RNG is keyed by the global ray index: the render is invariant to max_rays_per_launch."""
    from stnerf_amd.utils import layered_batchify_ray
    meta, a = load_golden("fwd_c3")
    model = build_model(meta)
    K, T = syn.camera(40, 64, 10.0)
    from stnerf_amd import ops
    rays = ops.generate_rays(K, T, 40, 64, frame_ids=[1.0, 2.5, 2.5])
    model.seed = 7
    with torch.no_grad():
        a1 = layered_batchify_ray(model, rays, None, None, chuncks=256)
        model.max_rays_per_launch = 700
        a2 = layered_batchify_ray(model, rays, None, None, chuncks=256)
    assert torch.equal(a1[0][0], a2[0][0]) and torch.equal(a1[0][1], a2[0][1])
    for i in range(3):
        assert torch.equal(a1[2][i][0], a2[2][i][0])


def test_cpu_tensors_are_refused():
    meta, a = load_golden("fwd_c1")
    model = build_model(meta)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(a["rays"], None, None)


def test_sharded_render_is_bitwise_the_single_gpu_render():
    """Ray-tile sharding (stnerf_amd.parallel): shards rendered separately (as ranks would) and
    concatenated == the whole view rendered at once, bit for bit (RNG keyed by global ray index)."""
    from stnerf_amd.parallel import make_row_renderer, shard_range
    meta, _ = load_golden("fwd_c3")
    model = build_model(meta)
    model.seed = 11
    h, w = 48, 80
    K, T = syn.camera(h, w, -12.0)
    render_rows = make_row_renderer(model, K, T, h, w, [1.0, 2.5, 2.5], chuncks=512)
    whole = render_rows(0, h * w)
    for world in (2, 3):
        parts = [render_rows(*(lambda s, e: (s, e - s))(*shard_range(h * w, r, world))) for r in range(world)]
        assert torch.equal(torch.cat(parts, 0), whole)
    assert bool(torch.isfinite(whole).all()) and float(whole[:, 4].max()) > 0.5
