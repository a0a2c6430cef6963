"""End-to-end parity of the drop-in boundary (LayeredRFRender / layered_batchify_ray mirrors running on
the HIP kernels) against fixtures produced by the reference's own forward, with the reference's
torch.rand draws replayed.  Needs an MI355X: `pytest -m gpu`.

Stated fp32 tolerance for composited outputs: |color|, |acc| <= 5e-5 abs, depth <= 5e-4 abs (depths reach
~10 and hit-less layers carry t = -1000 samples), ray masks bit-exact.  Coarse-stage outputs must meet it
on EVERY ray.

Fine-stage outputs cannot have a fixed per-ray tolerance: the reference's inverse CDF divides by cdf differences
down to 1e-5 (and has a hard `den < 1e-5 -> 1` switch, utils/sample_pdf.py:59), so a last-ulp difference in a coarse
weight can move a fine sample, which the 2^9 positional-encoding frequency and a sharp density turn into a visible
per-ray difference -- for ANY two fp32 evaluations of the reference.  The bar is therefore MEASURED, per case
(`fine_stage_bar`): the same render is also evaluated in fp64 by the oracle (the exact answer), and the HIP output's
distance from it must not exceed the distance of the reference's own fp32 evaluation (the fixture / the fp32
oracle): number of rays above the stated tolerance <= 1.5x the reference's (+2), median and 90th percentile <= 2x.
test_gpu_round2.py::test_fine_stage_error_is_within_the_reference_fp32_spread prints the distributions on 1024
performer rays (reference fp32: ~16 % of such rays are further than 5e-5 from the exact image; the kernels: the same).
Stage-level parity with identical stage inputs (bit-exact where the arithmetic allows) is in test_gpu_ops.py.
"""
import types

import pytest
import torch

from conftest import load_golden
from stnerf_amd import synthetic as syn

pytestmark = pytest.mark.gpu

FWD_CASES = ["fwd_c1", "fwd_c3", "fwd_edit", "fwd_hide", "fwd_nonretime", "fwd_only_coarse",
             "batchify_chunked", "batchify_small", "fwd_bkgd_time", "fwd_bkgd_time_mixed_ids", "fwd_same_spacenet", "fwd_deep_rgb", "fwd_no_raw_no_dir", "fwd_c4", "fwd_c5"]
COLOR_ATOL, DEPTH_ATOL = 5e-5, 5e-4


def fine_stage_bar(got, ref32, exact, tol, what):
    """got (HIP), ref32 (the reference's own fp32 evaluation: fixture or fp32 oracle), exact (fp64 oracle): (n, c) tensors.
    The HIP output must not be further from the exact result than the reference's fp32 evaluation is."""
    err_ref = (ref32.double() - exact.double()).abs().reshape(ref32.shape[0], -1).amax(-1)
    err_hip = (got.double() - exact.double()).abs().reshape(got.shape[0], -1).amax(-1)
    n = err_ref.numel()
    out_ref, out_hip = int((err_ref > tol).sum()), int((err_hip > tol).sum())
    # (round 2 allowed 1.5 x + max(2, n / 200) outliers and 2 x on the quantiles; the kernels measure 162 vs 161 of 1024 and
    # 33 vs 36 -- the bar is now what they need: 1.1 x + 2 rays, median / p90 within 1.25 x)
    assert out_hip <= 1.1 * out_ref + 2, \
        f"{what}: {out_hip} of {n} rays above {tol} vs the exact result, the reference's fp32 evaluation has {out_ref}"
    for q in (0.5, 0.9):
        qh, qr = float(torch.quantile(err_hip, q)), float(torch.quantile(err_ref, q))
        assert qh <= 1.25 * qr + tol / 50, f"{what}: p{int(100 * q)} error {qh:.2e} vs the reference's {qr:.2e}"
    return out_hip, out_ref, float(err_hip.max()), float(err_ref.max())


def oracle_model_from_meta(meta, dtype=torch.float32):
    from oracle import stnerf_oracle as O
    L = meta["L"]
    bk, per = syn.scene_boxes(L)
    fl = meta.get("flags", {})
    sd = syn.state_dict_for_flags(L, meta["space_time"], meta["deform_time"], meta["weight_seed"], fl)
    m = O.OracleModel(layer_num=L, n_coarse=meta["n1"], n_fine=meta["n2"], params={k: v.to(dtype) for k, v in sd.items()},
                      use_deform_time=meta["deform_time"], use_space_time=meta["space_time"],
                      bkgd_use_deform_time=fl.get("BKGD_USE_DEFORM_TIME", False),
                      bkgd_use_space_time=fl.get("BKGD_USE_SPACE_TIME", False), bkgd_bbox=bk.to(dtype), bboxes=per.to(dtype))
    e = meta.get("edit", {})
    m.scale, m.shift = e.get("scale"), e.get("shift")
    m.alpha, m.near = e.get("alpha", 1.0), e.get("near", 0.0)
    m.hidden = set(e.get("hide", []))
    return m


def oracle_render(meta, rays, draws, dtype=torch.float32, chunk=None, only_coarse=False, **kw):
    """The oracle on `rays` with the uniform draws `draws` (the reference's call order), evaluated in `dtype`."""
    from oracle import stnerf_oracle as O
    it = iter(draws)
    m = oracle_model_from_meta(meta, dtype)
    with torch.no_grad():
        if chunk is None:
            return O.render_chunk(m, rays.to(dtype), only_coarse=only_coarse, rand=lambda shape: next(it), **kw)
        return O.layered_batchify_ray(m, rays.to(dtype), chuncks=chunk, rand=lambda shape: next(it), **kw)


_EXACT = {}


def exact_forward(name):
    """fp64 evaluation of a forward fixture (same rays, weights, recorded draws): the exact answer both fp32 evaluations
    (the reference's = the fixture, and the HIP path's) are measured against."""
    if name not in _EXACT:
        meta, a = load_golden(name)
        draws = [a[f"draw{i}"] for i in range(meta["n_draws"])]
        _EXACT[name] = flatten(oracle_render(meta, a["rays"], draws, torch.float64, chunk=meta["chunk"],
                                             only_coarse=meta["only_coarse"], **meta["call_kwargs"]))
    return _EXACT[name]


def make_cfg(layer_num, n1, n2, space_time, deform_time, flags=None):
    m = types.SimpleNamespace(BOARDER_WEIGHT=1e10, SAMPLE_METHOD="BBOX", SAME_SPACENET=False, TKERNEL_INC_RAW=True,
                              POSE_REFINEMENT=False, USE_DIR=True, USE_DEFORM_VIEW=False, USE_DEFORM_TIME=deform_time,
                              USE_SPACE_TIME=space_time, BKGD_USE_DEFORM_TIME=False, BKGD_USE_SPACE_TIME=False,
                              DEEP_RGB=False, COARSE_RAY_SAMPLING=n1, FINE_RAY_SAMPLING=n2)
    for k, v in (flags or {}).items():
        setattr(m, k, v)
    return types.SimpleNamespace(MODEL=m, DATASETS=types.SimpleNamespace(LAYER_NUM=layer_num))


def build_model(meta):
    from stnerf_amd.modeling import build_layered_model
    L = meta["L"]
    fl = meta.get("flags", {})
    model = build_layered_model(make_cfg(L, meta["n1"], meta["n2"], meta["space_time"], meta["deform_time"], fl), camera_num=1)
    model.load_state_dict(syn.state_dict_for_flags(L, meta["space_time"], meta["deform_time"], meta["weight_seed"], fl))
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


@pytest.mark.parametrize("precision", ["bf16x3", "fp32"])     # the library default, and the exact f32 MFMA arithmetic
@pytest.mark.parametrize("name", FWD_CASES)
def test_forward_matches_reference(name, precision):
    run_forward_case(name, precision=precision)


def run_forward_case(name, precision):
    from stnerf_amd.utils import layered_batchify_ray
    meta, a = load_golden(name)
    model = build_model(meta).set_precision(precision)
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
    exact = exact_forward(name)
    worst, bars = {}, []
    for k in keys:
        g = got[k].cpu()
        if k.startswith("mask"):
            assert g.dtype == torch.bool and torch.equal(g, a[k]), k
            continue
        assert g.shape == a[k].shape, (k, g.shape, a[k].shape)
        err = float((g - a[k]).abs().max())
        worst[k.split("_")[-1]] = max(worst.get(k.split("_")[-1], 0.0), err)
        tol = DEPTH_ATOL if k.endswith("depth") else COLOR_ATOL
        if k.startswith("coarse") or meta["only_coarse"]:
            assert err <= tol, f"{name}/{k}: max abs err {err:.3e} > {tol}"          # coarse stage: every ray
        else:
            bars.append(fine_stage_bar(g, a[k], exact[k], tol, f"{name}/{k}"))
    print(f"{name}: max abs err vs the reference " + ", ".join(f"{k}={v:.2e}" for k, v in worst.items())
          + (f"; fine-stage rays above tol vs fp64 (HIP / reference): {sum(b[0] for b in bars)} / {sum(b[1] for b in bars)}" if bars else ""))


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_chunking_and_launch_size_do_not_change_the_image(precision):
    """Device RNG is keyed by the global ray index: the render is invariant to max_rays_per_launch (in both arithmetics: a
    sample's evaluation does not depend on which 128-row work item it lands in)."""
    from stnerf_amd.utils import layered_batchify_ray
    meta, a = load_golden("fwd_c3")
    model = build_model(meta).set_precision(precision)
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


@pytest.mark.parametrize("n", [1, 31, 63])
def test_ragged_ray_counts_match_the_reference_rows(n):
    """Ray counts that fill neither a wave, a 128-row MLP tile nor a chunk: the first n rays of the
    reference fixture rendered alone equal the fixture's first n rows (rays are independent, boxes come
    from row 0 which is kept)."""
    meta, a = load_golden("fwd_c3")
    model = build_model(meta)
    full = assemble_replay(meta, a, a["rays"].shape[0])
    model.replay = {k: v[:, :n].contiguous() for k, v in full.items()}
    with torch.no_grad():
        got = flatten(model(a["rays"][:n].cuda(), None, None, **meta["call_kwargs"]))
    for k, g in got.items():
        ref = a[k][:n]
        if k.startswith("mask"):
            assert torch.equal(g.cpu(), ref), k
        elif k.startswith("coarse"):
            tol = DEPTH_ATOL if k.endswith("depth") else COLOR_ATOL
            assert float((g.cpu() - ref).abs().max()) <= tol, k
        else:
            fine_stage_bar(g.cpu(), ref, exact_forward("fwd_c3")[k][:n], DEPTH_ATOL if k.endswith("depth") else COLOR_ATOL, k)


@pytest.mark.parametrize("n", [65, 130, 257])
def test_ragged_ray_counts_above_a_wave_match_the_oracle(n):
    """n rays that straddle wave (64) and MLP-tile (128 rows = 128/ns rays) boundaries, odd sample counts."""
    from oracle import stnerf_oracle as O
    meta = dict(L=2, n1=13, n2=7, space_time=True, deform_time=True, weight_seed=41, edit={}, H=36, W=64)
    model = build_model(meta)
    sd = syn.make_state_dict(2, True, True, 41)
    K, T = syn.camera(36, 64, 12.0)
    g = torch.Generator().manual_seed(n)
    full = O.generate_rays(K, T, 36, 64)
    pick = torch.randperm(full.shape[0], generator=g)[:n].sort()[0]
    rays = torch.cat([full[pick], syn.frame_id_columns(n, 2)], -1)
    jitter, u = torch.rand(3, n, 13, generator=g), torch.rand(3, n, 7, generator=g)
    model.replay = {"jitter": jitter.cuda(), "u": u.cuda()}
    with torch.no_grad():
        out = model(rays.cuda(), None, None)
    ref = oracle_render(meta, rays, list(jitter) + list(u))
    ref64 = oracle_render(meta, rays, list(jitter) + list(u), torch.float64)
    for i in range(3):
        assert torch.equal(out[4][i].cpu(), ref[4][i])
        assert float((out[3][i][0].cpu() - ref[3][i][0]).abs().max()) <= COLOR_ATOL
    assert float((out[1][0].cpu() - ref[1][0]).abs().max()) <= COLOR_ATOL
    fine_stage_bar(out[0][0].cpu(), ref[0][0], ref64[0][0], COLOR_ATOL, f"fine mixed colour, {n} rays")


def test_empty_ray_batch_raises_like_the_reference():
    meta, a = load_golden("fwd_c3")
    model = build_model(meta)
    with pytest.raises(IndexError):
        model(a["rays"][:0].cuda(), None, None)


def test_background_space_time_with_mixed_frame_ids_is_per_sample_time():
    """BKGD_USE_SPACE_TIME (off in both shipped ymls): the reference tiles the background's frame ids over the samples when they
    differ across the rays of a call (modeling/spacenet.py:117-118 with the 1-D ids of layered_rfrender.py:380): sample j of ray i is
    evaluated at the id of ray (i ns + j) mod n.  fwd_bkgd_time_mixed_ids (FWD_CASES, both arithmetics) holds the outputs; here: the
    scramble is really there (the call differs from one with a ray's own id on every sample), and a call with ONE id stays on the fused
    pipeline and agrees with the per-sample path fed the same ids."""
    meta, a = load_golden("fwd_bkgd_time_mixed_ids")
    model = build_model(meta).set_precision("fp32")
    rays = a["rays"].cuda()
    n = rays.shape[0]
    assert rays.shape[1] == 7 and len(set(rays[:, 6].tolist())) > 1
    model.replay = assemble_replay(meta, a, n)
    with torch.no_grad():
        mixed = model(rays, None, None)
        same = rays.clone()
        same[:, 6] = rays[0, 6]
        one = model(same, None, None)                      # one id for the whole call: the fused pipeline
        from stnerf_amd.modeling import training as T
        fused, T.mixed_bkgd_ids = one, (lambda r: True)    # ... and the same call forced through the per-sample path
        try:
            per_sample = model(same, None, None)
        finally:
            T.mixed_bkgd_ids = lambda r: bool((r[:, 6] != r[0, 6]).any())
    assert float((mixed[0][0] - one[0][0]).abs().max()) > 1e-3
    assert torch.allclose(per_sample[1][0], fused[1][0], atol=COLOR_ATOL) and torch.allclose(per_sample[3][0][0], fused[3][0][0], atol=COLOR_ATOL)


def test_cpu_tensors_are_refused():
    meta, a = load_golden("fwd_c1")
    model = build_model(meta)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(a["rays"], None, None)


def test_sharded_render_is_bitwise_the_single_gpu_render():
    """Ray-tile sharding (stnerf_amd.parallel): shards rendered separately (as ranks would) and
    concatenated == the whole view rendered at once, bit for bit (RNG keyed by global ray index)."""
    from stnerf_amd.parallel import make_row_renderer, shard_range, stripe_spans
    meta, _ = load_golden("fwd_c3")
    model = build_model(meta)
    model.seed = 11
    h, w = 48, 80
    K, T = syn.camera(h, w, -12.0)
    render_rows = make_row_renderer(model, K, T, h, w, [1.0, 2.5, 2.5], density_threshold=0.05, chuncks=512)
    whole = render_rows(0, h * w)
    for world in (2, 3):
        parts = [render_rows(*(lambda s, e: (s, e - s))(*shard_range(h * w, r, world))) for r in range(world)]
        assert torch.equal(torch.cat(parts, 0), whole)
    # interleaved stripes of 2 image rows (160 rays < one 512-ray chunk: the thresholds must still apply)
    for world in (2, 5):
        out = torch.empty_like(whole)
        for r in range(world):
            for s, e in stripe_spans(h * w, 2 * w, r, world):
                out[s:e] = render_rows(s, e - s)
        assert torch.equal(out, whole)
    assert bool(torch.isfinite(whole).all()) and float(whole[:, 4].max()) > 0.5


# ----------------------------------------------------------------------------------------------------
# BASELINE.json configs at (or sampled from) their full sizes
# ----------------------------------------------------------------------------------------------------
def _oracle_model(meta, sd):
    from oracle import stnerf_oracle as O
    bk, per = syn.scene_boxes(meta["L"])
    return O.OracleModel(layer_num=meta["L"], n_coarse=meta["n1"], n_fine=meta["n2"], params=sd,
                         use_deform_time=meta["deform_time"], use_space_time=meta["space_time"], bkgd_bbox=bk,
                         bboxes=per)


@pytest.mark.parametrize("cfg_name, meta", [
    ("C2: 512x512, L=1, 64+64", dict(L=1, n1=64, n2=64, space_time=True, deform_time=False, weight_seed=40, edit={}, H=512, W=512)),
    ("C3: 1080p, L=2, 64+64", dict(L=2, n1=64, n2=64, space_time=True, deform_time=True, weight_seed=41, edit={}, H=1080, W=1920)),
    ("C4: 1080p, L=4, 64+64, deformation only", dict(L=4, n1=64, n2=64, space_time=False, deform_time=True, weight_seed=42, edit={}, H=1080, W=1920)),
])
def test_full_sample_counts_on_a_ray_subset_match_the_oracle(cfg_name, meta):
    """The real sample counts of C2 / C3 on 768 rays spread over the view (what the oracle finishes in seconds)."""
    from oracle import stnerf_oracle as O
    model = build_model(meta)
    sd = syn.make_state_dict(meta["L"], meta["space_time"], meta["deform_time"], meta["weight_seed"])
    K, T = syn.camera(meta["H"], meta["W"], 8.0)
    full = O.generate_rays(K, T, meta["H"], meta["W"])
    g = torch.Generator().manual_seed(5)
    pick = torch.randperm(full.shape[0], generator=g)[:768].sort()[0]
    l = meta["L"] + 1
    rays = torch.cat([full[pick], syn.frame_id_columns(768, meta["L"])], -1)
    jitter, u = torch.rand(l, 768, meta["n1"], generator=g), torch.rand(l, 768, meta["n2"], generator=g)
    model.replay = {"jitter": jitter.cuda(), "u": u.cuda()}
    with torch.no_grad():
        out = model(rays.cuda(), None, None)
    ref = oracle_render(meta, rays, list(jitter) + list(u))
    ref64 = oracle_render(meta, rays, list(jitter) + list(u), torch.float64)
    for i in range(l):
        assert torch.equal(out[4][i].cpu(), ref[4][i])
    assert float((out[1][0].cpu() - ref[1][0]).abs().max()) <= COLOR_ATOL            # coarse: every ray
    stats = fine_stage_bar(out[0][0].cpu(), ref[0][0], ref64[0][0], COLOR_ATOL, cfg_name)
    print(f"{cfg_name}: fine-stage rays above {COLOR_ATOL} vs fp64: HIP {stats[0]}, reference fp32 {stats[1]} of 768")


def test_c2_full_view_properties_and_determinism():
    """C2 at its full size (262,144 rays, device RNG): size-independent properties."""
    from stnerf_amd.render.render_pose import render_pose, to_uint8
    meta = dict(L=1, n1=64, n2=64, space_time=True, deform_time=False, weight_seed=40, edit={})
    model = build_model(meta)
    K, T = syn.camera(512, 512, 8.0)
    model.seed = 3
    c1, d1, cl1, dl1 = render_pose(model, T, K, 512, 512, [(0, 1), (1, 2.5)], far=20.0)
    c2, d2, _, _ = render_pose(model, T, K, 512, 512, [(0, 1), (1, 2.5)], far=20.0)
    assert torch.equal(c1, c2) and torch.equal(d1, d2)                       # same inputs twice -> bitwise equal
    assert c1.shape == (512, 512, 3) and len(cl1) == 2 and dl1[1].shape == (512, 512, 1)
    assert bool(torch.isfinite(c1).all()) and float(c1.min()) >= 0.0 and float(c1.max()) <= 1.0 + 1e-5
    assert float(d1.min()) >= 0.0
    img = to_uint8(c1)
    assert img.dtype == torch.uint8 and img.shape == (512, 512, 3)
    model.seed = 4                                                           # another noise stream: close, not equal
    c3, _, _, _ = render_pose(model, T, K, 512, 512, [(0, 1), (1, 2.5)], far=20.0)
    assert not torch.equal(c1, c3)
    assert float(-10 * torch.log10(torch.mean((c1 - c3) ** 2))) > 15.0


def test_stage_invariants_at_c3_sample_counts():
    """Ordering / range invariants of every stage on 20k rays of the 1080p C3 view."""
    from stnerf_amd import ops
    meta = dict(L=2, n1=64, n2=64, space_time=True, deform_time=True, weight_seed=41, edit={})
    K, T = syn.camera(1080, 1920, 12.0)
    rays = ops.generate_rays(K, T, 1080, 1920, frame_ids=[1.0, 2.5, 2.5], first_ray=1000000, n=20000)
    bk, per = syn.scene_boxes(2)
    boxes = torch.cat([bk, per[1]], 0).cuda()
    t, xyz, mask = ops.sample_coarse(rays, boxes, 64, seed=9, ray_index_base=1000000)
    hit = mask.bool()
    assert bool(hit[:, 0].all()) and 0.05 < float(hit[:, 1].float().mean()) < 0.95
    assert bool((t[..., 1:] >= t[..., :-1])[hit].all())                      # ascending inside a hit layer
    miss = t[~hit]                                                           # no usable interval: all samples coincide
    assert bool(((miss.max(-1)[0] - miss.min(-1)[0]).abs() <= 1e-2).all()) and bool((miss[:, 0] == -1000.0).any())
    raw = torch.randn(20000, 3, 64, 4, device="cuda")
    lo, mix, w, order = ops.composite(t, raw, mask, cut_negative_t=True, want_weights=True, want_order=True)
    assert bool((w >= 0).all()) and bool((w.sum(-1) <= 1 + 1e-4).all())
    assert bool((mix[:, 4] <= 1 + 1e-4).all()) and bool((lo[..., 4] <= 1 + 1e-4).all())
    srt = torch.gather(t.reshape(20000, -1), 1, order.long())
    assert bool((srt[:, 1:] >= srt[:, :-1]).all())                           # merged order is a sort
    assert torch.equal(order.long().sort(-1)[0], torch.arange(192, device="cuda").expand(20000, 192))  # a permutation
    tf, xf = ops.resample(t, w, 64, rays, seed=9, ray_index_base=1000000)
    assert bool((tf[..., 1:] >= tf[..., :-1]).all())
    lo_t, hi_t = t.min(-1)[0], t.max(-1)[0]
    assert bool((tf.min(-1)[0] >= lo_t - 1e-4).all()) and bool((tf.max(-1)[0] <= hi_t + 1e-4).all())


def test_c5_shape_eight_performers_192_samples():
    """C5's per-ray shape (8 performer layers + background, 128 coarse + 64 fine samples = 1728 merged samples
    per ray) on a small ray window: every kernel handles l = 9, S = 192, and the result matches the oracle."""
    from oracle import stnerf_oracle as O
    meta = dict(L=8, n1=128, n2=64, space_time=False, deform_time=True, weight_seed=45, edit={})
    model = build_model(meta)
    sd = syn.make_state_dict(8, False, True, 45)
    K, T = syn.camera(2160, 3840, 5.0)
    full_rows = O.generate_rays(K.clone(), T, 2, 3840)          # two image rows of the 4K view
    g = torch.Generator().manual_seed(9)
    pick = torch.randperm(full_rows.shape[0], generator=g)[:96].sort()[0]
    Kc = K.clone()
    Kc[1, 2] -= 1079                                             # rows 1079..1080 (image centre)
    rays = torch.cat([O.generate_rays(Kc, T, 2, 3840)[pick], syn.frame_id_columns(96, 8)], -1)
    jitter, u = torch.rand(9, 96, 128, generator=g), torch.rand(9, 96, 64, generator=g)
    model.replay = {"jitter": jitter.cuda(), "u": u.cuda()}
    with torch.no_grad():
        out = model(rays.cuda(), None, None)
    ref = oracle_render(meta, rays, list(jitter) + list(u))
    ref64 = oracle_render(meta, rays, list(jitter) + list(u), torch.float64)
    for i in range(9):
        assert torch.equal(out[4][i].cpu(), ref[4][i])
    assert sum(int(m.sum()) for m in ref[4][1:]) > 20            # performers are actually hit
    assert float((out[1][0].cpu() - ref[1][0]).abs().max()) <= COLOR_ATOL
    fine_stage_bar(out[0][0].cpu(), ref[0][0], ref64[0][0], COLOR_ATOL, "C5-shaped fine mixed colour")


def test_per_chunk_boxes_follow_row0_of_every_reference_chunk():
    """Retiming quirk (layered_rfrender.py:195-200): each reference chunk takes its boxes from ITS row 0.
    Rays whose frame ids change between chunks must be grouped accordingly (one launch per run of equal ids)."""
    from oracle import stnerf_oracle as O
    from stnerf_amd.utils import layered_batchify_ray
    meta = dict(L=2, n1=12, n2=6, space_time=True, deform_time=True, weight_seed=47, edit={})
    model = build_model(meta)
    sd = syn.make_state_dict(2, True, True, 47)
    K, T = syn.camera(8, 12, 12.0)
    base = O.generate_rays(K, T, 8, 12)                           # 96 rays = 4 chunks of 24
    fid = torch.zeros(96, 3)
    fid[:, 0] = 1.0
    for c, (f1, f2) in enumerate([(1.0, 2.0), (1.0, 2.0), (2.5, 1.5), (3.0, 3.0)]):
        fid[24 * c:24 * (c + 1), 1], fid[24 * c:24 * (c + 1), 2] = f1, f2
    fid[30, 1] = 3.0                                              # NOT row 0 of its chunk: must not move the box
    rays = torch.cat([base, fid], -1)
    g = torch.Generator().manual_seed(3)
    jitter, u = torch.rand(3, 96, 12, generator=g), torch.rand(3, 96, 6, generator=g)
    model.replay = {"jitter": jitter.cuda(), "u": u.cuda()}
    with torch.no_grad():
        out = layered_batchify_ray(model, rays.cuda(), None, None, chuncks=24, density_threshold=0.05)
    order = []                                                    # oracle draws per chunk: l jitter then l u tensors
    for c in range(4):
        order += [jitter[i, 24 * c:24 * (c + 1)] for i in range(3)] + [u[i, 24 * c:24 * (c + 1)] for i in range(3)]
    ref = oracle_render(meta, rays, order, chunk=24, density_threshold=0.05)
    ref64 = oracle_render(meta, rays, order, torch.float64, chunk=24, density_threshold=0.05)
    for i in range(3):
        assert torch.equal(out[4][i].cpu(), ref[4][i])
    assert float((out[1][0].cpu() - ref[1][0]).abs().max()) <= COLOR_ATOL
    fine_stage_bar(out[0][0].cpu(), ref[0][0], ref64[0][0], COLOR_ATOL, "per-chunk boxes, fine mixed colour")


def test_bad_inputs_raise_instead_of_exiting():
    meta, a = load_golden("fwd_c3")
    model = build_model(meta)
    with pytest.raises(ValueError, match="undefined ray format"):     # the reference prints and calls exit(-1)
        model(torch.zeros(10, 8, device="cuda"), None, None)
    fresh = build_model(meta)
    fresh.bboxes = None
    with pytest.raises(RuntimeError, match="set_bkgd_bbox / set_bboxes"):
        fresh(a["rays"].cuda(), None, None)
    with pytest.raises(ValueError):
        fresh.set_precision("bf16")


def test_user_facing_render_path_runs_the_demo_flow():
    """demo/taekwondo_demo.py:46-53 on the mirror: smooth path + retiming + per-frame edit schedule + render_path."""
    import types as _t
    from stnerf_amd.render import LayeredNeuralRenderer
    meta = dict(L=2, n1=16, n2=8, space_time=True, deform_time=True, weight_seed=51, edit={})
    model = build_model(meta)
    C, h, w = 4, 36, 64
    poses, Ks = [], []
    for i in range(C):
        K, T = syn.camera(h, w, orbit_deg=-20.0 + 12.0 * i)
        poses.append(T)
        Ks.append(K)
    cfg = _t.SimpleNamespace(DATASETS=_t.SimpleNamespace(LAYER_NUM=2, FRAME_NUM=3, FRAME_OFFSET=0),
                             INPUT=_t.SimpleNamespace(SIZE_TEST=[w, h]), OUTPUT_DIR="")
    r = LayeredNeuralRenderer(cfg, s_alpha=[1.0, 0.3], model=model, gt_poses=torch.stack(poses), gt_Ks=Ks)
    r.set_fps(25)
    r.set_smooth_path_poses(3, around=True, smooth_time=True)
    r.retime_by_key_frames(1, [1, 3], [2, 3])
    seen = []
    images, depths = r.render_path(density_threshold=0.01, bkgd_density_threshold=0.0,
                                   on_frame=lambda idx, c, d, cl, dl: seen.append((idx, c.is_cuda, len(cl))))
    assert len(images) == 3 and images[0].shape == (h, w, 3) and depths[0].shape == (h, w, 1)
    assert seen == [(0, True, 3), (1, True, 3), (2, True, 3)]
    assert all(bool(torch.isfinite(im).all()) for im in images) and float(images[1].max()) <= 1 + 1e-5
    assert not torch.equal(images[0], images[2])                 # the camera actually moved
    assert len(r.images_layer[1]) == 3 and r.image_num == 3
    assert model.alpha == pytest.approx(0.3)                     # last frame of the alpha schedule was poked
    # the walking demo's loop (occlusion composite of layer 2 over the background, :550-618) and the video hook (:624-637)
    r.render_path_walking(density_threshold=0.01)
    assert len(r.images_hide) == 3 and r.images_hide[0].shape == (h, w, 3)
    cl0, cl2 = r.images_layer[0][1], r.images_layer[2][1]
    dl0, dl2 = r.depths_layer[0][1], r.depths_layer[2][1]
    want = cl0.clone()
    idx = torch.logical_and(torch.cat([dl2 < dl0] * 3, 2), cl2 != 0)
    want[idx] = cl2[idx]
    assert torch.equal(r.images_hide[1], want)
    written = []
    assert r.save_video(lambda kind, i, frames, fps: written.append((kind, i, len(frames), fps))) is True
    assert written == [("color", 0, 3, 25), ("depth", 0, 3, 25)] and r.save_count == 1
