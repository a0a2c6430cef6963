"""The production compositor (csrc/render.hip: composite_single_kernel + composite_merge_kernel -- layers in registers,
merged order built by inserting one layer at a time) against
  * the LDS-staged kernel (the `order` parity call still takes it): same merged order, same lanes, same arithmetic ->
    bit-identical images and weights, at every BASELINE shape (3 x 64 / 3 x 128, 5 x 128, 9 x 128 / 9 x 192) and at ragged
    ones, with missed / hidden / grazing / descending layers and ties between layers;
  * the CPU oracle (layers/render_layer.py:8-58 restated in oracle/stnerf_oracle.py) at the stated fp32 tolerance.
Needs an MI355X: `pytest -m gpu`."""
import pytest
import torch

from oracle import stnerf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from stnerf_amd import ops as _ops
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return _ops


def bits(x):
    return x.contiguous().view(torch.int32)


def scene(n, l, S, seed, hit=0.5, ties=True):
    """Depth lists as the sampler leaves them: ascending per layer inside the layer's own interval, -1000 everywhere on a
    ray the layer misses; plus the special rows: a ray that misses the background box (0 .. -1000, strictly descending),
    a descending performer (edited box), grazing hits (mask clear, all samples at one real depth), exact ties between
    layers (a performer sample copied from the background list)."""
    g = torch.Generator().manual_seed(seed)
    lo = torch.rand(n, l, 1, generator=g) * 3.0
    hi = lo + 0.2 + torch.rand(n, l, 1, generator=g) * 3.0
    t = torch.sort(lo + (hi - lo) * torch.rand(n, l, S, generator=g), -1)[0]
    t[:, 0] = torch.sort(torch.rand(n, S, generator=g) * 6.5 - 0.3, -1)[0]
    hitm = torch.rand(n, l, generator=g) < hit
    hitm[:, 0] = True
    t[~hitm] = -1000.0
    mask = hitm.clone()
    bk_miss = torch.rand(n, generator=g) < 0.05
    t[bk_miss, 0] = -(torch.arange(S).float() + torch.rand(int(bk_miss.sum()), S, generator=g)) * (1000.0 / S)
    mask[bk_miss, 0] = torch.rand(int(bk_miss.sum()), generator=g) < 0.5
    if l > 1:
        rev = (torch.rand(n, generator=g) < 0.05) & hitm[:, 1]
        t[rev, 1] = t[rev, 1].flip(-1) + torch.linspace(0.0, -1e-3, S)      # strictly descending
        graze = (torch.rand(n, generator=g) < 0.05) & ~hitm[:, l - 1]
        t[graze, l - 1] = (torch.rand(int(graze.sum()), 1, generator=g) * 4.0).expand(-1, S)
        if ties and S >= 3:
            tie = (torch.rand(n, generator=g) < 0.2) & hitm[:, 1] & ~rev & ~bk_miss
            k = S // 3
            t[tie, 1, k] = t[tie, 0, k].clamp(min=t[tie, 1, k - 1], max=t[tie, 1, k + 1])
    raw = torch.randn(n, l, S, 4, generator=g) * torch.tensor([2.0, 2.0, 2.0, 4.0])
    return t, raw, mask.to(torch.uint8)


# (16, 192): two launches (lists of four layers, then sixteen), scratch cleared by the library (no pre-pass instantiation for that
# shape), more than 64 KB of dynamic LDS per workgroup
SHAPES = [(3, 64), (3, 128), (5, 128), (9, 128), (9, 192), (2, 40), (4, 150), (3, 17), (16, 64), (6, 1), (3, 2), (16, 192)]


@pytest.mark.parametrize("fine", [False, True])
@pytest.mark.parametrize("l, S", SHAPES)
def test_merge_kernel_is_bit_identical_to_the_staged_kernel(ops, l, S, fine):
    n = 3000 if l * S <= 1000 else 1200 if l * S <= 2000 else 500
    t, raw, mask = scene(n, l, S, seed=100 * l + S + fine, hit=0.6 if l <= 5 else 0.35)
    ev = [2] + [1] * (l - 1)
    if l > 2:
        ev[2] = 0                                           # a hidden performer: real depths, no network output
    kw = dict(near=0.4, fine=fine, cut_negative_t=not fine, thresholds=[0.3 if fine else None] + [0.5] * (l - 1),
              sigma_scale=[1.0] * (l - 1) + [0.4 if fine else 1.0], evaluated=ev, want_weights=True)
    td, rd, md = t.cuda(), raw.cuda(), mask.cuda()
    ref = ops.composite(td, rd, md, want_order=True, **kw)            # LDS-staged kernel (parity path)
    for two_pass in (True, False):
        for activated in (False, True):
            r_in = rd.clone()
            if activated:
                r_in[..., :3] = torch.sigmoid(r_in[..., :3])
            got = ops.composite(td, r_in, md, want_order=False, two_pass=two_pass, rgb_activated=activated, **kw)
            want = ref if not activated else ops.composite(td, r_in, md, want_order=True, rgb_activated=True, **kw)
            for name, a, b in zip(("layer_out", "mixed", "weights"), got[:3], want[:3]):
                bad = (bits(a) != bits(b))
                assert not bool(bad.any()), (name, two_pass, activated, int(bad.sum()), bad.nonzero()[:5].tolist())
    multi = ((t[:, :, 0] > -999).sum(1) >= 2).sum()
    assert int(multi) > n // 4


@pytest.mark.parametrize("fine", [False, True])
def test_merge_kernel_vs_oracle(ops, fine):
    """Five layers x 96 samples against the oracle's composite of the stably sorted union (torch.sort(stable=True) over
    the concatenation, layered_rfrender.py:425-448 / :587-606)."""
    n, l, S = 500, 5, 96
    t, raw, mask = scene(n, l, S, seed=7 + fine, hit=0.7)
    near, thr, bthr, alpha = 0.6, 0.4, 0.2, 0.5
    sig = [raw[:, i, :, 3:].clone() for i in range(l)]
    rgb = [raw[:, i, :, :3].clone() for i in range(l)]
    for i in range(1, l):
        dead = mask[:, i] == 0
        sig[i][dead] = 0
        rgb[i][dead] = 0
    if fine:
        sig[0][sig[0] < bthr] = 0
    for i in range(1, l):
        if not fine:
            sig[i][t[:, i].unsqueeze(-1) < 0] = 0
        sig[i][sig[i] < thr] = 0
        if fine and i == l - 1:
            sig[i] = sig[i] * alpha
    if not fine:
        sig[0][t[:, 0].unsqueeze(-1) < near] = 0
    ts = [t[:, i].unsqueeze(-1) for i in range(l)]
    t_mix, order = torch.sort(torch.cat(ts, -2), dim=-2, stable=True)
    rgb_mix = torch.cat(rgb, -2).gather(1, order.repeat(1, 1, 3))
    sig_mix = torch.cat(sig, -2).gather(1, order)
    if fine:
        sig_mix[t_mix < near] = 0
    mix = O.composite(t_mix, rgb_mix, sig_mix)
    lo, mo, w, _ = ops.composite(t.cuda(), raw.cuda(), mask.cuda(), near=near, fine=fine, cut_negative_t=not fine,
                                 thresholds=[bthr if fine else None] + [thr] * (l - 1), evaluated=[2] + [1] * (l - 1),
                                 sigma_scale=[1.0] * (l - 1) + [alpha if fine else 1.0], want_weights=True)
    ok = torch.isfinite(mix[0]).all(-1) & torch.isfinite(mix[2]).all(-1)     # descending rows: inf / NaN in the reference too
    assert float(ok.float().mean()) > 0.8
    torch.testing.assert_close(mo[:, :3].cpu()[ok], mix[0][ok], rtol=1e-5, atol=3e-6)
    torch.testing.assert_close(mo[:, 4:5].cpu()[ok], mix[2][ok], rtol=1e-5, atol=3e-6)
    for i in range(l):
        per = O.composite(ts[i], rgb[i], sig[i])
        good = torch.isfinite(per[3]).all(-1).all(-1)
        # (alpha = 1 - exp(-sigma delta) on the 1-ulp hardware exponential: 1.2e-7 absolute where sigma delta is small, times a
        # transmittance that exceeds 1 on the descending rows)
        torch.testing.assert_close(w[:, i].cpu()[good], per[3].squeeze(-1)[good], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(lo[:, i, :3].cpu()[good], per[0][good], rtol=1e-5, atol=3e-6)


def test_merge_kernel_unsorted_layer_takes_the_general_rank(ops):
    """A layer that is neither ascending nor strictly descending: the ray's order is the stable sort of the concatenation."""
    torch.manual_seed(5)
    n, l, S = 200, 3, 33
    t = torch.sort(torch.rand(n, l, S) * 4.0, -1)[0]
    t[:100, 1] = t[:100, 1][:, torch.randperm(S)]
    t[100:150, 2] = t[100:150, 2].flip(-1)
    t[100:150, 2, 3] = t[100:150, 2, 2]                             # a tie inside a descending list
    raw = torch.randn(n, l, S, 4)
    lo, mo, _, _ = ops.composite(t.cuda(), raw.cuda(), None)
    _, order = torch.sort(t.reshape(n, l * S), dim=-1, stable=True)
    t_mix = t.reshape(n, l * S, 1).gather(1, order.unsqueeze(-1))
    rgb_mix = raw[..., :3].reshape(n, l * S, 3).gather(1, order.unsqueeze(-1).repeat(1, 1, 3))
    sig_mix = raw[..., 3:].reshape(n, l * S, 1).gather(1, order.unsqueeze(-1))
    mix = O.composite(t_mix, rgb_mix, sig_mix)
    torch.testing.assert_close(mo[:, :3].cpu(), mix[0], rtol=1e-5, atol=3e-6)
    torch.testing.assert_close(mo[:, 4:5].cpu(), mix[2], rtol=1e-5, atol=3e-6)


@pytest.mark.parametrize("L, n1", [(2, 64), (4, 64), (8, 128), (2, 90)])
def test_sampler_missed_hint_is_bitwise_neutral_for_the_compositor(L, n1):
    """The sampler's "missed" hint (mask bit 1: both slab hits -1000, every depth of the pair exactly -1000) lets the
    compositor skip reading those depths; the composites must be the SAME BITS as with a plain 0 / 1 mask, in all three
    kernels (single-layer, merge, staged), and the hint must be set exactly on the pairs whose depths are all -1000."""
    import numpy as np
    from stnerf_amd import ops, synthetic as syn
    h, w = 40, 64
    K, T = syn.camera(h, w, 17.0)
    rays = ops.generate_rays(K, T, h, w, frame_ids=[1.0] + [2.5] * L)
    bk, per = syn.scene_boxes(L)
    boxes = torch.cat([bk, per[1]], 0).cuda()
    t, _, raw_mask = ops.sample_coarse(rays, boxes, n1, seed=3, raw_mask=True)
    mask = raw_mask & 1
    hinted = (raw_mask & 2) != 0
    all_missed = (t == -1000.0).all(-1)
    assert torch.equal(hinted, all_missed) and int(hinted.sum()) > 0.2 * hinted.numel() / (L + 1)
    assert not bool((hinted & (mask != 0)).any()) and not bool(hinted[:, 0].any())
    torch.manual_seed(L)
    raw = torch.randn(t.shape + (4,), device="cuda")
    kw = dict(near=0.1, fine=False, cut_negative_t=True, thresholds=[None] + [0.2] * L, evaluated=[2] + [1] * L, want_weights=True)
    bits = lambda x: x.contiguous().view(torch.int32)
    for two_pass, want_order in ((True, False), (False, False), (False, True)):
        a = ops.composite(t, raw, mask, two_pass=two_pass, want_order=want_order, **kw)
        b = ops.composite(t, raw, raw_mask, two_pass=two_pass, want_order=want_order, **kw)
        for x, y in zip(a[:3], b[:3]):
            assert torch.equal(bits(x), bits(y)), (two_pass, want_order)


@pytest.mark.parametrize("L, n1, n2", [(2, 64, 64), (4, 64, 64), (8, 128, 64), (2, 90, 30), (3, 200, 40)])
def test_resampler_skips_exactly_the_pairs_the_sampler_flagged(L, n1, n2):
    """stnerf_resample with the sampler's mask: a pair flagged "missed" is left UNWRITTEN (sentinel intact), every other pair is
    the bits of the call without a mask -- and on the flagged pairs that call wrote nothing but -1000 depths."""
    from stnerf_amd import ops, synthetic as syn
    import stnerf_amd.ops as O_
    h, w = 24, 48
    K, T = syn.camera(h, w, 17.0)
    rays = ops.generate_rays(K, T, h, w, frame_ids=[1.0] + [2.5] * L)
    bk, per = syn.scene_boxes(L)
    boxes = torch.cat([bk, per[1]], 0).cuda()
    t, _, raw_mask = ops.sample_coarse(rays, boxes, n1, seed=3, raw_mask=True)
    torch.manual_seed(n1)
    wts = torch.rand(t.shape, device="cuda") ** 6
    plain_t, plain_xyz = ops.resample(t, wts, n2, rays, seed=9)
    # a sentinel-filled output: call the entry point with pre-filled buffers through the wrapper's allocator
    orig = torch.empty
    torch.empty = lambda *a, **k: orig(*a, **k).fill_(7.0) if k.get("dtype", torch.float32) == torch.float32 else orig(*a, **k)
    try:
        hint_t, hint_xyz = ops.resample(t, wts, n2, rays, seed=9, mask=raw_mask)
    finally:
        torch.empty = orig
    flagged = (raw_mask & 2) != 0
    assert int(flagged.sum()) > 0 and int((~flagged).sum()) > 0
    assert bool((hint_t[flagged] == 7.0).all()) and bool((hint_xyz[flagged] == 7.0).all())
    assert bool((plain_t[flagged] == -1000.0).all())
    assert torch.equal(hint_t[~flagged], plain_t[~flagged]) and torch.equal(hint_xyz[~flagged], plain_xyz[~flagged])
