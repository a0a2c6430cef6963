"""the backward pass of SpaceNet / MotionNet (csrc/train.hip, stnerf_amd.modeling.autograd) --
what engine/layered_trainer.py:192-217's loss.backward() asks of modeling/spacenet.py:101-160 and
modeling/motion_net.py:34-71.  Needs an MI355X: `pytest -m gpu`.

Reference = torch.autograd through the CPU oracle's restatement of the two networks in float64 (the same graph the
reference's nn.Modules build).  Bars:
  * every gradient tensor (all weights, all biases, the sample points) within 2e-5 of its fp64 value, relative to the
    tensor's largest entry;
  * no further from fp64 than 12 x the reference's own fp32 autograd is (measured 2.2 - 7.2 x -- the reference's side of the
    ratio moves with the host's BLAS blocking and thread count: the f32 MFMA accumulates a 256-deep dot product as one chain
    of K = 2 steps where ATen's blocked sgemm sums partial blocks; both ~1e-6, the 2e-5 bar above is the one that matters);
  * ReLU'(0): a hidden unit whose pre-activation is within fp32 rounding of zero has a gradient that is 0 in one evaluation
    and passes in another -- in the reference's own fp32 autograd just as much.  Samples with such a unit (|pre-activation|
    < 1e-5 in the fp64 evaluation: a few percent of them) get a ZERO cotangent, in the HIP run and in both references, so
    that the comparison is about arithmetic and not about which side of zero a rounding error fell;
  * the GEMM building blocks against fp64 matmuls on ragged shapes, masks, accumulation, strided column blocks."""
import numpy as np
import pytest
import torch

from oracle import stnerf_oracle as O
from stnerf_amd import synthetic as syn

pytestmark = pytest.mark.gpu

GRAD_RTOL = 2e-5


@pytest.fixture(scope="module")
def ops():
    from stnerf_amd import ops as _ops
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return _ops


def _padded(t, extra=0):
    """A (rows, cols) view of 16-byte aligned padded storage holding t."""
    rows, cols = t.shape
    buf = torch.full((rows, (cols + 3) // 4 * 4 + extra), float("nan"), device="cuda")
    buf[:, :cols] = t.cuda()
    return buf[:, :cols]


def _rel(got, ref):
    return float((got.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("m, n, k", [(1, 1, 4), (127, 3, 128), (129, 256, 63), (1000, 128, 319), (300, 256, 256), (2500, 1, 256), (64, 130, 84),
                                     (128 * 70001 + 5, 3, 4)])     # (the last one: more than 65535 row tiles)
def test_linear_forward_dx_dw_match_fp64_matmuls(ops, m, n, k):
    g = torch.Generator().manual_seed(m + n + k)
    x, w, b = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) / k ** 0.5, torch.randn(n, generator=g)
    dy = torch.randn(m, n, generator=g)
    xin = torch.relu(torch.randn(m, k, generator=g))                     # the stored post-ReLU tensor that fed the layer
    xd, wd, dyd, maskd = _padded(x, 4), _padded(w), _padded(dy), _padded(xin)
    x64, w64, dy64 = x.double(), w.double(), dy.double()
    for relu in (False, True):
        y = torch.full((m, n + 5), 7.0, device="cuda")                  # a column block of a wider matrix
        ops.train_linear_fwd(xd, wd, b.cuda(), y[:, 2:2 + n], relu)
        ref = x64 @ w64.T + b.double()
        assert _rel(y[:, 2:2 + n], torch.relu(ref) if relu else ref) <= 2e-6
        assert bool((y[:, :2] == 7.0).all()) and bool((y[:, 2 + n:] == 7.0).all())
    dx = torch.full((m, k + 3), 7.0, device="cuda")
    ops.train_linear_dx(dyd, wd, dx[:, 1:1 + k], mask=maskd)
    ref = (dy64 @ w64) * (xin > 0)
    assert _rel(dx[:, 1:1 + k], ref) <= 2e-6 and bool((dx[:, :1] == 7.0).all()) and bool((dx[:, 1 + k:] == 7.0).all())
    ops.train_linear_dx(dyd, wd, dx[:, 1:1 + k], accumulate=True)        # second consumer, no mask
    assert _rel(dx[:, 1:1 + k], ref + dy64 @ w64) <= 2e-6
    dw, db = torch.full((n, k), 0.5, device="cuda"), torch.full((n,), -0.25, device="cuda")
    ops.train_linear_dw(dyd, xd, dw, db, accumulate=True)
    assert _rel(dw, 0.5 + dy64.T @ x64) <= (4e-6 if m < 1_000_000 else 4e-5) and _rel(db, -0.25 + dy64.sum(0)) <= (4e-6 if m < 1_000_000 else 2e-4)
    dw2 = torch.empty(n, k, device="cuda")
    ops.train_linear_dw(dyd, xd, dw2, None, accumulate=False)
    ops.train_linear_dw(dyd, xd, dw, db, accumulate=False)
    assert torch.equal(dw, dw2)                                          # deterministic: same bits every time


def test_dw_reduction_over_many_samples_is_deterministic_and_accurate(ops):
    g = torch.Generator().manual_seed(3)
    m, n, k = 150_000, 128, 256                                          # > 64 slices of 2048: the capped split
    x, dy = torch.relu(torch.randn(m, k, generator=g)), torch.randn(m, n, generator=g)
    xd, dyd = x.cuda(), dy.cuda()
    a, b = torch.empty(n, k, device="cuda"), torch.empty(n, k, device="cuda")
    da, db = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
    ops.train_linear_dw(dyd, xd, a, da, False)
    ops.train_linear_dw(dyd, xd, b, db, False)
    assert torch.equal(a, b) and torch.equal(da, db)
    ref = dy.double().T @ x.double()
    assert _rel(a, ref) <= 1e-5 and _rel(da, dy.double().sum(0)) <= 1e-5
    fp32 = dy.T @ x
    assert float((a.cpu().double() - ref).abs().max()) <= 3 * float((fp32.double() - ref).abs().max()) + 1e-6


@pytest.mark.parametrize("m", [1, 300, 7001, 300_000])
def test_dw_batch_matches_fp64_matmuls_layer_by_layer(ops, m):
    """stnerf_train_dw_batch: every layer shape of the two networks (SpaceNet's ten, then MotionNet's six) in one launch each -- 319-
    and 304-wide inputs (two 128-column tiles and a 64-column one), 84 (a 96-column tile), 1- and 3-row heads, layers without a bias gradient, dw as a column block of
    a wider matrix, accumulate -- against fp64 matmuls; the same bits on every run and under accumulate = False after a dirty
    destination; the per-layer entry point's result within rounding."""
    g = torch.Generator().manual_seed(m)
    space = [(256, 63), (256, 256), (256, 256), (256, 256), (256, 319), (256, 256), (256, 256), (1, 256), (128, 304), (3, 128)]
    motion = [(128, 84), (128, 128), (128, 128), (128, 128), (128, 128), (3, 128)]
    # thin layers (<= 4 outputs: one 32-row block) of every width, and neighbours; several row blocks with remainders on both sides;
    # bias sums of more than 256 outputs
    odd = [(4, 319), (2, 63), (1, 4), (5, 130), (130, 5), (512, 520), (300, 300)]
    for shapes in (space, motion, odd):
        dys = [torch.randn(m, n, generator=g) for n, _ in shapes]
        xs = [torch.relu(torch.randn(m, k, generator=g)) for _, k in shapes]
        dyd, xd = [_padded(t) for t in dys], [_padded(t, 4) for t in xs]
        wide = [torch.full((n, k + 5), 0.5, device="cuda") for n, k in shapes]
        dws = [w[:, 2:2 + k] for w, (_, k) in zip(wide, shapes)]
        dbs = [None if i == 2 else torch.full((n,), -0.25, device="cuda") for i, (n, _) in enumerate(shapes)]
        ops.train_dw_batch(list(zip(dyd, xd, dws, dbs)), accumulate=True)
        tol = 4e-6 if m < 100_000 else 1e-5
        for i, (n, k) in enumerate(shapes):
            ref = dys[i].double().T @ xs[i].double()
            assert _rel(dws[i], 0.5 + ref) <= tol, (i, n, k)
            assert bool((wide[i][:, :2] == 0.5).all()) and bool((wide[i][:, 2 + k:] == 0.5).all())
            if dbs[i] is not None:
                assert _rel(dbs[i], -0.25 + dys[i].double().sum(0)) <= max(tol, 2e-5 if m > 100_000 else 0), (i, n)
        a = [torch.full((n, k), float("nan"), device="cuda") for n, k in shapes]
        b = [torch.empty(n, k, device="cuda") for n, k in shapes]
        da = [torch.full((n,), float("nan"), device="cuda") for n, _ in shapes]
        db_ = [torch.empty(n, device="cuda") for n, _ in shapes]
        ops.train_dw_batch(list(zip(dyd, xd, a, da)), accumulate=False)
        ops.train_dw_batch(list(zip(dyd, xd, b, db_)), accumulate=False)
        for i, (n, k) in enumerate(shapes):
            assert torch.equal(a[i], b[i]) and (da[i] is None or torch.equal(da[i], db_[i])), i
            one, one_b = torch.empty(n, k, device="cuda"), torch.empty(n, device="cuda")
            ops.train_linear_dw(dyd[i], xd[i], one, one_b, False)
            assert _rel(a[i], one.double().cpu()) <= (2e-6 if m < 100_000 else 5e-6), i    # (two fp32 summation orders of m terms)
            assert da[i] is None or _rel(da[i], one_b.double().cpu()) <= (2e-6 if m < 100_000 else 2e-5), i


def test_dw_batch_refuses_what_it_cannot_run(ops):
    dy, x = torch.zeros(8, 4, device="cuda"), torch.zeros(8, 4, device="cuda")
    dw = torch.zeros(4, 4, device="cuda")
    with pytest.raises(ValueError):
        ops.train_dw_batch([(dy, x, dw, None)] * 17, False)
    with pytest.raises(ValueError):
        ops.train_dw_batch([(dy, x, dw, None), (dy[:4], x[:4], dw, None)], False)      # different sample counts
    with pytest.raises(RuntimeError):
        ops.train_dw_batch([(dy.cpu(), x, dw, None)], False)
    ops.train_dw_batch([], False)
    wide = torch.zeros(8, 1024, device="cuda")
    with pytest.raises(ValueError, match="tiles"):                                     # 8 x 8 + 8 x 1 tiles of 128 x 128
        ops.train_dw_batch([(wide, wide, torch.zeros(1024, 1024, device="cuda"), None), (wide, x, torch.zeros(1024, 4, device="cuda"), None)], False)


def test_encode_and_its_chain_rule(ops):
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(500, 3, generator=g) - 0.5) * 6.0
    for inc in (True, False):
        w = 3 * (int(inc) + 20)
        y = torch.full((500, w + 9), 7.0, device="cuda")
        ops.train_encode(x.cuda(), y[:, 4:4 + w], 10, inc)
        ref = O.positional_encoding(x.double(), 10, inc)
        assert float((y[:, 4:4 + w].cpu().double() - ref).abs().max()) <= 2e-6 and bool((y[:, :4] == 7.0).all())
        dy = torch.randn(500, w, generator=g)
        x64 = x.double().requires_grad_(True)
        (O.positional_encoding(x64, 10, inc) * dy.double()).sum().backward()
        dx = torch.zeros(500, 3, device="cuda")
        ops.train_encode_bwd(x.cuda(), dy.cuda(), dx, 10, inc)
        assert _rel(dx, x64.grad) <= 2e-6
    # a ray's encoding repeated on its samples, rgb_net's leading ReLU applied
    d = torch.nn.functional.normalize(torch.randn(20, 3, generator=g), dim=-1)
    y = torch.empty(20 * 7, 27, device="cuda")
    ops.train_encode(d.cuda(), y, 4, True, rows_per_src=7, relu=True)
    ref = torch.relu(O.positional_encoding(d.double(), 4)).repeat_interleave(7, 0)
    assert float((y.cpu().double() - ref).abs().max()) <= 2e-6
    # MotionNet's fractional-time lerp on the frame-id column
    xt = torch.cat([x[:64], torch.tensor([1.0, 2.25, 7.5, 30.0]).repeat(16).reshape(-1, 1)], -1)
    y = torch.empty(64, 84, device="cuda")
    ops.train_encode(xt.cuda(), y, 10, True, lerp_col=3)
    lo = torch.floor(xt[:, 3:]).double()
    wt = xt[:, 3:].double() - lo
    ref = (1 - wt) * O.positional_encoding(torch.cat([xt[:, :3].double(), lo], -1), 10) + wt * O.positional_encoding(
        torch.cat([xt[:, :3].double(), lo + 1], -1), 10)
    assert float((y.cpu().double() - ref).abs().max()) <= 4e-6


def _reference_grads(fn, params, inputs, cots, dtype):
    ps = {k: v.to(dtype).clone().requires_grad_(True) for k, v in params.items()}
    ins = [t.to(dtype).clone().requires_grad_(True) if t is not None and t.is_floating_point() and g else (t.to(dtype) if t is not None else None)
           for t, g in inputs]
    outs = fn(ps, *ins)
    outs = outs if isinstance(outs, tuple) else (outs,)
    sum((o * c.to(dtype)).sum() for o, c in zip(outs, cots)).backward()
    return {k: v.grad for k, v in ps.items()}, [t.grad if (t is not None and t.requires_grad) else None for t in ins]


def _check(name, got, ref64, ref32):
    e = float((got.double().cpu() - ref64).abs().max())
    scale = float(ref64.abs().max())
    e32 = float((ref32.double() - ref64).abs().max())
    assert e <= GRAD_RTOL * scale + 1e-12, f"{name}: |err| {e:.3e} = {e / max(scale, 1e-30):.2e} of the largest entry"
    assert e <= 12 * e32 + 2e-7 * scale, f"{name}: {e:.3e} vs the fp32 autograd's {e32:.3e}"
    return e / max(scale, 1e-30), e / max(e32, 1e-30)


class _MinPreactivation:
    """While active, F.relu records per ROW the smallest |input| it has seen (rows = samples in both oracle networks)."""

    def __init__(self):
        self.smallest = None

    def __enter__(self):
        import torch.nn.functional as F
        self._orig = F.relu

        def relu(x, *a, **k):
            mag = x.detach().abs().reshape(x.shape[0], -1)
            m = torch.where(mag == 0, torch.ones_like(mag), mag).min(1)[0]     # (exact zeros: outputs of an earlier ReLU fed to rgb_net's leading one)
            self.smallest = m if self.smallest is None else torch.minimum(self.smallest, m)
            return self._orig(x, *a, **k)
        F.relu = relu
        return self

    def __exit__(self, *exc):
        import torch.nn.functional as F
        F.relu = self._orig


def _safe_samples(fn, eps=1e-5):
    """bool per sample: no ReLU input of the fp64 evaluation within eps of zero."""
    with _MinPreactivation() as rec, torch.no_grad():
        fn()
    return rec.smallest > eps


@pytest.mark.parametrize("use_time, deep, inc, use_dir, n, ns, chunk", [(True, False, True, True, 200, 64, None), (False, False, True, True, 97, 90, 3000),
                                                                         (True, True, True, True, 150, 12, None), (True, False, False, False, 60, 33, 700),
                                                                         (False, False, True, False, 40, 128, None)])
def test_spacenet_backward_matches_fp64_autograd(use_time, deep, inc, use_dir, n, ns, chunk, monkeypatch):
    from stnerf_amd.modeling import autograd as ag
    from stnerf_amd.modeling.spacenet import SpaceNet
    if chunk:
        monkeypatch.setattr(ag, "CHUNK_SAMPLES", chunk)                   # several chunks: gradients accumulate across them
    sd = syn.spacenet_state("net", np.random.RandomState(n + ns), use_time, deep_rgb=deep, include_input=inc, use_dir=use_dir)
    net = SpaceNet(include_input=inc, use_dir=use_dir, use_time=use_time, deep_rgb=deep)
    net.load_state_dict({k[4:]: v for k, v in sd.items()})
    net = net.cuda()
    g = torch.Generator().manual_seed(ns)
    pos = (torch.rand(n, ns, 3, generator=g) - 0.5) * 4.0
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    times = torch.rand(n, 1, generator=g) * 20 + 1
    c_rgb, c_sig = torch.randn(n, ns, 3, generator=g), torch.randn(n, ns, 1, generator=g) * 0.1
    sd64 = {k: v.double() for k, v in sd.items()}
    # (rgb_net's leading ReLU also sees the direction / time encodings, which are not differentiated: only the 256 backbone
    # columns count, and those are >= 0 already -- their zeros are exact zeros of stage2.4's ReLU, recorded there)
    safe = _safe_samples(lambda: O.space_net(sd64, "net", pos.double(), dirs.double(), times.double() if use_time else None)).reshape(n, ns, 1)
    assert 0.5 < float(safe.float().mean()) < 1.0
    c_rgb, c_sig = c_rgb * safe, c_sig * safe
    rays = torch.cat([torch.zeros(n, 3), dirs], -1).cuda()
    pd = pos.cuda().requires_grad_(True)
    rgb, sig = net(pd, rays, times.cuda() if use_time else None)
    assert rgb.shape == (n, ns, 3) and sig.shape == (n, ns, 1)
    ((rgb * c_rgb.cuda()).sum() + (sig * c_sig.cuda()).sum()).backward()
    fn = lambda ps, p, d, t: O.space_net(ps, "net", p, d, t)
    ins = [(pos, True), (dirs, False), (times if use_time else None, False)]
    g64, i64 = _reference_grads(fn, sd, ins, (c_rgb, c_sig), torch.float64)
    g32, i32 = _reference_grads(fn, sd, ins, (c_rgb, c_sig), torch.float32)
    worst = {}
    for k, p in net.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape, k
        worst[k] = _check(k, p.grad, g64["net." + k], g32["net." + k])
    worst["pos"] = _check("pos", pd.grad, i64[0], i32[0])
    print(f"SpaceNet time={use_time} deep={deep} inc={inc} dir={use_dir} {n}x{ns}: worst relative error "
          f"{max(v[0] for v in worst.values()):.2e}, worst ratio to the fp32 autograd {max(v[1] for v in worst.values()):.2f}")
    # sigma-only and rgb-only cotangents (the other output's gradient is None)
    for which in (0, 1):
        net.zero_grad()
        out = net(pos.cuda(), rays, times.cuda() if use_time else None)
        (out[which] * (c_rgb, c_sig)[which].cuda()).sum().backward()
        cots = (c_rgb if which == 0 else torch.zeros_like(c_rgb), c_sig if which == 1 else torch.zeros_like(c_sig))
        g64b, _ = _reference_grads(fn, sd, ins, cots, torch.float64)
        for k, p in net.named_parameters():
            ref = g64b["net." + k]
            got = p.grad if p.grad is not None else torch.zeros_like(p)
            assert float((got.double().cpu() - ref).abs().max()) <= GRAD_RTOL * float(ref.abs().max()) + 1e-9, (which, k)


@pytest.mark.parametrize("input_time, inc, rows, chunk", [(True, True, 5000, None), (True, True, 3001, 1024), (False, True, 777, None), (True, False, 900, None)])
def test_motionnet_backward_matches_fp64_autograd(input_time, inc, rows, chunk, monkeypatch):
    from stnerf_amd.modeling import autograd as ag
    from stnerf_amd.modeling.motion_net import MotionNet
    if chunk:
        monkeypatch.setattr(ag, "CHUNK_SAMPLES", chunk)
    sd = syn.motionnet_state("net", np.random.RandomState(rows), include_input=inc)
    net = MotionNet(c_input=4, include_input=inc, input_time=input_time)
    net.load_state_dict({k[4:]: v for k, v in sd.items()})
    net = net.cuda()
    g = torch.Generator().manual_seed(rows)
    xyz = (torch.rand(rows, 3, generator=g) - 0.5) * 4.0
    t = torch.where(torch.rand(rows, 1, generator=g) < 0.5, torch.floor(torch.rand(rows, 1, generator=g) * 30), torch.rand(rows, 1, generator=g) * 30) + 1
    xt = torch.cat([xyz, t], -1)
    cot = torch.randn(rows, 3, generator=g)
    sd64 = {k: v.double() for k, v in sd.items()}
    safe = _safe_samples(lambda: O.motion_net(sd64, "net", xt.double(), input_time=input_time)).reshape(rows, 1)
    assert 0.5 < float(safe.float().mean()) <= 1.0
    cot = cot * safe
    xd = xt.cuda().requires_grad_(True)
    flow = net(xd)
    (flow * cot.cuda()).sum().backward()
    fn = lambda ps, x: O.motion_net(ps, "net", x, input_time=input_time)
    g64, i64 = _reference_grads(fn, sd, [(xt, True)], (cot,), torch.float64)
    g32, i32 = _reference_grads(fn, sd, [(xt, True)], (cot,), torch.float32)
    worst = {k: _check(k, p.grad, g64["net." + k], g32["net." + k]) for k, p in net.named_parameters()}
    worst["xyz"] = _check("xyz", xd.grad[:, :3], i64[0][:, :3], i32[0][:, :3])
    assert float(xd.grad[:, 3].abs().max()) == 0.0                       # the frame id is data
    print(f"MotionNet input_time={input_time} inc={inc} rows={rows}: worst relative error {max(v[0] for v in worst.values()):.2e}, "
          f"worst ratio to the fp32 autograd {max(v[1] for v in worst.values()):.2f}")


def test_deformed_spacenet_chain_trains_both_networks():
    """modeling/layered_rfrender.py:340-356 + :382-410: pos = xyz + MotionNet([xyz, t]), then SpaceNet(pos): the SpaceNet's
    point gradient is the MotionNet's output gradient; one optimiser step lowers the loss."""
    from stnerf_amd.modeling.motion_net import MotionNet
    from stnerf_amd.modeling.spacenet import SpaceNet
    rs = np.random.RandomState(8)
    sd_s, sd_m = syn.spacenet_state("net", rs, True), syn.motionnet_state("net", rs)
    space, motion = SpaceNet(use_time=True), MotionNet(c_input=4, input_time=True)
    space.load_state_dict({k[4:]: v for k, v in sd_s.items()})
    motion.load_state_dict({k[4:]: v for k, v in sd_m.items()})
    space, motion = space.cuda(), motion.cuda()
    g = torch.Generator().manual_seed(2)
    n, ns = 128, 32
    xyz = (torch.rand(n, ns, 3, generator=g) - 0.5) * 3.0
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    times = torch.full((n, 1), 2.5)
    target = torch.rand(n, ns, 3, generator=g)
    rays = torch.cat([torch.zeros(n, 3), dirs], -1).cuda()

    def loss_of(space_fn, motion_fn, dev, keep):
        x = xyz.to(dev)
        flow = motion_fn(torch.cat([x, times.to(dev).view(n, 1, 1).repeat(1, ns, 1)], -1))
        rgb, sig = space_fn(x + flow)
        k = keep.to(dev)
        return (k * (torch.sigmoid(rgb) - target.to(dev)) ** 2).mean() + 1e-3 * (k * sig ** 2).mean()
    # samples with a hidden unit within rounding of ReLU's kink are left out of the loss (see the module docstring)
    sm64, ss64 = {k: v.double() for k, v in sd_m.items()}, {k: v.double() for k, v in sd_s.items()}
    keep = _safe_samples(lambda: loss_of(lambda p: O.space_net(ss64, "net", p, dirs.double(), times.double()),
                                         lambda x: O.motion_net(sm64, "net", x.double()), "cpu", torch.ones(n, ns, 1))).reshape(n, ns, 1).float()
    assert 0.5 < float(keep.mean()) < 1.0
    loss = loss_of(lambda p: space(p, rays, times.cuda()), motion, "cuda", keep)
    loss.backward()
    ps = {"s." + k: v.double().requires_grad_(True) for k, v in sd_s.items()}
    pm = {"m." + k: v.double().requires_grad_(True) for k, v in sd_m.items()}
    ref = loss_of(lambda p: O.space_net({k[2:]: v for k, v in ps.items()}, "net", p, dirs.double(), times.double()),
                  lambda x: O.motion_net({k[2:]: v for k, v in pm.items()}, "net", x.double()), "cpu", keep)
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref)) + 1e-7
    for k, p in motion.named_parameters():
        r = pm["m.net." + k].grad
        assert float((p.grad.double().cpu() - r).abs().max()) <= 5e-5 * float(r.abs().max()) + 1e-12, k
    for k, p in space.named_parameters():
        r = ps["s.net." + k].grad
        assert float((p.grad.double().cpu() - r).abs().max()) <= 5e-5 * float(r.abs().max()) + 1e-12, k
    opt = torch.optim.SGD(list(space.parameters()) + list(motion.parameters()), lr=1e-2)
    opt.step()
    with torch.no_grad():
        after = loss_of(lambda p: space(p, rays, times.cuda()), motion, "cuda", keep)
    assert float(after) < float(loss)


# ---- round 5: the fused launches (csrc/train_wave.hip, csrc/mlp_wave.hip StoreTap) ------------------------------------------------
def _space_case(use_time, n=37, ns=19, seed=0):
    from stnerf_amd.modeling.spacenet import SpaceNet
    rs = np.random.RandomState(seed)
    net = SpaceNet(use_time=use_time)
    sd = syn.spacenet_state("net", rs, use_time)
    net.load_state_dict({k[4:]: v for k, v in sd.items()})
    net = net.cuda()
    g = torch.Generator().manual_seed(seed)
    pos = ((torch.rand(n, ns, 3, generator=g) - 0.5) * 4).cuda()
    rays = torch.cat([torch.zeros(n, 3), torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)], -1).cuda()
    tm = (torch.rand(n, 1, generator=g) * 20 + 1).cuda() if use_time else None
    net.oracle_state = sd
    return net, pos, rays, tm


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("use_time, n", [(False, 37), (True, 37), (True, 128), (True, 1)])
def test_fused_forward_writes_the_activations_the_layerwise_recompute_builds(ops, use_time, n, precision):
    """stnerf_train_spacenet_fwd[_bf16x3] = the inference stage kernel of that arithmetic + a tap: its outputs are stnerf_spacenet_fwd's
    bit for bit, and every layer input it writes out is what the per-layer recomputation (train_encode + train_linear_fwd, round 4,
    exact f32) produces -- within the rounding of an fp32 layer for both arithmetics -- with a ragged last work item (703 rows),
    with whole ones (2432) and with a single ray (19 rows: less than one wave)."""
    from stnerf_amd.modeling import autograd as A
    net, pos, rays, tm = _space_case(use_time, n=n)
    n, ns = pos.shape[0], pos.shape[1]
    M = n * ns
    packed = net._packed(precision)
    dir_w, time_w = 27, 21 if use_time else 0
    Cc, R, T0 = A._buf(M, 320, "cuda"), A._buf(M, 256 + dir_w + time_w, "cuda"), A._buf(M, 128, "cuda")
    Hs, Gs = [A._buf(M, 256, "cuda") for _ in range(3)], [A._buf(M, 256, "cuda") for _ in range(2)]
    for b in [Cc, R, T0] + Hs + Gs:
        b.fill_(float("nan"))
    acts = [Hs[0], Hs[1], Hs[2], Cc[:, :256], Gs[0], Gs[1], R[:, :256], T0]
    raw = torch.empty(n, ns, 4, device="cuda")
    bits = torch.full((8, M, 8), -1, dtype=torch.int32, device="cuda")
    ops.train_spacenet_fwd(packed, pos, rays[:, 3:6], tm.reshape(n) if use_time else None, raw, acts, Cc[:, 256:320], bits)
    want = torch.empty(n, ns, 4, device="cuda")
    ops.spacenet_fwd(packed, pos, rays[:, 3:6], tm.reshape(n) if use_time else None, want)
    assert torch.equal(raw, want)
    # without mask planes (relu_bits is optional): the same outputs and activations, bit for bit
    again = [torch.full_like(b, float("nan")) for b in [Cc, R, T0] + Hs + Gs]
    acts2 = [again[3], again[4], again[5], again[0][:, :256], again[6], again[7], again[1][:, :256], again[2]]
    raw2 = torch.empty_like(raw)
    ops.train_spacenet_fwd(packed, pos, rays[:, 3:6], tm.reshape(n) if use_time else None, raw2, acts2, again[0][:, 256:320], None)
    assert torch.equal(raw2, raw)
    for a_, b_ in zip(acts2, acts):
        assert torch.equal(a_[:, :128], b_[:, :128]) and torch.equal(a_, b_)
    assert torch.equal(again[0][:, 256:320], Cc[:, 256:320])
    # nothing beyond a matrix's 256 columns was touched (rgb_net.1's input keeps its direction / time columns for train_encode)
    assert bool(torch.isnan(R[:, 256:]).all())
    # the round-4 recomputation, layer by layer
    params = [p.detach() for p in net.training_parameters()]
    W = [A._padded_weight(params[2 * i]) for i in range(10)]
    B = [params[2 * i + 1].float().contiguous() for i in range(10)]
    x = pos.reshape(M, 3)
    C2 = A._buf(M, 320, "cuda")
    P = C2[:, 256:319]
    ops.train_encode(x, P, 10, True)
    assert torch.allclose(Cc[:, 256:319], P, rtol=0, atol=2e-7) and float(Cc[:, 319].abs().max()) == 0.0
    h = [A._buf(M, 256, "cuda") for _ in range(3)]
    ops.train_linear_fwd(P, W[0], B[0], h[0][:, :256], True)
    ops.train_linear_fwd(h[0][:, :256], W[1], B[1], h[1][:, :256], True)
    ops.train_linear_fwd(h[1][:, :256], W[2], B[2], h[2][:, :256], True)
    ops.train_linear_fwd(h[2][:, :256], W[3], B[3], C2[:, :256], True)
    g_ = [A._buf(M, 256, "cuda") for _ in range(3)]
    ops.train_linear_fwd(C2[:, :319], W[4], B[4], g_[0][:, :256], True)
    ops.train_linear_fwd(g_[0][:, :256], W[5], B[5], g_[1][:, :256], True)
    ops.train_linear_fwd(g_[1][:, :256], W[6], B[6], g_[2][:, :256], True)
    for got, ref in zip(acts[:7], [h[0], h[1], h[2], C2[:, :256], g_[0], g_[1], g_[2]]):
        assert torch.allclose(got[:, :256], ref[:, :256], rtol=1e-5, atol=1e-6)
        assert float(got[:, :256].min()) >= 0.0
    assert bool(torch.isfinite(T0).all()) and float(T0.min()) >= 0.0 and float(T0.max()) > 0.0
    # the ReLU masks as bit planes: row r, stage s: 8 words; the lane (h, r % 32) of the wave that owns the row holds register i of block
    # fb <-> column 32 fb + 8 (i >> 2) + 4 h + (i & 3), stored as bit (16 fb + i) & 31 of word 4 h + (fb >> 1)
    fb, i, h = torch.meshgrid(torch.arange(8), torch.arange(16), torch.arange(2), indexing="ij")
    col = (32 * fb + 8 * (i >> 2) + 4 * h + (i & 3)).reshape(-1)
    word, bit = (4 * h + (fb >> 1)).reshape(-1), ((16 * fb + i) & 31).reshape(-1)
    for s_, act in enumerate(acts):
        width = act.shape[1] if s_ < 7 else 128
        keep = col < width
        got = (bits[s_].cpu()[:, word[keep]] >> bit[keep]) & 1
        assert torch.equal(got.bool(), act[:, :width].cpu()[:, col[keep]] > 0), s_


@pytest.mark.parametrize("use_time, want_dpos", [(True, True), (False, True), (True, False)])
def test_fused_backward_matches_the_layerwise_backward(ops, monkeypatch, use_time, want_dpos):
    """Same gradients from the two fused launches + per-layer dW as from the round-4 chain of per-layer GEMMs (every weight, bias and
    the sample points), several chunks with a ragged last one."""
    from stnerf_amd.modeling import autograd as A
    net, pos, rays, tm = _space_case(use_time, n=301, ns=23, seed=3)
    monkeypatch.setattr(A, "CHUNK_SAMPLES", 2048)
    g = torch.Generator().manual_seed(9)
    # (the two recomputations round differently: a hidden unit within fp32 rounding of zero passes its cotangent in one and not in the
    # other -- such samples get a zero cotangent, as in the fp64 comparisons above)
    sd64 = {k: v.double() for k, v in net.oracle_state.items()}
    safe = _safe_samples(lambda: O.space_net(sd64, "net", pos.cpu().double(), rays[:, 3:6].cpu().double(),
                                             tm.cpu().double() if use_time else None)).reshape(301, 23, 1).cuda()
    c_rgb, c_sig = torch.randn(301, 23, 3, generator=g).cuda() * safe, torch.randn(301, 23, 1, generator=g).cuda() * safe
    grads = {}
    for fused in (True, "recompute", False):      # activations kept by the forward / recomputed chunk by chunk / the per-layer GEMM chain
        monkeypatch.setattr(A, "FUSED_BACKWARD", bool(fused))
        monkeypatch.setattr(A, "KEEP_BYTES", 0 if fused == "recompute" else 1 << 35)
        net.zero_grad(set_to_none=True)
        p = pos.clone().requires_grad_(want_dpos)
        rgb, sig = net(p, rays, tm)
        ((rgb * c_rgb).sum() + (sig * c_sig).sum()).backward()
        grads[fused] = {k: v.grad.clone() for k, v in net.named_parameters()}
        grads[fused]["pos"] = p.grad.clone() if want_dpos else None
    for k, a in grads[True].items():
        b, r = grads[False][k], grads["recompute"][k]
        if a is None:
            assert b is None and r is None
            continue
        assert float((a - b).abs().max()) <= GRAD_RTOL * float(b.abs().max()), k
        assert torch.equal(a, r), k                     # the same kernels on the same rows: bit-identical


def _motion_case(input_time, rows=1237, seed=4):
    from stnerf_amd.modeling.motion_net import MotionNet
    sd = syn.motionnet_state("net", np.random.RandomState(seed))
    net = MotionNet(c_input=4, input_time=input_time)
    net.load_state_dict({k[4:]: v for k, v in sd.items()})
    net = net.cuda()
    g = torch.Generator().manual_seed(seed)
    xyz = (torch.rand(rows, 3, generator=g) - 0.5) * 4.0
    t = torch.where(torch.rand(rows, 1, generator=g) < 0.5, torch.floor(torch.rand(rows, 1, generator=g) * 30), torch.rand(rows, 1, generator=g) * 30) + 1
    net.oracle_state = sd
    return net, torch.cat([xyz, t], -1).cuda()


@pytest.mark.parametrize("input_time", [True, False])
def test_fused_motionnet_forward_writes_what_the_layerwise_recompute_builds(ops, input_time):
    """stnerf_train_motionnet_fwd: the flow of stnerf_motionnet_fwd (same arithmetic: within fp32 rounding of the stand-alone kernel's
    different summation order), and every matrix it writes -- the staged encoding with the fractional-time blend, the five post-ReLU
    outputs, their masks as bit planes -- against the per-layer recomputation (train_encode + train_linear_fwd)."""
    from stnerf_amd.modeling import autograd as A
    net, xt = _motion_case(input_time)
    rows = xt.shape[0]
    packed = net._packed("fp32")
    bufs = A._motion_buffers(rows, "cuda")
    for b in bufs[:6]:
        b.fill_(float("nan"))
    bufs[6].fill_(-1)
    flow = torch.full((rows, 3), float("nan"), device="cuda")
    acts = [b[:, :128] for b in bufs[1:6]]
    ops.train_motionnet_fwd(packed, xt, flow, bufs[0], acts, bufs[6], plain_time=not input_time)
    want = net(xt.detach())
    assert torch.allclose(flow, want, rtol=1e-5, atol=2e-7)
    params = [p.detach() for p in net.parameters()]
    W = [A._padded_weight(params[2 * i]) for i in range(6)]
    B = [params[2 * i + 1].float().contiguous() for i in range(6)]
    E = A._buf(rows, 84, "cuda")
    ops.train_encode(xt, E[:, :84], 10, True, lerp_col=3 if input_time else -1)
    assert torch.allclose(bufs[0][:, :84], E[:, :84], rtol=0, atol=2e-7) and float(bufs[0][:, 84:88].abs().max()) == 0.0
    src = E[:, :84]
    for j in range(5):
        ref = A._buf(rows, 128, "cuda")
        ops.train_linear_fwd(src, W[j], B[j], ref[:, :128], True)
        assert torch.allclose(acts[j], ref[:, :128], rtol=1e-5, atol=1e-6), j
        assert float(acts[j].min()) >= 0.0 and float(acts[j].max()) > 0.0
        src = ref[:, :128]
    # bit (16 fb + i) & 31 of word 2 h + (fb >> 1) <-> column 32 fb + 8 (i >> 2) + 4 h + (i & 3)
    fb, i, h = torch.meshgrid(torch.arange(4), torch.arange(16), torch.arange(2), indexing="ij")
    col = (32 * fb + 8 * (i >> 2) + 4 * h + (i & 3)).reshape(-1)
    word, bit = (2 * h + (fb >> 1)).reshape(-1), ((16 * fb + i) & 31).reshape(-1)
    for s_, act in enumerate(acts):
        got = (bufs[6][s_].cpu()[:, word] >> bit) & 1
        assert torch.equal(got.bool(), act.cpu()[:, col] > 0), s_


@pytest.mark.parametrize("input_time, want_dx", [(True, True), (False, True), (True, False)])
def test_fused_motionnet_backward_matches_the_layerwise_backward(ops, monkeypatch, input_time, want_dx):
    """Same gradients from the fused launches as from the per-layer GEMM chain (every weight, bias and the points), several chunks with
    a ragged last one; activations kept by the forward and recomputed: the same bits."""
    from stnerf_amd.modeling import autograd as A
    net, xt = _motion_case(input_time, rows=5003, seed=6)
    monkeypatch.setattr(A, "CHUNK_SAMPLES", 2048)
    g = torch.Generator().manual_seed(2)
    sd64 = {k: v.double() for k, v in net.oracle_state.items()}
    safe = _safe_samples(lambda: O.motion_net(sd64, "net", xt.cpu().double(), input_time=input_time)).reshape(-1, 1).cuda()
    cot = torch.randn(xt.shape[0], 3, generator=g).cuda() * safe
    grads = {}
    for fused in (True, "recompute", False):
        monkeypatch.setattr(A, "FUSED_BACKWARD", bool(fused))
        monkeypatch.setattr(A, "KEEP_BYTES", 0 if fused == "recompute" else 1 << 35)
        net.zero_grad(set_to_none=True)
        x = xt.clone().requires_grad_(want_dx)
        (net(x) * cot).sum().backward()
        grads[fused] = {k: v.grad.clone() for k, v in net.named_parameters()}
        grads[fused]["x"] = x.grad.clone() if want_dx else None
    for k, a in grads[True].items():
        b, r = grads[False][k], grads["recompute"][k]
        if a is None:
            assert b is None and r is None
            continue
        assert float((a - b).abs().max()) <= GRAD_RTOL * float(b.abs().max()), k
        assert torch.equal(a, r), k


def test_pack_transposed_builds_the_sections_torch_builds(ops):
    """stnerf_pack_transposed (one launch per network after every optimizer.step()): [out / 4][n_pad][4] sections with zero padding, a
    column block of a wider weight as the source, heads copied as they are -- against zeros / copy / permute in torch."""
    g = torch.Generator().manual_seed(5)
    wide = torch.randn(128, 304, generator=g).cuda()
    ws = [(wide[:, :256], 256), (torch.randn(256, 63, generator=g).cuda(), 64), (torch.randn(256, 319, generator=g).cuda(), 320),
          (torch.randn(128, 84, generator=g).cuda(), 128), (torch.randn(1, 256, generator=g).cuda(), 0), (torch.randn(3, 128, generator=g).cuda(), 0)]
    total = sum(w.shape[0] * (p if p else w.shape[1]) for w, p in ws)
    dst = torch.full((total + 64,), float("nan"), device="cuda")
    offsets = ops.pack_transposed(ws, dst)
    off = 0
    for (w, p), o in zip(ws, offsets):
        assert o == off
        if p:
            wp = torch.zeros(w.shape[0], p, device="cuda")
            wp[:, :w.shape[1]] = w
            want = wp.reshape(w.shape[0] // 4, 4, p).permute(0, 2, 1).contiguous().reshape(-1)
        else:
            want = w.reshape(-1)
        assert torch.equal(dst[off:off + want.numel()], want)
        off += want.numel()
    assert bool(torch.isnan(dst[off:]).all())
    with pytest.raises(ValueError):
        ops.pack_transposed(ws, dst[:100])


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("kind", ["space", "space_time", "space_time_deep", "space_noinc", "motion"])
def test_device_packer_writes_the_host_packers_blob(ops, kind, precision):
    """stnerf_pack_net_device / stnerf_pack_net_bf16x3_device (what a training loop calls after every optimizer.step(): no host round
    trip) write the blobs of stnerf_pack_net / stnerf_pack_net_bf16x3 bit for bit, for every network flavour."""
    rs = np.random.RandomState(11)
    if kind == "motion":
        sd = syn.motionnet_state("net", rs)
        pack = ops.pack_motionnet
    else:
        sd = syn.spacenet_state("net", rs, "time" in kind, deep_rgb="deep" in kind, include_input="noinc" not in kind)
        pack = ops.pack_spacenet
    host = pack(sd, "net", "cuda", precision)
    dev = pack({k: v.cuda() for k, v in sd.items()}, "net", "cuda", precision)
    assert dev.blob.data_ptr() % (1024 if precision == "bf16x3" else 16) == 0
    assert host.kind == dev.kind and host.blob.shape == dev.blob.shape
    assert torch.equal(host.blob.view(torch.int32), dev.blob.view(torch.int32))


@pytest.mark.parametrize("use_time, n, want_dpos", [(True, 37, True), (False, 37, False), (True, 128, True), (True, 1, True), (False, 7, False)])
def test_split_bf16_backward_chain_matches_the_exact_f32_chain(ops, use_time, n, want_dpos):
    """stnerf_train_spacenet_dx_bf16x3 against stnerf_train_spacenet_dx on the same cotangent and the same ReLU masks (a split-bf16 forward
    tap's): every layer's pre-activation gradient within 2e-5 of the matrix's largest entry (both are fp32-faithful evaluations of the
    same chain: measured ~1e-6), exact zeros where the mask is zero in both, d PE(pos) = dpe + dpe_skip likewise; a ragged last work
    item (703 rows), whole ones (2432), 19 and 133 rows; rows past the end untouched."""
    from stnerf_amd.modeling import autograd as A
    net, pos, rays, tm = _space_case(use_time, n=n, seed=5)
    n, ns = pos.shape[0], pos.shape[1]
    M = n * ns
    bufs = A._activation_buffers(M, 48, "cuda")
    raw = torch.empty(n, ns, 4, device="cuda")
    ops.train_spacenet_fwd(net._packed("bf16x3"), pos, rays[:, 3:6], tm.reshape(n) if use_time else None, raw, A._act_views(bufs),
                           bufs[0][:, 256:320], bufs[8])
    g = torch.Generator().manual_seed(11)
    d_raw = torch.randn(M, 4, generator=g).cuda()
    params = [p.detach() for p in net.training_parameters()]
    wt, offsets = A.transposed_spacenet(net, params)
    mk = lambda: [torch.full((M + 3, 256), float("nan"), device="cuda")[:M, :256] for _ in range(7)] + [torch.full((M + 3, 128), float("nan"), device="cuda")[:M]]
    want, got = mk(), mk()
    dpe_w = torch.empty(M, 64, device="cuda") if want_dpos else None
    ops.train_spacenet_dx(wt, offsets, d_raw, bufs[8], want, dpe_w)
    blob = A.dx_blob_bf16x3(net, params, want_dpos)
    dpe, dpe_skip = (torch.empty(M, 64, device="cuda"), torch.empty(M, 64, device="cuda")) if want_dpos else (None, None)
    ops.train_spacenet_dx_bf16x3(blob, d_raw, bufs[8], got, dpe, dpe_skip)
    torch.cuda.synchronize()
    for s_, (a_, b_) in enumerate(zip(got, want)):
        assert bool(torch.isfinite(a_).all()), s_
        scale = float(b_.abs().max())
        assert scale > 0 and float((a_ - b_).abs().max()) <= 2e-5 * scale, (s_, float((a_ - b_).abs().max()), scale)
        assert torch.equal(a_ == 0, b_ == 0) or float(((a_ == 0) != (b_ == 0)).float().mean()) < 1e-4, s_
        assert bool(torch.isnan(a_._base[M:]).all()) if a_._base is not None else True      # (rows past the end: untouched)
    if want_dpos:
        total = dpe + dpe_skip
        scale = float(dpe_w.abs().max())
        assert float((total - dpe_w).abs().max()) <= 2e-5 * scale, (float((total - dpe_w).abs().max()), scale)
        assert float(dpe_w[:, 63].abs().max()) == 0.0 and float(total[:, 63].abs().max()) == 0.0      # (the pad column)


def test_device_bf16x3_packer_refuses_unsplittable_weights_when_packing_to_render(ops):
    """A model whose tensors live on the GPU is packed there; packing to render, the packer refuses what the host packer refuses -- NaN,
    inf, |w| > 3.3895e38 (include/stnerf.h) -- with a ValueError; inside a training step (ops.training_pack(), modeling/autograd.py) it
    does not synchronise and packs (NaN pieces: a NaN loss)."""
    sd = {k: v.cuda() for k, v in syn.spacenet_state("net", np.random.RandomState(3), True).items()}
    if True:
        ops.pack_spacenet(sd, "net", "cuda", "bf16x3")
        for bad in (float("nan"), float("inf"), -3.39e38):
            sd2 = dict(sd)
            w = sd["net.stage1.4.weight"].clone()
            w[5, 7] = bad
            sd2["net.stage1.4.weight"] = w
            with pytest.raises(ValueError, match="not finite or exceeds"):
                ops.pack_spacenet(sd2, "net", "cuda", "bf16x3")
            ops.pack_spacenet(sd2, "net", "cuda", "fp32")          # (the exact-f32 packing takes any value)
    with ops.training_pack():
        ops.pack_spacenet(sd2, "net", "cuda", "bf16x3")
