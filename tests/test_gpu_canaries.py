"""Overrun canaries (the substitute for a race / bounds checker on this path): every device buffer
the op wrappers allocate for a kernel to write -- sampler, ray generation, compaction, compositor (all three kernels),
resampler (every flavour), the network stages and the whole pipeline's outputs and workspace -- is placed between two 4 KB
guard zones filled with a NaN bit pattern; after the launch every guard must be intact.  Shapes are chosen ragged (ray
counts that are not a multiple of the wave / block size, sample counts that leave partial 64-lane blocks) because that is
where an unguarded tail store would land.  Needs an MI355X: `pytest -m gpu`."""
import math

import numpy as np
import pytest
import torch

from stnerf_amd import synthetic as syn

pytestmark = pytest.mark.gpu

GUARD_BYTES = 4096
PATTERN = 0x7FC00A5A                      # a quiet-NaN payload no kernel produces


class Canaries:
    """While active, ``torch.empty`` / ``torch.zeros`` / ``torch.full`` on the GPU return the inside of a guarded buffer."""

    def __init__(self):
        self.buffers = []

    def _guarded(self, shape, dtype, device):
        n = int(math.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        body = (n + 255) // 256 * 256                      # the inner view keeps the allocator's 256-byte alignment
        buf = self._empty(body + 2 * GUARD_BYTES, dtype=torch.uint8, device=device)
        buf.view(torch.int32).fill_(PATTERN)
        self.buffers.append((buf, n))
        return buf[GUARD_BYTES:GUARD_BYTES + n].view(dtype).reshape(shape)

    def _wrap(self, orig, fill):
        def alloc(*args, dtype=None, device=None, **kw):
            if device is None or torch.device(device).type != "cuda" or kw:
                return orig(*args, dtype=dtype, device=device, **kw)          # host tensors / exotic calls: untouched
            size, extra = (args[:-1], args[-1:]) if fill == "full" else (args, ())
            if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
                size = tuple(size[0])
            t = self._guarded(tuple(int(x) for x in size), dtype or torch.float32, device)
            if fill == "zeros":
                t.zero_()
            elif fill == "full":
                t.fill_(extra[0])
            return t
        return alloc

    def __enter__(self):
        self._empty, self._zeros, self._full = torch.empty, torch.zeros, torch.full
        torch.empty, torch.zeros, torch.full = self._wrap(self._empty, "empty"), self._wrap(self._zeros, "zeros"), self._wrap(self._full, "full")
        return self

    def __exit__(self, *exc):
        torch.empty, torch.zeros, torch.full = self._empty, self._zeros, self._full

    def check(self, what):
        torch.cuda.synchronize()
        assert self.buffers, what
        for buf, n in self.buffers:
            words = buf.view(torch.int32)
            head = words[: GUARD_BYTES // 4]
            tail_start = (GUARD_BYTES + n + 3) // 4          # first whole word behind the body
            tail = words[tail_start:]
            assert bool((head == PATTERN).all()), f"{what}: a kernel wrote IN FRONT of a {n}-byte buffer"
            assert bool((tail == PATTERN).all()), f"{what}: a kernel wrote BEHIND a {n}-byte buffer"
        count = len(self.buffers)
        self.buffers = []
        return count


@pytest.fixture(scope="module")
def ops():
    from stnerf_amd import ops as _ops
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return _ops


def _rays(n, L, seed=0):
    g = torch.Generator().manual_seed(seed)
    o = torch.tensor([0.0, 0.0, -4.0]) + 0.05 * torch.randn(n, 3, generator=g)
    d = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, 1.0]) + 0.35 * torch.randn(n, 3, generator=g), dim=-1)
    return torch.cat([o, d, syn.frame_id_columns(n, L)], -1).cuda()


def _boxes(L):
    bk, per = syn.scene_boxes(L)
    return torch.cat([bk, per[1]], 0).cuda()


@pytest.mark.parametrize("n, L, n1", [(1, 1, 3), (63, 2, 13), (257, 2, 64), (1000, 4, 90), (333, 8, 128), (77, 2, 200)])
def test_sampler_ray_generation_and_compaction_stay_inside_their_buffers(ops, n, L, n1):
    with Canaries() as c:
        K, T = syn.camera(7, 11, 20.0)
        ops.generate_rays(K, T, 7, 11, frame_ids=[1.0] + [2.5] * L)
        ops.generate_rays(K, T, 7, 11, frame_ids=[1.0] + [2.5] * L, first_ray=11, stripe=11, period=33)
        assert c.check("generate_rays") == 2
        rays, boxes = _rays(n, L, n), _boxes(L)
        ops.intersect(rays, boxes)
        t, xyz, mask = ops.sample_coarse(rays, boxes, n1, seed=3)
        ops.sample_coarse(rays, boxes, n1, seed=3, want_xyz=False, edits=[(None, None)] + [([0.1, 0.0, 0.0], 1.1)] * L,
                          pivot=torch.tensor([0.1, 0.2, 0.3]))
        ops.compact_rays(mask)
        assert c.check(f"sampler n={n} L={L} n1={n1}") >= 8
    assert bool(torch.isfinite(t).all()) and int(mask[:, 0].sum()) > 0


@pytest.mark.parametrize("n, l, S", [(1, 1, 3), (65, 1, 64), (130, 2, 90), (1023, 3, 128), (500, 3, 192), (90, 5, 64), (70, 9, 16),
                                     (40, 2, 300), (64, 16, 192)])
@pytest.mark.parametrize("fine", [False, True])
def test_compositor_stays_inside_its_buffers(ops, n, l, S, fine):
    torch.manual_seed(n + S)
    t = torch.sort(torch.rand(n, l, S) * 6.0, -1)[0]
    mask = (torch.rand(n, l) < 0.6).to(torch.uint8)
    mask[:, 0] = 1
    t[~mask.bool()] = -1000.0
    raw = torch.randn(n, l, S, 4)
    kw = dict(near=0.2, fine=fine, cut_negative_t=not fine, thresholds=[0.1] * l, evaluated=[2] + [1] * (l - 1))
    with Canaries() as c:
        td, rd, md = t.cuda(), raw.cuda(), mask.cuda()
        c.buffers = []                                           # (inputs are not under test)
        for want_weights, want_order, two_pass in [(True, False, True), (False, False, True), (True, True, False), (False, False, False)]:
            lo, mo, w, od = ops.composite(td, rd, md, want_weights=want_weights, want_order=want_order, two_pass=two_pass, **kw)
            c.check(f"composite n={n} l={l} S={S} fine={fine} weights={want_weights} order={want_order} two_pass={two_pass}")
    assert lo.shape == (n, l, 5) and mo.shape == (n, 5)


@pytest.mark.parametrize("n, l, n1, n2", [(1, 1, 3, 2), (65, 2, 64, 64), (300, 3, 90, 30), (129, 2, 128, 64), (50, 9, 12, 4), (33, 2, 200, 40),
                                          (20, 1, 300, 20), (10, 2, 512, 512), (64, 2, 64, 0), (257, 3, 10, 64)])
def test_resampler_stays_inside_its_buffers(ops, n, l, n1, n2):
    torch.manual_seed(n1 + n2)
    t = torch.sort(torch.rand(n, l, n1) * 5.0, -1)[0]
    w = torch.rand(n, l, n1) ** 8
    rays = _rays(n, 1, 5)
    with Canaries() as c:
        td, wd = t.cuda(), w.cuda()
        u = torch.rand(l, n, n2).cuda()
        c.buffers = []
        tf, xyz = ops.resample(td, wd, n2, rays, seed=4)                                   # production flavour (device RNG)
        c.check(f"resample {n1}+{n2} plain")
        ops.resample(td, wd, n2, rays, seed=4, want_xyz=False)
        c.check(f"resample {n1}+{n2} no xyz")
        ops.resample(td, wd, n2, rays, u=u, debug=True, edits=[([0.1, 0.0, 0.0], 1.2)] * l, pivot=torch.tensor([0.0, 0.1, 0.2]))
        c.check(f"resample {n1}+{n2} debug")
    assert bool((tf[..., 1:] >= tf[..., :-1]).all())


@pytest.mark.parametrize("precision", ["bf16x3", "fp32"])
@pytest.mark.parametrize("n, ns", [(1, 3), (37, 64), (200, 90), (129, 192)])
def test_network_stage_stays_inside_its_buffers(ops, precision, n, ns):
    rs = np.random.RandomState(n)
    sd_b, sd_p, sd_m = syn.spacenet_state("net", rs, False), syn.spacenet_state("net", rs, True), syn.motionnet_state("net", rs)
    torch.manual_seed(n)
    xyz = ((torch.rand(n, 2, ns, 3) - 0.5) * 5.0).cuda()
    rays = _rays(n, 1, 9)
    mask = (torch.rand(n, 2) < 0.5).to(torch.uint8).cuda()
    mask[0, 1] = 1
    bk, sp, mo = (ops.pack_spacenet(sd_b, "net", precision=precision), ops.pack_spacenet(sd_p, "net", precision=precision),
                  ops.pack_motionnet(sd_m, "net", precision=precision))
    with Canaries() as c:
        lst, cnt = ops.compact_rays(mask)
        raw = torch.full((n, 2, ns, 4), 7.0, device="cuda")
        ops.mlp_stage([dict(space=sp, motion=mo, xyz=xyz[:, 1], raw=raw[:, 1], times=rays[:, 7], ray_list=lst[1], ray_count=cnt[1:2]),
                       dict(space=bk, motion=None, xyz=xyz[:, 0], raw=raw[:, 0], times=None, plain_time=True)], rays[:, 3:6], ns)
        c.check(f"mlp_stage {precision} n={n} ns={ns}")
    assert bool(torch.isfinite(raw).all()) and bool((raw[~mask.bool()[:, 1], 1] == 7.0).all())


@pytest.mark.parametrize("precision", ["bf16x3", "fp32"])
@pytest.mark.parametrize("h, w, L, n1, n2, only_coarse", [(5, 7, 1, 8, 0, False), (9, 13, 2, 64, 64, False), (6, 11, 2, 90, 30, False),
                                                          (4, 9, 4, 12, 6, True), (3, 23, 8, 16, 8, False)])
def test_whole_pipeline_stays_inside_its_outputs_and_workspace(precision, h, w, L, n1, n2, only_coarse):
    """stnerf_render_rays: the five output tensors AND the caller-provided workspace (whose size stnerf_render_workspace_bytes
    promises is enough -- the fine network outputs reuse the dead coarse block) between guards."""
    import test_gpu_render as R
    from stnerf_amd import ops
    meta = dict(L=L, n1=n1, n2=n2, space_time=True, deform_time=True, weight_seed=90 + L, edit={})
    model = R.build_model(meta).set_precision(precision)
    K, T = syn.camera(h, w, 14.0)
    rays = ops.generate_rays(K, T, h, w, frame_ids=[1.0] + [2.5] * L)
    model._workspace = None
    with Canaries() as c, torch.no_grad():
        out = model(rays, None, None, only_coarse=only_coarse)
        n_buf = c.check(f"render_rays {precision} {h}x{w} L={L} {n1}+{n2}")
    assert n_buf >= (4 if only_coarse else 6)                      # mask + coarse (+ fine) outputs + the workspace
    assert bool(torch.isfinite(out[0][0]).all()) and out[0][0].shape == (h * w, 3)
