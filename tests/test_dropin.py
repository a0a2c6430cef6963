"""The drop-in boundary, executed against the REAL reference tree.

Runs in the build container only (``/root/reference`` does not exist on the GPU box).  Only the third-party
packages the reference imports but this image lacks (yacs, torchvision, imageio, kornia, robopy, open3d, pyrender)
are stubbed; every ``modeling`` / ``utils`` / ``layers`` / ``render`` / ``data`` / ``config`` module below is the
reference's own file.

1. after ``stnerf_amd.dropin.patch_reference()`` the import statements of the reference's own callers
   (render/layered_neural_renderer.py:1-10, demo/taekwondo_demo.py:16-23) all resolve, the render-path symbols
   among them to this framework, the rest (``add_two_dim_dict``, ``make_loss``, ``setup_logger`` ...) to the reference;
2. the reference's OWN ``LayeredNeuralRenderer.render_pose`` (render/layered_neural_renderer.py:364-391, the same
   function object before and after patching) produces the same images when it drives the reference's model and
   when it drives this framework's ``LayeredRFRender`` host logic.  There is no GPU here, so in the second run the
   ONE call that leaves the host (``_render_launch`` -> ``stnerf_render_rays``) is answered by the CPU oracle with
   the boxes, pivot, thresholds and launch pieces this framework's host code computed;
   tests/test_gpu_render.py::test_reference_render_pose_call_sequence drives the same call sequence into the HIP
   library on the GPU box.
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

import pytest
import torch

REFERENCE = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "modeling")),
                                reason="the reference checkout is only present in the build container")

ABSENT_THIRD_PARTY = ("torchvision", "imageio", "kornia", "robopy", "open3d", "pyrender", "cv2")
REFERENCE_TOP_LEVEL = ("config", "data", "engine", "layers", "modeling", "render", "solver", "utils")


class _Anything:
    """Placeholder for whatever an absent third-party package would have exported (callable, attribute-able)."""

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return self


class _StubModule(types.ModuleType):
    """Stand-in for an absent third-party package: any attribute is a placeholder, ``import *`` exports nothing."""
    __all__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        top = fullname.split(".")[0]
        if top in ABSENT_THIRD_PARTY:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


@pytest.fixture()
def reference_env(monkeypatch):
    """The reference importable with its absent third-party deps stubbed; everything is undone afterwards (the
    reference's package names ``utils`` / ``config`` / ... must not leak into the other test modules)."""
    import stnerf_amd.dropin as dropin
    for name in ABSENT_THIRD_PARTY:
        assert importlib.util.find_spec(name) is None, f"{name} is installed: drop it from ABSENT_THIRD_PARTY"
    before = set(sys.modules)
    finder = _StubFinder()
    sys.meta_path.append(finder)
    monkeypatch.setattr(sys, "dont_write_bytecode", True)       # /root/reference is read-only by contract
    monkeypatch.syspath_prepend(REFERENCE)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)   # no GPU here (SURVEY 8c shim 2)
    shim = dropin.install_yacs_shim()
    yield dropin
    dropin.unpatch_reference()
    sys.meta_path.remove(finder)
    for name in set(sys.modules) - before:
        top = name.split(".")[0]
        if top in ABSENT_THIRD_PARTY or top in REFERENCE_TOP_LEVEL or (shim and top == "yacs"):
            del sys.modules[name]


def _is_ours(obj):
    return obj.__module__.startswith("stnerf_amd.")


def test_reference_callers_import_after_patch(reference_env):
    dropin = reference_env
    patch = dropin.patch_reference(REFERENCE)
    assert patch.rebound, "nothing was rebound"
    ns = {}
    # render/layered_neural_renderer.py:1-10 (importing the module runs exactly those statements)
    lnr = importlib.import_module("render.layered_neural_renderer")
    assert lnr.__file__.startswith(REFERENCE)
    assert _is_ours(lnr.build_layered_model) and _is_ours(lnr.layered_batchify_ray)
    assert lnr.add_two_dim_dict.__module__ == "utils.high_dim_dics"
    assert lnr.get_iteration_path.__module__ == "data.datasets.utils"
    # demo/taekwondo_demo.py:16-23
    for stmt in ("from config import cfg",
                 "from engine.layered_trainer import do_train",
                 "from solver import make_optimizer, WarmupMultiStepLR,build_scheduler",
                 "from layers import make_loss",
                 "from utils.logger import setup_logger",
                 "from layers.RaySamplePoint import RaySamplePoint",
                 "from utils import batchify_ray, vis_density",
                 "from render import LayeredNeuralRenderer"):
        exec(stmt, ns)
    assert _is_ours(ns["RaySamplePoint"])
    for name in ("do_train", "make_loss", "setup_logger", "batchify_ray", "vis_density"):
        assert not _is_ours(ns[name]) and sys.modules[ns[name].__module__].__file__.startswith(REFERENCE), name
    assert ns["LayeredNeuralRenderer"] is lnr.LayeredNeuralRenderer
    assert ns["LayeredNeuralRenderer"].__module__ == "render.layered_neural_renderer"
    # the trainer's evaluator binds the renderer by name too (engine/layered_trainer.py:36)
    trainer = sys.modules["engine.layered_trainer"]
    assert _is_ours(trainer.layered_batchify_ray)
    # every symbol of SURVEY 8(b) now resolves to this framework through the reference's own package names
    import layers
    import modeling
    import utils
    for obj in (modeling.build_layered_model, modeling.LayeredRFRender, modeling.spacenet.SpaceNet,
                modeling.motion_net.MotionNet, utils.layered_batchify_ray, utils.sample_pdf,
                utils.Trigonometric_kernel, layers.RaySamplePoint, layers.VolumeRenderer,
                layers.render_layer.gen_weight, utils.batchify_rays.layered_batchify_ray):
        assert _is_ours(obj), obj
    # ... and the rest of those packages is untouched
    assert utils.generate_rays.__module__ == "utils.render_helpers"      # CPU ray generation unless asked for
    assert utils.add_two_dim_dict.__module__ == "utils.high_dim_dics"
    # undo restores the originals
    patch.undo()
    assert not _is_ours(modeling.build_layered_model) and not _is_ours(lnr.layered_batchify_ray)


def test_patched_models_are_inference_only_and_draw_fresh_numbers(reference_env):
    """after patching, models built by the reference's code (a) advance their RNG seed on every forward, as the
    reference's torch.rand does, (b) refuse to run with autograd enabled on trainable parameters (no backward pass exists);
    undo() restores the class default."""
    from stnerf_amd.modeling.layered_rfrender import LayeredRFRender
    dropin = reference_env
    assert LayeredRFRender.FRESH_DRAWS_DEFAULT is False
    dropin.patch_reference(REFERENCE)
    assert LayeredRFRender.FRESH_DRAWS_DEFAULT is True
    import modeling
    m = modeling.build_layered_model(_cfg(8, 4, 2), camera_num=1)
    assert isinstance(m, LayeredRFRender) and m.fresh_draws_per_call is True
    rays = torch.zeros(4, 9)
    with pytest.raises(RuntimeError, match="GPU"):               # CPU tensors are refused ...
        m(rays)
    dropin.unpatch_reference()
    assert LayeredRFRender.FRESH_DRAWS_DEFAULT is False
    cfg = _our_cfg(8, 4, 2)
    cfg.MODEL.SAMPLE_METHOD, cfg.MODEL.POSE_REFINEMENT, cfg.MODEL.DEEP_RGB = "BBOX", False, False
    assert LayeredRFRender(cfg, camera_num=1).fresh_draws_per_call is False


def _our_cfg(n1, n2, L):
    from stnerf_amd.config.defaults import get_cfg_defaults
    cfg = get_cfg_defaults()
    cfg.MODEL.COARSE_RAY_SAMPLING, cfg.MODEL.FINE_RAY_SAMPLING, cfg.DATASETS.LAYER_NUM = n1, n2, L
    cfg.MODEL.USE_DEFORM_TIME = cfg.MODEL.USE_SPACE_TIME = True
    return cfg


def test_device_ray_generation_patch(reference_env):
    dropin = reference_env
    dropin.patch_reference(REFERENCE, device_ray_generation=True)
    import utils
    from data.datasets.ray_dataset import Ray_Dataset_Render
    assert _is_ours(utils.generate_rays) and _is_ours(utils.ray_sampling)
    assert Ray_Dataset_Render.get_rays_by_pose_and_K is dropin._device_rays_by_pose_and_K


def _scene(L):
    from stnerf_amd import synthetic as syn
    sd = syn.make_state_dict(L, True, True, seed=5)
    bk, per = syn.scene_boxes(L)
    return sd, bk, per


def _cfg(n1, n2, L):
    from config import cfg as ref_cfg                      # the reference's own defaults tree (config/defaults.py)
    cfg = ref_cfg.clone()
    cfg.merge_from_file(os.path.join(REFERENCE, "configs", "config_taekwondo.yml"))
    cfg.MODEL.COARSE_RAY_SAMPLING, cfg.MODEL.FINE_RAY_SAMPLING = n1, n2
    assert cfg.DATASETS.LAYER_NUM == L and cfg.MODEL.SAMPLE_METHOD == "BBOX"
    return cfg


class _RandReplay:
    """Feeds recorded uniforms to the reference's ``torch.rand`` calls of one chunk sequence: per chunk, l jitter
    tensors (layers/RaySamplePoint.py:98) then l resampling tensors (utils/sample_pdf.py:31)."""

    def __init__(self, jitter, u, chunk):
        self.jitter, self.u, self.chunk = jitter, u, chunk          # (l,N,N1), (l,N,N2)
        self.calls = 0

    def __call__(self, *shape, **kw):
        shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        l = self.jitter.shape[0]
        per_chunk = 2 * l
        c, k = divmod(self.calls, per_chunk)
        self.calls += 1
        src = self.jitter if k < l else self.u
        out = src[k % l, c * self.chunk: c * self.chunk + shape[0]]
        assert tuple(out.shape) == shape, (out.shape, shape)
        return out.clone()


def test_reference_render_pose_runs_through_this_framework(reference_env, monkeypatch):
    dropin = reference_env
    from oracle import stnerf_oracle as O
    from stnerf_amd import synthetic as syn
    L, n1, n2, H, W = 2, 8, 4, 48, 80                       # 3840 rays = one full 3584-ray chunk + a ragged tail
    l, N = L + 1, 48 * 80
    sd, bk, per = _scene(L)
    K, T = syn.camera(H, W, 12.0)
    pairs = [(0, 1), (1, 2.5), (2, 1)]                      # (layer, frame) pairs; a fractional retimed frame id
    g = torch.Generator().manual_seed(7)
    jitter, u = torch.rand(l, N, n1, generator=g), torch.rand(l, N, n2, generator=g)
    thr, bthr = 0.05, 0.02

    # ---- the reference, unpatched: its own model, batchify loop and render_pose -------------------------------
    lnr = importlib.import_module("render.layered_neural_renderer")
    ray_dataset = importlib.import_module("data.datasets.ray_dataset")
    render_pose = lnr.LayeredNeuralRenderer.render_pose
    cfg = _cfg(n1, n2, L)

    def make_renderer(model):
        ds = object.__new__(ray_dataset.Ray_Dataset_Render)   # the dataset object minus its file loading
        ds.height, ds.width, ds.layer_num = H, W, L
        ds.use_deform_time = ds.use_space_time = True
        ds.near_far = torch.tensor([[-1.0, -1.0]])
        r = object.__new__(lnr.LayeredNeuralRenderer)
        r.dataset, r.model, r.far = ds, model, 20.0
        return r

    import modeling
    assert not _is_ours(modeling.build_layered_model)
    ref_model = modeling.build_layered_model(cfg, camera_num=1).eval()
    ref_model.load_state_dict(sd)
    ref_model.set_bkgd_bbox(bk)
    ref_model.set_bboxes(per)
    ref_model.shift, ref_model.scale, ref_model.alpha = [[0.0, 0.0, 0.0], [0.1, 0.0, 0.05], None], [1.0, 1.1, 0.9], 0.7
    monkeypatch.setattr(torch, "rand", _RandReplay(jitter, u, 3584))
    want = render_pose(make_renderer(ref_model), T.numpy(), K, pairs, density_threshold=thr, bkgd_density_threshold=bthr)
    monkeypatch.undo()
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)

    # ---- patched: the same render_pose function object now reaches this framework ----------------------------
    dropin.patch_reference(REFERENCE)
    assert lnr.LayeredNeuralRenderer.render_pose is render_pose
    assert _is_ours(lnr.layered_batchify_ray) and _is_ours(modeling.build_layered_model)
    from stnerf_amd.modeling.layered_rfrender import LayeredRFRender
    launches = []

    class HostLogicOnCpu(LayeredRFRender):
        """This framework's model with the device check lifted and the one native call answered by the oracle."""

        def render_rays(self, rays, *a, **k):
            monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda t: True))
            try:
                return super().render_rays(rays, *a, **k)
            finally:
                monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda t: False))

        def _render_launch(self, rays, boxes, pivot, retiming, only_coarse, thr_, bthr_, index_base, replay):
            n = rays.shape[0]
            launches.append(dict(n=n, boxes=boxes.clone(), thr=thr_, bthr=bthr_, base=index_base, width=rays.shape[1],
                                 dtype=rays.dtype))
            om = O.OracleModel(layer_num=L, n_coarse=n1, n_fine=n2, params=self.state_dict(), bkgd_bbox=bk, bboxes=per,
                               near=self.near, alpha=self.alpha, scale=self.scale, shift=self.shift)
            bx = boxes.unsqueeze(0).expand(n, l, 8, 3) if boxes.dim() == 3 else boxes
            monkeypatch.setattr(O, "layer_boxes", lambda m, r: (bx.clone(), pivot, retiming, r[:, 6:] if retiming else r[:, -1]))
            draws = iter(list(replay["jitter"]) + list(replay["u"]))
            out = O.render_chunk(om, rays, only_coarse, thr_, bthr_, rand=lambda shape: next(draws))
            cat = lambda trip: torch.cat(list(trip), -1)
            return (cat(out[0]), cat(out[1]), torch.stack([cat(t) for t in out[2]], 1),
                    torch.stack([cat(t) for t in out[3]], 1), torch.stack(out[4], 1).to(torch.uint8))

    model = modeling.build_layered_model(cfg, camera_num=1).eval()        # the reference's name, this framework's class
    assert type(model) is LayeredRFRender
    model.__class__ = HostLogicOnCpu
    model.load_state_dict(sd)                                             # reference checkpoint key names
    model.set_bkgd_bbox(bk)
    model.set_bboxes(per)
    model.shift, model.scale, model.alpha = [[0.0, 0.0, 0.0], [0.1, 0.0, 0.05], None], [1.0, 1.1, 0.9], 0.7
    model.replay = {"jitter": jitter, "u": u}
    got = render_pose(make_renderer(model), T.numpy(), K, pairs, density_threshold=thr, bkgd_density_threshold=bthr)

    # what the reference's caller handed over is what this framework's entry point takes
    assert sum(x["n"] for x in launches) == N and all(x["width"] == 6 + l and x["dtype"] == torch.float32 for x in launches)
    assert all(x["thr"] == thr and x["bthr"] == bthr for x in launches)    # N >= one chunk: thresholds are passed on
    assert len(launches) == 1, "both reference chunks share row-0 frame ids: one merged launch"
    # same images out of the reference's own post-processing (reshape, depth clamp, / far)
    color, depth, color_layer, depth_layer = got
    assert color.shape == (H, W, 3) and depth.shape == (H, W, 1) and len(color_layer) == l and len(depth_layer) == l
    tol = dict(rtol=0, atol=2e-5)
    torch.testing.assert_close(color, want[0], **tol)
    torch.testing.assert_close(depth, want[1], **tol)
    for i in range(l):
        torch.testing.assert_close(color_layer[i], want[2][i], **tol)
        torch.testing.assert_close(depth_layer[i], want[3][i], **tol)
    assert float(color.std()) > 0.01 and float(depth.max()) > 0            # a non-trivial image


def test_small_call_drops_thresholds_like_the_reference(reference_env, monkeypatch):
    """N < one chunk: utils/batchify_rays.py:52-54 calls the model WITHOUT the thresholds; the patched entry point
    must do the same when the reference's caller passes them."""
    dropin = reference_env
    dropin.patch_reference(REFERENCE)
    import modeling
    import utils
    cfg = _cfg(8, 4, 2)
    model = modeling.build_layered_model(cfg, camera_num=1)
    seen = {}

    def fake_forward(rays, labels=None, bboxes=None, only_coarse=False, near_far=None, near_far_points=[],
                     density_threshold=0.0001, bkgd_density_threshold=0):
        seen.update(thr=density_threshold, bthr=bkgd_density_threshold, near_far=near_far)
        return "sentinel"
    monkeypatch.setattr(model, "forward", fake_forward)
    rays = torch.zeros(100, 9)
    out = utils.layered_batchify_ray(model, rays, torch.zeros(100), torch.zeros(100, 8, 3), near_far=torch.zeros(100, 2),
                                     density_threshold=20, bkgd_density_threshold=0.8)
    assert out == "sentinel" and seen["thr"] == 0.0001 and seen["bthr"] == 0 and seen["near_far"] is not None


def test_launcher_runs_a_reference_script_unmodified(reference_env, tmp_path, monkeypatch, capsys):
    """``python -m stnerf_amd.dropin script.py`` = patch, then run the script as __main__ (no edits to the script)."""
    dropin = reference_env
    script = tmp_path / "user_script.py"
    script.write_text(
        "import sys\n"
        "sys.path.append('.')\n"                                  # demo/taekwondo_demo.py:15
        "from config import cfg\n"
        "from modeling import build_layered_model\n"
        "from utils import layered_batchify_ray, add_two_dim_dict\n"
        "print('MODEL', build_layered_model.__module__, 'BATCHIFY', layered_batchify_ray.__module__, sys.argv[1:])\n")
    monkeypatch.chdir(REFERENCE)
    dropin.main([str(script), "-c", "configs/config_taekwondo.yml"])
    out = capsys.readouterr().out
    assert "MODEL stnerf_amd.modeling BATCHIFY stnerf_amd.utils.batchify_rays ['-c', 'configs/config_taekwondo.yml']" in out


# ---- N ranks: `torchrun ... -m stnerf_amd.dropin demo/...` -- the launcher joins the process group and the reference's own
# render_pose, unmodified, renders 1 / N of the frame per rank -------------------------------------------------------------
def _dropin_rank(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      STNERF_DIST_BACKEND="gloo")
    sys.dont_write_bytecode = True
    sys.meta_path.append(_StubFinder())
    sys.path.insert(0, REFERENCE)
    torch.Tensor.cuda = lambda self, *a, **k: self               # no GPU here (SURVEY 8c shim 2)
    import torch.distributed as dist
    import stnerf_amd.dropin as dropin
    from oracle import stnerf_oracle as O
    from stnerf_amd import parallel, synthetic as syn
    dropin.patch_reference(REFERENCE)
    assert dropin.join_process_group() == (rank, world)          # what `python -m stnerf_amd.dropin` does before the script runs
    try:
        assert dist.is_initialized() and dist.get_world_size() == world
        lnr = importlib.import_module("render.layered_neural_renderer")
        ray_dataset = importlib.import_module("data.datasets.ray_dataset")
        import modeling
        from stnerf_amd.modeling.layered_rfrender import LayeredRFRender
        L, n1, n2, H, W = 2, 8, 4, 48, 80                        # 3840 rays = one full 3584-ray chunk + a ragged second one
        l, N = L + 1, H * W
        sd, bk, per = _scene(L)
        K, T = syn.camera(H, W, 12.0)
        pairs = [(0, 1), (1, 2.5), (2, 1)]
        g = torch.Generator().manual_seed(7)
        jitter, u = torch.rand(l, N, n1, generator=g), torch.rand(l, N, n2, generator=g)
        launches = []

        class HostLogicOnCpu(LayeredRFRender):
            def render_rays_raw(self, rays, *a, **k):
                torch.Tensor.is_cuda = property(lambda t: True)
                try:
                    return super().render_rays_raw(rays, *a, **k)
                finally:
                    torch.Tensor.is_cuda = property(lambda t: False)

            def _render_launch(self, rays, boxes, pivot, retiming, only_coarse, thr_, bthr_, window, replay):
                n = rays.shape[0]
                launches.append((n, tuple(window)))
                om = O.OracleModel(layer_num=L, n_coarse=n1, n_fine=n2, params=self.state_dict(), bkgd_bbox=bk, bboxes=per, near=self.near,
                                   alpha=self.alpha, scale=self.scale, shift=self.shift)
                bx = boxes.unsqueeze(0).expand(n, l, 8, 3) if boxes.dim() == 3 else boxes
                saved = O.layer_boxes
                O.layer_boxes = lambda m, r: (bx.clone(), pivot, retiming, r[:, 6:] if retiming else r[:, -1])
                try:
                    draws = iter(list(replay["jitter"]) + list(replay["u"]))
                    out = O.render_chunk(om, rays, only_coarse, thr_, bthr_, rand=lambda shape: next(draws))
                finally:
                    O.layer_boxes = saved
                cat = lambda trip: torch.cat(list(trip), -1)
                return (cat(out[0]), cat(out[1]), torch.stack([cat(t) for t in out[2]], 1), torch.stack([cat(t) for t in out[3]], 1),
                        torch.stack(out[4], 1).to(torch.uint8))

        model = modeling.build_layered_model(_cfg(n1, n2, L), camera_num=1).eval()    # the reference's name, this framework's class
        assert type(model) is LayeredRFRender
        model.__class__ = HostLogicOnCpu
        model.load_state_dict(sd)
        model.set_bkgd_bbox(bk)
        model.set_bboxes(per)
        model.replay = {"jitter": jitter, "u": u}
        ds = object.__new__(ray_dataset.Ray_Dataset_Render)
        ds.height, ds.width, ds.layer_num = H, W, L
        ds.use_deform_time = ds.use_space_time = True
        ds.near_far = torch.tensor([[-1.0, -1.0]])
        r = object.__new__(lnr.LayeredNeuralRenderer)
        r.dataset, r.model, r.far = ds, model, 20.0
        render_pose = lnr.LayeredNeuralRenderer.render_pose                       # the reference's own function object
        model.shard_views = False
        whole = render_pose(r, T.numpy(), K, pairs, density_threshold=0.05, bkgd_density_threshold=0.02)
        n_whole = sum(n for n, _ in launches)
        launches.clear()
        model.shard_views = True
        split = render_pose(r, T.numpy(), K, pairs, density_threshold=0.05, bkgd_density_threshold=0.02)
        mine = sum(e - s for s, e in parallel.stripe_spans(N, 3584, rank, world))
        flat = lambda o: [o[0], o[1]] + list(o[2]) + list(o[3])
        same = all(a.shape == b.shape and torch.allclose(a, b, rtol=0, atol=2e-6) for a, b in zip(flat(whole), flat(split)))
        q.put((rank, dict(same=bool(same), whole_rays=n_whole, my_rays=sum(n for n, _ in launches), my_share=mine,
                          windows=[w for _, w in launches], picture=float(whole[0].std()) > 0.01)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_reference_render_pose_is_sharded_across_ranks_with_no_edit(world):
    """What ``python -m torch.distributed.run --nproc-per-node N -m stnerf_amd.dropin demo/taekwondo_demo.py`` sets up, on
    gloo: the reference's own render_pose (render/layered_neural_renderer.py:364-391), unmodified, on every rank; each rank's
    model is handed only its chunks (ray window of stripes of one reference chunk), and every rank returns the full colour,
    depth and per-layer images of the single-process render."""
    import torch.multiprocessing as mp
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dropin_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    N = 48 * 80
    assert sorted(results) == list(range(world))
    for rank, res in results.items():
        assert res["same"] and res["picture"], (rank, res)
        assert res["whole_rays"] == N and res["my_rays"] == res["my_share"] < N, (rank, res)
        assert all(w == (rank * 3584, 3584, world * 3584) for w in res["windows"]), (rank, res)
    assert sum(res["my_rays"] for res in results.values()) == N


# ---------------------------------------------------------------------------------------------------------------------
# the reference's OWN training loop drives this framework's model
# ---------------------------------------------------------------------------------------------------------------------
def test_reference_do_train_runs_one_iteration_on_the_patched_model(reference_env, tmp_path, monkeypatch):
    """``engine.layered_trainer.do_train`` -- the reference's function object, unmodified -- with the reference's ``make_loss``
    (layers/loss.py:4) and ``make_optimizer`` (solver/build.py:10-27), one batch, one epoch, on the model
    ``modeling.build_layered_model`` returns after ``patch_reference`` (= this framework's ``LayeredRFRender`` in train() mode under
    autograd: stnerf_amd.modeling.training).  There is no GPU here, so the ONE native piece, ``render_rays_train`` (sampler ->
    networks -> compositor, with autograd history), is answered by the CPU oracle evaluated on the model's own nn.Parameters with the
    boxes / thresholds / draws this framework's host code hands it; everything around it -- the batch unpacking, model.train(),
    the loss with its outlier / inlier terms, loss.backward(), optimizer.step(), scheduler.step(), the psnr monitor, the checkpoint
    -- is the reference's code.  The parameters after the step are the ones the reference's own model ends up with
    (tests/golden/train_c3.npz, recorded by make_golden.py --grads).  tests/test_gpu_training.py runs the same iteration on the
    HIP kernels against the same fixture."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from train_step_common import compare_digest, load_fixture, replay_of
    from oracle import stnerf_oracle as O
    from stnerf_amd import synthetic as syn
    dropin = reference_env
    dropin.patch_reference(REFERENCE)
    trainer = importlib.import_module("engine.layered_trainer")
    modeling = importlib.import_module("modeling")
    make_loss = importlib.import_module("layers").make_loss
    make_optimizer = importlib.import_module("solver").make_optimizer
    assert trainer.__file__.startswith(REFERENCE) and make_loss.__module__ == "layers.loss"
    from stnerf_amd.modeling import training as ours_training
    from stnerf_amd.modeling.layered_rfrender import LayeredRFRender

    z, meta = load_fixture("train_c3")
    L, n1, n2 = meta["L"], meta["n1"], meta["n2"]
    cfg = _cfg(n1, n2, L)
    cfg.SOLVER.MAX_EPOCHS, cfg.SOLVER.COARSE_STAGE, cfg.SOLVER.LOG_PERIOD, cfg.SOLVER.CHECKPOINT_PERIOD = 2, 0, 1, 10
    cfg.SOLVER.BASE_LR, cfg.SOLVER.WEIGHT_DECAY, cfg.SOLVER.OPTIMIZER_NAME = meta["lr"], 0.0, "Adam"
    cfg.MODEL.REMOVE_OUTLIERS = True
    cfg.OUTPUT_DIR = str(tmp_path / "out")
    model = modeling.build_layered_model(cfg, camera_num=1)
    assert type(model) is LayeredRFRender
    model.load_state_dict(syn.make_state_dict(L, True, True, seed=meta["weight_seed"]))
    bk, per = syn.scene_boxes(L)
    model.set_bkgd_bbox(bk)
    model.set_bboxes(per)
    _, model.replay = replay_of(z, meta)
    model.fresh_draws_per_call = False
    calls = []

    def oracle_render_rays_train(self, rays, boxes, pivot, retiming, only_coarse, thr, bthr, window, replay):
        n, l = rays.shape[0], self.layer_num + 1
        calls.append((n, retiming, only_coarse, tuple(window), self.training, torch.is_grad_enabled()))
        om = O.OracleModel(layer_num=L, n_coarse=n1, n_fine=n2, params=dict(self.named_parameters()), bkgd_bbox=bk, bboxes=per,
                           near=self.near, alpha=self.alpha, scale=self.scale, shift=self.shift)
        bx = boxes.unsqueeze(0).expand(n, l, 8, 3) if boxes.dim() == 3 else boxes
        saved = O.layer_boxes
        O.layer_boxes = lambda m_, r: (bx.clone(), pivot, retiming, r[:, 6:] if retiming else r[:, -1])
        try:
            draws = iter(list(replay["jitter"]) + list(replay.get("u", [])))
            out = O.render_chunk(om, rays, only_coarse, thr, bthr, rand=lambda shape: next(draws))
        finally:
            O.layer_boxes = saved
        cat = lambda trip: torch.cat(list(trip), -1)
        return (cat(out[0]), cat(out[1]), torch.stack([cat(t) for t in out[2]], 1), torch.stack([cat(t) for t in out[3]], 1),
                torch.stack(out[4], 1).to(torch.uint8))
    monkeypatch.setattr(ours_training, "render_rays_train", oracle_render_rays_train)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda t: True))          # (the host logic refuses CPU rays)
    # the trainer's surroundings that need a dataset / files: validation images and checkpoints
    checkpoints = []
    monkeypatch.setattr(trainer, "val_vis", lambda *a, **k: None)
    monkeypatch.setattr(trainer, "ModelCheckpoint", lambda model_, opt_, sch_, out_dir, epoch, global_step=0: checkpoints.append((epoch, global_step)))

    class Writer:
        def __init__(self):
            self.scalars = {}

        def add_scalar(self, tag, value, step):
            self.scalars[tag] = float(value)

    class Scheduler:
        steps = 0

        def step(self):
            type(self).steps += 1

    n = meta["n_rays"]
    batch = (torch.from_numpy(z["rays"]), torch.from_numpy(z["rgbs"]), torch.from_numpy(z["labels"]), torch.zeros(n), torch.zeros(n, 8, 3),
             torch.zeros(n, 2))
    writer = Writer()
    anomaly = torch.is_anomaly_enabled()
    try:
        trainer.do_train(cfg, model, [batch], None, make_optimizer(cfg, model), Scheduler(), make_loss(cfg), writer)
    finally:
        torch.autograd.set_detect_anomaly(anomaly)                                  # (do_train switches it on, :160)
    assert len(calls) == 1 and calls[0] == (n, False, False, (0, 0, 0), True, True), calls
    assert Scheduler.steps == 1 and checkpoints, (Scheduler.steps, checkpoints)
    assert writer.scalars["Loss/train_loss"] == pytest.approx(float(z["loss"][0]), rel=1e-6)
    assert writer.scalars["Loss/mask_loss"] == pytest.approx(float(z["loss_mask_0"][0] + z["loss_mask_1"][0]), rel=1e-5)
    named = dict(model.named_parameters())
    for k in z.files:
        if k.startswith("stepped|"):
            p = k.split("|", 1)[1]
            assert compare_digest(p, syn.tensor_digest(p, named[p], meta["grad_samples"]), z[k], rel=2e-6) <= 1.0, p
        if k.startswith("grad|"):
            p = k.split("|", 1)[1]
            assert compare_digest(p, syn.tensor_digest(p, named[p].grad, meta["grad_samples"]), z[k], rel=2e-5) <= 1.0, p
