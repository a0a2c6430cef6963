"""Debug helper: per-stage comparison of the HIP pipeline with the CPU oracle on the smoke scene."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import stnerf_oracle as O
from stnerf_amd import ops, synthetic as syn
from stnerf_amd.modeling import build_layered_model

L, n1, n2, H, W = 2, 16, 8, 24, 32
m = types.SimpleNamespace(BOARDER_WEIGHT=1e10, SAMPLE_METHOD="BBOX", SAME_SPACENET=False, TKERNEL_INC_RAW=True,
                          POSE_REFINEMENT=False, USE_DIR=True, USE_DEFORM_VIEW=False, USE_DEFORM_TIME=True,
                          USE_SPACE_TIME=True, BKGD_USE_DEFORM_TIME=False, BKGD_USE_SPACE_TIME=False,
                          DEEP_RGB=False, COARSE_RAY_SAMPLING=n1, FINE_RAY_SAMPLING=n2)
cfg = types.SimpleNamespace(MODEL=m, DATASETS=types.SimpleNamespace(LAYER_NUM=L))
sd = syn.make_state_dict(L, True, True, seed=3)
bk, per = syn.scene_boxes(L)
model = build_layered_model(cfg, camera_num=1); model.load_state_dict(sd); model.set_bkgd_bbox(bk); model.set_bboxes(per)
model = model.cuda().eval()
K, T = syn.camera(H, W, 15.0)
rays = ops.generate_rays(K, T, H, W, frame_ids=[1.0, 2.5, 1.0])
n = H * W
g = torch.Generator().manual_seed(0)
jitter, u = torch.rand(L + 1, n, n1, generator=g), torch.rand(L + 1, n, n2, generator=g)
om = O.OracleModel(layer_num=L, n_coarse=n1, n_fine=n2, params=sd, bkgd_bbox=bk, bboxes=per)
draws = iter(list(jitter) + list(u)); tr = {}
with torch.no_grad():
    ref = O.render_chunk(om, rays.cpu(), rand=lambda s: next(draws), trace=tr)
# HIP stages
boxes, pivot = model._retimed_boxes(rays[0, 6:].cpu())
boxes = boxes.cuda()
l = L + 1
t_c, xyz_c, mask = ops.sample_coarse(rays, boxes, n1, jitter=jitter.cuda())
print("t_c exact", all(torch.equal(t_c[:, i].cpu(), tr["t_coarse"][i].squeeze(-1)) for i in range(l)))
lst, cnt = ops.compact_rays(mask)
raw_c = torch.zeros(n, l, n1, 4, device="cuda")
model._stage(rays, xyz_c, raw_c, lst, cnt, lambda i: 6 + i, False)
for i in range(l):
    mk = tr["mask"][i]
    print(i, "xyz_c", float((xyz_c[:, i].cpu() - tr["xyz_coarse"][i]).abs()[mk].max()) if mk.any() else None)
thr = 1e-4
lo_c, mix_c, w_c, od = ops.composite(t_c, raw_c, mask, near=0.0, fine=False, cut_negative_t=True, thresholds=[None, thr, thr], want_weights=True, want_order=True)
for i in range(l):
    sg = raw_c[:, i, :, 3].cpu().clone()
    if i > 0:
        sg[t_c[:, i].cpu() < 0] = 0; sg[sg < thr] = 0
    print(i, "sigma_c", float((sg - tr["sigma_coarse"][i].squeeze(-1)).abs().max()), "w_c", float((w_c[:, i].cpu() - tr["w_coarse"][i].squeeze(-1)).abs().max()))
print("coarse mixed color err", float((mix_c[:, :3].cpu() - ref[1][0]).abs().max()))
t_f, xyz_f, z, inds, cdf = ops.resample(t_c, w_c, n2, rays, u=u.cuda(), debug=True)
for i in range(l):
    d = (t_f[:, i].cpu() - tr["t_fine"][i]).abs()
    print(i, "t_f err", float(d.max()), "rays>1e-4:", int((d.max(-1)[0] > 1e-4).sum()))
raw_f = torch.zeros(n, l, n1 + n2, 4, device="cuda")
model._stage(rays, xyz_f, raw_f, lst, cnt, lambda i: 6 + i, True)
for i in range(l):
    sg = raw_f[:, i, :, 3].cpu().clone()
    d = (sg - tr["sigma_fine"][i].squeeze(-1)).abs()
    print(i, "sigma_f raw err (evaluated rays)", float(d[tr["mask"][i]].max()) if i else float(d.max()))
lo_f, mix_f, _, _ = ops.composite(t_f, raw_f, mask, near=0.0, fine=True, thresholds=[0.0, thr, thr], sigma_scale=[1, 1, 1.0])
e = (mix_f[:, :3].cpu() - ref[0][0]).abs().max(-1)[0]
j = int(e.argmax()); print("fine mixed color err", float(e.max()), "ray", j)
for i in range(l):
    el = (lo_f[:, i, :3].cpu() - ref[2][i][0]).abs().max(-1)[0]
    print(i, "layer color err", float(el.max()), "at ray", int(el.argmax()), "err at worst ray", float(el[j]))
    sf = raw_f[j, i, :, 3].cpu(); so = tr["sigma_fine"][i][j].squeeze(-1)
    print("   sigma last (hip, oracle):", float(sf[-1]), float(so[-1]), " min|sigma|:", float(so.abs().min()))
    print("   t_f diff at ray", float((t_f[j, i].cpu() - tr["t_fine"][i][j]).abs().max()))
