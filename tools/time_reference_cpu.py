#!/usr/bin/env python3
"""Time the REFERENCE itself (DarlingHang/st-nerf imported from /root/reference) on the CPU of the build container, on the
same sample bench.py's `cpu_baseline` leg times with the oracle port on the GPU box: 8 reference chunks of 3584 rays spread
evenly over the rows of the 1080p benchmark view.  /root/reference does not exist on the GPU
box, so this figure can only be produced here; it is committed beside the port's figure.

    PYTHONDONTWRITEBYTECODE=1 python tools/time_reference_cpu.py > profiles/r02_reference_cpu_build_container.json
"""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.argv = [sys.argv[0]]
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
import make_golden as G          # noqa: E402  (imports /root/reference with the SURVEY 8c shims)
from bench import WORKLOADS, _oracle_ray_window   # noqa: E402
from oracle import stnerf_oracle as O   # noqa: E402

syn = G.syn
workload = "taekwondo-1080p-64+64"
H, W, L, n1, n2, st, dt = WORKLOADS[workload]
model = G.build_ref_model(L, n1, n2, st, dt, 0)
K, T = syn.camera(H, W, 10.0)
chunk, n_chunks = 3584, 8
torch.manual_seed(0)
rows = []
with torch.no_grad():
    warm = torch.cat([_oracle_ray_window(O, K, T, H, W, (H // 2) * W, 256), syn.frame_id_columns(256, L)], -1)
    model(warm, torch.zeros(256), torch.zeros(256, 8, 3), near_far=torch.zeros(256, 2))
    for i in range(n_chunks):
        first = min(H * W - chunk, max(0, int((i + 0.5) / n_chunks * H * W) - chunk // 2))
        rays = torch.cat([_oracle_ray_window(O, K, T, H, W, first, chunk), syn.frame_id_columns(chunk, L)], -1)
        t0 = time.perf_counter()
        out = G.ref_utils.layered_batchify_ray(model, rays, torch.zeros(chunk), torch.zeros(chunk, 8, 3), chuncks=chunk,
                                               near_far=torch.zeros(chunk, 2))
        sec = time.perf_counter() - t0
        t1 = time.perf_counter()
        om = O.OracleModel(layer_num=L, n_coarse=n1, n_fine=n2, params=syn.make_state_dict(L, st, dt, seed=0),
                           use_deform_time=dt, use_space_time=st, bkgd_bbox=syn.scene_boxes(L)[0], bboxes=syn.scene_boxes(L)[1])
        O.layered_batchify_ray(om, rays, chuncks=chunk)
        sec_port = time.perf_counter() - t1
        rows.append(dict(first_row=first // W, reference_seconds=round(sec, 3), port_seconds=round(sec_port, 3),
                         performer_hit_fraction=round(sum(int(m.sum()) for m in out[4][1:]) / (chunk * L), 3)))
tot = sum(r["reference_seconds"] for r in rows)
tot_p = sum(r["port_seconds"] for r in rows)
cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
print(json.dumps(dict(workload=workload, what="the reference's own utils.layered_batchify_ray + modeling.build_layered_model on CPU",
                      reference_rays_per_s=n_chunks * chunk / tot, oracle_port_rays_per_s=n_chunks * chunk / tot_p,
                      port_over_reference_time=tot_p / tot, host=dict(nproc=os.cpu_count(), torch_threads=torch.get_num_threads(), cpu=cpu,
                                                                     torch=torch.__version__),
                      extrapolated_frame_seconds=H * W / (n_chunks * chunk / tot), chunks=rows), indent=1))
