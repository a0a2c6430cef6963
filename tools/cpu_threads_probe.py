#!/usr/bin/env python3
"""Which torch thread count suits the CPU baseline on this host?  One reference chunk of the benchmark view per setting."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import WORKLOADS, _oracle_ray_window
from oracle import stnerf_oracle as O
from stnerf_amd import synthetic as syn
H, W, L, n1, n2, st, dt = WORKLOADS["taekwondo-1080p-64+64"]
K, T = syn.camera(H, W, 10.0)
bk, per = syn.scene_boxes(L)
m = O.OracleModel(layer_num=L, n_coarse=n1, n_fine=n2, params=syn.make_state_dict(L, st, dt, seed=0), use_deform_time=dt,
                  use_space_time=st, bkgd_bbox=bk, bboxes=per)
rays = torch.cat([_oracle_ray_window(O, K, T, H, W, (H // 2) * W, 3584), syn.frame_id_columns(3584, L)], -1)
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    with torch.no_grad():
        O.render_chunk(m, rays[:256])
        t0 = time.perf_counter()
        O.layered_batchify_ray(m, rays, chuncks=3584)
        print(th, "threads:", round(3584 / (time.perf_counter() - t0), 1), "rays/s", flush=True)
