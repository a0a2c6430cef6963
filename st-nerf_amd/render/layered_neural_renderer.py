"""User-facing layered renderer: camera paths, retiming, per-frame edit schedule, per-pose rendering.

Mirror of ``LayeredNeuralRenderer`` (render/layered_neural_renderer.py:17-741) for the methods the shipped
demos use (demo/taekwondo_demo.py:46-53): same method names, argument meaning and bookkeeping
(``poses`` / ``Ks`` / ``layer_frame_pairs`` / ``s_*_frame`` lists), same scipy calls for the pose path
(Slerp of the rotations + ``splprep``/``splev`` of the camera centres).  Differences:

* the reference constructor reads a dataset and a checkpoint from ``cfg.OUTPUT_DIR``
  (:96-121, needs open3d / torchvision and the absent data); here the model, the camera poses and the
  intrinsics are passed in (``model=``, ``gt_poses=``, ``gt_Ks=``).  ``from_checkpoint`` covers the
  checkpoint half (reference key names);
* rays are generated on the device and images stay there until the caller asks for them
  (``render_path`` returns them; writing jpg/png/mp4 is left to ``on_frame`` -- imageio is absent here);
* per-frame rendering is ``stnerf_amd.render.render_pose.render_pose`` (HIP kernels).

The host logic below is pinned against fixtures produced by the reference's own methods
(tests/golden/make_golden.py: ``g_path``).
"""
from __future__ import annotations

from typing import Callable, List, Optional

import numpy as np
import torch
from scipy.interpolate import splev, splprep
from scipy.spatial.transform import Rotation as R
from scipy.spatial.transform import Slerp

from stnerf_amd.render.render_pose import render_pose as _render_pose


class LayeredNeuralRenderer:

    def __init__(self, cfg, scale=None, shift=None, rotation=None, s_shift=None, s_scale=None, s_alpha=None, *,
                 model=None, gt_poses=None, gt_Ks=None):
        if model is None or gt_poses is None or gt_Ks is None:
            raise NotImplementedError(
                "dataset / checkpoint discovery from cfg.OUTPUT_DIR (render/layered_neural_renderer.py:96-121) is "
                "outside the MI355X hot path: pass model=, gt_poses= (C,4,4) and gt_Ks= (C x (3,3)) explicitly")
        self.alpha = None
        self.cfg = cfg
        self.scale, self.shift, self.rotation = scale, shift, rotation
        self.s_shift, self.s_scale, self.s_alpha = s_shift, s_scale, s_alpha
        if s_shift is not None:
            self.shift = self.s_shift[0]
        if s_scale is not None:
            self.scale = self.s_scale[0]
        if s_alpha is not None:
            self.alpha = self.s_alpha[0]
        self.model = model
        self.model.scale, self.model.shift = self.scale, self.shift
        self.layer_num = cfg.DATASETS.LAYER_NUM
        self.frame_num = cfg.DATASETS.FRAME_NUM
        self.display_layers = {i: 1 for i in range(self.layer_num + 1)}
        self.gt_poses = torch.as_tensor(gt_poses, dtype=torch.float32)
        self.gt_Ks = [torch.as_tensor(k, dtype=torch.float32) for k in gt_Ks]
        self.far = 20.0
        off = cfg.DATASETS.FRAME_OFFSET
        self.min_frame = [1 + off for _ in range(self.layer_num + 1)]
        self.max_frame = [self.frame_num + off for _ in range(self.layer_num + 1)]
        self.images, self.depths = [], []
        self.image_num = 0
        self.camera_num = self.gt_poses.shape[0]
        self.min_camera_id, self.max_camera_id = 0, self.camera_num - 1
        self.fps = 25
        self.height, self.width = cfg.INPUT.SIZE_TEST[1], cfg.INPUT.SIZE_TEST[0]
        self.save_count = 0
        self.poses: List = []
        self.Ks: List = []
        self.layer_frame_pairs: List = []
        self.trace_layer = -1
        self.dir_name = ''

    # ---- layer display / knobs (:643-686, :740-741) ----------------------------------------------------
    def hide_layer(self, layer_id):
        self.model.hide_layer(layer_id)
        self.display_layers[layer_id] = 0

    def show_layer(self, layer_id):
        self.model.show_layer(layer_id)
        self.display_layers[layer_id] = 1

    def is_shown_layer(self, layer_id):
        return self.display_layers[layer_id] == 1

    def set_save_dir(self, dir_name):
        self.dir_name = dir_name

    def set_fps(self, fps):
        self.fps = fps

    def set_near(self, near):
        self.model.near = near

    def set_frame_duration(self, min_frame, max_frame, layer_id=-1):
        if layer_id == -1:
            self.min_frame = [min_frame for _ in range(self.layer_num + 1)]
            self.max_frame = [max_frame for _ in range(self.layer_num + 1)]
        else:
            self.min_frame[layer_id], self.max_frame[layer_id] = min_frame, max_frame

    def set_pose_duration(self, min_camera_id, max_camera_id):
        self.min_camera_id, self.max_camera_id = min_camera_id, max_camera_id

    def invert_poses(self):
        self.poses.reverse()
        self.Ks.reverse()

    def save_poses(self, path):
        np.save(path, self.poses)

    # ---- (layer, frame) bookkeeping shared by every path setter (:161-168, :180-186, ...) ----------------
    def _append_layer_frame_pairs(self, n_poses, smooth_time=False):
        for idx in range(n_poses + 1):
            pair = []
            for layer_id in range(self.layer_num + 1):
                if self.is_shown_layer(layer_id):
                    span = (self.max_frame[layer_id] - self.min_frame[layer_id]) / n_poses * idx
                    frame_id = (span if smooth_time else int(span)) + self.min_frame[layer_id]
                    pair.append((layer_id, frame_id))
            self.layer_frame_pairs.append(pair)

    def _edit_schedule(self, n):
        """Linear per-frame schedules of the editing knobs (:232-243, :279-301)."""
        if self.s_shift is not None:
            a, b = np.array(self.s_shift[0]), np.array(self.s_shift[1])
            step = (b - a) / (n - 1)
            self.s_shift_frame = [(a + i * step).tolist() for i in range(n)]
        if self.s_scale is not None:
            a, b = np.array(self.s_scale[0]), np.array(self.s_scale[1])
            step = (b - a) / (n - 1)
            self.s_scale_frame = [(a + i * step).tolist() for i in range(n)]
        if self.s_alpha is not None:
            a, b = self.s_alpha[0], self.s_alpha[1]
            step = (b - a) / (n - 1)
            self.s_alpha_frame = [(a + i * step) for i in range(n)]

    # ---- camera paths ---------------------------------------------------------------------------------------
    def set_smooth_path_poses(self, step_num, around=False, smooth_time=False):
        """Slerp of the key rotations + cubic B-spline through the key camera centres, linear intrinsics
        (:230-319).  ``around=False`` keeps only the first and last rotation (:254-257)."""
        self._edit_schedule(step_num)
        lo, hi = self.min_camera_id, self.max_camera_id
        Rs = self.gt_poses[lo:hi + 1, :3, :3].cpu().numpy()
        Ts = self.gt_poses[lo:hi + 1, :3, 3].cpu().numpy()
        key_frames = [i for i in range(lo, hi + 1)]
        if not around:
            Rs = np.array([Rs[0], Rs[-1]])
            key_frames = [lo, hi]
        interp_frames = [(i * (hi - lo) / (step_num - 1) + lo) for i in range(step_num)]
        interp_Rs = Slerp(key_frames, R.from_matrix(Rs))(interp_frames).as_matrix()
        tck, _ = splprep([Ts[:, 0], Ts[:, 1], Ts[:, 2]])
        new_points = np.stack(splev([i / (step_num - 1) for i in range(step_num)], tck), axis=1)
        K0, K1 = self.gt_Ks[lo], self.gt_Ks[hi]
        poses = []
        for i in range(step_num):
            pose = np.zeros((4, 4))
            pose[:3, :3] = interp_Rs[i]
            pose[:3, 3] = new_points[i]
            pose[3, 3] = 1
            poses.append(pose)
            self.Ks.append((K1 - K0) * i / (step_num - 1) + K0)
        self.poses = self.poses + poses
        self._append_layer_frame_pairs(len(poses), smooth_time)

    def set_path_gt_poses(self):
        """One frame per ground-truth camera (:171-186)."""
        poses = [self.gt_poses[i] for i in range(self.gt_poses.shape[0])]
        self.poses = self.poses + poses
        self.Ks = self.Ks + self.gt_Ks
        self._append_layer_frame_pairs(len(poses))

    def set_path_fixed_gt_poses(self, id, num=None):
        """``num`` frames from ground-truth camera ``id`` (a time sweep from a fixed view, :188-228)."""
        self._edit_schedule(num)
        self.poses = self.poses + [self.gt_poses[id] for _ in range(num)]
        self.Ks = self.Ks + [self.gt_Ks[id] for _ in range(num)]
        self._append_layer_frame_pairs(num)

    def load_path_poses(self, poses):
        """Externally supplied poses, intrinsics lerped between the first and last-but-one camera (:321-337)."""
        self.poses = poses
        n = len(poses)
        K0, K1 = self.gt_Ks[self.min_camera_id], self.gt_Ks[self.max_camera_id - 1]
        for i in range(n):
            self.Ks.append((K1 - K0) * i / (n - 1) + K0)
        self._append_layer_frame_pairs(n)

    # ---- retiming (:495-545) ------------------------------------------------------------------------------
    def retime_by_key_frames(self, layer_id, key_frames_layer, key_frames):
        """Piecewise-linear remap of one layer's frame ids: global key frame key_frames[i] shows the layer's
        frame key_frames_layer[i]."""
        assert len(key_frames_layer) == len(key_frames)
        for i in range(len(self.layer_frame_pairs)):
            for j in range(len(self.layer_frame_pairs[i])):
                layer, frame = self.layer_frame_pairs[i][j]
                if layer != layer_id:
                    continue
                idx_start, idx_end, weight = -1, -1, 0
                for idx in range(len(key_frames)):
                    if frame <= key_frames[idx]:
                        idx_end, idx_start = idx, idx - 1
                        end = key_frames[idx]
                        start = self.min_frame[layer] if idx == 0 else key_frames[idx - 1]
                        weight = (frame - start) / (end - start)
                        break
                if idx_start == -1 and idx_end == 0:
                    weight = (frame - self.min_frame[layer]) / (key_frames[0] - self.min_frame[layer])
                    new_start, new_end = self.min_frame[layer], key_frames_layer[0]
                elif idx_start >= -1 and idx_end != -1:
                    new_start, new_end = key_frames_layer[idx_start], key_frames_layer[idx_start + 1]
                elif idx_start == -1 and idx_end == -1:
                    weight = (frame - key_frames[-1]) / (self.max_frame[layer] - key_frames[-1])
                    new_start, new_end = key_frames_layer[-1], self.max_frame[layer]
                else:
                    raise RuntimeError(f"Undefined branch: start idx {idx_start}, end idx {idx_end}")  # exit(-1), :537-539
                self.layer_frame_pairs[i][j] = (layer, round(weight * (new_end - new_start) + new_start))

    # ---- rendering -------------------------------------------------------------------------------------------
    def render_pose(self, pose, K, layer_frame_pair, density_threshold=0, bkgd_density_threshold=0):
        """-> color (H,W,3), depth (H,W,1), color_layer, depth_layer on the device (:364-391)."""
        return _render_pose(self.model, pose, K, self.height, self.width, layer_frame_pair, self.far, density_threshold,
                            bkgd_density_threshold)

    def render_path(self, inverse_y_axis=False, density_threshold=0, bkgd_density_threshold=0, auto_save=True,
                    on_frame: Optional[Callable] = None):
        """Render every pose of the path with its (layer, frame) pairs and edit schedule (:401-488).  With
        ``auto_save`` the frames are kept in ``self.images`` / ``self.depths`` (CPU tensors, as the reference
        keeps them for ``save_video``); ``on_frame(idx, color, depth, color_layer, depth_layer)`` is called with
        the device tensors (write files there)."""
        self.images, self.depths = [], []
        self.images_layer = [[] for _ in range(self.layer_num + 1)]
        self.depths_layer = [[] for _ in range(self.layer_num + 1)]
        self.image_num = 0
        for idx in range(len(self.poses)):
            if self.s_shift is not None:
                self.model.shift = self.s_shift_frame[idx]
            if self.s_scale is not None:
                self.model.scale = self.s_scale_frame[idx]
            if self.s_alpha is not None:
                self.model.alpha = self.s_alpha_frame[idx]
            color, depth, color_layer, depth_layer = self.render_pose(self.poses[idx], self.Ks[idx],
                                                                     self.layer_frame_pairs[idx], density_threshold,
                                                                     bkgd_density_threshold)
            if inverse_y_axis:
                color, depth = torch.flip(color, [0]), torch.flip(depth, [0])
                color_layer = [torch.flip(i, [0]) for i in color_layer]
                depth_layer = [torch.flip(i, [0]) for i in depth_layer]
            if on_frame is not None:
                on_frame(idx, color, depth, color_layer, depth_layer)
            if auto_save:
                self.images.append(color.cpu())
                self.depths.append(depth.cpu())
                for layer_id in range(self.layer_num + 1):
                    if self.is_shown_layer(layer_id):
                        self.images_layer[layer_id].append(color_layer[layer_id].cpu())
                        self.depths_layer[layer_id].append(depth_layer[layer_id].cpu())
            self.image_num += 1
        return self.images, self.depths

    def render_path_walking(self, inverse_y_axis=False, density_threshold=0, bkgd_density_threshold=0, auto_save=True,
                            on_frame: Optional[Callable] = None):
        """``render_path`` without the per-frame edit schedule plus the occlusion composite of the walking demo
        (:550-618): layer 2 is pasted over the background image wherever it is in front of it
        (``depth_layer[2] < depth_layer[0]``) and has colour.  The composites are kept in ``self.images_hide``."""
        self.images, self.depths, self.images_hide = [], [], []
        self.images_layer = [[] for _ in range(self.layer_num + 1)]
        self.depths_layer = [[] for _ in range(self.layer_num + 1)]
        self.image_num = 0
        for idx in range(len(self.poses)):
            color, depth, color_layer, depth_layer = self.render_pose(self.poses[idx], self.Ks[idx],
                                                                     self.layer_frame_pairs[idx], density_threshold,
                                                                     bkgd_density_threshold)
            if inverse_y_axis:
                color, depth = torch.flip(color, [0]), torch.flip(depth, [0])
                color_layer = [torch.flip(i, [0]) for i in color_layer]
                depth_layer = [torch.flip(i, [0]) for i in depth_layer]
            color_hide = None
            if self.layer_num >= 2:
                color_hide = color_layer[0].clone()                                   # :606-611
                index = depth_layer[2] < depth_layer[0]
                index = torch.cat([index, index, index], dim=2)
                index = torch.logical_and(index, color_layer[2] != 0)
                color_hide[index] = color_layer[2][index]
            if on_frame is not None:
                on_frame(idx, color, depth, color_layer, depth_layer)
            if auto_save:
                self.images.append(color.cpu())
                self.depths.append(depth.cpu())
                if color_hide is not None:
                    self.images_hide.append(color_hide.cpu())
                for layer_id in range(self.layer_num + 1):
                    self.images_layer[layer_id].append(color_layer[layer_id].cpu())
                    self.depths_layer[layer_id].append(depth_layer[layer_id].cpu())
            self.image_num += 1
        return self.images, self.depths

    def save_video(self, writer: Optional[Callable] = None):
        """The reference writes ``color_<n>.mp4`` / ``depth_<n>.mp4`` with imageio (:624-637); imageio is not a
        dependency here, so the frames are handed to ``writer(kind, index, frames, fps)`` (``kind`` in
        {"color", "depth"}), e.g. ``lambda kind, i, frames, fps: imageio.mimwrite(f"{kind}_{i}.mp4", frames, fps=fps,
        quality=8)``.  As in the reference an empty renderer only warns."""
        if len(self.images) == 0:
            print("Warning: Cannot generate video for all rendered images, data is empty.")
            return False
        if writer is None:
            raise NotImplementedError("pass writer=...: video encoding (imageio) is outside the MI355X hot path")
        writer("color", self.save_count, self.images, self.fps)
        writer("depth", self.save_count, self.depths, self.fps)
        self.save_count += 1
        return True
