"""torch.autograd.Functions for the two networks: what ``loss.backward()`` of
engine/layered_trainer.py:192-217 needs from modeling/spacenet.py:101-160 and modeling/motion_net.py:34-71.

Forward = the fused inference kernel (no activation is stored).  Backward recomputes the forward a chunk of samples at a time
with every layer's input kept in a bounded workspace (``CHUNK_SAMPLES`` x ~11 KB), then walks the layers backwards; every
matrix product -- the recomputed layers, dX = dY W, dW += dY^T X -- is the hand-written f32 MFMA GEMM of csrc/train.hip
(``ops.train_linear_*``), the encodings and their chain rule are ``ops.train_encode*``.  PyTorch only owns the buffers.

Gradients are returned for the sample points (``pos`` / MotionNet's xyz) and for every weight and bias; directions and
frame ids are inputs of the renderer, not parameters (the reference's POSE_REFINEMENT / USE_DEFORM_VIEW are outside this
path), and get none.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch

from stnerf_amd import ops

# Samples whose activations are kept at a time while recomputing (~11 KB each for a SpaceNet: 2.9 GB of a 288 GB part).  Large
# enough that the dW contraction's 256 slices are 1024 samples each with the chip full (csrc/train.hip: dw_slices); measured on
# 524,288 samples: 0.54 / 0.60 / 0.63 / 0.64 of the f32 MFMA peak at 2^16 / 2^17 / 2^18 / 2^19.  STNERF_TRAIN_CHUNK_SAMPLES overrides.
# Clamped to [1024, 2^29 / 320]: the widest activation row is 320 floats and a GEMM operand may hold at most 2^29 of them
# (csrc/train.hip), and a zero or negative value would make the chunk loops below step by nothing.
# STNERF_TRAIN_FUSED=0: the round-4 backward (one GEMM launch per layer and direction) instead of the fused launches of
# csrc/train_wave.hip -- kept for A/B measurements and for the configurations the fused kernels do not cover (deep_rgb, no raw input
# columns in the encodings, no view directions)
FUSED_BACKWARD = os.environ.get("STNERF_TRAIN_FUSED", "1") != "0"
# The training forward of a SpaceNet KEEPS every layer's input (8.1 KB per sample, written by the forward kernel itself) when the call's
# samples fit this budget -- as the reference's autograd does, and what 288 GB of HBM are for: the backward then starts without any
# recomputation.  Above the budget (or with STNERF_TRAIN_KEEP_GB=0) nothing is kept and the backward recomputes chunk by chunk.
KEEP_BYTES = int(float(os.environ.get("STNERF_TRAIN_KEEP_GB", "32")) * (1 << 30))
# ... per CALL; and over all calls of an iteration that hold their activations at the same time (every network of both stages until its
# backward has run: 2 l SpaceNets + deformation nets) at most this much, after which further calls fall back to recomputation instead
# of running the 288 GB part out of memory.  The counter follows the kept buffers' lifetime (weakref.finalize).
KEEP_TOTAL_BYTES = int(float(os.environ.get("STNERF_TRAIN_KEEP_TOTAL_GB", "160")) * (1 << 30))
_kept_now = [0]


def _may_keep(nbytes: int) -> bool:
    return 0 < nbytes <= KEEP_BYTES and _kept_now[0] + nbytes <= KEEP_TOTAL_BYTES


def _account_kept(bufs, nbytes: int) -> None:
    import weakref
    _kept_now[0] += nbytes

    def release(n=nbytes):
        _kept_now[0] -= n
    weakref.finalize(bufs[0], release)
ACT_FLOATS_PER_SAMPLE = 320 + 5 * 256 + 304 + 128 + 64
CHUNK_SAMPLES = min(max(int(os.environ.get("STNERF_TRAIN_CHUNK_SAMPLES", 1 << 18)), 1024), (1 << 29) // 320)


# every weight / bias gradient of a network in one launch (stnerf_train_dw_batch); "0": one launch group per layer (the A/B of
# tools/bench_backward.py)
DW_BATCH = os.environ.get("STNERF_TRAIN_DW_BATCH", "1") != "0"
# Arithmetic of the fused SpaceNet forward under autograd ("" = the module's own, which is split bf16 unless the model was built with
# precision="fp32"): the stage kernel that renders, with the activation tap (csrc/mlp_bf16x3.hip, round 6; 2.3 -> ~1.6 ms per 262,144
# samples).  "fp32": csrc/mlp_wave.hip's exact-f32 kernel with its tap, as up to round 5.
TRAIN_FWD = os.environ.get("STNERF_TRAIN_FWD", "")
DX_BF16X3 = os.environ.get("STNERF_TRAIN_DX_BF16X3", "1") != "0"   # (A/B: the exact-f32 d x chain behind a split-bf16 forward)


def _train_fwd_precision(module) -> str:
    """Arithmetic of the fused SpaceNet forward under autograd: the module's own (``precision``: "bf16x3" unless the model was
    built for exact f32), or what STNERF_TRAIN_FWD names ("fp32" / "bf16x3")."""
    p = TRAIN_FWD or module.precision
    if p not in ops.PRECISIONS:
        raise ValueError(f"STNERF_TRAIN_FWD / precision must be one of {ops.PRECISIONS}, got {p!r}")
    return p


def _dw(dy, x, dw, db, accumulate: bool) -> None:
    """One layer's dW / db on the per-layer path (deep_rgb, no-raw-input and no-direction networks): through the batch kernel of
    csrc/train_dw.hip as a batch of one (output-stationary wave tiles; 2 - 3 x the round-4 GEMM's rate), or -- STNERF_TRAIN_DW_BATCH=0 --
    the round-4 split-K GEMM (stnerf_train_linear_dw)."""
    if DW_BATCH:
        ops.train_dw_batch([(dy, x, dw, db)], accumulate)
    else:
        ops.train_linear_dw(dy, x, dw, db, accumulate)


def _weight_gradients(layers, accumulate: bool) -> None:
    if DW_BATCH:
        ops.train_dw_batch(layers, accumulate)
    else:
        for dy, x, dw, db in layers:
            ops.train_linear_dw(dy, x, dw, db, accumulate)


def _pad4(n: int) -> int:
    return (n + 3) // 4 * 4


def _padded_weight(w: torch.Tensor) -> torch.Tensor:
    """(n, k) nn.Linear weight -> a view (n, k) of zero-padded (n, round4(k)) storage: 16-byte aligned rows."""
    n, k = w.shape
    buf = torch.zeros(n, _pad4(k), dtype=torch.float32, device=w.device)
    buf[:, :k] = w.detach()
    return buf[:, :k]


def _buf(m: int, cols: int, device) -> torch.Tensor:
    return torch.empty(m, _pad4(cols), dtype=torch.float32, device=device)


def _transposed(module, params, sections) -> tuple:
    """(wt, offsets) of `sections` = [(W view, n_pad)] in one launch (stnerf_pack_transposed) into the module's own blob, rebuilt when a
    parameter changed (every optimizer.step(): ~40 torch launches per network when this was zeros / copy / permute / cat)."""
    key = tuple((p.data_ptr(), p._version) for p in params)
    cache = getattr(module, "_wt_cache", None)
    if cache is not None and cache[0] == key:
        return cache[1], cache[2]
    views = [(w.detach(), n_pad) for w, n_pad in sections]
    if any(w.dtype != torch.float32 for w, _ in views):
        views = [(w.float(), n_pad) for w, n_pad in views]
    total = sum(w.shape[0] * (n_pad if n_pad else w.shape[1]) for w, n_pad in views)
    wt = cache[1] if cache is not None and cache[1].numel() == total + 4096 and cache[1].device == views[0][0].device else \
        torch.zeros(total + 4096, dtype=torch.float32, device=views[0][0].device)     # (slack: operand prefetches run past a section's end)
    offsets = ops.pack_transposed(views, wt)
    module._wt_cache = (key, wt, offsets)
    return wt, offsets


def transposed_spacenet(module, params) -> tuple:
    """(wt, offsets): the A operands of the fused backward chain (csrc/train_wave.hip), one section [out / 4][N][4] per product
    d x = d y W -- wt[(o // 4), n, o % 4] = W[o][n], N = the layer's inputs padded to a multiple of 32 -- in the order
    stnerf_train_spacenet_dx takes them: rgb_net.1's 256 backbone columns, stage1.0 .. stage2.4, then density_net.0's weights and
    the colour head as they are.  Cached on the module until a parameter changes."""
    W = [params[2 * i] for i in range(len(params) // 2)]
    pad32 = lambda w: (w.shape[1] + 31) // 32 * 32
    return _transposed(module, params, [(W[8][:, :256], 256)] + [(W[i], pad32(W[i])) for i in range(7)] + [(W[7], 0), (W[9], 0)])


def dx_blob_bf16x3(module, params, with_dpos: bool) -> torch.Tensor:
    """The split-bf16 backward chain's weights (ops.pack_dx_bf16x3), cached on the module until a parameter changes."""
    key = (tuple((p.data_ptr(), p._version) for p in params), bool(with_dpos))
    cache = getattr(module, "_dx_bx_cache", None)
    if cache is None or cache[0] != key:
        W = [params[2 * i] for i in range(len(params) // 2)]
        cache = module._dx_bx_cache = (key, ops.pack_dx_bf16x3(ops.hip.NET_SPACE_TIME if module.use_time else ops.hip.NET_SPACE, W, with_dpos))
    return cache[1]


def _activation_buffers(rows: int, tail: int, device) -> List[torch.Tensor]:
    """Row-major storage of a SpaceNet's layer inputs for `rows` samples: [h4 | PE(pos) + pad] (320), the outputs of stage1.0, 1.2, 1.4
    and stage2.0, 2.2 (256 each), [g3 | relu PE(dir) | relu PE(t)] (256 + tail, padded), rgb_net.1's output (128); and, last, the eight
    ReLU masks as bit planes (8, rows, 8) int32 -- what the backward chain reads instead of the activations."""
    return ([_buf(rows, 320, device)] + [_buf(rows, 256, device) for _ in range(5)] + [_buf(rows, 256 + tail, device), _buf(rows, 128, device)]
            + [torch.empty(8, rows, 8, dtype=torch.int32, device=device)])


def _act_views(bufs: List[torch.Tensor]) -> List[torch.Tensor]:
    """The eight matrices stnerf_train_spacenet_fwd writes / stnerf_train_spacenet_dx masks with (the post-ReLU outputs of
    stage1.0 .. stage2.4 and rgb_net.1), as views of ``_activation_buffers``."""
    Cc, h0, h1, h2, g0, g1, R, t0 = bufs[:8]
    return [h0[:, :256], h1[:, :256], h2[:, :256], Cc[:, :256], g0[:, :256], g1[:, :256], R[:, :256], t0[:, :128]]


class SpaceNetFunction(torch.autograd.Function):
    """(rgb, sigma) = SpaceNet(pos, dirs, times).  ``flavour`` = (include_input, use_dir, use_time, deep_rgb); ``params`` =
    weight, bias of stage1.{0,2,4,6}, stage2.{0,2,4}, density_net.0, rgb_net.{1,3[,5,7]} in that order."""

    @staticmethod
    def forward(ctx, module, pos, dirs, times, *params):
        n, ns = pos.shape[0], pos.shape[1]
        dev = pos.device
        raw = torch.empty(n, ns, 4, dtype=torch.float32, device=dev)
        fused = FUSED_BACKWARD and module.include_input and module.use_dir and not module.deep_rgb
        kept = []
        with torch.no_grad():
            # The backward walks back through the ReLU masks of THIS evaluation.  Fused path: the stage kernel of the module's own
            # arithmetic (split bf16 by default -- the kernel that renders, with the tap; STNERF_TRAIN_FWD=fp32: exact f32 whatever
            # the module renders with, as up to round 5); per-layer path: exact f32.
            ctx.fwd_precision = _train_fwd_precision(module) if fused else "fp32"
            with ops.training_pack():
                packed = module._packed(ctx.fwd_precision)
            if fused and _may_keep(n * ns * ACT_FLOATS_PER_SAMPLE * 4):
                dir_w_, time_w_ = 27, (21 if module.use_time else 0)
                kept = _activation_buffers(n * ns, dir_w_ + time_w_, dev)
                _account_kept(kept, n * ns * ACT_FLOATS_PER_SAMPLE * 4)
                ops.train_spacenet_fwd(packed, pos.detach(), dirs.detach(), times, raw, _act_views(kept), kept[0][:, 256:320], kept[8])
                # rgb_net.1's direction / time columns of its input matrix (the right operand of its weight gradient): written HERE, so
                # that the backward reads the saved tensors and never writes into them
                ops.train_encode(dirs.detach(), kept[6][:, 256:256 + dir_w_], 4, True, rows_per_src=ns, relu=True)
                if module.use_time:
                    ops.train_encode(times.detach().reshape(-1, 1).float(), kept[6][:, 256 + dir_w_:256 + dir_w_ + time_w_], 10, True,
                                     rows_per_src=ns, relu=True)
            else:
                ops.spacenet_fwd(packed, pos.detach().contiguous(), dirs.detach(), times, raw)
        ctx.module, ctx.has_times, ctx.kept = module, times is not None, bool(kept)
        ctx.save_for_backward(pos.detach(), dirs.detach(), times.detach() if times is not None else pos.new_empty(0),
                              *[p.detach() for p in params], *kept)
        ctx.set_materialize_grads(False)
        return raw[..., :3], raw[..., 3:]

    @staticmethod
    def backward(ctx, d_rgb, d_sigma):
        pos, dirs, times, *params = ctx.saved_tensors
        kept = []
        if ctx.kept:
            params, kept = params[:-9], list(params[-9:])
        m_ = ctx.module
        inc, use_dir, use_time, deep = m_.include_input, m_.use_dir, m_.use_time, m_.deep_rgb
        n, ns = pos.shape[0], pos.shape[1]
        dev = pos.device
        fused = ctx.kept or (FUSED_BACKWARD and inc and use_dir and not deep)
        W = [] if fused else [_padded_weight(params[2 * i]) for i in range(len(params) // 2)]
        B = [] if fused else [params[2 * i + 1].detach().float().contiguous() for i in range(len(params) // 2)]
        # (the fused path's first chunk writes every gradient whole: nothing to clear -- 20 fill launches less per backward)
        fresh = torch.empty_like if fused and n > 0 else torch.zeros_like
        gW = [fresh(params[2 * i], dtype=torch.float32) for i in range(len(params) // 2)]
        gB = [fresh(params[2 * i + 1], dtype=torch.float32) for i in range(len(params) // 2)]
        d_pos = torch.zeros(n * ns, 3, dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        pe = 3 * (int(inc) + 20)
        dir_w = 3 * (int(inc) + 8) if use_dir else 0
        time_w = (int(inc) + 20) if use_time else 0
        n_tail = len(params) // 2 - 8                         # rgb_net: 2 linear layers, 4 with deep_rgb
        flat_pos = pos.reshape(n * ns, 3)
        rays_per_chunk = max(1, CHUNK_SAMPLES // ns)
        if fused:
            # ---- fused launches (csrc/train_wave.hip): the layers' inputs come from the forward itself (kept) or from one more run of
            # its stage kernel per chunk (recomputation); then the whole d x chain with the gradient carried in registers; only the
            # weight gradients and the encodings' chain rule stay per layer
            # the d x chain in the forward's arithmetic: split bf16 (csrc/mlp_bf16x3.hip) or exact f32 (csrc/train_wave.hip)
            dx_bx = ctx.fwd_precision == "bf16x3" and DX_BF16X3
            if dx_bx:
                dx_blob = dx_blob_bf16x3(m_, params, d_pos is not None)
            else:
                wt, offsets = transposed_spacenet(m_, params)
            with ops.training_pack():
                packed = m_._packed(ctx.fwd_precision)      # (the recomputation: the arithmetic of the forward, bit for bit)
            for r0 in range(0, n, rays_per_chunk):
                r1 = min(n, r0 + rays_per_chunk)
                M = (r1 - r0) * ns
                x = flat_pos[r0 * ns:r1 * ns]
                if kept:
                    bufs = [b[r0 * ns:r1 * ns] for b in kept[:8]] + [kept[8][:, r0 * ns:r1 * ns]]
                else:
                    bufs = _activation_buffers(M, dir_w + time_w, dev)
                    raw_tmp = torch.empty(r1 - r0, ns, 4, dtype=torch.float32, device=dev)
                    ops.train_spacenet_fwd(packed, pos[r0:r1], dirs[r0:r1], times[r0:r1] if use_time else None, raw_tmp, _act_views(bufs),
                                           bufs[0][:, 256:320], bufs[8])
                Cc, R = bufs[0], bufs[6]
                acts = _act_views(bufs)
                if not kept:                      # (kept: the forward wrote these columns)
                    ops.train_encode(dirs[r0:r1], R[:, 256:256 + dir_w], 4, inc, rows_per_src=ns, relu=True)
                    if use_time:
                        ops.train_encode(times[r0:r1].reshape(-1, 1).float(), R[:, 256 + dir_w:256 + dir_w + time_w], 10, inc, rows_per_src=ns,
                                         relu=True)
                d_raw = torch.zeros(M, 4, dtype=torch.float32, device=dev)
                if d_rgb is not None:
                    d_raw[:, :3] = d_rgb[r0:r1].reshape(M, 3)
                if d_sigma is not None:
                    d_raw[:, 3:] = d_sigma[r0:r1].reshape(M, 1)
                dys = [_buf(M, 256, dev)[:, :256] for _ in range(7)] + [_buf(M, 128, dev)[:, :128]]
                dpe = _buf(M, 64, dev)[:, :64] if d_pos is not None else None
                if dx_bx:
                    dpe_skip = _buf(M, 64, dev)[:, :64] if d_pos is not None else None
                    ops.train_spacenet_dx_bf16x3(dx_blob, d_raw, bufs[8], dys, dpe, dpe_skip)
                    if dpe is not None:
                        dpe += dpe_skip          # (d y0 W_stage1.0 + d y4 W_stage2.0[:, 256:]: see stnerf_train_spacenet_dx_bf16x3)
                else:
                    ops.train_spacenet_dx(wt, offsets, d_raw, bufs[8], dys, dpe)
                acc = r0 > 0
                xin = [Cc[:, 256:256 + pe], acts[0], acts[1], acts[2], Cc[:, :256 + pe], acts[4], acts[5]]
                dS = _buf(M, 1, dev)
                dS[:, :1] = d_raw[:, 3:]
                # every weight and bias gradient of the network: one launch + one reduction (stnerf_train_dw_batch)
                batch = ([(dys[i], xin[i], gW[i], gB[i]) for i in range(7)] +
                         [(dS[:, :1], acts[6], gW[7], gB[7]), (dys[7], R[:, :256 + dir_w + time_w], gW[8], gB[8]),
                          (d_raw[:, :3], acts[7], gW[9], gB[9])])
                assert len(batch) == len(gW) == len(gB)     # (gW / gB start as torch.empty: the first chunk must write every one whole)
                _weight_gradients(batch, acc)
                if d_pos is not None:
                    ops.train_encode_bwd(x, dpe[:, :pe], d_pos[r0 * ns:r1 * ns], 10, inc)
        for r0 in ([] if fused else range(0, n, rays_per_chunk)):
            r1 = min(n, r0 + rays_per_chunk)
            M = (r1 - r0) * ns
            x = flat_pos[r0 * ns:r1 * ns]
            # ---- recompute, keeping every layer's input (modeling/spacenet.py:101-160) -----------------------------------
            Cc = _buf(M, 256 + pe, dev)                       # [h4 | PE(pos)]: stage2.0's input, the skip connection in place
            P = Cc[:, 256:256 + pe]
            ops.train_encode(x, P, 10, inc)
            H = [_buf(M, 256, dev) for _ in range(3)]
            ops.train_linear_fwd(P, W[0], B[0], H[0][:, :256], True)
            ops.train_linear_fwd(H[0][:, :256], W[1], B[1], H[1][:, :256], True)
            ops.train_linear_fwd(H[1][:, :256], W[2], B[2], H[2][:, :256], True)
            ops.train_linear_fwd(H[2][:, :256], W[3], B[3], Cc[:, :256], True)
            G = [_buf(M, 256, dev) for _ in range(2)]
            R = _buf(M, 256 + dir_w + time_w, dev)            # [g3 | relu(PE(dir)) | relu(PE(t))]: rgb_net's input (:141-151, :80)
            ops.train_linear_fwd(Cc[:, :256 + pe], W[4], B[4], G[0][:, :256], True)
            ops.train_linear_fwd(G[0][:, :256], W[5], B[5], G[1][:, :256], True)
            ops.train_linear_fwd(G[1][:, :256], W[6], B[6], R[:, :256], True)
            if use_dir:
                ops.train_encode(dirs[r0:r1], R[:, 256:256 + dir_w], 4, inc, rows_per_src=ns, relu=True)
            if use_time:
                ops.train_encode(times[r0:r1].reshape(-1, 1).float(), R[:, 256 + dir_w:256 + dir_w + time_w], 10, inc, rows_per_src=ns,
                                 relu=True)
            T = [_buf(M, 128, dev) for _ in range(n_tail - 1)]   # hidden activations of rgb_net
            src = R[:, :256 + dir_w + time_w]
            for j in range(n_tail - 1):
                ops.train_linear_fwd(src, W[8 + j], B[8 + j], T[j][:, :128], True)
                src = T[j][:, :128]
            # ---- backwards --------------------------------------------------------------------------------------------------
            first = r0 == 0
            acc = not first
            dA, dB_ = _buf(M, 256, dev), _buf(M, 256, dev)
            g3 = R[:, :256]
            have = False                                       # dA[:, :256] holds d(stage2.4 pre-activation) contributions
            if d_rgb is not None:
                dO = _buf(M, 3, dev)
                dO[:, :3] = d_rgb[r0:r1].reshape(M, 3)
                dy = dO[:, :3]
                for j in range(n_tail - 1, -1, -1):            # rgb_net's linear layers, last first
                    xin = T[j - 1][:, :128] if j > 0 else R[:, :256 + dir_w + time_w]
                    _dw(dy, xin, gW[8 + j], gB[8 + j], acc)
                    if j > 0:
                        dT = _buf(M, 128, dev)
                        ops.train_linear_dx(dy, W[8 + j], dT[:, :128], mask=T[j - 1][:, :128])
                        dy = dT[:, :128]
                    else:                                      # into g3 only: the encodings are not differentiated
                        ops.train_linear_dx(dy, W[8][:, :256], dA[:, :256], mask=g3)
                        have = True
            if d_sigma is not None:
                dS = _buf(M, 1, dev)
                dS[:, :1] = d_sigma[r0:r1].reshape(M, 1)
                _dw(dS[:, :1], g3, gW[7], gB[7], acc)
                ops.train_linear_dx(dS[:, :1], W[7], dA[:, :256], mask=g3, accumulate=have)
                have = True
            if not have:
                continue
            # stage2 (dA = d pre-activation of stage2.4)
            _dw(dA[:, :256], G[1][:, :256], gW[6], gB[6], acc)
            ops.train_linear_dx(dA[:, :256], W[6], dB_[:, :256], mask=G[1][:, :256])
            _dw(dB_[:, :256], G[0][:, :256], gW[5], gB[5], acc)
            ops.train_linear_dx(dB_[:, :256], W[5], dA[:, :256], mask=G[0][:, :256])
            _dw(dA[:, :256], Cc[:, :256 + pe], gW[4], gB[4], acc)
            dP = _buf(M, pe, dev)
            ops.train_linear_dx(dA[:, :256], W[4][:, :256], dB_[:, :256], mask=Cc[:, :256])       # -> h4 (ReLU of stage1.6)
            ops.train_linear_dx(dA[:, :256], W[4][:, 256:256 + pe], dP[:, :pe])                   # -> PE(pos), the skip connection
            # stage1
            _dw(dB_[:, :256], H[2][:, :256], gW[3], gB[3], acc)
            ops.train_linear_dx(dB_[:, :256], W[3], dA[:, :256], mask=H[2][:, :256])
            _dw(dA[:, :256], H[1][:, :256], gW[2], gB[2], acc)
            ops.train_linear_dx(dA[:, :256], W[2], dB_[:, :256], mask=H[1][:, :256])
            _dw(dB_[:, :256], H[0][:, :256], gW[1], gB[1], acc)
            ops.train_linear_dx(dB_[:, :256], W[1], dA[:, :256], mask=H[0][:, :256])
            _dw(dA[:, :256], P, gW[0], gB[0], acc)
            if d_pos is not None:
                ops.train_linear_dx(dA[:, :256], W[0], dP[:, :pe], accumulate=True)
                ops.train_encode_bwd(x, dP[:, :pe], d_pos[r0 * ns:r1 * ns], 10, inc)
        grads: List[Optional[torch.Tensor]] = []
        for i in range(len(gW)):
            grads += [gW[i] if ctx.needs_input_grad[4 + 2 * i] else None, gB[i] if ctx.needs_input_grad[5 + 2 * i] else None]
        return (None, d_pos.reshape(n, ns, 3) if d_pos is not None else None, None, None, *grads)


MOTION_ACT_FLOATS_PER_SAMPLE = 96 + 5 * 128 + 5 * 4


def transposed_motionnet(module, params) -> tuple:
    """(wt, offsets): the A operands of the MotionNet's backward chain (stnerf_train_motionnet_dx): sections [128 / 4][128][4] of
    motion_net.0 (84 inputs zero-padded to 128), .2, .4, .6, .8, then the flow head [3][128] as it is.  Cached per parameter version."""
    W = [params[2 * i] for i in range(6)]
    return _transposed(module, params, [(W[i], 128) for i in range(5)] + [(W[5], 0)])


def _motion_buffers(rows: int, device) -> List[torch.Tensor]:
    """The staged encoding (rows, 96), the five post-ReLU outputs (rows, 128) and the masks as bit planes (5, rows, 4) int32."""
    return ([_buf(rows, 96, device)] + [_buf(rows, 128, device) for _ in range(5)] + [torch.empty(5, rows, 4, dtype=torch.int32, device=device)])


class MotionNetFunction(torch.autograd.Function):
    """flow = MotionNet([xyz, t]) (modeling/motion_net.py:34-71).  ``params`` = weight, bias of motion_net.{0,2,4,6,8,10}."""

    @staticmethod
    def forward(ctx, module, xt, *params):
        rows = xt.shape[0]
        fused = FUSED_BACKWARD and module.pos_dim == 84
        kept = []
        with torch.no_grad():
            if fused and _may_keep(rows * MOTION_ACT_FLOATS_PER_SAMPLE * 4):
                # the forward itself keeps what the backward needs (3 KB per row): no recomputation
                kept = _motion_buffers(rows, xt.device)
                _account_kept(kept, rows * MOTION_ACT_FLOATS_PER_SAMPLE * 4)
                flow = torch.empty(rows, 3, dtype=torch.float32, device=xt.device)
                ops.train_motionnet_fwd(module._packed("fp32"), xt.detach().float().contiguous(), flow, kept[0], [b[:, :128] for b in kept[1:6]],
                                        kept[6], plain_time=not module.input_time)
            else:
                xyz = xt[:, :3].detach().reshape(rows, 1, 3).contiguous()
                flow = torch.empty_like(xyz)
                ops.motionnet_fwd(module._packed("fp32"), xyz, xt[:, 3].detach().contiguous(), flow=flow, add_to_xyz=False,
                                  plain_time=not module.input_time)
        ctx.module, ctx.kept = module, bool(kept)
        ctx.save_for_backward(xt.detach(), *[p.detach() for p in params], *kept)
        return flow.reshape(rows, 3)

    @staticmethod
    def backward(ctx, d_flow):
        xt, *params = ctx.saved_tensors
        kept = []
        if ctx.kept:
            params, kept = params[:-7], list(params[-7:])
        m_ = ctx.module
        inc = m_.pos_dim == 84
        dev = xt.device
        rows = xt.shape[0]
        L = len(params) // 2
        fused = ctx.kept or (FUSED_BACKWARD and inc)
        if fused:
            # ---- two fused launches (csrc/train_wave.hip) + one for every weight gradient: the layers' inputs come from the forward
            # (kept) or from one more run of it per chunk; the d x chain carries the gradient in registers
            fresh = torch.empty_like if rows > 0 else torch.zeros_like
            gW = [fresh(params[2 * i], dtype=torch.float32) for i in range(L)]
            gB = [fresh(params[2 * i + 1], dtype=torch.float32) for i in range(L)]
            d_xt = torch.zeros(rows, 4, dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
            wt, offsets = transposed_motionnet(m_, params)
            x4 = xt.float().contiguous()
            for r0 in range(0, rows, CHUNK_SAMPLES):
                r1 = min(rows, r0 + CHUNK_SAMPLES)
                M = r1 - r0
                x = x4[r0:r1]
                if kept:
                    bufs = [b[r0:r1] for b in kept[:6]] + [kept[6][:, r0:r1]]
                else:
                    bufs = _motion_buffers(M, dev)
                    ops.train_motionnet_fwd(m_._packed("fp32"), x, torch.empty(M, 3, dtype=torch.float32, device=dev), bufs[0],
                                            [b[:, :128] for b in bufs[1:6]], bufs[6], plain_time=not m_.input_time)
                E, A = bufs[0], [b[:, :128] for b in bufs[1:6]]
                dO = _buf(M, 3, dev)
                dO[:, :3] = d_flow[r0:r1]
                dys = [_buf(M, 128, dev)[:, :128] for _ in range(5)]
                dE = _buf(M, 96, dev) if d_xt is not None else None
                ops.train_motionnet_dx(wt, offsets, dO[:, :3], bufs[6], dys, dE)
                batch = ([(dys[0], E[:, :84], gW[0], gB[0])] + [(dys[j], A[j - 1], gW[j], gB[j]) for j in range(1, 5)] +
                         [(dO[:, :3], A[4], gW[5], gB[5])])
                assert len(batch) == len(gW) == len(gB)     # (as above: the empty gradient buffers are written whole)
                _weight_gradients(batch, r0 > 0)
                if d_xt is not None:
                    # (the frame-id column gets no gradient: the lerp weights are data)
                    ops.train_encode_bwd(x, dE[:, :84], d_xt[r0:r1, :3], 10, inc)
            grads: List[Optional[torch.Tensor]] = []
            for i in range(L):
                grads += [gW[i] if ctx.needs_input_grad[2 + 2 * i] else None, gB[i] if ctx.needs_input_grad[3 + 2 * i] else None]
            return (None, d_xt, *grads)
        W = [_padded_weight(params[2 * i]) for i in range(L)]
        B = [params[2 * i + 1].detach().float().contiguous() for i in range(L)]
        fresh = torch.empty_like if rows > 0 else torch.zeros_like     # (the first chunk writes every gradient whole)
        gW = [fresh(params[2 * i], dtype=torch.float32) for i in range(L)]
        gB = [fresh(params[2 * i + 1], dtype=torch.float32) for i in range(L)]
        d_xt = torch.zeros(rows, 4, dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        pe = m_.pos_dim
        x4 = xt.float().contiguous()
        for r0 in range(0, rows, CHUNK_SAMPLES):
            r1 = min(rows, r0 + CHUNK_SAMPLES)
            M = r1 - r0
            x = x4[r0:r1]
            E = _buf(M, pe, dev)
            ops.train_encode(x, E[:, :pe], 10, inc, lerp_col=3 if m_.input_time else -1)
            A = [_buf(M, 128, dev) for _ in range(L - 1)]
            src = E[:, :pe]
            for j in range(L - 1):
                ops.train_linear_fwd(src, W[j], B[j], A[j][:, :128], True)
                src = A[j][:, :128]
            acc = r0 > 0
            dO = _buf(M, 3, dev)
            dO[:, :3] = d_flow[r0:r1]
            dy = dO[:, :3]
            layers = []
            for j in range(L - 1, -1, -1):
                xin = A[j - 1][:, :128] if j > 0 else E[:, :pe]
                layers.append((dy, xin, gW[j], gB[j]))
                if j > 0:
                    d_prev = _buf(M, 128, dev)
                    ops.train_linear_dx(dy, W[j], d_prev[:, :128], mask=A[j - 1][:, :128])
                    dy = d_prev[:, :128]
                elif d_xt is not None:
                    dE = _buf(M, pe, dev)
                    ops.train_linear_dx(dy, W[0], dE[:, :pe])
                    # (the frame-id column gets no gradient: the lerp weights are data)
                    ops.train_encode_bwd(x, dE[:, :pe], d_xt[r0:r1, :3], 10, inc)
            _weight_gradients(layers, acc)
        grads: List[Optional[torch.Tensor]] = []
        for i in range(L):
            grads += [gW[i] if ctx.needs_input_grad[2 + 2 * i] else None, gB[i] if ctx.needs_input_grad[3 + 2 * i] else None]
        return (None, d_xt, *grads)
