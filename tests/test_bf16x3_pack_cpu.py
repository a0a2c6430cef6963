"""The bf16x3 packer (stnerf_pack_net_bf16x3, csrc/mlp_bf16x3.hip -- host code, no GPU): the blob it writes is decoded here
with an independent numpy restatement of the layout in csrc/mlp_bf16x3.h and checked value by value:
  * the three bf16 pieces of every weight add up to the fp32 weight EXACTLY (8 + 8 + 8 significand bits), zero padding is zero;
  * every MFMA layer's weights sit where the kernel's K loops read them: [pass][K step][block of 32 outputs][piece][lane][8],
    K positions through the accumulator -> B-operand map of the hidden layers / the staged-encoding map of the first layers;
  * the consts section holds the bias vectors and head weights in the kernel's LDS order; the f32 section is the exact-f32 blob.
Reference: modeling/spacenet.py:45-86 (layer list), modeling/motion_net.py:20-32."""
import ctypes as C

import numpy as np
import pytest

from stnerf_amd import hip, synthetic as syn

SLOT, UNIT, CHUNK = 24576, 3072, 1024


def bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def kmap_hidden(t, h, j):
    return 32 * (t >> 1) + 8 * (2 * (t & 1) + (j >> 2)) + 4 * h + (j & 3)


def kmap_enc(t, h, j):
    return 16 * t + 8 * h + j


def _tensors(kind, rs):
    if kind == hip.NET_MOTION:
        sd = syn.motionnet_state("net", rs)
        keys = [f"motion_net.{j}" for j in (0, 2, 4, 6, 8, 10)]
    else:
        deep = kind in (hip.NET_SPACE_DEEP, hip.NET_SPACE_TIME_DEEP)
        sd = syn.spacenet_state("net", rs, kind in (hip.NET_SPACE_TIME, hip.NET_SPACE_TIME_DEEP), deep_rgb=deep)
        keys = ["stage1.0", "stage1.2", "stage1.4", "stage1.6", "stage2.0", "stage2.2", "stage2.4", "density_net.0", "rgb_net.1", "rgb_net.3"]
        if deep:
            keys += ["rgb_net.5", "rgb_net.7"]
    ws = [np.ascontiguousarray(sd[f"net.{k}.weight"].numpy(), dtype=np.float32) for k in keys]
    bs = [np.ascontiguousarray(sd[f"net.{k}.bias"].numpy(), dtype=np.float32) for k in keys]
    return ws, bs


def _pack(fn_bytes, fn_pack, kind, ws, bs):
    lib = hip.lib()
    nbytes = getattr(lib, fn_bytes)(kind)
    assert nbytes > 0
    dst = np.full(nbytes, 0xAB, dtype=np.uint8)
    wp = (C.c_void_p * len(ws))(*(w.ctypes.data for w in ws))
    bp = (C.c_void_p * len(bs))(*(b.ctypes.data for b in bs))
    rc = getattr(lib, fn_pack)(kind, wp, bp, len(ws), C.c_void_p(dst.ctypes.data), nbytes)
    assert rc == 0, hip.last_error()
    return dst


def _decode_pass(stream, off, n0, ksteps, W, col):
    """Checks one pass (128 output rows from n0) of `ksteps` K steps at byte offset `off`; returns the bytes consumed."""
    u16 = stream[off: off + ksteps * 4 * UNIT].view(np.uint16).reshape(ksteps, 4, 3, 64, 8)      # [t][fb][piece][lane][j]
    pieces = bf16_to_f32(u16)
    total = pieces[:, :, 0].astype(np.float64) + pieces[:, :, 1] + pieces[:, :, 2]
    want = np.zeros((ksteps, 4, 64, 8), dtype=np.float64)
    for t in range(ksteps):
        for lane in range(64):
            h, c = lane >> 5, lane & 31
            for j in range(8):
                k = col(t, h, j)
                if k >= 0:
                    want[t, :, lane, j] = W[n0 + 32 * np.arange(4) + c, k]
    assert np.array_equal(total, want), "pieces do not add up to the fp32 weights at the positions the K loops read"
    # each piece is what round-to-nearest of the remainder gives: |p1| <= half an ulp of p0's 8-bit grid, |p2| likewise of p1
    p0, p1, p2 = pieces[:, :, 0].astype(np.float64), pieces[:, :, 1].astype(np.float64), pieces[:, :, 2].astype(np.float64)
    nz = p0 != 0
    assert np.all(np.abs(p1[nz]) <= np.abs(p0[nz]) * 2.0 ** -8) and np.all(np.abs(p2[nz]) <= np.abs(p0[nz]) * 2.0 ** -16)
    return ksteps * 4 * UNIT


@pytest.mark.parametrize("kind", [hip.NET_SPACE, hip.NET_SPACE_TIME, hip.NET_SPACE_TIME_DEEP, hip.NET_MOTION])
def test_bf16x3_blob_layout_and_exact_split(kind):
    rs = np.random.RandomState(17 + kind)
    ws, bs = _tensors(kind, rs)
    blob = _pack("stnerf_packed_bytes_bf16x3", "stnerf_pack_net_bf16x3", kind, ws, bs)
    f32 = _pack("stnerf_packed_bytes", "stnerf_pack_net", kind, ws, bs)
    # ---- f32 section = the exact-f32 blob
    assert np.array_equal(blob[: f32.size], f32)
    consts_off = (f32.size + 1023) // 1024 * 1024
    space = kind != hip.NET_MOTION
    consts = blob[consts_off: consts_off + (3072 if space else 1024) * 4].view(np.float32)
    stream_off = consts_off + consts.size * 4
    stream = blob[stream_off:]
    hidden = kmap_hidden
    off = 0
    if space:
        deep = kind in (hip.NET_SPACE_DEEP, hip.NET_SPACE_TIME_DEEP)
        nt = len(ws)
        for i in range(7):
            assert np.array_equal(consts[256 * i: 256 * i + 256], bs[i])
        assert np.array_equal(consts[2048:2304], ws[7].ravel()) and np.array_equal(consts[2304:2688], ws[nt - 1].ravel())
        if deep:
            assert np.array_equal(consts[1792:1920], bs[9]) and np.array_equal(consts[1920:2048], bs[10])
        for i in range(7):
            for half in range(2):
                if i == 0:
                    off += _decode_pass(stream, off, 128 * half, 4, ws[0], lambda t, h, j: kmap_enc(t, h, j) if kmap_enc(t, h, j) < 63 else -1)
                elif i == 4:   # stage2.0: the 256 features, then PE(pos) (modeling/spacenet.py:56-57,136-138)
                    off += _decode_pass(stream, off, 128 * half, 20, ws[4],
                                        lambda t, h, j: hidden(t, h, j) if t < 16 else (256 + kmap_enc(t - 16, h, j) if kmap_enc(t - 16, h, j) < 63 else -1))
                else:
                    off += _decode_pass(stream, off, 128 * half, 16, ws[i], hidden)
        off += _decode_pass(stream, off, 0, 16, ws[8], hidden)           # rgb_net.1: the 256 backbone columns only
        for i in range(2 if deep else 0):
            off += _decode_pass(stream, off, 0, 8, ws[9 + i], hidden)
        assert off == (112 + (8 if deep else 0)) * SLOT
    else:
        for i in range(5):
            assert np.array_equal(consts[128 * i: 128 * i + 128], bs[i])
        assert np.array_equal(consts[640:1024], ws[5].ravel())
        off += _decode_pass(stream, off, 0, 6, ws[0], lambda t, h, j: kmap_enc(t, h, j) if kmap_enc(t, h, j) < 84 else -1)
        for i in range(1, 5):
            off += _decode_pass(stream, off, 0, 8, ws[i], hidden)
        assert off == 19 * SLOT
    assert off == stream.size, (off, stream.size)


def test_bf16x3_packer_takes_any_fp32_weight():
    """No range limit (fp16x3's packer refuses |W| >= 234): huge, tiny and denormal-adjacent weights split exactly."""
    rs = np.random.RandomState(3)
    ws, bs = _tensors(hip.NET_MOTION, rs)
    ws[1] = ws[1].copy()
    ws[1][0, :8] = [3.0e38, -1.7e30, 65520.0, 234.0, 1e-30, -3e-38, 1.17549435e-38, 0.0]
    blob = _pack("stnerf_packed_bytes_bf16x3", "stnerf_pack_net_bf16x3", hip.NET_MOTION, ws, bs)
    f32_bytes = hip.lib().stnerf_packed_bytes(hip.NET_MOTION)
    stream = blob[(f32_bytes + 1023) // 1024 * 1024 + 4096:]
    u16 = stream[6 * 4 * UNIT: 6 * 4 * UNIT + 8 * 4 * UNIT].view(np.uint16).reshape(8, 4, 3, 64, 8)   # motion_net.2
    pieces = bf16_to_f32(u16).astype(np.float64)
    total = pieces[:, :, 0] + pieces[:, :, 1] + pieces[:, :, 2]
    # row 0 = block 0, c = 0 (lanes 0 and 32): input column k sits at K step t, lane half h, element j with kmap_hidden(t, h, j) == k
    got = {}
    for t in range(8):
        for h in range(2):
            for j in range(8):
                got[kmap_hidden(t, h, j)] = total[t, 0, 32 * h, j]
    for k in range(8):
        w = float(ws[1][0, k])
        if abs(w) > 1e-30:
            assert got[k] == w, (k, w, got[k])
        else:   # below 2^-126 * 2^16 the third piece leaves the normal range: exact to the bf16 denormal grid
            assert abs(got[k] - w) <= 2.0 ** -133, (k, w, got[k])



@pytest.mark.parametrize("bad", [float("nan"), float("inf"), -float("inf"), 3.3999e38, -3.4028235e38])
def test_bf16x3_packer_refuses_what_it_cannot_split(bad):
    """include/stnerf.h: NaN, +-inf and |w| above bf16's largest finite value (3.3895e38) cannot be written as three bf16 pieces
    that add up (the leading piece rounds to inf): STNERF_EINVAL instead of a network that silently returns NaN.  The exact-f32
    packer takes the same tensors."""
    rs = np.random.RandomState(4)
    lib = hip.lib()
    for kind, which in ((hip.NET_MOTION, "w"), (hip.NET_SPACE_TIME, "b"), (hip.NET_SPACE, "w")):
        ws, bs = _tensors(kind, rs)
        tgt = ws if which == "w" else bs
        tgt[2] = tgt[2].copy()
        tgt[2].reshape(-1)[5] = bad
        nbytes = lib.stnerf_packed_bytes_bf16x3(kind)
        dst = np.zeros(nbytes, dtype=np.uint8)
        wp = (C.c_void_p * len(ws))(*(w.ctypes.data for w in ws))
        bp = (C.c_void_p * len(bs))(*(b.ctypes.data for b in bs))
        assert lib.stnerf_pack_net_bf16x3(kind, wp, bp, len(ws), C.c_void_p(dst.ctypes.data), nbytes) == hip.EINVAL
        assert "bf16" in hip.last_error()
        f32 = np.zeros(lib.stnerf_packed_bytes(kind), dtype=np.uint8)
        assert lib.stnerf_pack_net(kind, wp, bp, len(ws), C.c_void_p(f32.ctypes.data), f32.size) == 0
    # the largest value that CAN be split is accepted
    ws, bs = _tensors(hip.NET_MOTION, rs)
    ws[1] = ws[1].copy()
    ws[1][0, 0] = 3.3895313892515355e38
    _pack("stnerf_packed_bytes_bf16x3", "stnerf_pack_net_bf16x3", hip.NET_MOTION, ws, bs)
