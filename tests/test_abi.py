"""The C-ABI library: loads, exports every symbol include/stnerf.h declares, host-side entry points
(weight packing, argument validation) behave.  CPU only -- no kernel is launched."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import REPO
from stnerf_amd import hip, synthetic as syn


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(hip.LIB_PATH):
        import importlib.util
        spec = importlib.util.spec_from_file_location("stnerf_build", os.path.join(REPO, "st-nerf_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()
    return hip.lib()


def test_exports_every_declared_symbol(lib):
    header = open(os.path.join(REPO, "include", "stnerf.h")).read()
    declared = set(re.findall(r"\b(stnerf_[a-z_0-9]+)\s*\(", header))
    assert declared, "header parse failed"
    assert declared == set(hip.exported_symbols()), declared ^ set(hip.exported_symbols())
    for name in declared:
        assert getattr(lib, name) is not None
    assert b"gfx950" in lib.stnerf_version()


def test_struct_layouts_match_header():
    assert C.sizeof(hip.LayerEdit) == 24
    assert C.sizeof(hip.CompositeParams) == 16 + 4 + 4 * 4 * hip.MAX_LAYERS


def test_packed_sizes_and_bad_kind(lib):
    # 7 backbone layers [Kq][256][4] + biases + heads (DESIGN.md "packed weight layout")
    kq = [16, 64, 64, 64, 80, 64, 64]
    base = sum(k * 256 * 4 + 256 for k in kq) + 256 + 4 + 128 + 3 * 128 + 4
    assert lib.stnerf_packed_bytes(hip.NET_SPACE) == 4 * (base + 72 * 128 * 4)
    assert lib.stnerf_packed_bytes(hip.NET_SPACE_TIME) == 4 * (base + 76 * 128 * 4)
    assert lib.stnerf_packed_bytes(hip.NET_MOTION) == 4 * ((22 + 4 * 32) * 128 * 4 + 5 * 128 + 3 * 128 + 4)
    assert lib.stnerf_packed_bytes(99) == hip.EINVAL
    assert "unknown net kind" in hip.last_error()


def _pack(lib, kind, ws, bs):
    nbytes = lib.stnerf_packed_bytes(kind)
    dst = np.full(nbytes // 4, np.nan, dtype=np.float32)
    wp = (C.c_void_p * len(ws))(*(w.ctypes.data for w in ws))
    bp = (C.c_void_p * len(bs))(*(b.ctypes.data for b in bs))
    rc = lib.stnerf_pack_net(kind, wp, bp, len(ws), C.c_void_p(dst.ctypes.data), nbytes)
    return rc, dst


def test_pack_spacenet_layout(lib):
    from stnerf_amd import ops
    sd = syn.spacenet_state("net", np.random.RandomState(3), True)
    ws = [np.ascontiguousarray(sd[f"net.{k}.weight"].numpy()) for k in ops.SPACENET_KEYS]
    bs = [np.ascontiguousarray(sd[f"net.{k}.bias"].numpy()) for k in ops.SPACENET_KEYS]
    rc, dst = _pack(lib, hip.NET_SPACE_TIME, ws, bs)
    assert rc == 0 and np.isfinite(dst).all()
    off = 0
    for i, kq in enumerate([16, 64, 64, 64, 80, 64, 64]):
        blk = dst[off:off + kq * 256 * 4].reshape(kq, 256, 4)          # [k/4][n][k%4]
        w = ws[i]                                                      # (256, in)
        dense = blk.transpose(1, 0, 2).reshape(256, kq * 4)
        assert np.array_equal(dense[:, :w.shape[1]], w) and not dense[:, w.shape[1]:].any()
        off += kq * 256 * 4
        assert np.array_equal(dst[off:off + 256], bs[i])
        off += 256
    assert np.array_equal(dst[off:off + 256], ws[7][0]) and dst[off + 256] == bs[7][0]
    off += 260
    blk = dst[off:off + 76 * 128 * 4].reshape(76, 128, 4).transpose(1, 0, 2).reshape(128, 304)
    assert np.array_equal(blk, ws[8])
    # wrong tensor count / too-small destination are argument errors, not crashes
    rc, _ = _pack(lib, hip.NET_SPACE_TIME, ws[:9], bs[:9])
    assert rc == hip.EINVAL and "10 tensors" in hip.last_error()
    wp = (C.c_void_p * 10)(*(w.ctypes.data for w in ws))
    bp = (C.c_void_p * 10)(*(b.ctypes.data for b in bs))
    small = np.zeros(16, dtype=np.float32)
    assert lib.stnerf_pack_net(hip.NET_SPACE_TIME, wp, bp, 10, C.c_void_p(small.ctypes.data), 64) == hip.EINVAL


def test_argument_errors_before_any_launch(lib):
    null = C.c_void_p(0)
    assert lib.stnerf_sample_coarse(null, 4, 9, null, 0, 3, 8, null, 0, 0, 0, 0, None, None, null, null, null, null) == hip.EINVAL
    assert lib.stnerf_spacenet_fwd(7, null, 1, 1, null, null, null, 0, null, 0, null, 0, null, 0, null, null) == hip.EINVAL
    assert "bad kind" in hip.last_error()


def test_product_path_refuses_cpu_tensors():
    """No silent fallback: handing the HIP path a CPU tensor is an error."""
    from stnerf_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.intersect(torch.zeros(4, 9), torch.zeros(3, 8, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.composite(torch.zeros(2, 1, 4), torch.zeros(2, 1, 4, 4), None)


def test_render_structs_match_header_layout():
    # stnerf_nets: 2 + 3*16 pointers; stnerf_render_params: 13 ints, 16 ints, 5 floats, u64, 3 x i64 (ray window), 2*16 edits, 3 floats (+pad)
    assert C.sizeof(hip.Nets) == 8 * (2 + 3 * hip.MAX_LAYERS)
    assert C.sizeof(hip.RenderParams) == 52 + 64 + 20 + 0 + 8 + 3 * 8 + 2 * 16 * 24 + 12 + 4
    assert C.sizeof(hip.ProfileRecord) == 40


def test_ctypes_structs_agree_with_the_c_compiler(tmp_path):
    """include/stnerf.h compiled as plain C (gcc): sizeof and offsetof of every struct field == the ctypes mirror."""
    import os, shutil, subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    pairs = {"stnerf_layer_edit": hip.LayerEdit, "stnerf_composite_params": hip.CompositeParams, "stnerf_nets": hip.Nets,
             "stnerf_render_params": hip.RenderParams, "stnerf_profile_record": hip.ProfileRecord, "stnerf_dw_problem": hip.DwProblem,
             "stnerf_transpose_section": hip.TransposeSection}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "stnerf.h"', 'int main(void){']
    for cname, cls in pairs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append('return 0;}')
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", inc, str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_plain_c_program_links_and_calls_the_library(tmp_path):
    """The boundary is a C ABI: a C99 translation unit including only include/stnerf.h links against
    libstnerf_hip.so and calls the entry points that need no GPU (sizes, argument errors, error string)."""
    import os, shutil, subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "st-nerf_amd")
    if not os.path.exists(os.path.join(libdir, "libstnerf_hip.so")):
        pytest.skip("library not built")
    src = tmp_path / "client.c"
    src.write_text(r"""
#include <stdio.h>
#include <string.h>
#include "stnerf.h"
int main(void) {
    printf("version %s\n", stnerf_version());
    printf("space %lld time %lld motion %lld deep %lld\n", (long long)stnerf_packed_bytes(STNERF_NET_SPACE),
           (long long)stnerf_packed_bytes(STNERF_NET_SPACE_TIME), (long long)stnerf_packed_bytes(STNERF_NET_MOTION),
           (long long)stnerf_packed_bytes(STNERF_NET_SPACE_TIME_DEEP));
    printf("workspace %lld\n", (long long)stnerf_render_workspace_bytes(3584, 3, 90, 30, 0));
    int rc = stnerf_composite(NULL, NULL, NULL, 8, 3, 64, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
    printf("rc %d err %s\n", rc, stnerf_last_error());
    return rc == STNERF_EINVAL && strlen(stnerf_last_error()) > 0 ? 0 : 1;
}
""")
    exe = tmp_path / "client"
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-lstnerf_hip", f"-Wl,-rpath,{libdir}", f"-Wl,-rpath-link,{torch_lib}",
                    f"-Wl,-rpath,{torch_lib}"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert "version stnerf-hip" in out and "rc -1" in out
    sizes = dict(zip(("space", "time", "motion", "deep"), map(int, out.splitlines()[1].split()[1::2])))
    assert sizes["space"] == hip.lib().stnerf_packed_bytes(hip.NET_SPACE) and sizes["deep"] > sizes["time"] > sizes["space"] > sizes["motion"]


def test_render_workspace_query_and_argument_errors(lib):
    nb = lib.stnerf_render_workspace_bytes(1000, 3, 64, 64, 0)
    # per (ray, layer): the coarse block t + xyz + raw + weights = 9 n1 floats, which the fine network outputs (4 S) are written
    # over once the resampler has read it; fine depths + points (1 + 3) S; rgb_net.1's per-ray part (128)
    floats = 1000 * 3 * (max(64 * (1 + 3 + 4 + 1), 128 * 4) + 128 * (1 + 3) + 128)
    assert nb >= 4 * floats and nb < 4 * floats + 3 * 1000 * 4 + 1000 + 8192      # + ray lists, one flag byte per ray, counters, alignment
    old = 1000 * 3 * (64 * (1 + 3 + 4 + 1) + 128 * (1 + 3 + 4) + 128)             # (round 3: every buffer on its own)
    assert nb < 0.72 * 4 * old
    assert lib.stnerf_render_workspace_bytes(1000, 3, 64, 64, 1) < nb
    assert lib.stnerf_render_workspace_bytes(10, 99, 64, 64, 0) == hip.EINVAL
    null = C.c_void_p(0)
    assert lib.stnerf_render_rays(null, 4, null, 0, None, None, null, null, null, 0, null, null, null, null, null, null) == hip.EINVAL


def test_training_batch_entries_check_their_arguments_on_the_host(lib):
    """stnerf_train_dw_batch_workspace_bytes / stnerf_train_dw_batch / stnerf_pack_transposed: host arithmetic and argument errors
    before any launch (no GPU here): the workspace of a SpaceNet's ten layers, the limits (16 layers, 64 tiles, 16-byte aligned rows),
    the section table's bounds."""
    fake = 1 << 20                                     # (never dereferenced)
    def problems(shapes, db=True, lddy=None):
        arr = (hip.DwProblem * len(shapes))()
        for i, (n, k) in enumerate(shapes):
            arr[i] = hip.DwProblem(fake, lddy or (n + 3) // 4 * 4, fake, (k + 3) // 4 * 4, fake, k, fake if db else None, n, k)
        return arr
    space = [(256, 63), (256, 256), (256, 256), (256, 256), (256, 319), (256, 256), (256, 256), (1, 256), (128, 304), (3, 128)]
    m = 262144

    def tiles(shapes, db=True):
        """csrc/train_dw.hip's tiling: (rows, cols, carries the bias sums) per wave tile."""
        out = []
        for n, k in shapes:
            rows = 32 if n <= 4 else 128
            for n0 in range(0, n, rows):
                for k0 in range(0, k, 128):
                    left = k - k0
                    out.append((rows, 32 * (4 if left > 96 else 3 if left > 64 else 2 if left > 32 else 1), db and k0 == 0))
        return out
    one_range = lambda shapes, db=True: sum(r * c + (2 * r if b else 0) for r, c, b in tiles(shapes, db))
    nb = lib.stnerf_train_dw_batch_workspace_bytes(problems(space), len(space), m)
    # <= 1024 wave tiles x ranges in all (one wave per SIMD), ~36 ranges per 128 x 128 tile of a SpaceNet: tens of MB, not round 5's 476
    t = tiles(space)
    assert len(t) == 34 and (nb - 512) % 16 == 0
    assert 20 * 4 * one_range(space) < nb - 512 <= 4 * 1024 * (128 * 128 + 256) and nb < 80e6
    # ranges are not cut below 64 samples: 64 samples = one range per tile, 100 = two
    assert lib.stnerf_train_dw_batch_workspace_bytes(problems(space), len(space), 64) == 4 * one_range(space) + 512
    assert lib.stnerf_train_dw_batch_workspace_bytes(problems(space), len(space), 100) == 2 * 4 * one_range(space) + 512
    assert lib.stnerf_train_dw_batch_workspace_bytes(problems(space, db=False), len(space), 64) == 4 * one_range(space, False) + 512
    assert lib.stnerf_train_dw_batch_workspace_bytes(problems(space), len(space), 0) == 4 * one_range(space) + 512
    assert lib.stnerf_train_dw_batch_workspace_bytes(problems([(8, 8)] * 17), 17, m) == hip.EINVAL
    assert "16 problems" in hip.last_error()
    assert lib.stnerf_train_dw_batch_workspace_bytes(problems([(512, 4096)] * 2), 2, m) == hip.EINVAL      # operands of 4 GiB
    assert lib.stnerf_train_dw_batch_workspace_bytes(problems([(512, 4096)] * 2), 2, 1000) == hip.EINVAL   # 2 x 4 x 32 tiles
    assert "tiles" in hip.last_error()
    assert lib.stnerf_train_dw_batch_workspace_bytes(problems([(300, 8)]), 1, m) > 0                       # (no limit on the bias width any more)
    assert lib.stnerf_train_dw_batch_workspace_bytes(problems([(8, 8)], lddy=6), 1, m) == hip.EINVAL       # rows not 16-byte aligned
    assert lib.stnerf_train_dw_batch(problems(space), len(space), m, 0, fake, 1024, None) == hip.EINVAL
    assert "workspace too small" in hip.last_error()
    sec = (hip.TransposeSection * 2)(hip.TransposeSection(fake, 63, 0, 256, 63, 64), hip.TransposeSection(fake, 128, 256 * 64, 3, 128, 0))
    assert lib.stnerf_pack_transposed(sec, 2, fake, 256 * 64 + 383, None) == hip.EINVAL                  # the last section ends beyond dst
    assert lib.stnerf_pack_transposed(sec, 0, fake, 1 << 20, None) == hip.EINVAL
    sec[0].n_pad = 32                                                                                      # narrower than the inputs
    assert lib.stnerf_pack_transposed(sec, 2, fake, 1 << 20, None) == hip.EINVAL


def test_split_bf16_training_entries_check_their_arguments_on_the_host(lib):
    """Round 6's split-bf16 training entries: sizes and argument errors before any launch (no GPU here)."""
    fake = 1 << 20                                     # (1 KB aligned, never dereferenced)
    null = C.c_void_p(0)
    # the backward chain's blob: 4 KB of head weights + 24 KB slots: rgb_net.1 8, six 256 x 256 layers 16 each, with d pos two half passes of 4
    assert lib.stnerf_packed_bytes_dx_bf16x3(hip.NET_SPACE, 0) == 4096 + (8 + 96) * 24576
    assert lib.stnerf_packed_bytes_dx_bf16x3(hip.NET_SPACE_TIME, 1) == 4096 + (8 + 96 + 8) * 24576
    assert lib.stnerf_packed_bytes_dx_bf16x3(hip.NET_SPACE_TIME_DEEP, 0) == hip.EINVAL and "deep_rgb" in hip.last_error()
    assert lib.stnerf_packed_bytes_dx_bf16x3(hip.NET_MOTION, 0) == hip.EINVAL
    ptrs = (C.c_void_p * 10)(*([fake] * 10))
    nbytes = lib.stnerf_packed_bytes_dx_bf16x3(hip.NET_SPACE, 1)
    assert lib.stnerf_pack_dx_bf16x3_device(hip.NET_SPACE, ptrs, 9, 1, fake, nbytes, None) == hip.EINVAL and "10 weight tensors" in hip.last_error()
    assert lib.stnerf_pack_dx_bf16x3_device(hip.NET_SPACE, ptrs, 10, 1, fake, nbytes - 1, None) == hip.EINVAL
    assert lib.stnerf_pack_dx_bf16x3_device(hip.NET_SPACE, ptrs, 10, 1, fake + 16, nbytes, None) == hip.EINVAL and "1 KB aligned" in hip.last_error()
    assert lib.stnerf_pack_net_bf16x3_device(hip.NET_SPACE, ptrs, ptrs, 10, fake + 16, 1 << 30, None) == hip.EINVAL and "1 KB aligned" in hip.last_error()
    assert lib.stnerf_pack_net_bf16x3_device(99, ptrs, ptrs, 10, fake, 1 << 30, None) == hip.EINVAL
    dys = (C.c_void_p * 8)(*([fake] * 8))
    lds = (C.c_int32 * 8)(*([256] * 7 + [128]))
    dx = lambda blob, dpos, dpe, skip, rows=1000, stride=8000: lib.stnerf_train_spacenet_dx_bf16x3(blob, dpos, fake, rows, fake, stride, dys, lds, dpe, 64, skip, 64, None)
    assert dx(fake + 16, 0, null, null) == hip.EINVAL and "1 KB aligned" in hip.last_error()
    assert dx(fake, 1, null, null) == hip.EINVAL and "with_dpos" in hip.last_error()              # a d pos stream without its two matrices
    assert dx(fake, 0, fake, fake) == hip.EINVAL
    assert dx(fake, 1, fake, null) == hip.EINVAL
    assert dx(fake, 0, null, null, stride=7000) == hip.EINVAL                                       # stage stride < 8 x rows
    assert dx(fake, 0, null, null, rows=0) == hip.OK                                                # nothing to do: no launch
    lds[3] = 250
    assert dx(fake, 0, null, null) == hip.EINVAL and "matrix 3" in hip.last_error()
    # the forward: as stnerf_train_spacenet_fwd, plus the blob's alignment
    acts = (C.c_void_p * 8)(*([fake] * 8))
    lda = (C.c_int32 * 8)(*([256] * 7 + [128]))
    fwd = lambda packed, kind=hip.NET_SPACE: lib.stnerf_train_spacenet_fwd_bf16x3(kind, packed, 10, 4, fake, 12, fake, 3, null, 0, fake, 16, acts, lda, fake, 64, null, 0, fake, fake, None)
    assert fwd(fake + 16) == hip.EINVAL and "1 KB aligned" in hip.last_error()
    assert fwd(fake, hip.NET_SPACE_TIME) == hip.EINVAL and "frame-id" in hip.last_error()
    assert fwd(fake, hip.NET_SPACE_DEEP) == hip.EINVAL


def test_composite_launch_plan(lib):
    """stnerf_composite_plan: the sizing arithmetic of the compositor's launches (render.hip: plan_composite) on the CPU.
    Every BASELINE shape takes the register / insertion-merge kernels; the single-layer pre-pass needs scratch; a merged
    list of all nine 192-sample layers would cost occupancy, so with scratch there are two launches and the first holds
    four layers; the `order` output and layers of more than 192 samples take the LDS-staged kernel; nothing plans more than
    the 160 KiB of a CU, and a ray that cannot fit is refused with the number of bytes it needs."""
    def plan(l, S, scratch=True, order=False):
        out = (C.c_int64 * 9)()
        rc = lib.stnerf_composite_plan(l, S, int(scratch), int(order), out)
        return rc, list(out)

    for l, S in ((2, 64), (3, 64), (3, 128), (3, 90), (3, 120), (5, 64), (5, 128), (9, 128), (9, 192), (16, 192), (1, 1), (16, 1)):
        for scratch in (True, False):
            rc, (staged, single, tiers, cap, clear, wpb0, lds0, wpb1, lds1) = plan(l, S, scratch)
            assert rc == 0 and not staged, (l, S, scratch)
            assert (single != 0) == (scratch and (l - 1) * ((S + 63) // 64) <= 24), (l, S, scratch, single)
            assert 1 <= wpb0 <= 4 and lds0 <= 150 * 1024 and lds1 <= 150 * 1024
            assert tiers in (1, 2) and (tiers == 1 or (scratch and 2 <= cap < l and 1 <= wpb1 <= 4 and lds1 > lds0))
            assert tiers == 2 or cap == l
            assert clear == (1 if tiers == 2 and not single else 0)
    assert plan(3, 128)[1][:5] == [0, 1, 1, 3, 0]                       # C3 fine: pre-pass <2, 6>, one merge launch
    assert plan(5, 128)[1][:5] == [0, 2, 1, 5, 0]                       # C4 fine: pre-pass <3, 24>
    assert plan(9, 192)[1][:5] == [0, 2, 2, 4, 0]                       # C5 fine: lists of four layers first, then nine
    assert plan(9, 192, scratch=False)[1][:5] == [0, 0, 1, 9, 0]
    assert plan(16, 192)[1][:5] == [0, 0, 2, 4, 1]                      # no pre-pass instantiation: scratch cleared instead
    assert plan(3, 128, order=True)[1][0] == 1 and plan(3, 256)[1][0] == 1 and plan(3, 193)[1][0] == 1
    rc, _ = plan(16, 4000, order=True)                                  # 64,000 samples x 22 B do not fit
    assert rc == hip.EINVAL and "B of LDS per wave" in hip.last_error()
    rc, _ = plan(17, 64)
    assert rc == hip.EINVAL
