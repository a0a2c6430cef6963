"""ctypes binding of the C ABI in ``include/stnerf.h`` (libstnerf_hip.so).

This is the only place Python touches the native library.  There is NO fallback: if the shared
library is missing, cannot be loaded, or a call returns an error code, a ``RuntimeError`` /
``ValueError`` is raised -- nothing in this package computes the hot path in PyTorch.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch  # imported first on purpose: the HIP runtime of the process must be torch's (same SONAME)

_PKG = os.path.dirname(os.path.abspath(__file__))
# STNERF_LIB (development): load a variant build (st-nerf_amd/build.py with STNERF_LIB_TAG) instead of the product library
LIB_PATH = os.environ.get("STNERF_LIB") or os.path.join(_PKG, "libstnerf_hip.so")

OK, EINVAL, ELAUNCH, EARCH = 0, -1, -2, -3
MAX_LAYERS = 16
NET_SPACE, NET_SPACE_TIME, NET_MOTION, NET_SPACE_DEEP, NET_SPACE_TIME_DEEP = 0, 1, 2, 3, 4

c_f32p = C.c_void_p  # device pointers travel as void*
c_i64 = C.c_int64


class LayerEdit(C.Structure):
    _fields_ = [("shift", C.c_float * 3), ("scale", C.c_float), ("has_shift", C.c_int32), ("has_scale", C.c_int32)]


class CompositeParams(C.Structure):
    _fields_ = [("border", C.c_float), ("near", C.c_float), ("fine", C.c_int32), ("cut_negative_t", C.c_int32),
                ("threshold", C.c_float * MAX_LAYERS), ("use_threshold", C.c_int32 * MAX_LAYERS),
                ("sigma_scale", C.c_float * MAX_LAYERS), ("rgb_activated", C.c_int32),
                ("evaluated", C.c_int32 * MAX_LAYERS)]


class Nets(C.Structure):
    _fields_ = [("bkgd", C.c_void_p), ("bkgd_fine", C.c_void_p), ("space", C.c_void_p * MAX_LAYERS),
                ("space_fine", C.c_void_p * MAX_LAYERS), ("motion", C.c_void_p * MAX_LAYERS)]


class StageLayer(C.Structure):
    _fields_ = [("space", C.c_void_p), ("motion", C.c_void_p), ("ray_list", C.c_void_p), ("ray_count", C.c_void_p),
                ("xyz", C.c_void_p), ("raw", C.c_void_p), ("times", C.c_void_p), ("use_time", C.c_int32),
                ("motion_flags", C.c_int32)]


class RenderParams(C.Structure):
    _fields_ = [("l", C.c_int32), ("n1", C.c_int32), ("n2", C.c_int32), ("ray_stride", C.c_int32),
                ("retiming", C.c_int32), ("only_coarse", C.c_int32), ("use_deform_time", C.c_int32),
                ("use_space_time", C.c_int32), ("precision", C.c_int32), ("has_edits", C.c_int32),
                ("bkgd_use_deform_time", C.c_int32), ("bkgd_use_space_time", C.c_int32), ("deep_rgb", C.c_int32),
                ("shown", C.c_int32 * MAX_LAYERS), ("border", C.c_float), ("near", C.c_float), ("alpha", C.c_float),
                ("density_threshold", C.c_float), ("bkgd_density_threshold", C.c_float), ("seed", C.c_uint64),
                ("ray_index_base", C.c_int64), ("ray_index_stripe", C.c_int64), ("ray_index_period", C.c_int64),
                ("edits_coarse", LayerEdit * MAX_LAYERS),
                ("edits_fine", LayerEdit * MAX_LAYERS), ("pivot", C.c_float * 3)]


MOTION_ADD_TO_XYZ, MOTION_PLAIN_TIME = 1, 2   # STNERF_MOTION_* bits of stnerf_motionnet_fwd's add_to_xyz argument


class DwProblem(C.Structure):
    """stnerf_dw_problem (include/stnerf.h): one layer of stnerf_train_dw_batch."""
    _fields_ = [("dy", C.c_void_p), ("lddy", C.c_int64), ("x", C.c_void_p), ("ldx", C.c_int64), ("dw", C.c_void_p), ("lddw", C.c_int64),
                ("db", C.c_void_p), ("n", C.c_int32), ("k", C.c_int32)]


class TransposeSection(C.Structure):
    """stnerf_transpose_section (include/stnerf.h)."""
    _fields_ = [("w", C.c_void_p), ("ldw", C.c_int64), ("dst_off", C.c_int64), ("n_out", C.c_int32), ("n_in", C.c_int32), ("n_pad", C.c_int32)]


class ProfileRecord(C.Structure):
    _fields_ = [("kernel", C.c_int32), ("kind", C.c_int32), ("ns", C.c_int32), ("tag", C.c_int32),
                ("n_rays", C.c_int64), ("bytes_per_ray", C.c_int64), ("ms", C.c_float), ("pad_", C.c_int32)]


_PROTOS = {
    "stnerf_profile_begin": (C.c_int, []),
    "stnerf_profile_end": (C.c_int, [C.POINTER(ProfileRecord), C.c_int, C.POINTER(C.c_int)]),
    "stnerf_version": (C.c_char_p, []),
    "stnerf_last_error": (C.c_char_p, []),
    "stnerf_device_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "stnerf_generate_rays": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, c_i64, c_i64, c_i64,
                                       c_i64, C.POINTER(C.c_float), C.c_int, c_f32p, C.c_int, C.c_void_p]),
    "stnerf_intersect": (C.c_int, [c_f32p, c_i64, C.c_int, c_f32p, c_i64, C.c_int, c_f32p, C.c_void_p]),
    "stnerf_sample_coarse": (C.c_int, [c_f32p, c_i64, C.c_int, c_f32p, c_i64, C.c_int, C.c_int, c_f32p, C.c_uint64,
                                       c_i64, c_i64, c_i64, C.POINTER(LayerEdit), C.POINTER(C.c_float), c_f32p, c_f32p,
                                       C.c_void_p, C.c_void_p]),
    "stnerf_compact_rays": (C.c_int, [C.c_void_p, c_i64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "stnerf_packed_bytes": (c_i64, [C.c_int]),
    "stnerf_pack_net": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_void_p, c_i64]),
    "stnerf_pack_net_device": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_void_p, c_i64, C.c_void_p]),
    "stnerf_pack_net_bf16x3_device": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_void_p, c_i64, C.c_void_p]),
    "stnerf_spacenet_fwd": (C.c_int, [C.c_int, C.c_void_p, c_i64, C.c_int, C.c_void_p, C.c_void_p, c_f32p, c_i64,
                                      c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64, c_f32p, C.c_void_p]),
    "stnerf_rgb_ray_bias": (C.c_int, [C.c_int, C.c_void_p, c_i64, C.c_void_p, C.c_void_p, c_f32p, c_i64, c_f32p, c_i64,
                                      c_f32p, C.c_void_p]),
    "stnerf_packed_bytes_bf16x3": (c_i64, [C.c_int]),
    "stnerf_pack_net_bf16x3": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_void_p,
                                         c_i64]),
    "stnerf_motionnet_fwd": (C.c_int, [C.c_void_p, c_i64, C.c_int, C.c_void_p, C.c_void_p, c_f32p, c_i64, c_f32p,
                                       c_i64, c_f32p, c_i64, C.c_int, C.c_void_p]),
    "stnerf_mlp_stage": (C.c_int, [C.POINTER(StageLayer), C.c_int, c_i64, C.c_int, c_f32p, c_i64, c_i64, c_i64, c_i64, C.c_int,
                                   C.c_void_p, c_f32p, C.c_void_p]),
    "stnerf_train_linear_fwd": (C.c_int, [c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64, C.c_int, C.c_int, C.c_int, c_f32p, c_i64, C.c_void_p]),
    "stnerf_train_linear_dx": (C.c_int, [c_f32p, c_i64, c_f32p, c_i64, c_i64, C.c_int, C.c_int, c_f32p, c_i64, C.c_int, c_f32p, c_i64,
                                         C.c_void_p]),
    "stnerf_train_dw_workspace_bytes": (c_i64, [c_i64, C.c_int, C.c_int]),
    "stnerf_train_linear_dw": (C.c_int, [c_f32p, c_i64, c_f32p, c_i64, c_i64, C.c_int, C.c_int, c_f32p, c_i64, c_f32p, C.c_int, C.c_void_p,
                                         c_i64, C.c_void_p]),
    "stnerf_train_dw_batch_workspace_bytes": (c_i64, [C.POINTER(DwProblem), C.c_int32, c_i64]),
    "stnerf_train_dw_batch": (C.c_int, [C.POINTER(DwProblem), C.c_int32, c_i64, C.c_int32, C.c_void_p, c_i64, C.c_void_p]),
    "stnerf_train_encode": (C.c_int, [c_f32p, c_i64, C.c_int, C.c_int, C.c_int, c_i64, C.c_int, C.c_int, C.c_int, c_f32p, c_i64, C.c_int,
                                      C.c_void_p]),
    "stnerf_train_encode_bwd": (C.c_int, [c_f32p, c_i64, C.c_int, C.c_int, C.c_int, c_i64, c_f32p, c_i64, C.c_int, C.c_int, C.c_int, c_f32p,
                                          c_i64, C.c_void_p]),
    "stnerf_train_spacenet_fwd": (C.c_int, [C.c_int, C.c_void_p, c_i64, C.c_int, c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64,
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_int32), c_f32p, C.c_int32, C.c_void_p, c_i64, C.c_void_p, c_f32p, C.c_void_p]),
    "stnerf_train_spacenet_fwd_bf16x3": (C.c_int, [C.c_int, C.c_void_p, c_i64, C.c_int, c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64,
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_int32), c_f32p, C.c_int32, C.c_void_p, c_i64, C.c_void_p, c_f32p, C.c_void_p]),
    "stnerf_train_spacenet_dx": (C.c_int, [c_f32p, C.POINTER(C.c_uint32), c_f32p, c_i64, C.c_void_p, c_i64, C.POINTER(C.c_void_p), C.POINTER(C.c_int32),
                                           c_f32p, C.c_int32, C.c_void_p]),
    "stnerf_packed_bytes_dx_bf16x3": (c_i64, [C.c_int, C.c_int]),
    "stnerf_pack_dx_bf16x3_device": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, c_i64, C.c_void_p]),
    "stnerf_train_spacenet_dx_bf16x3": (C.c_int, [C.c_void_p, C.c_int, c_f32p, c_i64, C.c_void_p, c_i64, C.POINTER(C.c_void_p), C.POINTER(C.c_int32),
                                                  c_f32p, C.c_int32, c_f32p, C.c_int32, C.c_void_p]),
    "stnerf_pack_transposed": (C.c_int, [C.POINTER(TransposeSection), C.c_int, c_f32p, c_i64, C.c_void_p]),
    "stnerf_train_motionnet_fwd": (C.c_int, [C.c_void_p, c_f32p, c_i64, C.c_int, c_f32p, c_f32p, C.c_int32, C.POINTER(C.c_void_p),
                                             C.POINTER(C.c_int32), C.c_void_p, c_i64, C.c_void_p]),
    "stnerf_train_motionnet_dx": (C.c_int, [c_f32p, C.POINTER(C.c_uint32), c_f32p, c_i64, C.c_void_p, c_i64, C.POINTER(C.c_void_p),
                                            C.POINTER(C.c_int32), c_f32p, C.c_int32, C.c_void_p]),
    "stnerf_encode": (C.c_int, [c_f32p, c_i64, C.c_int, C.c_int, C.c_int, c_f32p, C.c_void_p]),
    "stnerf_gen_weight": (C.c_int, [c_f32p, c_f32p, c_i64, C.c_int, c_f32p, C.c_void_p]),
    "stnerf_composite": (C.c_int, [c_f32p, c_f32p, C.c_void_p, c_i64, C.c_int, C.c_int, C.POINTER(CompositeParams),
                                   c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "stnerf_composite_bwd": (C.c_int, [c_f32p, c_f32p, C.c_void_p, C.c_void_p, c_i64, C.c_int, C.c_int, C.POINTER(CompositeParams),
                                       c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "stnerf_composite_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(c_i64)]),
    "stnerf_render_workspace_bytes": (c_i64, [c_i64, C.c_int, C.c_int, C.c_int, C.c_int]),
    "stnerf_render_rays": (C.c_int, [c_f32p, c_i64, c_f32p, c_i64, C.POINTER(Nets), C.POINTER(RenderParams), c_f32p, c_f32p,
                                     C.c_void_p, c_i64, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_void_p]),
    "stnerf_resample": (C.c_int, [c_f32p, c_f32p, c_i64, C.c_int, C.c_int, C.c_int, c_f32p, C.c_uint64, c_i64, c_i64, c_i64, c_f32p,
                                  C.c_int, C.POINTER(LayerEdit), C.POINTER(C.c_float), C.c_void_p, c_f32p, c_f32p, c_f32p,
                                  C.c_void_p, c_f32p, C.c_void_p]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """The loaded library (loads on first use; raises if it is not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python st-nerf_amd/build.py` "
                "(or __graft_entry__.build()).  There is no PyTorch fallback for the render path.")
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def exported_symbols():
    return sorted(_PROTOS)


def last_error() -> str:
    return lib().stnerf_last_error().decode()


def check(rc: int, what: str) -> None:
    if rc == OK:
        return
    msg = f"{what}: {last_error()} (code {rc})"
    if rc == EINVAL:
        raise ValueError(msg)
    raise RuntimeError(msg)


def stream_ptr() -> C.c_void_p:
    """The current torch stream as a hipStream_t."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dptr(t: Optional[torch.Tensor], dtype=torch.float32, name: str = "tensor") -> C.c_void_p:
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return C.c_void_p(t.data_ptr())


def fatbin_sha256(path: Optional[str] = None) -> Optional[str]:
    """sha256 of the library's ``.hip_fatbin`` section (the device code objects): what identifies the KERNELS of a build
    independently of where it was linked.  bench.py prints committed PMC counter bytes only when they were measured on
    exactly these kernels (tools/summarise.py records the digest beside them).  None when the file is not an ELF64 with
    that section."""
    import hashlib
    import struct
    try:
        with open(path or LIB_PATH, "rb") as f:
            ident = f.read(64)
            if ident[:4] != b"\x7fELF" or ident[4] != 2:
                return None
            shoff, = struct.unpack_from("<Q", ident, 0x28)
            shentsize, shnum, shstrndx = struct.unpack_from("<HHH", ident, 0x3A)
            f.seek(shoff)
            table = f.read(shentsize * shnum)
            sec = lambda i: struct.unpack_from("<IIQQQQ", table, i * shentsize)   # name, type, flags, addr, offset, size
            _, _, _, _, str_off, str_size = sec(shstrndx)
            f.seek(str_off)
            names = f.read(str_size)
            for i in range(shnum):
                name, _, _, _, off, size = sec(i)
                if names[name:names.index(b"\0", name)] == b".hip_fatbin":
                    f.seek(off)
                    return hashlib.sha256(f.read(size)).hexdigest()
    except (OSError, struct.error, ValueError):
        pass
    return None


def device_info() -> dict:
    cu, lds, clk = C.c_int(0), C.c_int(0), C.c_int(0)
    arch = C.create_string_buffer(64)
    rc = lib().stnerf_device_info(C.byref(cu), C.byref(lds), C.byref(clk), arch, 64)
    info = dict(cu_count=cu.value, lds_bytes_per_cu=lds.value, clock_khz=clk.value, arch=arch.value.decode())
    check(rc, "stnerf_device_info")
    return info
