"""Build libstnerf_hip.so (in-tree) with hipcc for gfx950.  No GPU needed (cross-compile).

    python st-nerf_amd/build.py [--force] [--verbose]
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
# STNERF_LIB_TAG (development): build a variant beside the product library, e.g. STNERF_LIB_TAG=dbg
# STNERF_EXTRA_FLAGS=-DSTNERF_WAVE_DEBUG -> libstnerf_hip_dbg.so (objects under build_dbg/); load it with STNERF_LIB=<path>.
TAG = os.environ.get("STNERF_LIB_TAG", "")
LIB = os.path.join(PKG, "libstnerf_hip%s.so" % ("_" + TAG if TAG else ""))
OBJDIR = os.path.join(PKG, "build" + ("_" + TAG if TAG else ""))

# (source, extra flags).  The sampler/compositor set is compiled with contraction OFF so that its
# elementwise arithmetic is bit-identical to the reference's ATen CPU ops (no implicit FMA).
SOURCES = [
    ("lib.hip", []),
    ("sampler.hip", ["-ffp-contract=off"]),
    ("render.hip", ["-ffp-contract=off"]),
    ("mlp.hip", []),
    ("mlp_raybias.hip", []),
    # (no -mllvm -amdgpu-mfma-vgpr-form=1 here: it saves the v_accvgpr_read of every ReLU (+0.3 %), but with it two of
    # three instrumented variants of this file computed wrong, run-to-run varying results on the MI355X -- hipcc 7.2)
    ("mlp_wave.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"] if os.environ.get("STNERF_WAVE_VGPR_FORM") else []),
    ("mlp_bf16x3.hip", []),
    ("stage_entry.hip", []),
    ("pipeline.hip", []),
    ("train.hip", []),
    ("train_dw.hip", []),
    ("render_bwd.hip", []),
    ("train_wave.hip", []),
]
EXTRA = os.environ.get("STNERF_EXTRA_FLAGS", "").split()   # e.g. -DSTNERF_PHASE_PROF (development only)
# STNERF_FLAGS_<source stem> (development): extra flags for ONE source of a variant build, e.g.
# STNERF_LIB_TAG=noslp STNERF_FLAGS_mlp_bf16x3=-fno-slp-vectorize (the A/B of profiles/retired_designs.md)
SOURCES = [(s, f + os.environ.get("STNERF_FLAGS_" + s[:-4], "").split()) for s, f in SOURCES]
COMMON = EXTRA + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-I" + os.path.join(REPO, "include"), "-I" + CSRC]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(REPO, "include", "stnerf.h"))
    objs = []
    os.makedirs(OBJDIR, exist_ok=True)
    for src, extra in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        if force or _stale(obj, [sp] + headers + [os.path.abspath(__file__)]):
            cmd = [hipcc] + COMMON + extra + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or True))
