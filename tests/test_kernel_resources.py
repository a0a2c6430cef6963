"""Build-quality gate for the wave stage kernel (csrc/mlp_wave.hip), CPU only: hipcc cross-compiles it to gfx950 assembly.

The kernel runs one wave per SIMD with the whole unified register file; a change that pushes a handful of values into
scratch does not fail any numerical test, it just slows every phase down (scratch reloads queue up in front of the
operand loads of the K loop: -3 % measured).  So the resource usage is asserted here, not looked at by hand.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="no hipcc")
def test_wave_stage_kernel_uses_no_scratch_and_one_wave_per_simd(tmp_path):
    src = os.path.join(ROOT, "st-nerf_amd", "csrc", "mlp_wave.hip")
    asm = tmp_path / "mlp_wave.s"
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src), "-S", "--cuda-device-only", "-o", str(asm), src]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, timeout=600)
    text = asm.read_text()
    kernels = re.findall(r"^(_ZN6stnerf21mlp_wave_stage_kernelILb[01]E\S*):", text, re.M)
    assert len(kernels) == 2, kernels  # the plain and the deep_rgb variant
    scratch = [int(v) for v in re.findall(r"; ScratchSize: (\d+)", text)]
    vgprs = [int(v) for v in re.findall(r"; TotalNumVgprs: (\d+)", text)]
    occupancy = [int(v) for v in re.findall(r"; Occupancy: (\d+)", text)]
    assert scratch == [0, 0], f"the wave stage kernel spills to scratch: {scratch} bytes"
    assert all(v <= 512 for v in vgprs) and len(vgprs) == 2, vgprs
    assert occupancy == [1, 1], occupancy
    # the hot loop is what it is supposed to be: f32 MFMAs fed from registers, accumulators loaded by LDS reads directly
    assert text.count("v_mfma_f32_32x32x2_f32") >= 2 * 2400
    assert "scratch_load" not in text and "scratch_store" not in text
