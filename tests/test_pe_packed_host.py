"""sincos_pe2 (csrc/mlp_wave.hip: two PE arguments per packed-f32 instruction) against sincos_pe (csrc/mlp_common.h), on the
HOST: both function bodies are cut out of the kernel sources and compiled for the CPU with the ROCm clang (ext vectors,
__builtin_elementwise_*), then compared bit for bit on a million arguments over the range the encodings reach
(|x| <= 2^9 * a few units, utils/dimension_kernel.py:20-27) plus special values.  fma, multiply and rint are the same IEEE
operations on both machines, so this pins the source-level equivalence (operation order, the sign-bit xor in place of the
conditional negation); what the device compiler makes of it is covered by the bitwise kernel-vs-kernel GPU tests.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _cut(path, signature):
    text = open(path).read()
    start = text.index(signature)
    end = text.index("\n}\n", start) + 3
    return text[start:end]


@pytest.mark.skipif(not os.path.exists(CLANG), reason="no ROCm clang for the host build")
def test_packed_sincos_equals_scalar_sincos_bitwise(tmp_path):
    scalar = _cut(os.path.join(ROOT, "st-nerf_amd", "csrc", "mlp_common.h"), "__device__ __forceinline__ void sincos_pe(float x")
    wave = open(os.path.join(ROOT, "st-nerf_amd", "csrc", "mlp_wave_common.h")).read()
    typedef = re.search(r"typedef float f32x2 __attribute__\(\(ext_vector_type\(2\)\)\);", wave).group(0)
    packed = _cut(os.path.join(ROOT, "st-nerf_amd", "csrc", "mlp_wave_common.h"), "__device__ __forceinline__ void sincos_pe2(f32x2 x")
    src = tmp_path / "pe_host.cpp"
    src.write_text(r"""
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#define __device__
#define __forceinline__ inline
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
""" + scalar + "\n" + typedef + "\n" + packed + r"""
int main() {
    uint64_t state = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; };
    long bad = 0, n = 0;
    auto check = [&](float a, float b) {
        float s0, c0, s1, c1;
        sincos_pe(a, s0, c0);
        sincos_pe(b, s1, c1);
        f32x2 x = {a, b}, sn, cs;
        sincos_pe2(x, sn, cs);
        const float got[4] = {sn[0], cs[0], sn[1], cs[1]}, want[4] = {s0, c0, s1, c1};
        for (int i = 0; i < 4; ++i) bad += __float_as_uint(got[i]) != __float_as_uint(want[i]);
        ++n;
    };
    const float special[] = {0.f, -0.f, 1.f, -1.f, 1.5707963f, 3.1415927f, -3.1415927f, 6.2831855f, 512.f, -512.f, 1e-30f, -1e-30f,
                             0.78539816f, 2.3561945f, 1e-8f, 3000.f, -3000.f, 1.5707964f, 4.712389f};
    for (float a : special) for (float b : special) check(a, b);
    for (int i = 0; i < 1000000; ++i) {
        const int e = (int)(next() % 22) - 10;                                   // magnitudes 2^-10 .. 2^11
        const float a = std::ldexp((float)((double)(next() >> 11) / 9007199254740992.0 * 2.0 - 1.0), e);
        const float b = std::ldexp((float)((double)(next() >> 11) / 9007199254740992.0 * 2.0 - 1.0), (int)(next() % 22) - 10);
        check(a, b);
    }
    std::printf("%ld pairs, %ld mismatching words\n", n, bad);
    return bad != 0;
}
""")
    exe = tmp_path / "pe_host"
    subprocess.run([CLANG, "-O2", "-ffp-contract=off", "-mfma", "-std=c++17", "-o", str(exe), str(src)], check=True, timeout=300)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 mismatching words" in out.stdout
