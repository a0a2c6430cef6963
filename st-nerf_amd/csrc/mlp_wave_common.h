// Pieces shared by the two sample-split ("a wave owns 32 samples") stage kernels: mlp_wave.hip (exact f32 MFMA) and
// mlp_bf16x3.hip (split-bf16 MFMA): the wave-private encoding window in LDS, the positional encodings, the lane-pair
// reduction of the heads, a work item's inputs.
#pragma once
#include "mlp_stage.h"

namespace stnerf {

constexpr int WV_ROWS = 32;                       // samples per wave
constexpr int WV_NW = 4;                          // waves per workgroup: one per SIMD
constexpr int WV_THREADS = WV_NW * 64;
constexpr int WV_ITEM = WV_NW * WV_ROWS;          // rows per work item
constexpr int WV_ENC_QUADS = 22;                  // widest staged encoding: MotionNet's 84 (+4) features
constexpr int WV_ENC_FLOATS = WV_ENC_QUADS * WV_ROWS * 4;
constexpr int WV_BIAS_SLOTS = 11;                 // 256-float slots per wave: bias vectors of one network + its head weights
constexpr int WV_HEAD_SLOT = 8;                   // first head slot: density_net.0 (1 slot), then the 3 x 128 head (2 slots)
constexpr int WV_BIAS_FLOATS = WV_BIAS_SLOTS * 256;
constexpr int WV_WAVE_FLOATS = WV_ENC_FLOATS + WV_BIAS_FLOATS;   // a wave's private LDS window
constexpr int WV_LDS = WV_NW * WV_WAVE_FLOATS * 4 + 16 + STNERF_MAX_LAYERS * 8;   // + queue slots + rows per layer


// feature f of the lane's sample inside the wave-private staging window (col = window + 4 * c)
#define ENCW(col, f) (col)[((f) >> 2) * (WV_ROWS * 4) + ((f) & 3)]

// The window is private to one wave: LDS operations of a wave execute in issue order, the fences stop the compiler
// from moving accesses across the phase boundary.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sincos_pe (mlp_common.h) for TWO arguments at once on the packed-f32 instructions (v_pk_mul_f32 / v_pk_fma_f32: one
// issue slot for both): the same IEEE operations in the same order per element, so the results are those of the scalar
// function bit for bit.  A vector instruction costs this kernel the same 5 - 6 cycles of MFMA time whether it is packed or
// not, and PE(pos) is 15 evaluations per lane and item: 181 vector instructions less.  (MotionNet's encoding gains
// little from it -- 69 of ~800, against 140 more hazard nops: the lerp and the t + 1 branch keep it scalar.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sincos_pe2(f32x2 x, f32x2& sn, f32x2& cs) {
    const f32x2 k = __builtin_elementwise_rint(x * 0.63661977236758134308f);
    f32x2 r = __builtin_elementwise_fma(-k, (f32x2)(1.5707855225e+00f), x);
    r = __builtin_elementwise_fma(-k, (f32x2)(1.0804273188e-05f), r);
    r = __builtin_elementwise_fma(-k, (f32x2)(6.0770999344e-11f), r);
    const f32x2 z = r * r;
    f32x2 ps = __builtin_elementwise_fma(z, (f32x2)(1.5896910177e-10f), (f32x2)(-2.5050759689e-08f));
    ps = __builtin_elementwise_fma(z, ps, (f32x2)(2.7557314297e-06f));
    ps = __builtin_elementwise_fma(z, ps, (f32x2)(-1.9841270114e-04f));
    ps = __builtin_elementwise_fma(z, ps, (f32x2)(8.3333337680e-03f));
    ps = __builtin_elementwise_fma(z, ps, (f32x2)(-1.6666667163e-01f));
    const f32x2 s0 = __builtin_elementwise_fma(r * z, ps, r);
    f32x2 pc = __builtin_elementwise_fma(z, (f32x2)(-1.1359647598e-11f), (f32x2)(2.0875723372e-09f));
    pc = __builtin_elementwise_fma(z, pc, (f32x2)(-2.7557314297e-07f));
    pc = __builtin_elementwise_fma(z, pc, (f32x2)(2.4801587642e-05f));
    pc = __builtin_elementwise_fma(z, pc, (f32x2)(-1.3888889225e-03f));
    pc = __builtin_elementwise_fma(z, pc, (f32x2)(4.1666667908e-02f));
    const f32x2 c0 = __builtin_elementwise_fma(z * z, pc, __builtin_elementwise_fma(z, (f32x2)(-0.5f), (f32x2)(1.0f)));
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int q = (int)k[e];
        const float sv = (q & 1) ? c0[e] : s0[e];
        const float cv = (q & 1) ? s0[e] : c0[e];
        // (q & 2) ? -sv : sv and ((q + 1) & 2) ? -cv : cv as sign-bit xors: the same bits, without two more trips through vcc
        sn[e] = __uint_as_float(__float_as_uint(sv) ^ (((uint32_t)q << 30) & 0x80000000u));
        cs[e] = __uint_as_float(__float_as_uint(cv) ^ (((uint32_t)(q + 1) << 30) & 0x80000000u));
    }
}

// PE_10(pos): 63 features + one zero pad (utils/dimension_kernel.py:8-33); lane half h takes the frequencies 2 i + h
__device__ __forceinline__ void encode_pos(float* encw, int lane, const float (&p)[3]) {
    const int h = lane >> 5, c = lane & 31;
    float* col = encw + c * 4;
    if (h == 0) {
#pragma unroll
        for (int dmn = 0; dmn < 3; ++dmn) ENCW(col, dmn) = p[dmn];
    } else {
        ENCW(col, 63) = 0.f;
    }
    // the lane's 15 (frequency, dimension) evaluations e = 3 i + dmn, two at a time
#pragma unroll
    for (int e = 0; e < 15; e += 2) {
        const int i0 = e / 3, d0 = e - 3 * i0, i1 = (e + 1) / 3, d1 = (e + 1) - 3 * i1;
        const int fq0 = 2 * i0 + h, fq1 = 2 * i1 + h;
        if (e + 1 < 15) {
            f32x2 x = {p[d0], p[d1]}, fr = {(float)(1 << fq0), (float)(1 << fq1)}, sn, cs;
            sincos_pe2(x * fr, sn, cs);
            ENCW(col, 3 + fq0 * 6 + d0) = sn[0];
            ENCW(col, 6 + fq0 * 6 + d0) = cs[0];
            ENCW(col, 3 + fq1 * 6 + d1) = sn[1];
            ENCW(col, 6 + fq1 * 6 + d1) = cs[1];
        } else {
            float sn, cs;
            sincos_pe(p[d0] * (float)(1 << fq0), sn, cs);
            ENCW(col, 3 + fq0 * 6 + d0) = sn;
            ENCW(col, 6 + fq0 * 6 + d0) = cs;
        }
    }
}

// PE_10([x,y,z,t]) with the fractional-time lerp of modeling/motion_net.py:49-60: 84 features + 4 zero pads
__device__ __forceinline__ void encode_motion(float* encw, int lane, const float (&p)[3], float tv, int flags) {
    const int h = lane >> 5, c = lane & 31;
    float* col = encw + c * 4;
    const float lo = (flags & STNERF_MOTION_PLAIN_TIME) ? tv : floorf(tv);  // input_time=False: PE(input) as is
    const float wgt = tv - lo;
    const bool frac = wgt != 0.f;
    const float om = 1.f - wgt;
    if (h == 0) {
#pragma unroll
        for (int dmn = 0; dmn < 3; ++dmn) ENCW(col, dmn) = lerp_enc(frac, om, wgt, p[dmn], p[dmn]);
        ENCW(col, 3) = lerp_enc(frac, om, wgt, lo, lo + 1.f);
    } else {
#pragma unroll
        for (int f = 84; f < 88; ++f) ENCW(col, f) = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int fq = 2 * i + h;
        const float freq = (float)(1 << fq);
#pragma unroll
        for (int dmn = 0; dmn < 4; ++dmn) {
            float sn, cs, sn2, cs2;
            if (dmn < 3) {
                sincos_pe(p[dmn] * freq, sn, cs);
                sn2 = sn;
                cs2 = cs;
            } else {
                sincos_pe(lo * freq, sn, cs);
                sn2 = sn;
                cs2 = cs;
                if (frac) sincos_pe((lo + 1.f) * freq, sn2, cs2);
            }
            const int fs = 4 + fq * 8 + dmn, fc = fs + 4;
            ENCW(col, fs) = lerp_enc(frac, om, wgt, sn, sn2);
            ENCW(col, fc) = lerp_enc(frac, om, wgt, cs, cs2);
        }
    }
}

// (ca + partner's ca) + (cb + partner's cb), partner = the other lane of this sample (lane ^ 32), in every lane: two
// v_permlane32_swap (upper half of the first operand <-> lower half of the second) instead of two ds_bpermute round
// trips.  After swap(ca, cb) the lower lanes hold {own ca, partner's ca}, the upper lanes {partner's cb, own cb}; the
// second swap hands both half sums to both halves.  Same additions, same order as in the LDS kernels' reduction.
__device__ __forceinline__ float pair_sum(float ca, float cb) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(ca), __float_as_uint(cb), false, false);
    const float t = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// What a wave needs of a work item: its sample's point, direction and frame id, and where the result goes.
struct WaveInputs {
    float p[3], tv;
    int64_t raw_off;   // float offset of the sample's {r,g,b,sigma} in the layer's raw
    int32_t ray;       // row of the layer's ray-bias table
    bool valid;
};

}  // namespace stnerf
