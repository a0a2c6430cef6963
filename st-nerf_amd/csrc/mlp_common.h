// Pieces shared by the fused MLP kernels (mlp.hip, the stage kernels).
#pragma once
#include "common.h"

namespace stnerf {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// Optional per-phase cycle accounting (development builds: -DSTNERF_PHASE_PROF).  Thread 0 of every
// workgroup accumulates s_memtime deltas per phase; read back with stnerf_debug_read_phases().
#ifdef STNERF_PHASE_PROF
static __device__ unsigned long long g_phase[16];
#define PH_DECL unsigned long long ph_t = clock64(); unsigned long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PH(i) do { const unsigned long long n_ = clock64(); ph_acc[i] += n_ - ph_t; ph_t = n_; } while (0)
#define PH_FLUSH do { if (threadIdx.x == 0) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_phase[i_], ph_acc[i_]); atomicAdd(&g_phase[8], 1ull); } } while (0)
#define PH_PARAMS , unsigned long long& ph_t, unsigned long long (&ph_acc)[8]
#define PH_ARGS , ph_t, ph_acc
#else
#define PH_DECL
#define PH(i) do { } while (0)
#define PH_FLUSH do { } while (0)
#define PH_PARAMS
#define PH_ARGS
#endif
enum { PH_PE = 0, PH_MMA = 1, PH_BAR1 = 2, PH_EPI = 3, PH_BAR2 = 4, PH_ENC2 = 5, PH_HEAD = 6, PH_MISC = 7 };


// ---------------------------------------------------------------------------------------------
// Packed weight layouts (offsets in floats).  Shared by the host packer and the kernels.
// ---------------------------------------------------------------------------------------------
struct SpaceLayout {
    int64_t w[7], b[7];        // backbone: stage1.{0,2,4,6}, stage2.{0,2,4};  W as [Kq][256][4]
    int64_t w_sigma, b_sigma;  // density_net.0: 256 + 1(4)
    int64_t w_rgb1, b_rgb1;    // rgb_net.1 as [Kq][128][4]
    int64_t w_deep[2], b_deep[2];  // deep_rgb only: rgb_net.{3,5} as [32][128][4]  (modeling/spacenet.py:68-79)
    int64_t w_rgb2, b_rgb2;    // last rgb layer (rgb_net.3, or rgb_net.7 with deep_rgb) as [3][128] + 3(4)
    int64_t total;
    int kq[7];                 // K quads per backbone layer
    int kq_rgb1;               // 64 + 12 (time) | 64 + 8
};

__host__ __device__ inline SpaceLayout space_layout(bool use_time, bool deep = false) {
    SpaceLayout L;
    const int kq[7] = {16, 64, 64, 64, 80, 64, 64};
    int64_t off = 0;
    for (int i = 0; i < 7; ++i) {
        L.kq[i] = kq[i];
        L.w[i] = off;
        off += (int64_t)kq[i] * 256 * 4;
        L.b[i] = off;
        off += 256;
    }
    L.w_sigma = off; off += 256;
    L.b_sigma = off; off += 4;
    L.kq_rgb1 = 64 + (use_time ? 12 : 8);
    L.w_rgb1 = off; off += (int64_t)L.kq_rgb1 * 128 * 4;
    L.b_rgb1 = off; off += 128;
    for (int i = 0; i < 2; ++i) {
        L.w_deep[i] = off; off += deep ? 32 * 128 * 4 : 0;
        L.b_deep[i] = off; off += deep ? 128 : 0;
    }
    L.w_rgb2 = off; off += 3 * 128;
    L.b_rgb2 = off; off += 4;
    L.total = off;
    return L;
}

struct MotionLayout {
    int64_t w[5], b[5];  // motion_net.{0,2,4,6,8} as [Kq][128][4]
    int64_t w_out, b_out;  // motion_net.10 as [3][128] + 3(4)
    int64_t total;
    int kq[5];
};

__host__ __device__ inline MotionLayout motion_layout() {
    MotionLayout L;
    const int kq[5] = {22, 32, 32, 32, 32};
    int64_t off = 0;
    for (int i = 0; i < 5; ++i) {
        L.kq[i] = kq[i];
        L.w[i] = off;
        off += (int64_t)kq[i] * 128 * 4;
        L.b[i] = off;
        off += 128;
    }
    L.w_out = off; off += 3 * 128;
    L.b_out = off; off += 4;
    L.total = off;
    return L;
}


// ReLU as ONE VALU instruction (v_max_i32 on the bit pattern: negative floats are negative integers).  fmaxf(x, 0)
// costs two (a canonicalising v_max_f32 first), and on gfx950 every VALU instruction displaces f32 MFMA work
// (profiles/r01_dual_issue_microbench.md).  Differs from fmaxf only for NaNs with the sign bit set.
__device__ __forceinline__ float relu_bits(float x) {
    const int b = __float_as_int(x);
    return __int_as_float(b > 0 ? b : 0);
}

// sin and cos of one fp32 argument, |x| up to a few thousand (positional-encoding arguments are
// 2^f * coordinate, f <= 9).  Cody-Waite reduction by pi/2 in three fma steps (fdlibm's 17-bit splits of
// pi/2: the first step is exact, the total reduction error is < 1 ulp of the reduced argument), then
// fdlibm's float minimax kernels on [-pi/4, pi/4] (< 1 ulp).  ~40 VALU instructions with no slow path --
// ocml's sincosf spends about twice that and the encodings are ~45 % of this kernel's non-MFMA work.
__device__ __forceinline__ void sincos_pe(float x, float& sn, float& cs) {
    const float k = rintf(x * 0.63661977236758134308f);
    float r = fmaf(-k, 1.5707855225e+00f, x);
    r = fmaf(-k, 1.0804273188e-05f, r);
    r = fmaf(-k, 6.0770999344e-11f, r);
    const float z = r * r;
    float ps = fmaf(z, 1.5896910177e-10f, -2.5050759689e-08f);
    ps = fmaf(z, ps, 2.7557314297e-06f);
    ps = fmaf(z, ps, -1.9841270114e-04f);
    ps = fmaf(z, ps, 8.3333337680e-03f);
    ps = fmaf(z, ps, -1.6666667163e-01f);
    const float s0 = fmaf(r * z, ps, r);
    float pc = fmaf(z, -1.1359647598e-11f, 2.0875723372e-09f);
    pc = fmaf(z, pc, -2.7557314297e-07f);
    pc = fmaf(z, pc, 2.4801587642e-05f);
    pc = fmaf(z, pc, -1.3888889225e-03f);
    pc = fmaf(z, pc, 4.1666667908e-02f);
    const float c0 = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
    const int q = (int)k;
    const float sv = (q & 1) ? c0 : s0;
    const float cv = (q & 1) ? s0 : c0;
    sn = (q & 2) ? -sv : sv;
    cs = ((q + 1) & 2) ? -cv : cv;
}

struct WorkList {
    int64_t n_rays;
    int ns;
    const int32_t* ray_list;
    const int32_t* ray_count;
};

__device__ __forceinline__ int64_t worklist_rows(const WorkList& wl) {
    int64_t cnt = wl.n_rays;
    if (wl.ray_count) {
        const int64_t c = *wl.ray_count;
        cnt = c < cnt ? c : cnt;
    }
    return cnt * wl.ns;
}

struct SpaceArgs {
    const float* net;
    WorkList wl;
    const float* xyz;
    int64_t xyz_ray_stride;
    const float* dirs;
    int64_t dirs_ray_stride;
    const float* times;
    int64_t times_ray_stride;
    float* raw;
    int64_t raw_ray_stride;
    const float* raybias;  // exact-f32 kernels: [n_rays][128] C operands of rgb_net.1 (mlp_raybias.hip)
};

// rgb_net.1's direction / time columns once per ray (mlp_raybias.hip): out[ray][128] for the listed rays.
int launch_ray_bias(int kind, const float* net, int64_t n_rays, const int32_t* ray_list, const int32_t* ray_count,
                    const float* dirs, int64_t dirs_ray_stride, const float* times, int64_t times_ray_stride, float* out,
                    hipStream_t stream);

struct MotionArgs {
    const float* net;
    WorkList wl;
    float* xyz;
    int64_t xyz_ray_stride;
    const float* times;
    int64_t times_ray_stride;
    float* flow;
    int64_t flow_ray_stride;
    int add_to_xyz;  // STNERF_MOTION_* flag bits
};

// Positional-encoding feature f of a tile sample lives at col[(f >> 2) * TM * 4 + (f & 3)], col = encf + s * 4.
#define ENC_AT(col, f) (col)[((f) >> 2) * TM * 4 + ((f) & 3)]

// Wave -> (feature block, sample blocks) decomposition of a layer with N outputs on a TM-sample tile, NW waves.
//   N = 256: NW = 4 -> 64 features x all samples per wave;  NW = 8 -> 32 features x all samples;  NW = 16 -> 32 x half
//   N = 128: NW = 4 -> 32 features x all samples;           NW = 8 -> 32 features x half the samples;  NW = 16 -> 32 x quarter
// STNERF_WS_SQUARE (experiment): with 8 waves give a 256-wide layer 64 features x half the samples per wave
// (2 x 2 blocks: 4 global + 4 LDS operand loads per step instead of 2 + 8).
#ifndef STNERF_WS_SQUARE
#define STNERF_WS_SQUARE 0
#endif
template <int TM, int NW, int N>
struct WaveSplit {
    // FS feature slices x SG sample groups = NW waves
    static constexpr int NFB = (N == 256 && (NW == 4 || (STNERF_WS_SQUARE && NW == 8))) ? 2 : 1;
    static constexpr int FS = N / (32 * NFB);
    static constexpr int SG = NW / FS;
    static constexpr int NSB = (TM / 32) / SG;
    static_assert(FS * SG == NW && NSB >= 1 && NSB * SG * 32 == TM, "unsupported tile / wave decomposition");
    __device__ static __forceinline__ int n0(int wave) { return (wave % FS) * NFB * 32; }
    __device__ static __forceinline__ int sb0(int wave) { return (wave / FS) * NSB; }
};


}  // namespace stnerf
