// Fused SpaceNet / MotionNet kernels in "fp16x3" arithmetic: fp32-accurate matrix products on the fp16 MFMA
// pipe (16x the f32 MFMA rate per instruction, 3 instructions per product => 5.3x the f32 MFMA roofline).
//
// Every fp32 operand is split into two fp16 numbers, x = hi + lo with hi = fp16(x), lo = fp16(x - hi)
// (22 significand bits; the weights are pre-scaled by 2^8 so that their lo parts stay in the fp16 normal
// range).  A product a*b is evaluated as ah*bh + ah*bl + al*bh with v_mfma_f32_32x32x16_f16: the fp16
// products are exact in fp32 and are accumulated in fp32 (the dropped al*bl term is 2^-22 relative).  The
// accumulation rounds once per 16-k MFMA instead of once per k, so the end result is as close to an fp64
// evaluation as the plain fp32 chain is (measured: tests/test_gpu_f16x3.py) -- same parity tolerances.
// Everything that is not a matrix product (encodings, bias, ReLU, heads, outputs) is fp32 as in mlp.hip.
//
// Tile / LDS plan (TM = 128 samples, NW = 8 or 4 waves): activations as two fp16 planes in k-octet-major
// layout [K/8][TM] x 16 B (a lane's MFMA operand = one ds_read_b128 of 8 consecutive k for its sample):
// act_hi, act_lo 64 KiB each + enc_hi, enc_lo 16 KiB each = 160 KiB, exactly the fp32 kernel's footprint.
//
// Reference: modeling/spacenet.py:16-160, modeling/motion_net.py:7-71.
#include <stdlib.h>
#include <string.h>

#include "mlp_common.h"

namespace stnerf {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half4 = __attribute__((ext_vector_type(4))) _Float16;

constexpr float WSCALE = 256.0f;       // weights are stored as fp16 splits of W * 2^8
constexpr float WSCALE_INV = 1.0f / 256.0f;

// ---------------------------------------------------------------------------------------------
// Packed layout: [fp32 blob of mlp.hip (biases, heads; its fp32 weight matrices are unused here)]
//                [per MFMA layer: hi plane [K/8][N] x half8, lo plane [K/8][N] x half8]
// ---------------------------------------------------------------------------------------------
struct SpaceLayoutH {
    SpaceLayout f32;
    int64_t whi[10], wlo[10];  // offsets in half8 units from the start of the fp16 region; [7] = rgb_net.1,
                               // [8], [9] = rgb_net.{3,5} of the deep_rgb variant
    int oct[10];               // K octets per layer (K padded to a multiple of 16)
    int64_t half_region_bytes;
    int64_t total_bytes;
};

__host__ __device__ inline SpaceLayoutH space_layout_h(bool use_time, bool deep = false) {
    SpaceLayoutH L;
    L.f32 = space_layout(use_time, deep);
    const int oct[10] = {8, 32, 32, 32, 40, 32, 32, 32 + (use_time ? 6 : 4), deep ? 16 : 0, deep ? 16 : 0};
    int64_t off = 0;
    for (int i = 0; i < 10; ++i) {
        const int n = i < 7 ? 256 : 128;
        L.oct[i] = oct[i];
        L.whi[i] = off;
        off += (int64_t)oct[i] * n;
        L.wlo[i] = off;
        off += (int64_t)oct[i] * n;
    }
    L.half_region_bytes = off * 16;
    L.total_bytes = L.f32.total * 4 + L.half_region_bytes;
    return L;
}

struct MotionLayoutH {
    MotionLayout f32;
    int64_t whi[5], wlo[5];
    int oct[5];
    int64_t half_region_bytes;
    int64_t total_bytes;
};

__host__ __device__ inline MotionLayoutH motion_layout_h() {
    MotionLayoutH L;
    L.f32 = motion_layout();
    const int oct[5] = {12, 16, 16, 16, 16};
    int64_t off = 0;
    for (int i = 0; i < 5; ++i) {
        L.oct[i] = oct[i];
        L.whi[i] = off;
        off += (int64_t)oct[i] * 128;
        L.wlo[i] = off;
        off += (int64_t)oct[i] * 128;
    }
    L.half_region_bytes = off * 16;
    L.total_bytes = L.f32.total * 4 + L.half_region_bytes;
    return L;
}

// ---------------------------------------------------------------------------------------------
// Device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;                 // round to nearest
    lo = (_Float16)(v - (float)hi);   // exact difference, then rounded: |v - hi - lo| <= 2^-22 |v|
}

template <int NFB>
struct HFrag {  // a layer's first-step operands + bias, prefetched during the previous layer's tail
    half8 wh[NFB], wl[NFB];
    float4 b[NFB][4];
};

__device__ __forceinline__ const half8* hweight_lane_ptr(const half8* plane, int n_total, int n0, int lane) {
    return plane + ((int64_t)(lane >> 5) * n_total + n0 + (lane & 31));
}

template <int NFB>
__device__ __forceinline__ void load_hfrag(HFrag<NFB>& f, const half8* __restrict__ hi_ptr,
                                           const half8* __restrict__ lo_ptr, const float* __restrict__ lane_bias) {
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) {
        f.wh[fb] = hi_ptr[fb * 32];
        f.wl[fb] = lo_ptr[fb * 32];
#pragma unroll
        for (int q = 0; q < 4; ++q) f.b[fb][q] = *reinterpret_cast<const float4*>(lane_bias + fb * 32 + 8 * q);
    }
}

// 3 * NFB * NSB MFMAs of one 16-k step: ah*bh + ah*bl + al*bh per (feature block, sample block).  The three
// terms of one accumulator are issued NFB*NSB instructions apart (term-major order): back-to-back dependent
// 8-pass MFMAs would stall on the accumulator (issue 32 cycles, dependent latency ~40).
template <int NFB, int NSB, bool FIRST>
__device__ __forceinline__ void mma_step_h(f32x16 (&acc)[NFB][NSB], const half8 (&wh)[NFB], const half8 (&wl)[NFB],
                                           const half8 (&ah)[NSB], const half8 (&al)[NSB], const f32x16 (&cinit)[NFB]) {
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb)
            acc[fb][sb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[fb], ah[sb], FIRST ? cinit[fb] : acc[fb][sb], 0, 0, 0);
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb)
            acc[fb][sb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[fb], al[sb], acc[fb][sb], 0, 0, 0);
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb)
            acc[fb][sb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[fb], ah[sb], acc[fb][sb], 0, 0, 0);
}

// Two-stage ping-pong K loop (see mlp.hip: mma_segment); one step = 2 octet rows = 16 k values.
template <int TM, int NFB, int NSB, bool FIRST>
__device__ __forceinline__ void mma_segment_h(f32x16 (&acc)[NFB][NSB], const half8 (&wh_first)[NFB],
                                              const half8 (&wl_first)[NFB], const f32x16 (&cinit)[NFB],
                                              const half8* __restrict__ whp, const half8* __restrict__ wlp, int n_total,
                                              const half8* in_hi, const half8* in_lo, int steps) {
    half8 wh0[NFB], wl0[NFB], ah0[NSB], al0[NSB], wh1[NFB], wl1[NFB], ah1[NSB], al1[NSB];
    const int64_t wstep = 2 * (int64_t)n_total;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) {
        wh0[fb] = wh_first[fb];
        wl0[fb] = wl_first[fb];
    }
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
        ah0[sb] = in_hi[sb * 32];
        al0[sb] = in_lo[sb * 32];
    }
#if defined(STNERF_EXP_NOGLOBAL) || defined(STNERF_EXP_NOLDS)
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) { wh1[fb] = wh0[fb]; wl1[fb] = wl0[fb]; }
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) { ah1[sb] = ah0[sb]; al1[sb] = al0[sb]; }
#endif
#if defined(STNERF_EXP_NOGLOBAL)  /* development experiments only (wrong results): isolate a stall source */
#define H_LOAD_W(WH, WL, STEP) _Pragma("unroll") for (int fb = 0; fb < NFB; ++fb) { asm volatile("" : "+v"(WH[fb]), "+v"(WL[fb])); }
#else
#define H_LOAD_W(WH, WL, STEP)                                     \
    _Pragma("unroll") for (int fb = 0; fb < NFB; ++fb) {           \
        WH[fb] = whp[(STEP) * wstep + fb * 32];                    \
        WL[fb] = wlp[(STEP) * wstep + fb * 32];                    \
    }
#endif
#if defined(STNERF_EXP_NOLDS)
#define H_LOAD_A(AH, AL, STEP) _Pragma("unroll") for (int sb = 0; sb < NSB; ++sb) { asm volatile("" : "+v"(AH[sb]), "+v"(AL[sb])); }
#else
#define H_LOAD_A(AH, AL, STEP)                                     \
    _Pragma("unroll") for (int sb = 0; sb < NSB; ++sb) {           \
        AH[sb] = in_hi[(STEP) * 2 * TM + sb * 32];                 \
        AL[sb] = in_lo[(STEP) * 2 * TM + sb * 32];                 \
    }
#endif
#define H_LOAD_STEP(WH, WL, AH, AL, STEP) H_LOAD_W(WH, WL, STEP) H_LOAD_A(AH, AL, STEP)
    // one load per MFMA (2*NFB global + 2*NSB LDS loads <= 3*NFB*NSB MFMAs), fenced per half-iteration
#define H_INTERLEAVE()                                                                    \
    _Pragma("unroll") for (int i_ = 0; i_ < 2 * NFB; ++i_) {                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                \
    }                                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 2 * NSB; ++i_) {                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                \
    }                                                                                     \
    __builtin_amdgcn_sched_group_barrier(0x008, 3 * NFB * NSB - 2 * NFB - 2 * NSB, 0);    \
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_sched_barrier(0);
    {
        const int nx = 1 < steps ? 1 : steps - 1;
        H_LOAD_STEP(wh1, wl1, ah1, al1, nx)
    }
    mma_step_h<NFB, NSB, FIRST>(acc, wh0, wl0, ah0, al0, cinit);
    H_INTERLEAVE()
    int s = 1;
#pragma unroll 1
    for (; s + 2 <= steps; s += 2) {
        H_LOAD_STEP(wh0, wl0, ah0, al0, s + 1)
        mma_step_h<NFB, NSB, false>(acc, wh1, wl1, ah1, al1, cinit);
        H_INTERLEAVE()
        const int nx = (s + 2 < steps) ? (s + 2) : (steps - 1);
        H_LOAD_STEP(wh1, wl1, ah1, al1, nx)
        mma_step_h<NFB, NSB, false>(acc, wh0, wl0, ah0, al0, cinit);
        H_INTERLEAVE()
    }
    if (s < steps) mma_step_h<NFB, NSB, false>(acc, wh1, wl1, ah1, al1, cinit);
#undef H_LOAD_STEP
#undef H_LOAD_W
#undef H_LOAD_A
#undef H_INTERLEAVE
}

// One dense layer: out = relu(W in + b) for this wave's NFB*32 features x NSB*32 samples, written back as
// hi/lo fp16 planes.  Accumulators carry the 2^8 weight scale: C starts at 2^8 * bias, the epilogue
// multiplies by 2^-8 (exact).
// Range guard: an activation beyond the fp16 range (>= 65520) turns into inf in the split (and the NaNs that follow are
// flushed to 0 by the integer ReLU of the next layer: silently wrong, finite pixels), so the epilogue tracks the largest
// value it splits and raises the caller's `overflow` flag right there (nothing is kept live across layers: carrying
// a running maximum through the kernel cost 130+ VGPR spills).
template <int TM, int NFB, int NSB, int NFB_NEXT>
__device__ __forceinline__ void dense_layer_h(const half8* __restrict__ whi, const half8* __restrict__ wlo, int n_total,
                                              const half8* inA_hi, const half8* inA_lo, int octA, const half8* inB_hi,
                                              const half8* inB_lo, int octB, half8* out_hi, half8* out_lo, int n0,
                                              int sb0, int lane, const HFrag<NFB>& wfirst, const half8* next_hi,
                                              const half8* next_lo, const float* next_lane_bias,
                                              HFrag<NFB_NEXT>& wnext, uint32_t* overflow PH_PARAMS) {
    const int h = lane >> 5, c = lane & 31;
    const int s0 = sb0 * 32 + c;
    f32x16 cinit[NFB];
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cinit[fb][4 * q + 0] = wfirst.b[fb][q].x * WSCALE;
            cinit[fb][4 * q + 1] = wfirst.b[fb][q].y * WSCALE;
            cinit[fb][4 * q + 2] = wfirst.b[fb][q].z * WSCALE;
            cinit[fb][4 * q + 3] = wfirst.b[fb][q].w * WSCALE;
        }
    f32x16 acc[NFB][NSB];
    const half8* whp = hweight_lane_ptr(whi, n_total, n0, lane);
    const half8* wlp = hweight_lane_ptr(wlo, n_total, n0, lane);
    mma_segment_h<TM, NFB, NSB, true>(acc, wfirst.wh, wfirst.wl, cinit, whp, wlp, n_total, inA_hi + h * TM + s0,
                                      inA_lo + h * TM + s0, octA / 2);
    if (octB > 0) {
        half8 sh[NFB], sl[NFB];
        const half8* whp2 = whp + (int64_t)octA * n_total;
        const half8* wlp2 = wlp + (int64_t)octA * n_total;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) {
            sh[fb] = whp2[fb * 32];
            sl[fb] = wlp2[fb * 32];
        }
        mma_segment_h<TM, NFB, NSB, false>(acc, sh, sl, cinit, whp2, wlp2, n_total, inB_hi + h * TM + s0,
                                           inB_lo + h * TM + s0, octB / 2);
    }
    load_hfrag<NFB_NEXT>(wnext, next_hi, next_lo, next_lane_bias);
    PH(PH_MMA);
    __syncthreads();  // all waves done reading the input planes (out aliases in)
    PH(PH_BAR1);
    _Float16* oh = reinterpret_cast<_Float16*>(out_hi);
    _Float16* ol = reinterpret_cast<_Float16*>(out_lo);
    int vmax_bits = 0;  // max over the RAW accumulator bit patterns: negative floats are negative ints (the ReLU for free), a
                        // positive NaN beats every finite value; one v_max3_i32 per two values
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f = n0 + fb * 32 + 8 * q + 4 * h;  // this lane's 4 consecutive features: half an octet
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) {
                half4 vh, vl;
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const int b0 = __float_as_int(acc[fb][sb][4 * q + r]), b1 = __float_as_int(acc[fb][sb][4 * q + r + 1]);
                    const int m01 = b0 > b1 ? b0 : b1;
                    vmax_bits = vmax_bits > m01 ? vmax_bits : m01;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = relu_bits(acc[fb][sb][4 * q + r]) * WSCALE_INV;
                    _Float16 a_, b_;
                    split_f16(v, a_, b_);
                    vh[r] = a_;
                    vl[r] = b_;
                }
                const int at = ((f >> 3) * TM + sb * 32 + s0) * 8 + (f & 7);
                *reinterpret_cast<half4*>(oh + at) = vh;
                *reinterpret_cast<half4*>(ol + at) = vl;
            }
        }
    }
    if (overflow && vmax_bits >= __float_as_int(65520.f * WSCALE)) atomicOr(overflow, 1u);  // (accumulators carry the 2^8 weight scale)
    PH(PH_EPI);
}

// feature f of tile sample s in a plane: element ((f>>3)*TM + s)*8 + (f&7)
template <int TM>
__device__ __forceinline__ void put_feature(_Float16* ph, _Float16* pl, int s, int f, float v) {
    _Float16 a_, b_;
    split_f16(v, a_, b_);
    const int at = ((f >> 3) * TM + s) * 8 + (f & 7);
    ph[at] = a_;
    pl[at] = b_;
}

// partial dot products of heads (1..3 outputs) over an octet range, fp32 on hi+lo
template <int TM, int NOUT>
__device__ __forceinline__ void head_partial_h(const half8* act_hi, const half8* act_lo, int s, int o_begin, int o_end,
                                               const float* __restrict__ w, int ldw, float (&sum)[NOUT]) {
#pragma unroll
    for (int o = 0; o < NOUT; ++o) sum[o] = 0.f;
    for (int oc = o_begin; oc < o_end; ++oc) {
        const half8 vh = act_hi[oc * TM + s], vl = act_lo[oc * TM + s];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = (float)vh[j] + (float)vl[j];
#pragma unroll
            for (int o = 0; o < NOUT; ++o) sum[o] = fmaf(v, w[o * ldw + 8 * oc + j], sum[o]);
        }
    }
}

#define DENSE_H(TM_, NW_, N_, NN_, LI_, INA_, OCTA_, INB_, OCTB_, WFIRST_, LNEXT_, WNEXT_)                              \
    dense_layer_h<TM_, WaveSplit<TM_, NW_, N_>::NFB, WaveSplit<TM_, NW_, N_>::NSB, WaveSplit<TM_, NW_, NN_>::NFB>(      \
        hreg + L.whi[LI_], hreg + L.wlo[LI_], N_, INA_##_hi, INA_##_lo, OCTA_, INB_##_hi, INB_##_lo, OCTB_, act_hi,     \
        act_lo, WaveSplit<TM_, NW_, N_>::n0(wave), WaveSplit<TM_, NW_, N_>::sb0(wave), lane, WFIRST_,                   \
        hweight_lane_ptr(hreg + L.whi[LNEXT_], NN_, WaveSplit<TM_, NW_, NN_>::n0(wave), lane),                          \
        hweight_lane_ptr(hreg + L.wlo[LNEXT_], NN_, WaveSplit<TM_, NW_, NN_>::n0(wave), lane),                          \
        net + BIAS_OFF(LNEXT_) + WaveSplit<TM_, NW_, NN_>::n0(wave) + 4 * (lane >> 5), WNEXT_, a.overflow PH_ARGS)

// ---------------------------------------------------------------------------------------------
// SpaceNet
// ---------------------------------------------------------------------------------------------
template <int TM, int NW, bool USE_TIME, bool DEEP = false>
__global__ __launch_bounds__(NW * 64, (TM == 128 ? NW / 4 : 2)) void spacenet_h_kernel(SpaceArgs a) {
    constexpr int NTHREADS = NW * 64;
    constexpr int NPARTS = NTHREADS / TM;
    extern __shared__ __attribute__((aligned(16))) half8 smem_h[];
    half8* act_hi = smem_h;                 // [32][TM]
    half8* act_lo = smem_h + 32 * TM;       // [32][TM]
    half8* enc_hi = smem_h + 64 * TM;       // [8][TM]
    half8* enc_lo = smem_h + 72 * TM;       // [8][TM]
    half8* const null_hi = nullptr;
    half8* const null_lo = nullptr;
    float* scratch_sigma = reinterpret_cast<float*>(enc_hi + 6 * TM);  // enc_hi octets 6..7: 1024 floats
    float* scratch_rgb = reinterpret_cast<float*>(act_hi + 16 * TM);   // act_hi octets 16..31 (free after rgb1)
    const SpaceLayoutH L = space_layout_h(USE_TIME, DEEP);
#define BIAS_OFF(LI_) ((LI_) < 7 ? L.f32.b[(LI_) < 7 ? (LI_) : 0] : (LI_) == 7 ? L.f32.b_rgb1 : L.f32.b_deep[(LI_) >= 8 ? (LI_) - 8 : 0])
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int part = __builtin_amdgcn_readfirstlane(tid / TM);
    const int s = tid & (TM - 1);
    const int64_t rows = worklist_rows(a.wl);
    const int ns = a.wl.ns;
    PH_DECL
    const half8* hreg0 = reinterpret_cast<const half8*>(a.net + L.f32.total);
    HFrag<WaveSplit<TM, NW, 256>::NFB> wA, wB;
    HFrag<WaveSplit<TM, NW, 128>::NFB> wR, wR2;
    load_hfrag(wA, hweight_lane_ptr(hreg0 + L.whi[0], 256, WaveSplit<TM, NW, 256>::n0(wave), lane),
               hweight_lane_ptr(hreg0 + L.wlo[0], 256, WaveSplit<TM, NW, 256>::n0(wave), lane),
               a.net + L.f32.b[0] + WaveSplit<TM, NW, 256>::n0(wave) + 4 * (lane >> 5));

    for (int64_t tile = blockIdx.x; tile * TM < rows; tile += gridDim.x) {
        PH(PH_MISC);
        int64_t opaque_zero = 0;  // keeps loop-invariant weight/bias loads inside the iteration (see mlp.hip)
        asm volatile("" : "+s"(opaque_zero));
        const float* net = a.net + opaque_zero;
        const half8* hreg = hreg0 + opaque_zero;
        const int64_t row = tile * TM + s;
        const bool valid = row < rows;
        int64_t ray = 0;
        int k = 0;
        if (valid) {
            const int64_t slot = row / ns;
            k = (int)(row - slot * ns);
            ray = a.wl.ray_list ? (int64_t)a.wl.ray_list[slot] : slot;
        }
        _Float16* eh = reinterpret_cast<_Float16*>(enc_hi);
        _Float16* el = reinterpret_cast<_Float16*>(enc_lo);
        // ---- PE_10(pos): 63 features + zero pad -> enc planes
        {
            float p[3] = {0.f, 0.f, 0.f};
            if (valid) {
                const float* src = a.xyz + ray * a.xyz_ray_stride + 3 * k;
                p[0] = src[0];
                p[1] = src[1];
                p[2] = src[2];
            }
            if (part == 0) {
#pragma unroll
                for (int dmn = 0; dmn < 3; ++dmn) put_feature<TM>(eh, el, s, dmn, p[dmn]);
            }
            if (part == NPARTS - 1) put_feature<TM>(eh, el, s, 63, 0.f);
            for (int fq = part; fq < 10; fq += NPARTS) {
                const float freq = (float)(1 << fq);
#pragma unroll
                for (int dmn = 0; dmn < 3; ++dmn) {
                    float sn, cs;
                    sincos_pe(p[dmn] * freq, sn, cs);
                    put_feature<TM>(eh, el, s, 3 + fq * 6 + dmn, sn);
                    put_feature<TM>(eh, el, s, 6 + fq * 6 + dmn, cs);
                }
            }
        }
        PH(PH_PE);
        __syncthreads();
        PH(PH_BAR2);
        DENSE_H(TM, NW, 256, 256, 0, enc, 8, null, 0, wA, 1, wB);
        __syncthreads();
        PH(PH_BAR2);
        DENSE_H(TM, NW, 256, 256, 1, act, 32, null, 0, wB, 2, wA);
        __syncthreads();
        PH(PH_BAR2);
        DENSE_H(TM, NW, 256, 256, 2, act, 32, null, 0, wA, 3, wB);
        __syncthreads();
        PH(PH_BAR2);
        DENSE_H(TM, NW, 256, 256, 3, act, 32, null, 0, wB, 4, wA);
        __syncthreads();
        PH(PH_BAR2);
        DENSE_H(TM, NW, 256, 256, 4, act, 32, enc, 8, wA, 5, wB);
        // enc is free: relu(PE_4(dir)) (27) + relu(PE_10(time)) (21) -> enc features 0..47
        {
            float dv[3] = {0.f, 0.f, 0.f};
            if (valid) {
                const float* src = a.dirs + ray * a.dirs_ray_stride;
                dv[0] = src[0];
                dv[1] = src[1];
                dv[2] = src[2];
            }
            if (part == 0) {
#pragma unroll
                for (int dmn = 0; dmn < 3; ++dmn) put_feature<TM>(eh, el, s, dmn, fmaxf(dv[dmn], 0.f));
            }
            for (int fq = part; fq < 4; fq += NPARTS) {
                const float freq = (float)(1 << fq);
#pragma unroll
                for (int dmn = 0; dmn < 3; ++dmn) {
                    float sn, cs;
                    sincos_pe(dv[dmn] * freq, sn, cs);
                    put_feature<TM>(eh, el, s, 3 + fq * 6 + dmn, relu_bits(sn));
                    put_feature<TM>(eh, el, s, 6 + fq * 6 + dmn, relu_bits(cs));
                }
            }
            if (USE_TIME) {
                const float tv = valid ? a.times[ray * a.times_ray_stride] : 0.f;
                if (part == NPARTS - 1) put_feature<TM>(eh, el, s, 27, fmaxf(tv, 0.f));
                for (int fq = NPARTS - 1 - part; fq < 10; fq += NPARTS) {
                    float sn, cs;
                    sincos_pe(tv * (float)(1 << fq), sn, cs);
                    put_feature<TM>(eh, el, s, 28 + 2 * fq, relu_bits(sn));
                    put_feature<TM>(eh, el, s, 29 + 2 * fq, relu_bits(cs));
                }
            } else if (part == NPARTS - 1) {
#pragma unroll
                for (int f = 27; f < 32; ++f) put_feature<TM>(eh, el, s, f, 0.f);
            }
        }
        PH(PH_ENC2);
        __syncthreads();
        PH(PH_BAR2);
        DENSE_H(TM, NW, 256, 256, 5, act, 32, null, 0, wB, 6, wA);
        __syncthreads();
        PH(PH_BAR2);
        DENSE_H(TM, NW, 256, 128, 6, act, 32, null, 0, wA, 7, wR);
        __syncthreads();
        PH(PH_BAR2);
        // ---- sigma head (fp32)
        float sigma;
        {
            float ps[1];
            head_partial_h<TM, 1>(act_hi, act_lo, s, part * (32 / NPARTS), (part + 1) * (32 / NPARTS),
                                  net + L.f32.w_sigma, 256, ps);
            scratch_sigma[part * TM + s] = ps[0];
            __syncthreads();
            sigma = net[L.f32.b_sigma];
#pragma unroll
            for (int pp = 0; pp < NPARTS; ++pp) sigma += scratch_sigma[pp * TM + s];
        }
        PH(PH_HEAD);
        if constexpr (!DEEP) {
            DENSE_H(TM, NW, 128, 256, 7, act, 32, enc, (USE_TIME ? 6 : 4), wR, 0, wA);  // + next tile's layer 0
        } else {  // deep_rgb (modeling/spacenet.py:68-79): two more 128-wide hidden layers
            DENSE_H(TM, NW, 128, 128, 7, act, 32, enc, (USE_TIME ? 6 : 4), wR, 8, wR2);
            __syncthreads();
            DENSE_H(TM, NW, 128, 128, 8, act, 16, null, 0, wR2, 9, wR);
            __syncthreads();
            DENSE_H(TM, NW, 128, 256, 9, act, 16, null, 0, wR, 0, wA);
        }
        __syncthreads();
        PH(PH_BAR2);
        {
            float ps[3];
            head_partial_h<TM, 3>(act_hi, act_lo, s, part * (16 / NPARTS), (part + 1) * (16 / NPARTS),
                                  net + L.f32.w_rgb2, 128, ps);
            scratch_rgb[(part * 3 + 0) * TM + s] = ps[0];
            scratch_rgb[(part * 3 + 1) * TM + s] = ps[1];
            scratch_rgb[(part * 3 + 2) * TM + s] = ps[2];
            __syncthreads();
            if (part == 0 && valid) {
                float4 o;
                o.x = net[L.f32.b_rgb2 + 0];
                o.y = net[L.f32.b_rgb2 + 1];
                o.z = net[L.f32.b_rgb2 + 2];
#pragma unroll
                for (int pp = 0; pp < NPARTS; ++pp) {
                    o.x += scratch_rgb[(pp * 3 + 0) * TM + s];
                    o.y += scratch_rgb[(pp * 3 + 1) * TM + s];
                    o.z += scratch_rgb[(pp * 3 + 2) * TM + s];
                }
                o.w = sigma;
                if (a.overflow && !(isfinite(o.x) && isfinite(o.y) && isfinite(o.z) && isfinite(o.w))) atomicOr(a.overflow, 1u);
                *reinterpret_cast<float4*>(a.raw + ray * a.raw_ray_stride + 4 * k) = o;
            }
        }
        __syncthreads();
        PH(PH_HEAD);
    }
    PH_FLUSH;
#undef BIAS_OFF
}

// ---------------------------------------------------------------------------------------------
// MotionNet
// ---------------------------------------------------------------------------------------------
template <int TM, int NW>
constexpr int motion_h_lds_bytes() { return 16 * 2 * TM * 16 + 3 * NW * 64 * 4; }

#define DENSE_HM(TM_, NW_, LI_, INA_, OCTA_, WFIRST_, LNEXT_, WNEXT_)                                                   \
    dense_layer_h<TM_, WaveSplit<TM_, NW_, 128>::NFB, WaveSplit<TM_, NW_, 128>::NSB, WaveSplit<TM_, NW_, 128>::NFB>(    \
        hreg + L.whi[LI_], hreg + L.wlo[LI_], 128, INA_##_hi, INA_##_lo, OCTA_, null_hi, null_lo, 0, act_hi, act_lo,    \
        WaveSplit<TM_, NW_, 128>::n0(wave), WaveSplit<TM_, NW_, 128>::sb0(wave), lane, WFIRST_,                         \
        hweight_lane_ptr(hreg + L.whi[LNEXT_], 128, WaveSplit<TM_, NW_, 128>::n0(wave), lane),                          \
        hweight_lane_ptr(hreg + L.wlo[LNEXT_], 128, WaveSplit<TM_, NW_, 128>::n0(wave), lane),                          \
        net + L.f32.b[LNEXT_] + WaveSplit<TM_, NW_, 128>::n0(wave) + 4 * (lane >> 5), WNEXT_, a.overflow PH_ARGS)

template <int TM, int NW>
__global__ __launch_bounds__(NW * 64, (TM == 128 ? NW / 4 : 2)) void motionnet_h_kernel(MotionArgs a) {
    constexpr int NTHREADS = NW * 64;
    constexpr int NPARTS = NTHREADS / TM;
    extern __shared__ __attribute__((aligned(16))) half8 smem_h[];
    half8* act_hi = smem_h;             // [16][TM]
    half8* act_lo = smem_h + 16 * TM;   // [16][TM]
    half8* enc_hi = smem_h;             // [12][TM]: 84 features + 12 zero pads; ALIAS the activation planes (layer 0 is
    half8* enc_lo = smem_h + 16 * TM;   // their only reader and the layer's barrier precedes its first output write)
    half8* const null_hi = nullptr;
    half8* const null_lo = nullptr;
    float* scratch = reinterpret_cast<float*>(smem_h + 32 * TM);  // 3*NTHREADS floats
    const MotionLayoutH L = motion_layout_h();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int part = __builtin_amdgcn_readfirstlane(tid / TM);
    const int s = tid & (TM - 1);
    const int64_t rows = worklist_rows(a.wl);
    const int ns = a.wl.ns;
    PH_DECL
    const half8* hreg0 = reinterpret_cast<const half8*>(a.net + L.f32.total);
    HFrag<WaveSplit<TM, NW, 128>::NFB> wA, wB;
    load_hfrag(wA, hweight_lane_ptr(hreg0 + L.whi[0], 128, WaveSplit<TM, NW, 128>::n0(wave), lane),
               hweight_lane_ptr(hreg0 + L.wlo[0], 128, WaveSplit<TM, NW, 128>::n0(wave), lane),
               a.net + L.f32.b[0] + WaveSplit<TM, NW, 128>::n0(wave) + 4 * (lane >> 5));

    for (int64_t tile = blockIdx.x; tile * TM < rows; tile += gridDim.x) {
        int64_t opaque_zero = 0;
        asm volatile("" : "+s"(opaque_zero));
        const float* net = a.net + opaque_zero;
        const half8* hreg = hreg0 + opaque_zero;
        const int64_t row = tile * TM + s;
        const bool valid = row < rows;
        int64_t ray = 0;
        int k = 0;
        if (valid) {
            const int64_t slot = row / ns;
            k = (int)(row - slot * ns);
            ray = a.wl.ray_list ? (int64_t)a.wl.ray_list[slot] : slot;
        }
        float p[3] = {0.f, 0.f, 0.f};
        float tv = 0.f;
        if (valid) {
            const float* src = a.xyz + ray * a.xyz_ray_stride + 3 * k;
            p[0] = src[0];
            p[1] = src[1];
            p[2] = src[2];
            tv = a.times[ray * a.times_ray_stride];
        }
        {  // PE_10([x,y,z,t]) with the fractional-time lerp (modeling/motion_net.py:49-60), as in mlp.hip
            _Float16* eh = reinterpret_cast<_Float16*>(enc_hi);
            _Float16* el = reinterpret_cast<_Float16*>(enc_lo);
            const float lo = (a.add_to_xyz & STNERF_MOTION_PLAIN_TIME) ? tv : floorf(tv);  // input_time=False: PE(input) as is
            const float wgt = tv - lo;
            const bool frac = wgt != 0.f;
            const float om = 1.f - wgt;
            auto mix = [&](float va, float vb) { return frac ? om * va + wgt * vb : va; };
            if (part == 0) {
#pragma unroll
                for (int dmn = 0; dmn < 3; ++dmn) put_feature<TM>(eh, el, s, dmn, mix(p[dmn], p[dmn]));
                put_feature<TM>(eh, el, s, 3, mix(lo, lo + 1.f));
            }
            if (part == NPARTS - 1) {
#pragma unroll
                for (int f = 84; f < 96; ++f) put_feature<TM>(eh, el, s, f, 0.f);
            }
            for (int fq = part; fq < 10; fq += NPARTS) {
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
                    put_feature<TM>(eh, el, s, 4 + fq * 8 + dmn, mix(sn, sn2));
                    put_feature<TM>(eh, el, s, 8 + fq * 8 + dmn, mix(cs, cs2));
                }
            }
        }
        __syncthreads();
        DENSE_HM(TM, NW, 0, enc, 12, wA, 1, wB);
        __syncthreads();
        DENSE_HM(TM, NW, 1, act, 16, wB, 2, wA);
        __syncthreads();
        DENSE_HM(TM, NW, 2, act, 16, wA, 3, wB);
        __syncthreads();
        DENSE_HM(TM, NW, 3, act, 16, wB, 4, wA);
        __syncthreads();
        DENSE_HM(TM, NW, 4, act, 16, wA, 0, wB);  // + next tile's layer 0
        wA = wB;
        __syncthreads();
        {
            float ps[3];
            head_partial_h<TM, 3>(act_hi, act_lo, s, part * (16 / NPARTS), (part + 1) * (16 / NPARTS),
                                  net + L.f32.w_out, 128, ps);
            scratch[(part * 3 + 0) * TM + s] = ps[0];
            scratch[(part * 3 + 1) * TM + s] = ps[1];
            scratch[(part * 3 + 2) * TM + s] = ps[2];
            __syncthreads();
            if (part == 0 && valid) {
                float fl[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    fl[c] = net[L.f32.b_out + c];
#pragma unroll
                    for (int pp = 0; pp < NPARTS; ++pp) fl[c] += scratch[(pp * 3 + c) * TM + s];
                }
                if (a.overflow && !(isfinite(fl[0]) && isfinite(fl[1]) && isfinite(fl[2]))) atomicOr(a.overflow, 1u);
                if (a.flow) {
                    float* dst = a.flow + ray * a.flow_ray_stride + 3 * k;
                    dst[0] = fl[0];
                    dst[1] = fl[1];
                    dst[2] = fl[2];
                }
                if (a.add_to_xyz & STNERF_MOTION_ADD_TO_XYZ) {
                    float* dst = a.xyz + ray * a.xyz_ray_stride + 3 * k;
                    dst[0] = p[0] + fl[0];
                    dst[1] = p[1] + fl[1];
                    dst[2] = p[2] + fl[2];
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
static void pack_linear_h(const float* w, int out_f, int in_f, int oct, _Float16* hi, _Float16* lo) {
    // (out,in) fp32 row-major -> two planes [oct][out][8] of fp16: W*2^8 = hi + lo
    memset(hi, 0, sizeof(_Float16) * (size_t)oct * out_f * 8);
    memset(lo, 0, sizeof(_Float16) * (size_t)oct * out_f * 8);
    for (int n = 0; n < out_f; ++n)
        for (int k = 0; k < in_f; ++k) {
            const float v = w[(size_t)n * in_f + k] * WSCALE;
            const _Float16 h = (_Float16)v;
            const _Float16 l = (_Float16)(v - (float)h);
            const size_t at = ((size_t)(k >> 3) * out_f + n) * 8 + (k & 7);
            hi[at] = h;
            lo[at] = l;
        }
}

static float max_abs(const float* w, size_t n) {
    float m = 0.f;
    for (size_t i = 0; i < n; ++i) {
        const float a = w[i] < 0 ? -w[i] : w[i];
        if (!(a <= m)) m = a;  // also catches NaN
    }
    return m;
}

static int grid_for_h(int64_t n_rays, int ns, int tm) {
    const int64_t tiles = (n_rays * ns + tm - 1) / tm;
    return (int)(tiles < 8192 ? tiles : 8192);
}

}  // namespace stnerf

using namespace stnerf;

extern "C" int64_t stnerf_packed_bytes_f16x3(int kind) {
    switch (kind) {
        case STNERF_NET_SPACE: return space_layout_h(false).total_bytes;
        case STNERF_NET_SPACE_TIME: return space_layout_h(true).total_bytes;
        case STNERF_NET_SPACE_DEEP: return space_layout_h(false, true).total_bytes;
        case STNERF_NET_SPACE_TIME_DEEP: return space_layout_h(true, true).total_bytes;
        case STNERF_NET_MOTION: return motion_layout_h().total_bytes;
        default: set_error("packed_bytes_f16x3: unknown net kind %d", kind); return STNERF_EINVAL;
    }
}

extern "C" int stnerf_pack_net_f16x3(int kind, const float* const* W, const float* const* B, int n_tensors,
                                     void* dst_host, int64_t dst_bytes) {
    if (kind == STNERF_NET_MOTION) {
        const MotionLayoutH L = motion_layout_h();
        STNERF_REQUIRE(dst_host && dst_bytes >= L.total_bytes, "pack_net_f16x3: dst too small");
        const int rc = stnerf_pack_net(kind, W, B, n_tensors, dst_host, L.f32.total * 4);
        if (rc != STNERF_OK) return rc;
        const int in_f[5] = {84, 128, 128, 128, 128};
        _Float16* hreg = reinterpret_cast<_Float16*>(static_cast<char*>(dst_host) + L.f32.total * 4);
        for (int i = 0; i < 5; ++i) {
            const float m = max_abs(W[i], (size_t)128 * in_f[i]);
            STNERF_REQUIRE(m * WSCALE < 60000.f, "pack_net_f16x3: |weight| up to %g does not fit the fp16 split (use fp32)", m);
            pack_linear_h(W[i], 128, in_f[i], L.oct[i], hreg + L.whi[i] * 8, hreg + L.wlo[i] * 8);
        }
        return STNERF_OK;
    }
    STNERF_REQUIRE(STNERF_NET_IS_SPACE(kind), "pack_net_f16x3: unknown kind %d", kind);
    const bool ut = STNERF_NET_USES_TIME(kind), deep = STNERF_NET_IS_DEEP(kind);
    const SpaceLayoutH L = space_layout_h(ut, deep);
    STNERF_REQUIRE(dst_host && dst_bytes >= L.total_bytes, "pack_net_f16x3: dst too small");
    const int rc = stnerf_pack_net(kind, W, B, n_tensors, dst_host, L.f32.total * 4);
    if (rc != STNERF_OK) return rc;
    const int in_f[10] = {63, 256, 256, 256, 319, 256, 256, 256 + 27 + (ut ? 21 : 0), 128, 128};
    const int widx[10] = {0, 1, 2, 3, 4, 5, 6, 8, 9, 10};
    _Float16* hreg = reinterpret_cast<_Float16*>(static_cast<char*>(dst_host) + L.f32.total * 4);
    for (int i = 0; i < (deep ? 10 : 8); ++i) {
        const int n = i < 7 ? 256 : 128;
        const float m = max_abs(W[widx[i]], (size_t)n * in_f[i]);
        STNERF_REQUIRE(m * WSCALE < 60000.f, "pack_net_f16x3: |weight| up to %g does not fit the fp16 split (use fp32)", m);
        pack_linear_h(W[widx[i]], n, in_f[i], L.oct[i], hreg + L.whi[i] * 8, hreg + L.wlo[i] * 8);
    }
    return STNERF_OK;
}

extern "C" int stnerf_spacenet_fwd_f16x3(int kind, const void* packed, int64_t n_rays, int ns, const int32_t* ray_list,
                                         const int32_t* ray_count, const float* xyz, int64_t xyz_ray_stride,
                                         const float* dirs, int64_t dirs_ray_stride, const float* times,
                                         int64_t times_ray_stride, float* raw, int64_t raw_ray_stride,
                                         uint32_t* overflow, stnerf_stream_t stream) {
    STNERF_REQUIRE(STNERF_NET_IS_SPACE(kind), "spacenet_fwd_f16x3: bad kind %d", kind);
    STNERF_REQUIRE(packed && xyz && dirs && raw, "spacenet_fwd_f16x3: null pointer");
    STNERF_REQUIRE(!STNERF_NET_USES_TIME(kind) || times, "spacenet_fwd_f16x3: net takes time but times is null");
    STNERF_REQUIRE(n_rays >= 0 && ns >= 1, "spacenet_fwd_f16x3: bad shape");
    STNERF_REQUIRE((raw_ray_stride & 3) == 0 && ((uintptr_t)raw & 15) == 0 && ((uintptr_t)packed & 15) == 0,
                   "spacenet_fwd_f16x3: raw / packed must be 16-byte aligned");
    if (n_rays == 0) return STNERF_OK;
    SpaceArgs a{static_cast<const float*>(packed), {n_rays, ns, ray_list, ray_count}, xyz, xyz_ray_stride, dirs,
                dirs_ray_stride, times, times_ray_stride, raw, raw_ray_stride, overflow};
    const char* e = getenv("STNERF_TILE_H");
    const bool four = e && !strcmp(e, "128");
    const bool small = e && !strcmp(e, "64") && !STNERF_NET_IS_DEEP(kind);
    const int tm = small ? 64 : 128;
    const int lds = 80 * tm * 16;
    const int grid = grid_for_h(n_rays, ns, tm);
    const bool ut = STNERF_NET_USES_TIME(kind);
    auto launch = [&](auto kernel, int nthreads) -> int {
        if (const int rc = reserve_dynamic_lds(reinterpret_cast<const void*>(kernel), lds, "spacenet_fwd_f16x3")) return rc;
        LaunchTimer timer(PROF_SPACENET, kind, n_rays, ns, 0, as_stream(stream));
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(nthreads), lds, as_stream(stream), a);
        STNERF_CHECK_LAUNCH("spacenet_fwd_f16x3");
        return STNERF_OK;
    };
    if (STNERF_NET_IS_DEEP(kind)) {  // deep_rgb: compiled for the default tile configuration only (the knob is ignored)
        return ut ? launch(spacenet_h_kernel<128, 8, true, true>, 512)
                  : launch(spacenet_h_kernel<128, 8, false, true>, 512);
    }
    if (small)
        return ut ? launch(spacenet_h_kernel<64, 4, true>, 256)
                  : launch(spacenet_h_kernel<64, 4, false>, 256);
    if (four)
        return ut ? launch(spacenet_h_kernel<128, 4, true>, 256)
                  : launch(spacenet_h_kernel<128, 4, false>, 256);
    return ut ? launch(spacenet_h_kernel<128, 8, true>, 512)
              : launch(spacenet_h_kernel<128, 8, false>, 512);
}

extern "C" int stnerf_motionnet_fwd_f16x3(const void* packed, int64_t n_rays, int ns, const int32_t* ray_list,
                                          const int32_t* ray_count, float* xyz, int64_t xyz_ray_stride,
                                          const float* times, int64_t times_ray_stride, float* flow,
                                          int64_t flow_ray_stride, int add_to_xyz, uint32_t* overflow,
                                          stnerf_stream_t stream) {
    STNERF_REQUIRE(packed && xyz && times, "motionnet_fwd_f16x3: null pointer");
    STNERF_REQUIRE(flow || (add_to_xyz & STNERF_MOTION_ADD_TO_XYZ), "motionnet_fwd_f16x3: nothing to write");
    STNERF_REQUIRE(n_rays >= 0 && ns >= 1, "motionnet_fwd_f16x3: bad shape");
    STNERF_REQUIRE(((uintptr_t)packed & 15) == 0, "motionnet_fwd_f16x3: packed weights must be 16-byte aligned");
    if (n_rays == 0) return STNERF_OK;
    MotionArgs a{static_cast<const float*>(packed), {n_rays, ns, ray_list, ray_count}, xyz, xyz_ray_stride, times,
                 times_ray_stride, flow, flow_ray_stride, add_to_xyz, overflow};
    const char* e = getenv("STNERF_TILE_HM");
    const bool big = e && !strcmp(e, "128x8");
    auto launch = [&](auto kernel, int lds, int nthreads, int tm) -> int {
        if (const int rc = reserve_dynamic_lds(reinterpret_cast<const void*>(kernel), lds, "motionnet_fwd_f16x3")) return rc;
        LaunchTimer timer(PROF_MOTIONNET, STNERF_NET_MOTION, n_rays, ns, 0, as_stream(stream));
        hipLaunchKernelGGL(kernel, dim3(grid_for_h(n_rays, ns, tm)), dim3(nthreads), lds, as_stream(stream), a);
        STNERF_CHECK_LAUNCH("motionnet_fwd_f16x3");
        return STNERF_OK;
    };
    if (big) return launch(motionnet_h_kernel<128, 8>, motion_h_lds_bytes<128, 8>(), 512, 128);
    return launch(motionnet_h_kernel<64, 4>, motion_h_lds_bytes<64, 4>(), 256, 64);
}

#ifdef STNERF_PHASE_PROF
extern "C" int stnerf_debug_read_phases_h(unsigned long long* host16, int reset) {
    if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 16) != hipSuccess) return STNERF_ELAUNCH;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) != hipSuccess) return STNERF_ELAUNCH;
    }
    return STNERF_OK;
}
#endif
