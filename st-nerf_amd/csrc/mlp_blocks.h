// Building blocks of the fused fp32 MLP kernels with LDS-resident activations (mlp.hip: one network per launch -- the
// op-level entry points stnerf_spacenet_fwd / stnerf_motionnet_fwd): the pipelined MFMA K loop, one dense layer on an
// LDS-resident tile, the VALU heads.  (The stage kernels, mlp_wave.hip / mlp_bf16x3.hip, share only the head grouping.)
// Design notes: mlp.hip header and DESIGN.md section 4.1.
#pragma once
#include <type_traits>

#include "mlp_common.h"

namespace stnerf {

// ---------------------------------------------------------------------------------------------
// One dense layer on the tile:  out[:, n] = act(bias[n] + sum_k W[n][k] in[k])   for the wave's
// NFB*32 features and all TM samples.  K comes from up to two LDS segments (quads kqA then kqB).
// ---------------------------------------------------------------------------------------------
template <int NFB>
struct WFrag {  // what a layer needs before its first MFMA, prefetched during the previous layer's tail:
    float4 w[NFB];     // step-0 weight operands of this lane (8 k values x NFB feature blocks)
    float4 b[NFB][4];  // bias of this lane's 16*NFB features = the C operand of the first MFMAs
};

// This lane's pointer to quad row 0 of a packed [K/4][N][4] weight matrix (lane half h takes row h).
__device__ __forceinline__ const float4* weight_lane_ptr(const float* base, int64_t w_off, int n_total, int n0, int lane) {
    return reinterpret_cast<const float4*>(base + w_off) + ((int64_t)(lane >> 5) * n_total + n0 + (lane & 31));
}

// K-loop weight loads are BUFFER loads: resource descriptor of the packed blob (4 SGPRs) + wave-uniform byte offset
// (soffset, advanced on the scalar ALU) + per-lane byte offset in one VGPR that never changes.  With a per-lane 64-bit
// pointer (global_load) every step costs 64-bit VALU adds, and on gfx950 every VALU instruction displaces f32 MFMA
// work (profiles/r01_dual_issue_microbench.md; stubbing the weight loads out of the K loop was worth +1.9 % on
// SpaceNet and +5.6 % on MotionNet).
using i32x4 = __attribute__((ext_vector_type(4))) int;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t weight_rsrc(const float* blob) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(blob), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ uint32_t weight_lane_bytes(int n_total, int lane) {
    return (uint32_t)((lane >> 5) * n_total + (lane & 31)) * 16u;
}
__device__ __forceinline__ float4 load_weight(__amdgpu_buffer_rsrc_t rsrc, uint32_t lane_bytes, uint32_t wave_bytes) {
    const i32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_bytes, wave_bytes, 0);
    return make_float4(__int_as_float(r.x), __int_as_float(r.y), __int_as_float(r.z), __int_as_float(r.w));
}

// lane_bias = bias + n0 + 4*(lane>>5): register 4q+r of block fb <-> feature n0 + fb*32 + 8q + 4h + r
template <int NFB>
__device__ __forceinline__ void load_wfrag(WFrag<NFB>& f, const float4* __restrict__ lane_ptr,
                                           const float* __restrict__ lane_bias) {
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) {
        f.w[fb] = lane_ptr[fb * 32];
#pragma unroll
        for (int q = 0; q < 4; ++q) f.b[fb][q] = *reinterpret_cast<const float4*>(lane_bias + fb * 32 + 8 * q);
    }
}

// 4 * NFB * NSB MFMAs of one K step.  ZERO_C (first step of a layer): C = `cinit` = the bias, so neither an
// accumulator initialisation nor a bias add in the epilogue is needed.
// NSC = 1: the C operand is the bias, the same registers for every sample block; NSC = NSB: one C operand per sample
// block (rgb_net.1: bias + the ray's direction / time columns, mlp_raybias.hip).
template <int NFB, int NSB, bool ZERO_C, int NSC>
__device__ __forceinline__ void mma_step(f32x16 (&acc)[NFB][NSB], const float4 (&w)[NFB], const float4 (&a)[NSB],
                                         const f32x16 (&cinit)[NFB][NSC]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) {
            const float wv = kk == 0 ? w[fb].x : kk == 1 ? w[fb].y : kk == 2 ? w[fb].z : w[fb].w;
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) {
                const float av = kk == 0 ? a[sb].x : kk == 1 ? a[sb].y : kk == 2 ? a[sb].z : a[sb].w;
                if (ZERO_C && kk == 0) {  // first MFMAs of the layer: C = bias
                    acc[fb][sb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, av, cinit[fb][NSC == 1 ? 0 : sb], 0, 0, 0);
                } else {
                    acc[fb][sb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, av, acc[fb][sb], 0, 0, 0);
                }
            }
        }
    }
}

// Software-pipelined K loop: the operands of step s+1 are in flight (global weights, LDS activations)
// while the 4*NFB*NSB MFMAs of step s issue.  Written as an explicit two-stage ping-pong (two register
// sets, loop unrolled by two) -- a rotate-the-copy form gets collapsed by the compiler into
// load -> wait -> use, which with one wave per SIMD exposes the full L2 latency every step.
// `wfirst` = the step-0 weights, already loaded by the caller (prefetched during the previous layer's
// epilogue); requires steps >= 2.
template <int TM, int NFB, int NSB, bool FIRST, int NSC>
__device__ __forceinline__ void mma_segment(f32x16 (&acc)[NFB][NSB], const float4 (&wfirst)[NFB],
                                            const f32x16 (&cinit)[NFB][NSC], __amdgpu_buffer_rsrc_t rsrc, uint32_t wwave,
                                            uint32_t wlane, int n_total, const float4* in, int steps) {
    // rsrc + wwave (wave-uniform byte offset) + wlane (this lane's byte offset) / in point at the first quad row of the
    // segment; one step = 2 quad rows = 8 k values.
    float4 w0[NFB], a0[NSB], w1[NFB], a1[NSB];
    const uint32_t wstep = 2u * (uint32_t)n_total * 16u;  // bytes per step
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) w0[fb] = wfirst[fb];
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) a0[sb] = in[sb * 32];
#if defined(STNERF_EXP_NOGLOBAL) || defined(STNERF_EXP_NOLDS)
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) w1[fb] = w0[fb];
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) a1[sb] = a0[sb];
#endif
    // Scheduling of one half-iteration = {issue the operand loads of the NEXT step, 4*NFB*NSB MFMAs of this
    // step}.  Two things are pinned: (1) sched_barrier(0) fences keep the loads in the half-iteration they
    // were written in -- left alone the machine scheduler sinks each load to just before its first use and
    // exposes the L2 latency; (2) inside the half-iteration the loads are interleaved ONE per MFMA
    // (sched_group_barrier): issued as a clump, the 2 global + 4 LDS loads take ~120 issue cycles, more than
    // the 64-cycle shadow of one MFMA, and the MFMA pipe idles ~3 % (measured with the loads stubbed out).
#define STNERF_INTERLEAVE()                                                                         \
    _Pragma("unroll") for (int i_ = 0; i_ < NFB; ++i_) {                                            \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); /* MFMA */                               \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); /* VMEM read */                          \
    }                                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < NSB; ++i_) {                                            \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                          \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); /* DS read */                            \
    }                                                                                               \
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NFB * NSB - NFB - NSB, 0);                      \
    __builtin_amdgcn_sched_barrier(0);
#if defined(STNERF_EXP_NOGLOBAL)   /* development experiments only: wrong results, isolates a stall source */
#define STNERF_LOAD_W(W, STEP) _Pragma("unroll") for (int fb = 0; fb < NFB; ++fb) asm volatile("" : "+v"(W[fb].x), "+v"(W[fb].y), "+v"(W[fb].z), "+v"(W[fb].w));
#else
#define STNERF_LOAD_W(W, STEP) _Pragma("unroll") for (int fb = 0; fb < NFB; ++fb) W[fb] = load_weight(rsrc, wlane + fb * 512u, wwave + (uint32_t)(STEP) * wstep);
#endif
#if defined(STNERF_EXP_NOLDS)
#define STNERF_LOAD_A(A, STEP) _Pragma("unroll") for (int sb = 0; sb < NSB; ++sb) asm volatile("" : "+v"(A[sb].x), "+v"(A[sb].y), "+v"(A[sb].z), "+v"(A[sb].w));
#else
#define STNERF_LOAD_A(A, STEP) _Pragma("unroll") for (int sb = 0; sb < NSB; ++sb) A[sb] = in[(STEP) * 2 * TM + sb * 32];
#endif
#define STNERF_LOAD_STEP(W, A, STEP) STNERF_LOAD_W(W, STEP) STNERF_LOAD_A(A, STEP)
    __builtin_amdgcn_sched_barrier(0);
    // peeled first pair of steps (the very first MFMAs of a layer take C = 0)
    STNERF_LOAD_STEP(w1, a1, 1)
    mma_step<NFB, NSB, FIRST, NSC>(acc, w0, a0, cinit);
    STNERF_INTERLEAVE()
    {
        const int nx = 2 < steps ? 2 : steps - 1;
        STNERF_LOAD_STEP(w0, a0, nx)
    }
    mma_step<NFB, NSB, false, NSC>(acc, w1, a1, cinit);
    STNERF_INTERLEAVE()
    int s = 2;
#pragma unroll 1
    for (; s + 2 <= steps; s += 2) {
        STNERF_LOAD_STEP(w1, a1, s + 1)
        mma_step<NFB, NSB, false, NSC>(acc, w0, a0, cinit);
        STNERF_INTERLEAVE()
        const int nx = (s + 2 < steps) ? (s + 2) : (steps - 1);  // clamped: never out of bounds
        STNERF_LOAD_STEP(w0, a0, nx)
        mma_step<NFB, NSB, false, NSC>(acc, w1, a1, cinit);
        STNERF_INTERLEAVE()
    }
    if (s < steps) mma_step<NFB, NSB, false, NSC>(acc, w0, a0, cinit);
#undef STNERF_INTERLEAVE
#undef STNERF_LOAD_STEP
#undef STNERF_LOAD_W
#undef STNERF_LOAD_A
}

// One dense layer.  The wave computes features [n0, n0 + NFB*32) for the samples of blocks
// [sb0, sb0 + NSB) of the tile.  `wfirst` holds this layer's step-0 weights (already in flight);
// before the barrier/epilogue the step-0 weights of the NEXT layer (`next_lane_ptr`) are issued into
// `wnext`, so the next layer's pipeline fill overlaps this layer's epilogue instead of following it.
// The C operands of rgb_net.1 for this lane's samples: row ray_of[sample] of the layer's ray-bias table (bias + the
// direction / time columns of the ray, mlp_raybias.hip), fetched ahead of the layer.
template <int NFB, int NSB>
struct RayC {
    f32x16 c[NFB][NSB];
};
template <int NFB, int NSB>
__device__ __forceinline__ void load_rayc(RayC<NFB, NSB>& rc, const float* __restrict__ raybias, const int32_t* ray_of /* LDS, [TM] */,
                                          int n0, int sb0, int lane) {
    const int h = lane >> 5, c = lane & 31;
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
        const float* row = raybias + (int64_t)ray_of[(sb0 + sb) * 32 + c] * 128 + n0 + 4 * h;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(row + fb * 32 + 8 * q);
                rc.c[fb][sb][4 * q + 0] = v.x;
                rc.c[fb][sb][4 * q + 1] = v.y;
                rc.c[fb][sb][4 * q + 2] = v.z;
                rc.c[fb][sb][4 * q + 3] = v.w;
            }
    }
}

template <int TM, int NFB, int NSB, bool RELU, int NFB_NEXT, class RC = void>
__device__ __forceinline__ void dense_layer(const float* __restrict__ base, int64_t w_off, int64_t b_off, int n_total,
                                            const float4* inA, int kqA, const float4* inB, int kqB, float4* out,
                                            int n0, int sb0, int lane, const WFrag<NFB>& wfirst,
                                            const float4* next_lane_ptr, const float* next_lane_bias,
                                            WFrag<NFB_NEXT>& wnext PH_PARAMS, const RayC<NFB, NSB>* rayc = nullptr) {
    const int h = lane >> 5, c = lane & 31;
    const int s0 = sb0 * 32 + c;  // this lane's sample column within the tile (+ sb*32)
    f32x16 acc[NFB][NSB];
    const __amdgpu_buffer_rsrc_t rsrc = weight_rsrc(base);
    const uint32_t wwave = (uint32_t)(w_off * 4) + (uint32_t)n0 * 16u;
    const uint32_t wlane = weight_lane_bytes(n_total, lane);
    if constexpr (!std::is_void<RC>::value) {  // one C operand per sample block (rgb_net.1)
        mma_segment<TM, NFB, NSB, true, NSB>(acc, wfirst.w, rayc->c, rsrc, wwave, wlane, n_total, inA + h * TM + s0, kqA / 2);
    } else {
        f32x16 cinit[NFB][1];
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cinit[fb][0][4 * q + 0] = wfirst.b[fb][q].x;
                cinit[fb][0][4 * q + 1] = wfirst.b[fb][q].y;
                cinit[fb][0][4 * q + 2] = wfirst.b[fb][q].z;
                cinit[fb][0][4 * q + 3] = wfirst.b[fb][q].w;
            }
        mma_segment<TM, NFB, NSB, true, 1>(acc, wfirst.w, cinit, rsrc, wwave, wlane, n_total, inA + h * TM + s0, kqA / 2);
        if (kqB > 0) {
            const uint32_t wwave2 = wwave + (uint32_t)kqA * (uint32_t)n_total * 16u;
            float4 wseg[NFB];
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) wseg[fb] = load_weight(rsrc, wlane + fb * 512u, wwave2);
            mma_segment<TM, NFB, NSB, false, 1>(acc, wseg, cinit, rsrc, wwave2, wlane, n_total, inB + h * TM + s0, kqB / 2);
        }
    }
    load_wfrag<NFB_NEXT>(wnext, next_lane_ptr, next_lane_bias);
    PH(PH_MMA);
    // every wave has finished READING the input tile before anyone overwrites it (out may alias inA)
    __syncthreads();
    PH(PH_BAR1);
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f = n0 + fb * 32 + 8 * q + 4 * h;  // first of this lane's 4 consecutive features
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) {
                float4 v;
                v.x = acc[fb][sb][4 * q + 0];
                v.y = acc[fb][sb][4 * q + 1];
                v.z = acc[fb][sb][4 * q + 2];
                v.w = acc[fb][sb][4 * q + 3];
                if (RELU) {
                    v.x = relu_bits(v.x);
                    v.y = relu_bits(v.y);
                    v.z = relu_bits(v.z);
                    v.w = relu_bits(v.w);
                }
                out[(f >> 2) * TM + sb * 32 + s0] = v;
            }
        }
    }
    PH(PH_EPI);
}

// out[c] partial dot products over a quad range, for heads with 1..3 outputs (VALU; the weights are
// wave-uniform so they come through the scalar cache).
template <int TM, int NOUT>
__device__ __forceinline__ void head_partial(const float4* act, int s, int q_begin, int q_end,
                                             const float* __restrict__ w, int ldw, float (&sum)[NOUT]) {
    // 4 quads per iteration: the 4 LDS reads and the scalar weight loads are issued together, then 16*NOUT FMAs on
    // independent partial sums (a one-quad loop serialises on lgkmcnt(0) every iteration)
    float part[4][NOUT];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int o = 0; o < NOUT; ++o) part[u][o] = 0.f;
    for (int q = q_begin; q < q_end; q += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = act[(q + u) * TM + s];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                const float4 wv = *reinterpret_cast<const float4*>(w + o * ldw + 4 * (q + u));
                part[u][o] = fmaf(v[u].x, wv.x, part[u][o]);
                part[u][o] = fmaf(v[u].y, wv.y, part[u][o]);
                part[u][o] = fmaf(v[u].z, wv.z, part[u][o]);
                part[u][o] = fmaf(v[u].w, wv.w, part[u][o]);
            }
        }
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) sum[o] = (part[0][o] + part[1][o]) + (part[2][o] + part[3][o]);
}

// DENSE(TM, NW, N, N_NEXT, <dense_layer args up to `out`>, wfirst, next_w_off, wnext): one layer with N outputs;
// prefetches step 0 of the following layer (N_NEXT outputs, packed at next_w_off) into wnext.
#define DENSE(TM_, NW_, N_, NN_, BASE_, WOFF_, BOFF_, INA_, KQA_, INB_, KQB_, OUT_, WFIRST_, NEXT_WOFF_, NEXT_BOFF_, WNEXT_) \
    dense_layer<TM_, WaveSplit<TM_, NW_, N_>::NFB, WaveSplit<TM_, NW_, N_>::NSB, true, WaveSplit<TM_, NW_, NN_>::NFB>(   \
        BASE_, WOFF_, BOFF_, N_, INA_, KQA_, INB_, KQB_, OUT_, WaveSplit<TM_, NW_, N_>::n0(wave),                        \
        WaveSplit<TM_, NW_, N_>::sb0(wave), lane, WFIRST_,                                                               \
        weight_lane_ptr(BASE_, NEXT_WOFF_, NN_, WaveSplit<TM_, NW_, NN_>::n0(wave), lane),                               \
        (BASE_) + (NEXT_BOFF_) + WaveSplit<TM_, NW_, NN_>::n0(wave) + 4 * (lane >> 5), WNEXT_ PH_ARGS)

// rgb_net.1: one C operand per sample block (RAYC_ = the prefetched RayC), no bias, no second K segment.
#define DENSE_RAYC(TM_, NW_, N_, NN_, BASE_, WOFF_, INA_, KQA_, OUT_, WFIRST_, NEXT_WOFF_, NEXT_BOFF_, WNEXT_, RAYC_)             \
    dense_layer<TM_, WaveSplit<TM_, NW_, N_>::NFB, WaveSplit<TM_, NW_, N_>::NSB, true, WaveSplit<TM_, NW_, NN_>::NFB, int>(       \
        BASE_, WOFF_, 0, N_, INA_, KQA_, nullptr, 0, OUT_, WaveSplit<TM_, NW_, N_>::n0(wave),                                     \
        WaveSplit<TM_, NW_, N_>::sb0(wave), lane, WFIRST_,                                                                        \
        weight_lane_ptr(BASE_, NEXT_WOFF_, NN_, WaveSplit<TM_, NW_, NN_>::n0(wave), lane),                                        \
        (BASE_) + (NEXT_BOFF_) + WaveSplit<TM_, NW_, NN_>::n0(wave) + 4 * (lane >> 5), WNEXT_ PH_ARGS, &(RAYC_))

}  // namespace stnerf
