// The device-side building blocks of the sample-split f32 wave kernels: the K loop on registers (segment_r / segment_p), the
// layer boundary (relu_rebias), the heads, and the two networks on a wave's 32 samples (space_wave, motion_wave).  Shared by
// csrc/mlp_wave.hip (the inference stage kernel, and its activation-storing instantiation for training) and
// csrc/train_wave.hip (the fused dX chain of the backward pass).  See mlp_wave.hip for the organisation.
#pragma once
#include <stdlib.h>
#include <string.h>

#include "mlp_wave_common.h"

namespace stnerf {
// Optional per-phase cycle accounting (development builds: -DSTNERF_WAVE_PROF): every wave adds its s_memtime deltas per
// phase; read back with stnerf_debug_wave_phases().
#ifdef STNERF_WAVE_PROF
static __device__ unsigned long long g_wphase[16];
struct WaveProf {
    unsigned long long t, acc[16];
};
#define WP_PARAM , WaveProf& wp
#define WP_ARG , wp
#define WP(i) do { const unsigned long long n_ = clock64(); wp.acc[i] += n_ - wp.t; wp.t = n_; } while (0)
#else
#define WP_PARAM
#define WP_ARG
#define WP(i) do { } while (0)
#endif
enum { WP_TOP = 0, WP_M_ENC = 1, WP_M_LAYERS = 2, WP_M_HEAD = 3, WP_S_PE = 4, WP_S_L0 = 5, WP_S_LOOP = 6, WP_S_MID = 7,
       WP_S_RGB1 = 8, WP_S_HEAD = 9, WP_END = 10, WP_ITEMS = 11, WP_L_SEG = 12, WP_L_EPI = 13 };

// quad rows 2 s + h, s < STEPS, of the staged encoding -> B-operand registers (block s >> 2, registers 4 (s & 3) ..)
template <int NBLK, int STEPS>
__device__ __forceinline__ void read_enc_blocks(const float* encw, int lane, f32x16 (&blk)[NBLK]) {
    const float4* e4 = reinterpret_cast<const float4*>(encw);
    const int h = lane >> 5, c = lane & 31;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const float4 v = e4[(2 * s + h) * WV_ROWS + c];
        blk[s >> 2][4 * (s & 3) + 0] = v.x;
        blk[s >> 2][4 * (s & 3) + 1] = v.y;
        blk[s >> 2][4 * (s & 3) + 2] = v.z;
        blk[s >> 2][4 * (s & 3) + 3] = v.w;
    }
}

// ---------------------------------------------------------------------------------------------
// The K loop on registers.  w holds the A operands of one K step: 16 B per feature block, straight from the packed
// [K/4][N][4] blob (lane half h reads quad row 2 s + h, column 32 fb + c).
// ---------------------------------------------------------------------------------------------
// (the block offset fb * 512 travels in the per-lane address -- 8 loop-invariant VGPRs -- so that an unrolled layer needs
// one scalar offset per K step; with it on the scalar side the compiler materialises STEPS x NFB offsets up front and
// spills them through VGPR lanes)
struct LaneOfs {
    uint32_t v[8];
};
__device__ __forceinline__ LaneOfs lane_offsets(uint32_t wlane) {
    LaneOfs o;
#pragma unroll
    for (int fb = 0; fb < 8; ++fb) {
        o.v[fb] = wlane + fb * 512u;
        asm volatile("" : "+v"(o.v[fb]));  // opaque: keep eight registers instead of re-deriving the sums at every load
    }
    return o;
}
// Where the first operand fetch of whatever runs NEXT goes (issued once, in the last K step of a segment): its eight
// lane offsets are derived on the spot from one register -- a second LaneOfs kept through the layer loop pushes the loop's
// own offsets out to scratch, and a reload inside the K loop queues up in front of the operand loads.
struct NextOfs {
    uint32_t base;  // (h * N + c) * 16 of the next matrix
    bool paired;    // 128-wide layer in K-step pairs (blocks 4..7 = the second step's rows)
    uint32_t wstep; // its step stride (paired only)
};
template <int NFB>
__device__ __forceinline__ void load_w_next(float4 (&w)[8], __amdgpu_buffer_rsrc_t rsrc, const NextOfs& nx, uint32_t soff) {
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) {
        const uint32_t ofs = nx.paired ? (uint32_t)(fb & 3) * 512u + (uint32_t)(fb >> 2) * nx.wstep : (uint32_t)fb * 512u;  // (scalar)
        w[fb] = load_weight(rsrc, nx.base + ofs, soff);
    }
}
template <int NFB>
__device__ __forceinline__ void load_w(float4 (&w)[8], __amdgpu_buffer_rsrc_t rsrc, const LaneOfs& wl, uint32_t soff) {
#ifdef STNERF_WAVE_EXP_NOLOADW  /* development experiment: wrong results, isolates the cost of the operand loads */
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) asm volatile("" : "+v"(w[fb].x), "+v"(w[fb].y), "+v"(w[fb].z), "+v"(w[fb].w));
#else
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) w[fb] = load_weight(rsrc, wl.v[fb], soff);
#endif
}

// acc[fb] = this lane's 16 bias values of block fb: the C operand of the block's first MFMA (no accumulator
// initialisation, no bias add).  blane = 16 h bytes; register 4 q + r <-> feature 32 fb + 8 q + 4 h + r.
template <int NFB>
__device__ __forceinline__ void load_bias(f32x16 (&acc)[8], __amdgpu_buffer_rsrc_t rsrc, uint32_t blane, uint32_t boff) {
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b = load_weight(rsrc, blane, boff + (uint32_t)(fb * 32 + 8 * q) * 4u);
            acc[fb][4 * q + 0] = b.x;
            acc[fb][4 * q + 1] = b.y;
            acc[fb][4 * q + 2] = b.z;
            acc[fb][4 * q + 3] = b.w;
        }
}

// STEPS K steps whose B operands are blk[s >> 2][4 (s & 3) + kk].  The A operands ping-pong between wa and wb: step s
// reads buffer (s + PAR) & 1, which holds its weights on entry of the step; the loads of step s + 1 -- for the last
// step: NFB_NEXT blocks at (next_wlane, next_soff), the first step of whatever runs next -- are issued in front of the
// MFMAs of step s, one behind each of the first MFMAs (see mma_segment in mlp_blocks.h for the scheduling notes).
// The 4 * NFB MFMAs of one K step (B operands b0..b3 = the four k of this lane half), with the NL operand loads the caller
// has just issued for the following step pinned one behind each of the first MFMAs.
template <int NFB, int NL>
__device__ __forceinline__ void step_r(f32x16 (&acc)[8], const float4 (&wc)[8], float b0, float b1, float b2, float b3) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const float bv = kk == 0 ? b0 : kk == 1 ? b1 : kk == 2 ? b2 : b3;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) {
            const float wv = kk == 0 ? wc[fb].x : kk == 1 ? wc[fb].y : kk == 2 ? wc[fb].z : wc[fb].w;
            acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, bv, acc[fb], 0, 0, 0);
        }
    }
    // (spreading the loads over the whole step -- one behind every 2nd .. 4th MFMA -- measures the same)
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NFB - NL, 0);
    __builtin_amdgcn_sched_barrier(0);
}

template <int NFB, int NBLK, int STEPS, int PAR, int NFB_NEXT>
__device__ __forceinline__ void segment_r(f32x16 (&acc)[8], const f32x16 (&blk)[NBLK], float4 (&wa)[8], float4 (&wb)[8],
                                          __amdgpu_buffer_rsrc_t rsrc, const LaneOfs& wlane, uint32_t soff, uint32_t wstep,
                                          const NextOfs& next_wlane, uint32_t next_soff) {
    static_assert(STEPS >= 1 && STEPS <= 4 * NBLK, "segment_r: not enough input blocks");
    static_assert(NFB_NEXT <= 4 * NFB, "segment_r: more operand loads than MFMAs to hide them behind");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s + 1 < STEPS; ++s) {
        float4 (&wc)[8] = ((s + PAR) & 1) ? wb : wa;
        float4 (&wn)[8] = ((s + PAR) & 1) ? wa : wb;
        load_w<NFB>(wn, rsrc, wlane, soff + (uint32_t)(s + 1) * wstep);
        step_r<NFB, NFB>(acc, wc, blk[s >> 2][4 * (s & 3) + 0], blk[s >> 2][4 * (s & 3) + 1], blk[s >> 2][4 * (s & 3) + 2],
                         blk[s >> 2][4 * (s & 3) + 3]);
    }
    {
        constexpr int s = STEPS - 1;
        float4 (&wc)[8] = ((s + PAR) & 1) ? wb : wa;
        float4 (&wn)[8] = ((s + PAR) & 1) ? wa : wb;
        load_w_next<NFB_NEXT>(wn, rsrc, next_wlane, next_soff);
        step_r<NFB, NFB_NEXT>(acc, wc, blk[s >> 2][4 * (s & 3) + 0], blk[s >> 2][4 * (s & 3) + 1], blk[s >> 2][4 * (s & 3) + 2],
                              blk[s >> 2][4 * (s & 3) + 3]);
    }
}

// 128-wide layers (4 feature blocks: 16 MFMAs = 1024 cycles per K step) in PAIRS of K steps: one operand fetch of 8 x 16 B
// per lane covers two steps (blocks 0..3: step 2 m, blocks 4..7: step 2 m + 1 -- the lane offsets of `wlp` carry the
// extra row pair), so the fetch runs 2048 cycles ahead of its use like in the 256-wide layers instead of 1024 (the L2
// latency under load is of that order: the single-step form lost 6 .. 11 % of these layers' MFMA time to operand
// waits).  Per accumulator the MFMAs come in the same order as in the single-step form.
__device__ __forceinline__ LaneOfs lane_offsets_paired(uint32_t wlane, uint32_t wstep) {
    LaneOfs o;
#pragma unroll
    for (int fb = 0; fb < 8; ++fb) {
        o.v[fb] = wlane + (fb & 3) * 512u + (fb >> 2) * wstep;
        asm volatile("" : "+v"(o.v[fb]));
    }
    return o;
}
template <int NL, bool BOTH>
__device__ __forceinline__ void pair_r(f32x16 (&acc)[8], const float4 (&wc)[8], const f32x16& b0, int r0, const f32x16& b1, int r1) {
#pragma unroll
    for (int half = 0; half < (BOTH ? 2 : 1); ++half) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float bv = half == 0 ? b0[r0 + kk] : b1[r1 + kk];
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
                const float4& w4 = wc[4 * half + fb];
                const float wv = kk == 0 ? w4.x : kk == 1 ? w4.y : kk == 2 ? w4.z : w4.w;
                acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, bv, acc[fb], 0, 0, 0);
            }
        }
    }
    constexpr int NM = BOTH ? 32 : 16;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NM - NL, 0);
    __builtin_amdgcn_sched_barrier(0);
}
// STEPS K steps (an odd count ends on a half pair); pair m reads buffer (m + PAR) & 1; on entry that buffer holds pair 0;
// the last pair fetches NFB_NEXT blocks at (next_wlane, next_soff) into the other buffer.
template <int NBLK, int STEPS, int PAR, int NFB_NEXT>
__device__ __forceinline__ void segment_p(f32x16 (&acc)[8], const f32x16 (&blk)[NBLK], float4 (&wa)[8], float4 (&wb)[8],
                                          __amdgpu_buffer_rsrc_t rsrc, const LaneOfs& wlp, uint32_t soff, uint32_t wstep,
                                          const NextOfs& next_wlane, uint32_t next_soff) {
    static_assert(STEPS >= 1 && STEPS <= 4 * NBLK, "segment_p: not enough input blocks");
    constexpr int PAIRS = (STEPS + 1) / 2;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m + 1 < PAIRS; ++m) {
        float4 (&wc)[8] = ((m + PAR) & 1) ? wb : wa;
        float4 (&wn)[8] = ((m + PAR) & 1) ? wa : wb;
        load_w<8>(wn, rsrc, wlp, soff + (uint32_t)(m + 1) * 2u * wstep);
        pair_r<8, true>(acc, wc, blk[(2 * m) >> 2], 4 * ((2 * m) & 3), blk[(2 * m + 1) >> 2], 4 * ((2 * m + 1) & 3));
    }
    {
        constexpr int m = PAIRS - 1;
        constexpr bool both = (STEPS & 1) == 0;
        float4 (&wc)[8] = ((m + PAR) & 1) ? wb : wa;
        float4 (&wn)[8] = ((m + PAR) & 1) ? wa : wb;
        load_w_next<NFB_NEXT>(wn, rsrc, next_wlane, next_soff);
        pair_r<NFB_NEXT, both>(acc, wc, blk[(2 * m) >> 2], 4 * ((2 * m) & 3), blk[both ? (2 * m + 1) >> 2 : (2 * m) >> 2],
                               both ? 4 * ((2 * m + 1) & 3) : 0);
    }
}

// Biases through LDS.  Fetched from the blob at the layer boundary, a layer's bias costs one exposed L2 round trip
// (~900 cycles: the accumulators it goes into are busy until the previous layer's ReLU pass, and there are no registers
// to park it in) -- measured 1.6 .. 2.0 k cycles of overhead per layer whatever its size.  Instead every bias vector of
// the network is copied once, at the start of the network, into the wave's private LDS window (one 16-byte load per
// lane and vector, behind the encoding arithmetic); the layer boundary reads it back with ds_read_b128 (both lanes
// halves broadcast), ~100 cycles ahead of its first use.
template <int NB>
struct BiasStage {
    float4 v[NB];
};
template <int NB>
__device__ __forceinline__ void stage_bias_issue(BiasStage<NB>& st, __amdgpu_buffer_rsrc_t rsrc, int lane, const uint32_t (&boff)[NB],
                                                 int last_lanes = 64) {
    // floats 4 lane .. of vector i; the LAST vector may end with the blob: lanes >= last_lanes re-read its start
#pragma unroll
    for (int i = 0; i < NB; ++i)
        st.v[i] = load_weight(rsrc, (uint32_t)((i + 1 == NB && lane >= last_lanes) ? 0 : lane) * 16u, boff[i]);
}
template <int NB>
__device__ __forceinline__ void stage_bias_store(const BiasStage<NB>& st, float* biasw, int lane) {
#pragma unroll
    for (int i = 0; i < NB; ++i) reinterpret_cast<float4*>(biasw + i * 256)[lane] = st.v[i];
    wave_lds_sync();
}

// Layer boundary: in = relu(acc) (one v_max_i32 per value); as soon as a block is consumed, the bias of the NEXT
// layer's block (next_bias = that vector in the wave's LDS window, nullptr-free: NFB_NEXT = 0 after the last layer) is
// read into the freed accumulator registers.  Register 4 q + r of block fb <-> feature 32 fb + 8 q + 4 h + r.
// Layer loops are ROTATED around this pass -- { relu_rebias of the previous layer; K segment } -- so that what crosses the
// loop's back edge is the MFMAs' own output.  With the pass at the END of the body the freshly loaded bias crosses it,
// and the register allocator parks 76 of the 128 values in arch VGPRs to copy them into the accumulators at the loop
// head: 76 v_accvgpr_write per layer on the matrix pipe's time (-0.3 % for the whole kernel).  The barrier between a
// block's reads and its bias load keeps the ds_read from being hoisted (= from needing other registers than the ones
// just read).
template <int NFB, int NFB_NEXT>
__device__ __forceinline__ void relu_rebias(f32x16 (&acc)[8], f32x16 (&in)[8], const float* next_bias, int lane) {
    const float4* nb4 = reinterpret_cast<const float4*>(next_bias) + (lane >> 5);
#pragma unroll
    for (int fb = 0; fb < (NFB > NFB_NEXT ? NFB : NFB_NEXT); ++fb) {
        if (fb < NFB) {
#pragma unroll
            for (int i = 0; i < 16; ++i) in[fb][i] = relu_bits(acc[fb][i]);
        }
        __builtin_amdgcn_sched_barrier(0);  // (the bias goes into the registers just read: no earlier)
        if (fb < NFB_NEXT) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b = nb4[fb * 8 + 2 * q];
                acc[fb][4 * q + 0] = b.x;
                acc[fb][4 * q + 1] = b.y;
                acc[fb][4 * q + 2] = b.z;
                acc[fb][4 * q + 3] = b.w;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

#ifdef STNERF_WAVE_DEBUG
// development: this lane's registers of `nblk` blocks -> dbg[row][256] in feature order (register 4 q + r of block fb
// <-> feature 32 fb + 8 q + 4 h + r)
struct WaveDbg {
    float* buf;
    int stage;
    int64_t row;   // row of this lane's sample inside its layer, -1: do not dump
};
template <int NBLK>
__device__ __forceinline__ void dbg_dump(const WaveDbg& d, int stage, const f32x16 (&blk)[NBLK], int nblk, int lane) {
    if (!d.buf || d.stage != stage || d.row < 0) return;
    const int h = lane >> 5;
    for (int fb = 0; fb < nblk; ++fb)
        for (int i = 0; i < 16; ++i) d.buf[d.row * 256 + fb * 32 + 8 * (i >> 2) + 4 * h + (i & 3)] = blk[fb][i];
}
#define WV_DBG_PARAM , const WaveDbg& dbg
#define WV_DBG_ARG , dbg
#define WV_DBG(stage, blk, nblk) dbg_dump(dbg, stage, blk, nblk, lane)
#else
#define WV_DBG_PARAM
#define WV_DBG_ARG
#define WV_DBG(stage, blk, nblk) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// Heads.  The weights come out of the wave's LDS window (staged there with the biases), the fma chains run on registers,
// and the two lanes of a sample swap their chains with v_permlane32_swap.  (Written as load from the blob -> use ->
// ds_bpermute per part, a head is a chain of 4 .. 12 dependent L2 round trips plus as many LDS-crossbar round trips with
// nothing else for the single wave of the SIMD to do: measured 3.6 % of the kernel for 0.2 % worth of arithmetic.
// Fetched from the blob into registers EARLY instead, the weights hold 64 .. 192 registers across a layer and the
// allocator shuffles ~200 values between the register files around them: +0.45 .. 0.75 % for the LDS form.)
// ---------------------------------------------------------------------------------------------
// sigma head (256 -> 1) in the grouping of head_partial<TM, 1> with four parts of 16 quads: part pp, chain u runs over
// the quads 16 pp + u + 4 m, m = 0..3, four fmas each; S_pp = (c0 + c1) + (c2 + c3); sigma = (((b + S_0) + S_1) + S_2) + S_3.
// This lane holds the quads 2 s + h: its chains are u = h (s = 8 pp + 2 m) and u = h + 2 (s = 8 pp + 2 m + 1).
// The head is written in two halves (parts 0, 1 over the feature blocks 0..3, parts 2, 3 over blocks 4..7).
// sigma_in = the bias (half 0) or the result of half 0 (half 1)
template <int HALF>
__device__ __forceinline__ float head_sigma(const f32x16 (&in)[8], const float* wlds /* the 256 weights */, float sigma_in, int lane) {
    const float4* w4 = reinterpret_cast<const float4*>(wlds) + (lane >> 5);  // quad 2 s + h
    float ca[2], cb[2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        const int pp = 2 * HALF + pl;
        ca[pl] = cb[pl] = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int sa = 8 * pp + 2 * m, sb = sa + 1;
            const float4 wa4 = w4[2 * sa], wb4 = w4[2 * sb];
            ca[pl] = fmaf(in[sa >> 2][4 * (sa & 3) + 0], wa4.x, ca[pl]);
            ca[pl] = fmaf(in[sa >> 2][4 * (sa & 3) + 1], wa4.y, ca[pl]);
            ca[pl] = fmaf(in[sa >> 2][4 * (sa & 3) + 2], wa4.z, ca[pl]);
            ca[pl] = fmaf(in[sa >> 2][4 * (sa & 3) + 3], wa4.w, ca[pl]);
            cb[pl] = fmaf(in[sb >> 2][4 * (sb & 3) + 0], wb4.x, cb[pl]);
            cb[pl] = fmaf(in[sb >> 2][4 * (sb & 3) + 1], wb4.y, cb[pl]);
            cb[pl] = fmaf(in[sb >> 2][4 * (sb & 3) + 2], wb4.z, cb[pl]);
            cb[pl] = fmaf(in[sb >> 2][4 * (sb & 3) + 3], wb4.w, cb[pl]);
        }
    }
    float sigma = sigma_in;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) sigma += pair_sum(ca[pl], cb[pl]);
    return sigma;
}

// 128 -> 3 head (rgb_net's last layer, MotionNet's flow) in the grouping of head_partial<TM, 3> with four parts of 8
// quads: part pp, chain u over the quads 8 pp + u + 4 m, m = 0, 1.  Weights [3][128] in the LDS window, bias b3.
__device__ __forceinline__ void head3(const f32x16 (&in)[8], const float* wlds /* the 3 x 128 weights */, const float* __restrict__ b3,
                                      int lane, float (&out)[3]) {
    const float4* w4 = reinterpret_cast<const float4*>(wlds) + (lane >> 5);
    float ca[4][3], cb[4][3];
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            ca[pp][o] = cb[pp][o] = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int sa = 4 * pp + 2 * m, sb = sa + 1;
                const float4 wa4 = w4[o * 32 + 2 * sa], wb4 = w4[o * 32 + 2 * sb];
                ca[pp][o] = fmaf(in[sa >> 2][4 * (sa & 3) + 0], wa4.x, ca[pp][o]);
                ca[pp][o] = fmaf(in[sa >> 2][4 * (sa & 3) + 1], wa4.y, ca[pp][o]);
                ca[pp][o] = fmaf(in[sa >> 2][4 * (sa & 3) + 2], wa4.z, ca[pp][o]);
                ca[pp][o] = fmaf(in[sa >> 2][4 * (sa & 3) + 3], wa4.w, ca[pp][o]);
                cb[pp][o] = fmaf(in[sb >> 2][4 * (sb & 3) + 0], wb4.x, cb[pp][o]);
                cb[pp][o] = fmaf(in[sb >> 2][4 * (sb & 3) + 1], wb4.y, cb[pp][o]);
                cb[pp][o] = fmaf(in[sb >> 2][4 * (sb & 3) + 2], wb4.z, cb[pp][o]);
                cb[pp][o] = fmaf(in[sb >> 2][4 * (sb & 3) + 3], wb4.w, cb[pp][o]);
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        out[o] = b3[o];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) out[o] += pair_sum(ca[pp][o], cb[pp][o]);
    }
}

// ---------------------------------------------------------------------------------------------
// SpaceNet on the wave's 32 samples (point p of sample c in both lanes of the sample; `ray` = the sample's row of the
// layer's ray-bias table: rgb_net.1's bias + direction / time columns, mlp_raybias.hip).  Returns {r, g, b, sigma} (raw)
// in every lane.  `mid` is called once, in front of the sigma head's chains -- a stretch of ~1k cycles of vector
// arithmetic without a memory wait, where the caller issues the HBM loads of the next work item (the counters are in
// order: a load issued elsewhere stalls the next weight wait for its whole latency).
// ---------------------------------------------------------------------------------------------
// `tap` sees every layer's input as the layer reads it (training: csrc/mlp_wave.hip's StoreTap writes them out for the backward
// pass; NoTap compiles to nothing): stage TAP_PE = PE(pos) (2 blocks), 0 .. 5 = the inputs of stage1.2 .. stage2.4 (8 blocks each;
// 3 = h4, stage2.0's input next to PE), 6 = g3 (the heads' and rgb_net.1's input), 7 = rgb_net.3's input (4 blocks).
struct NoTap {
    template <int NBLK>
    __device__ __forceinline__ void blocks(int, const f32x16 (&)[NBLK], int, int) const {}
    __device__ __forceinline__ void flow(const float (&)[3], int) const {}
};
template <bool DEEP, class Mid, class Tap = NoTap>
__device__ __forceinline__ float4 space_wave(const float* net, const bool use_time, float* encw, const float (&p)[3],
                                             const float* __restrict__ raybias, int32_t ray, int lane, f32x16 (&acc)[8],
                                             f32x16 (&in)[8], float4 (&wa)[8], float4 (&wb)[8], Mid mid WV_DBG_PARAM WP_PARAM,
                                             const Tap& tap = Tap()) {
    const SpaceLayout L = space_layout(use_time, DEEP);
    const __amdgpu_buffer_rsrc_t rsrc = weight_rsrc(net);
    const int h = lane >> 5, c = lane & 31;
    constexpr uint32_t WSTEP256 = 2u * 256u * 16u, WSTEP128 = 2u * 128u * 16u;
    const LaneOfs wl256 = lane_offsets((uint32_t)(h * 256 + c) * 16u);
    const NextOfs nx256{(uint32_t)(h * 256 + c) * 16u, false, 0u}, nx128p{(uint32_t)(h * 128 + c) * 16u, true, WSTEP128};
    const uint32_t blane = (uint32_t)h * 16u;
    // ---- every later bias vector on its way into the wave's LDS window: slots 0..5 = stage1.2 .. stage2.4, 6 / 7 = the
    // deep_rgb layers (rgb_net.1 has none: its C operand is the ray's row of `raybias`); then stage1.0's bias + first
    // weights, all in flight behind the encoding arithmetic
    float* biasw = encw + WV_ENC_FLOATS;
    constexpr int NBS = DEEP ? 8 : 6;
    BiasStage<NBS> bst;
    {
        uint32_t bo[NBS];
#pragma unroll
        for (int i = 0; i < 6; ++i) bo[i] = (uint32_t)L.b[i + 1] * 4u;
        if constexpr (DEEP) {
            bo[6] = (uint32_t)L.b_deep[0] * 4u;
            bo[7] = (uint32_t)L.b_deep[1] * 4u;
        }
        stage_bias_issue<NBS>(bst, rsrc, lane, bo);
    }
    load_bias<8>(acc, rsrc, blane, (uint32_t)L.b[0] * 4u);
    load_w<8>(wa, rsrc, wl256, (uint32_t)L.w[0] * 4u);
    f32x16 pe[2];
    encode_pos(encw, lane, p);
    wave_lds_sync();
    read_enc_blocks<2, 8>(encw, lane, pe);
    wave_lds_sync();
    stage_bias_store<NBS>(bst, biasw, lane);
    // the head weights follow through slots 8 (density_net.0) and 9 / 10 (the colour head), in flight behind stage1.0
    // (whose input is the encoding: the eight activation blocks are not live yet, registers to spare)
    BiasStage<3> hst;
    {
        const uint32_t ho[3] = {(uint32_t)L.w_sigma * 4u, (uint32_t)L.w_rgb2 * 4u,
                                (uint32_t)L.w_rgb2 * 4u + 1024u};  // floats 256..383 of the 3 x 128 head: 32 lanes, then the blob ends
        stage_bias_issue<3>(hst, rsrc, lane, ho, 32);
    }
    WV_DBG(100, pe, 2);
    tap.template blocks<2>(TAP_PE, pe, 2, lane);
    WP(WP_S_PE);
    segment_r<8, 2, 8, 0, 8>(acc, pe, wa, wb, rsrc, wl256, (uint32_t)L.w[0] * 4u, WSTEP256, nx256, (uint32_t)L.w[1] * 4u);
    stage_bias_store<3>(hst, biasw + WV_HEAD_SLOT * 256, lane);
    WP(WP_S_L0);
    // ---- stage1.2 .. stage2.4: six 256-wide layers, stage2.0 (li == 4) with the PE(pos) skip segment behind its 256
    // features (modeling/spacenet.py:45-57,136-138)
    uint32_t soff = (uint32_t)L.w[1] * 4u;
#pragma unroll 1
    for (int li = 1; li <= 6; ++li) {
        const uint32_t kq = li == 4 ? 80u : 64u;
        const uint32_t boff = soff + kq * 4096u;           // this layer's bias
        const uint32_t after = boff + 1024u;               // the next layer's weights (stage2.4: density_net follows)
        const uint32_t next_w = li == 6 ? (uint32_t)L.w_rgb1 * 4u : after;
        const NextOfs next_wl{li == 6 ? nx128p.base : nx256.base, li == 6, WSTEP128};  // (rgb_net.1 runs in step pairs)
        // the previous layer's ReLU, this layer's bias (rotated: see relu_rebias)
        relu_rebias<8, 8>(acc, in, biasw + (li - 1) * 256, lane);
        WV_DBG(li - 1, in, 8);
        tap.template blocks<8>(li - 1, in, 8, lane);
        WP(WP_L_EPI);
        // (one copy of the 32-step body: behind stage2.0's 256 features the skip segment simply continues in the blob)
        segment_r<8, 8, 32, 0, 8>(acc, in, wa, wb, rsrc, wl256, soff, WSTEP256, next_wl,  // (li == 4: next_wl == wl256)
                                  li == 4 ? soff + 32u * WSTEP256 : next_w);
        if (li == 4) segment_r<8, 2, 8, 0, 8>(acc, pe, wa, wb, rsrc, wl256, soff + 32u * WSTEP256, WSTEP256, next_wl, next_w);
        WP(WP_L_SEG);
        soff = after;
    }
    relu_rebias<8, 0>(acc, in, biasw, lane);
    WV_DBG(6, in, 8);
    tap.template blocks<8>(6, in, 8, lane);
    WP(WP_L_EPI);
    // ---- rgb_net.1's C operands: this sample's row of the ray-bias table straight into the accumulators.  (Fetching them
    // inside the preceding ReLU pass, block by block, would cover their latency, but keeps the row pointer live through
    // the layer loop -- the loop's lane offsets then go to scratch and every phase slows down.)
    {
        const float* row = raybias + (int64_t)ray * 128 + 4 * h;
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(row + fb * 32 + 8 * q);
                acc[fb][4 * q + 0] = v.x;
                acc[fb][4 * q + 1] = v.y;
                acc[fb][4 * q + 2] = v.z;
                acc[fb][4 * q + 3] = v.w;
            }
    }
    // ---- sigma = density_net(h) (:139), raw, then rgb_net: relu -> Linear(283|304, 128) -> relu -> Linear(128, 3)
    // (:80-86); h is already >= 0, and only the 256 backbone columns are left of the first layer.  The next item's HBM
    // loads go out first (`mid`); the sigma chains (weights from the LDS window) cover them and the C-operand fetch above.
    const float b_sigma = net[L.b_sigma];
    const uint32_t wr = (uint32_t)L.w_rgb1 * 4u;
    const LaneOfs wl128p = lane_offsets_paired(nx128p.base, WSTEP128);  // (not live through the layer loop)
    mid();
    float sigma = head_sigma<0>(in, biasw + WV_HEAD_SLOT * 256, b_sigma, lane);
    sigma = head_sigma<1>(in, biasw + WV_HEAD_SLOT * 256, sigma, lane);
    WP(WP_S_MID);
    segment_p<4, 16, 0, 8>(acc, reinterpret_cast<const f32x16 (&)[4]>(in[0]), wa, wb, rsrc, wl128p, wr, WSTEP128, nx128p,
                           wr + 16u * WSTEP128);
    const uint32_t w_after = DEEP ? (uint32_t)L.w_deep[0] * 4u : wr;  // (not deep: nothing follows; the fetch is discarded)
    segment_p<4, 16, 0, 8>(acc, reinterpret_cast<const f32x16 (&)[4]>(in[4]), wa, wb, rsrc, wl128p, wr + 16u * WSTEP128, WSTEP128,
                           nx128p, w_after);
    if constexpr (DEEP) {  // deep_rgb (:68-79): two more 128-wide hidden layers
        relu_rebias<4, 4>(acc, in, biasw + 6 * 256, lane);
        segment_p<8, 16, 0, 8>(acc, in, wa, wb, rsrc, wl128p, (uint32_t)L.w_deep[0] * 4u, WSTEP128, nx128p, (uint32_t)L.w_deep[1] * 4u);
        relu_rebias<4, 4>(acc, in, biasw + 7 * 256, lane);
        segment_p<8, 16, 0, 8>(acc, in, wa, wb, rsrc, wl128p, (uint32_t)L.w_deep[1] * 4u, WSTEP128, nx128p, (uint32_t)L.w_deep[1] * 4u);
    }
    WP(WP_S_RGB1);
    relu_rebias<4, 0>(acc, in, biasw, lane);
    WV_DBG(7, in, 4);
    tap.template blocks<8>(7, in, 4, lane);
    float rgb[3];
    head3(in, biasw + (WV_HEAD_SLOT + 1) * 256, net + L.b_rgb2, lane, rgb);
    WP(WP_S_HEAD);
    return make_float4(rgb[0], rgb[1], rgb[2], sigma);
}

// ---------------------------------------------------------------------------------------------
// MotionNet on the wave's 32 samples: p += flow (modeling/layered_rfrender.py:356,510).
// ---------------------------------------------------------------------------------------------
// `tap` (training, csrc/train_wave.hip): stage TAP_PE = the staged encoding (3 blocks, 11 K steps), 0 .. 4 = the post-ReLU outputs of
// motion_net.0 .. .8 (4 blocks), and the flow itself.
template <class Tap = NoTap>
__device__ __forceinline__ void motion_wave(const float* net, float* encw, float (&p)[3], float tv, int flags, int lane,
                                            f32x16 (&acc)[8], f32x16 (&in)[8], float4 (&wa)[8], float4 (&wb)[8] WV_DBG_PARAM WP_PARAM,
                                            const Tap& tap = Tap()) {
    const MotionLayout L = motion_layout();
    const __amdgpu_buffer_rsrc_t rsrc = weight_rsrc(net);
    const int h = lane >> 5, c = lane & 31;
    constexpr uint32_t WSTEP128 = 2u * 128u * 16u;
    const LaneOfs wl128p = lane_offsets_paired((uint32_t)(h * 128 + c) * 16u, WSTEP128);
    const NextOfs nx128p{(uint32_t)(h * 128 + c) * 16u, true, WSTEP128};
    const uint32_t blane = (uint32_t)h * 16u;
    float* biasw = encw + WV_ENC_FLOATS;
    BiasStage<4> bst;  // motion_net.2 .. .8 -> slots 0..3 of the wave's LDS window
    {
        uint32_t bo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bo[i] = (uint32_t)L.b[i + 1] * 4u;
        stage_bias_issue<4>(bst, rsrc, lane, bo);
    }
    load_bias<4>(acc, rsrc, blane, (uint32_t)L.b[0] * 4u);
    load_w<8>(wa, rsrc, wl128p, (uint32_t)L.w[0] * 4u);
    f32x16 me[3];
    encode_motion(encw, lane, p, tv, flags);
    wave_lds_sync();
    read_enc_blocks<3, 11>(encw, lane, me);
    wave_lds_sync();
    stage_bias_store<4>(bst, biasw, lane);
    BiasStage<2> hst;  // the flow head's 3 x 128 weights -> slots 4, 5, in flight behind motion_net.0
    {
        const uint32_t ho[2] = {(uint32_t)L.w_out * 4u, (uint32_t)L.w_out * 4u + 1024u};  // (32 lanes, then the blob ends)
        stage_bias_issue<2>(hst, rsrc, lane, ho, 32);
    }
    WV_DBG(199, me, 3);
    tap.template blocks<3>(TAP_PE, me, 3, lane);
    WP(WP_M_ENC);
    // motion_net.0: 11 K steps (22 quads) = 5 step pairs + a half pair (every layer here runs in step pairs)
    segment_p<3, 11, 0, 8>(acc, me, wa, wb, rsrc, wl128p, (uint32_t)L.w[0] * 4u, WSTEP128, nx128p, (uint32_t)L.w[1] * 4u);
    stage_bias_store<2>(hst, biasw + 4 * 256, lane);
#pragma unroll 1
    for (int li = 1; li <= 4; ++li) {
        relu_rebias<4, 4>(acc, in, biasw + (li - 1) * 256, lane);  // the previous layer's ReLU, this layer's bias
        WV_DBG(199 + li, in, 4);
        tap.template blocks<8>(li - 1, in, 4, lane);
        const uint32_t soff = (uint32_t)L.w[1] * 4u + (uint32_t)(li - 1) * (32u * 128u * 16u + 512u);
        const uint32_t next_w = soff + 32u * 128u * 16u + 512u;
        // (behind motion_net.8 nothing follows: the fetch is discarded, so it re-reads this layer's first rows -- a full
        // operand fetch at the output layer's 1.5 KB would run past the end of the blob.  6 and 8 pairs per layer: every
        // layer starts in wa)
        segment_p<8, 16, 0, 8>(acc, in, wa, wb, rsrc, wl128p, soff, WSTEP128, nx128p, li < 4 ? next_w : soff);
    }
    relu_rebias<4, 0>(acc, in, biasw, lane);
    WV_DBG(204, in, 4);
    tap.template blocks<8>(4, in, 4, lane);
    WP(WP_M_LAYERS);
    float fl[3];
    head3(in, biasw + 4 * 256, net + L.b_out, lane, fl);
    tap.flow(fl, lane);
#pragma unroll
    for (int c3 = 0; c3 < 3; ++c3) p[c3] = p[c3] + fl[c3];
    WP(WP_M_HEAD);
}

}  // namespace stnerf
