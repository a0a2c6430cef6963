// One persistent launch per network stage of the pipeline (coarse or fine): every shown layer's MotionNet + SpaceNet
// evaluation of modeling/layered_rfrender.py:340-418 / :495-576 behind ONE kernel.
//
//   * Work queue.  The stage is cut into items: 128 consecutive sample rows of a layer without deformation (the
//     background), 256 rows of a layer with a MotionNet.  One workgroup per CU (8 waves, all 160 KiB of LDS) pops
//     items from a global atomic counter until the queue is dry: no launch per (layer, network), no per-launch tail
//     (a static grid-stride split leaves CUs idle for up to one workgroup's worth of tiles at the end of each of the
//     10 launches of a stage), and the item -> layer map is computed on the device from the compacted ray counts,
//     so there is still no host synchronisation.
//   * MotionNet fused in front of its SpaceNet.  A 256-row item runs the deformation net on all 256 samples at
//     once (activations [32 quads][256] = the 128 KiB `act` region; with 8 waves each wave then owns 32 features
//     x 128 samples, the same 16-MFMA-per-step shape as the 256-wide SpaceNet layers), keeps the deformed points
//     on chip and evaluates the SpaceNet on the two 128-row halves.  The flow never goes to HBM (the separate
//     launches wrote xyz += flow and read it back: 24 B per evaluation) and the deformed points are not written at all.
//   * Same arithmetic as the per-network kernels of mlp.hip (same dense_layer / K order / heads): outputs are
//     bit-identical to stnerf_motionnet_fwd followed by stnerf_spacenet_fwd (tests/test_gpu_ops.py).
//
// Reference: modeling/spacenet.py:16-160, modeling/motion_net.py:7-71, modeling/layered_rfrender.py:340-418,495-576.
#include <stdlib.h>
#include <string.h>

#include "mlp_stage.h"

#ifndef STNERF_STAGE_KERNEL_DEFAULT
#define STNERF_STAGE_KERNEL_DEFAULT STAGE_KERNEL_WAVE
#endif

namespace stnerf {

constexpr int ST_TM = 128;        // SpaceNet tile (samples)
constexpr int ST_MTM = 256;       // MotionNet tile = two SpaceNet tiles
constexpr int ST_NW = 8;
constexpr int ST_THREADS = ST_NW * 64;
constexpr int ST_LDS = (64 + 16) * ST_TM * 16;   // act [64][128] float4 + enc [16][128] float4 = 160 KiB

// First layer of whatever network runs next on this workgroup (its step-0 weights and bias are fetched during the
// current network's last layer).
struct NextFirst {
    const float* base;
    int64_t w_off, b_off;
    int n_total;   // 256: a SpaceNet's stage1.0 (8 feature slices), 128: a MotionNet's layer 0 (4 slices)
};
__device__ __forceinline__ const float4* next_w_ptr(const NextFirst& nf, int wave, int lane) {
    const int n0 = nf.n_total == 256 ? (wave & 7) * 32 : (wave & 3) * 32;
    return weight_lane_ptr(nf.base, nf.w_off, nf.n_total, n0, lane);
}
__device__ __forceinline__ const float* next_b_ptr(const NextFirst& nf, int wave, int lane) {
    const int n0 = nf.n_total == 256 ? (wave & 7) * 32 : (wave & 3) * 32;
    return nf.base + nf.b_off + n0 + 4 * (lane >> 5);
}

using WF = WFrag<1>;  // every wave split used here has one 32-feature block per wave
static_assert(WaveSplit<ST_TM, ST_NW, 256>::NFB == 1 && WaveSplit<ST_TM, ST_NW, 128>::NFB == 1 &&
              WaveSplit<ST_MTM, ST_NW, 128>::NFB == 1, "stage kernel assumes one feature block per wave");

// ---------------------------------------------------------------------------------------------
// SpaceNet on one 128-sample tile whose (already deformed) point p sits in registers; ray_idx = the sample's row of the
// layer's ray-bias table (rgb_net.1's bias + direction / time columns, mlp_raybias.hip).
// wA: step-0 weights + bias of stage1.0 (prefetched); on return wA holds those of `nf`.  Part 0 of every valid
// sample returns its {r,g,b,sigma}.  Body = spacenet_kernel<128, 8, USE_TIME, DEEP> of mlp.hip with USE_TIME a
// (workgroup-uniform) run-time value: one instance serves the background and the performers.
// ---------------------------------------------------------------------------------------------
template <bool DEEP>
__device__ __forceinline__ float4 space_tile(const float* net_in, const bool USE_TIME, float4* act, float4* enc,
                                             const float (&p)[3], const float* raybias, int32_t ray_idx, int lane, int wave,
                                             int part, int s, WF& wA, const NextFirst& nf PH_PARAMS) {
    constexpr int TM = ST_TM, NW = ST_NW, NTHREADS = ST_THREADS, NPARTS = NTHREADS / TM;
    float* encf = reinterpret_cast<float*>(enc);
    float* scratch = reinterpret_cast<float*>(enc + 12 * TM);
    const SpaceLayout L = space_layout(USE_TIME, DEEP);
    WF wB, wR, wR2;
    int64_t opaque_zero = 0;  // keeps the loop-invariant bias / head-weight loads inside the tile (see mlp.hip)
    asm volatile("" : "+s"(opaque_zero));
    const float* net = net_in + opaque_zero;
    float* col = encf + s * 4;
    // ---- PE_10(pos) -> enc quads 0..15; utils/dimension_kernel.py:8-33
    if (part == 0) {
#pragma unroll
        for (int dmn = 0; dmn < 3; ++dmn) ENC_AT(col, dmn) = p[dmn];
    }
    if (part == NPARTS - 1) ENC_AT(col, 63) = 0.f;
    for (int fq = part; fq < 10; fq += NPARTS) {
        const float freq = (float)(1 << fq);
#pragma unroll
        for (int dmn = 0; dmn < 3; ++dmn) {
            float sn, cs;
            sincos_pe(p[dmn] * freq, sn, cs);
            const int fs = 3 + fq * 6 + dmn, fc = fs + 3;
            ENC_AT(col, fs) = sn;
            ENC_AT(col, fc) = cs;
        }
    }
    __syncthreads();
    // ---- stage1 (modeling/spacenet.py:45-54)
    DENSE(TM, NW, 256, 256, net, L.w[0], L.b[0], enc, 16, nullptr, 0, act, wA, L.w[1], L.b[1], wB);
    __syncthreads();
    DENSE(TM, NW, 256, 256, net, L.w[1], L.b[1], act, 64, nullptr, 0, act, wB, L.w[2], L.b[2], wA);
    __syncthreads();
    DENSE(TM, NW, 256, 256, net, L.w[2], L.b[2], act, 64, nullptr, 0, act, wA, L.w[3], L.b[3], wB);
    __syncthreads();
    DENSE(TM, NW, 256, 256, net, L.w[3], L.b[3], act, 64, nullptr, 0, act, wB, L.w[4], L.b[4], wA);
    __syncthreads();
    // ---- stage2.0 on [h, PE(pos)] (:56-57, :137)
    DENSE(TM, NW, 256, 256, net, L.w[4], L.b[4], act, 64, enc, 16, act, wA, L.w[5], L.b[5], wB);
    // enc is free now; rgb_net.1's direction / time columns come with the ray's C operand (mlp_raybias.hip): the tile keeps
    // the ray of every sample for that fetch
    int32_t* ray_of = reinterpret_cast<int32_t*>(enc);
    if (part == 0) ray_of[s] = ray_idx;
    __syncthreads();
    DENSE(TM, NW, 256, 256, net, L.w[5], L.b[5], act, 64, nullptr, 0, act, wB, L.w[6], L.b[6], wA);
    __syncthreads();
    DENSE(TM, NW, 256, 128, net, L.w[6], L.b[6], act, 64, nullptr, 0, act, wA, L.w_rgb1, L.b_rgb1, wR);
    __syncthreads();
    // rgb_net.1's C operands on their way behind the sigma head
    RayC<1, WaveSplit<TM, NW, 128>::NSB> rayc;
    load_rayc(rayc, raybias, ray_of, WaveSplit<TM, NW, 128>::n0(wave), WaveSplit<TM, NW, 128>::sb0(wave), lane);
    // ---- sigma = density_net(h) (:139), raw
    float sigma;
    {
        float ps[1];
        head_partial<TM, 1>(act, s, part * (64 / NPARTS), (part + 1) * (64 / NPARTS), net + L.w_sigma, 256, ps);
        scratch[part * TM + s] = ps[0];
        __syncthreads();
        sigma = net[L.b_sigma];
#pragma unroll
        for (int pp = 0; pp < NPARTS; ++pp) sigma += scratch[pp * TM + s];
    }
    // ---- rgb_net (:80-86); the last layer prefetches the first layer of whatever runs next
    const float4* nwp = next_w_ptr(nf, wave, lane);
    const float* nbp = next_b_ptr(nf, wave, lane);
    if constexpr (!DEEP) {
        dense_layer<TM, 1, WaveSplit<TM, NW, 128>::NSB, true, 1, int>(net, L.w_rgb1, 0, 128, act, 64, nullptr, 0, act,
                                                                     WaveSplit<TM, NW, 128>::n0(wave), WaveSplit<TM, NW, 128>::sb0(wave),
                                                                     lane, wR, nwp, nbp, wA PH_ARGS, &rayc);
    } else {
        DENSE_RAYC(TM, NW, 128, 128, net, L.w_rgb1, act, 64, act, wR, L.w_deep[0], L.b_deep[0], wR2, rayc);
        __syncthreads();
        DENSE(TM, NW, 128, 128, net, L.w_deep[0], L.b_deep[0], act, 32, nullptr, 0, act, wR2, L.w_deep[1], L.b_deep[1], wR);
        __syncthreads();
        dense_layer<TM, 1, WaveSplit<TM, NW, 128>::NSB, true, 1>(net, L.w_deep[1], L.b_deep[1], 128, act, 32, nullptr, 0, act,
                                                                WaveSplit<TM, NW, 128>::n0(wave), WaveSplit<TM, NW, 128>::sb0(wave),
                                                                lane, wR, nwp, nbp, wA PH_ARGS);
    }
    __syncthreads();
    float4 o = make_float4(0.f, 0.f, 0.f, sigma);
    {
        float ps[3];
        head_partial<TM, 3>(act, s, part * (32 / NPARTS), (part + 1) * (32 / NPARTS), net + L.w_rgb2, 128, ps);
        float* sc = scratch + NTHREADS;
        sc[(part * 3 + 0) * TM + s] = ps[0];
        sc[(part * 3 + 1) * TM + s] = ps[1];
        sc[(part * 3 + 2) * TM + s] = ps[2];
        __syncthreads();
        if (part == 0) {
            o.x = net[L.b_rgb2 + 0];
            o.y = net[L.b_rgb2 + 1];
            o.z = net[L.b_rgb2 + 2];
#pragma unroll
            for (int pp = 0; pp < NPARTS; ++pp) {
                o.x += sc[(pp * 3 + 0) * TM + s];
                o.y += sc[(pp * 3 + 1) * TM + s];
                o.z += sc[(pp * 3 + 2) * TM + s];
            }
        }
    }
    __syncthreads();  // scratch / enc / act are rewritten by whatever comes next
    return o;
}

// ---------------------------------------------------------------------------------------------
// MotionNet on one 256-sample tile: p (sample point) and tv (frame id) of sample `s` in registers (both parts of
// the sample hold them).  On return threads of part 0 hold the deformed point in p; wA holds the step-0 weights
// of `nf`.  act = the 128 KiB region, scratch = 12 KiB elsewhere.  Body = motionnet_kernel<256, 8> of mlp.hip.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void motion_tile(const float* net_in, float4* act, float* scratch, float (&p)[3], float tv,
                                            int flags, int lane, int wave, int part, int s, WF& wA,
                                            const NextFirst& nf PH_PARAMS) {
    constexpr int TM = ST_MTM, NW = ST_NW, NTHREADS = ST_THREADS, NPARTS = NTHREADS / TM;
    float4* enc = act;  // [22][TM], aliases act: layer 0 is its only reader (see mlp.hip)
    float* encf = reinterpret_cast<float*>(enc);
    const MotionLayout L = motion_layout();
    WF wB;
    int64_t opaque_zero = 0;
    asm volatile("" : "+s"(opaque_zero));
    const float* net = net_in + opaque_zero;
    {
        const float lo = (flags & STNERF_MOTION_PLAIN_TIME) ? tv : floorf(tv);
        const float wgt = tv - lo;
        const bool frac = wgt != 0.f;
        const float om = 1.f - wgt;
        float* col = encf + s * 4;
        auto mix = [&](float va, float vb) { return frac ? om * va + wgt * vb : va; };
        if (part == 0) {
#pragma unroll
            for (int dmn = 0; dmn < 3; ++dmn) ENC_AT(col, dmn) = mix(p[dmn], p[dmn]);
            ENC_AT(col, 3) = mix(lo, lo + 1.f);
        }
        if (part == NPARTS - 1) {
#pragma unroll
            for (int f = 84; f < 88; ++f) ENC_AT(col, f) = 0.f;
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
                const int fs = 4 + fq * 8 + dmn, fc = fs + 4;
                ENC_AT(col, fs) = mix(sn, sn2);
                ENC_AT(col, fc) = mix(cs, cs2);
            }
        }
    }
    __syncthreads();
    DENSE(TM, NW, 128, 128, net, L.w[0], L.b[0], enc, 22, nullptr, 0, act, wA, L.w[1], L.b[1], wB);
    __syncthreads();
    DENSE(TM, NW, 128, 128, net, L.w[1], L.b[1], act, 32, nullptr, 0, act, wB, L.w[2], L.b[2], wA);
    __syncthreads();
    DENSE(TM, NW, 128, 128, net, L.w[2], L.b[2], act, 32, nullptr, 0, act, wA, L.w[3], L.b[3], wB);
    __syncthreads();
    DENSE(TM, NW, 128, 128, net, L.w[3], L.b[3], act, 32, nullptr, 0, act, wB, L.w[4], L.b[4], wA);
    __syncthreads();
    dense_layer<TM, 1, WaveSplit<TM, NW, 128>::NSB, true, 1>(net, L.w[4], L.b[4], 128, act, 32, nullptr, 0, act,
                                                            WaveSplit<TM, NW, 128>::n0(wave), WaveSplit<TM, NW, 128>::sb0(wave),
                                                            lane, wA, next_w_ptr(nf, wave, lane), next_b_ptr(nf, wave, lane),
                                                            wB PH_ARGS);
    wA = wB;
    __syncthreads();
    {
        // partial dot products in the grouping of the per-network kernel's default tile (motionnet_kernel<64, 4>: four
        // parts of 8 quads), so that the flow -- and everything downstream -- is bit-identical to that path
        float psa[3], psb[3];
        head_partial<TM, 3>(act, s, part * 16, part * 16 + 8, net + L.w_out, 128, psa);
        head_partial<TM, 3>(act, s, part * 16 + 8, part * 16 + 16, net + L.w_out, 128, psb);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            scratch[((2 * part + 0) * 3 + c) * TM + s] = psa[c];
            scratch[((2 * part + 1) * 3 + c) * TM + s] = psb[c];
        }
        __syncthreads();
        if (part == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float fl = net[L.b_out + c];
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) fl += scratch[(pp * 3 + c) * TM + s];
                p[c] = p[c] + fl;  // modeling/layered_rfrender.py:356,510
            }
        }
    }
    __syncthreads();
}

template <bool DEEP>
__global__ __launch_bounds__(ST_THREADS, 2) void mlp_stage_kernel(StageArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    float4* act = smem;                 // [64][128]
    float4* enc = smem + 64 * ST_TM;    // [16][128]
    float* encf = reinterpret_cast<float*>(enc);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    PH_DECL
    // ---- the queue: items of layer slot j are [pre[j], pre[j+1])
    uint32_t pre[STNERF_MAX_LAYERS + 1];
    pre[0] = 0;
#pragma unroll
    for (int j = 0; j < STNERF_MAX_LAYERS; ++j) {
        uint32_t items = 0;
        if (j < a.n_layers) {
            const int64_t rows = layer_rows(a.layer[j], a.n_rays, a.ns);
            const int per = a.layer[j].motion ? ST_MTM : ST_TM;
            items = (uint32_t)((rows + per - 1) / per);
        }
        pre[j + 1] = pre[j] + items;
    }
    const uint32_t total = pre[STNERF_MAX_LAYERS];
    auto slot_of = [&](uint32_t item) {
        int slot = 0;
#pragma unroll
        for (int j = 1; j < STNERF_MAX_LAYERS; ++j) slot += (item >= pre[j]) ? 1 : 0;
        return slot;
    };
    auto first_of = [&](uint32_t item) {  // first layer of the item's first network
        NextFirst nf;
        if (item >= total) {  // nothing follows: any valid address will do (the fetch is discarded)
            const StageLayer& l0 = a.layer[0];
            nf.base = l0.space;
            const SpaceLayout L = space_layout(false, DEEP);
            nf.w_off = L.w[0];
            nf.b_off = L.b[0];
            nf.n_total = 256;
            return nf;
        }
        const StageLayer& ly = a.layer[slot_of(item)];
        if (ly.motion) {
            const MotionLayout M = motion_layout();
            nf.base = ly.motion;
            nf.w_off = M.w[0];
            nf.b_off = M.b[0];
            nf.n_total = 128;
        } else {
            const SpaceLayout L = space_layout(ly.use_time != 0, DEEP);
            nf.base = ly.space;
            nf.w_off = L.w[0];
            nf.b_off = L.b[0];
            nf.n_total = 256;
        }
        return nf;
    };
    auto pop = [&]() {  // every thread gets the same fresh item; the word used sits in the (free) scratch area
        uint32_t* slot = reinterpret_cast<uint32_t*>(enc + 12 * ST_TM);
        if (tid == 0) *slot = atomicAdd(a.queue, 1u);
        __syncthreads();
        const uint32_t v = __builtin_amdgcn_readfirstlane(*slot);
        __syncthreads();
        return v;
    };

    uint32_t item = pop();
    WF wA;
    {
        const NextFirst nf = first_of(item);
        load_wfrag(wA, next_w_ptr(nf, wave, lane), next_b_ptr(nf, wave, lane));
    }
    while (item < total) {
        const uint32_t next_item = pop();   // one item of lookahead: its first layer is prefetched by our last one
        const NextFirst nf_next = first_of(next_item);
        const int slot = slot_of(item);
        const StageLayer& ly = a.layer[slot];
        const int64_t rows = layer_rows(ly, a.n_rays, a.ns);
        uint32_t base_item = 0;
#pragma unroll
        for (int j = 1; j < STNERF_MAX_LAYERS; ++j) base_item = (item >= pre[j]) ? pre[j] : base_item;
        const uint32_t t_in_layer = item - base_item;
        const bool use_time = ly.use_time != 0;
        const SpaceLayout L0 = space_layout(use_time, DEEP);
        const NextFirst nf_space{ly.space, L0.w[0], L0.b[0], 256};
        const int part4 = __builtin_amdgcn_readfirstlane(tid / ST_TM);
        const int s4 = tid & (ST_TM - 1);
        const bool deform = ly.motion != nullptr;
        const int64_t r0 = (int64_t)t_in_layer * (deform ? ST_MTM : ST_TM);
        float pa[3] = {0.f, 0.f, 0.f}, pb[3] = {0.f, 0.f, 0.f};
        if (deform) {
            // ---- 256 rows: the MotionNet on all of them; the deformed points stay on chip
            const int part2 = __builtin_amdgcn_readfirstlane(tid / ST_MTM);
            const int s2 = tid & (ST_MTM - 1);
            float p[3] = {0.f, 0.f, 0.f};
            float tv = 0.f;
            {
                const RowRef rr = locate_row(ly.ray_list, r0 + s2, rows, a.ns);
                if (rr.valid) {
                    const float* src = ly.xyz + rr.ray * a.xyz_ray_stride + 3 * rr.k;
                    p[0] = src[0];
                    p[1] = src[1];
                    p[2] = src[2];
                    tv = ly.times[rr.ray * a.times_ray_stride];
                }
            }
            motion_tile(ly.motion, act, encf /* 12 KiB of the idle enc region */, p, tv, ly.motion_flags, lane, wave, part2, s2,
                        wA, nf_space PH_ARGS);
            float* xyzp = reinterpret_cast<float*>(act);  // act is free: hand every thread its two SpaceNet samples
            if (part2 == 0) {
                xyzp[3 * s2 + 0] = p[0];
                xyzp[3 * s2 + 1] = p[1];
                xyzp[3 * s2 + 2] = p[2];
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                pa[c] = xyzp[3 * s4 + c];
                pb[c] = xyzp[3 * (ST_TM + s4) + c];
            }
            // (the first write into act sits behind the barriers of space_tile's prologue and first layer)
        }
        // ---- the SpaceNet on 128 rows (twice for a deformed item)
#pragma unroll 1
        for (int half = 0; half < (deform ? 2 : 1); ++half) {
            const int64_t rbase = r0 + half * ST_TM;
            if (rbase >= rows) break;   // (uniform) the second half of a layer's last item may be empty
            const RowRef rr = locate_row(ly.ray_list, rbase + s4, rows, a.ns);
            float p[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) p[c] = half == 0 ? pa[c] : pb[c];
            if (rr.valid && !deform) {
                const float* src = ly.xyz + rr.ray * a.xyz_ray_stride + 3 * rr.k;
                p[0] = src[0];
                p[1] = src[1];
                p[2] = src[2];
            }
            const bool more = deform && half == 0 && rbase + ST_TM < rows;   // the other half follows on this net
            const NextFirst nf = more ? nf_space : nf_next;
            float4 o = space_tile<DEEP>(ly.space, use_time, act, enc, p, ly.raybias, rr.valid ? (int32_t)rr.ray : 0, lane, wave,
                                        part4, s4, wA, nf PH_ARGS);
            if (part4 == 0 && rr.valid) {
                if (a.sigmoid_rgb) {  // torch.sigmoid(rgb): 1-ulp v_exp_f32 / v_rcp_f32, the same expression the compositor uses
                    o.x = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(o.x * -1.44269504088896340736f));
                    o.y = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(o.y * -1.44269504088896340736f));
                    o.z = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(o.z * -1.44269504088896340736f));
                }
                *reinterpret_cast<float4*>(ly.raw + rr.ray * a.raw_ray_stride + 4 * rr.k) = o;
            }
        }
        item = next_item;
    }
    PH_FLUSH;
}

}  // namespace stnerf

using namespace stnerf;

// Which of the two organisations runs behind stnerf_mlp_stage: STNERF_STAGE_KERNEL = "wave" (mlp_wave.hip: sample-split
// waves, activations in registers) | "lds" (this file: feature-split waves, activations in LDS).  Read on every call so
// that a test can compare the two inside one process; the results are bit-identical.
namespace {
enum { STAGE_KERNEL_LDS = 0, STAGE_KERNEL_WAVE = 1 };
int stage_kernel_choice() {
    const char* e = getenv("STNERF_STAGE_KERNEL");
    if (e && !strcmp(e, "lds")) return STAGE_KERNEL_LDS;
    if (e && !strcmp(e, "wave")) return STAGE_KERNEL_WAVE;
    return STNERF_STAGE_KERNEL_DEFAULT;
}
}  // namespace

// One network stage of the pipeline.  layers[i] describes slot i of the queue (heavier, deformed layers first);
// `queue` is a zeroed uint32 on the device.  Called by stnerf_render_rays only (csrc/pipeline.hip).
extern "C" int stnerf_mlp_stage(const stnerf_stage_layer* layers, int n_layers, int64_t n_rays, int ns, const float* dirs,
                                int64_t dirs_ray_stride, int64_t times_ray_stride, int64_t xyz_ray_stride,
                                int64_t raw_ray_stride, int flags, uint32_t* queue, float* ray_bias, stnerf_stream_t stream) {
    STNERF_REQUIRE(layers && dirs && queue && ray_bias, "mlp_stage: null pointer");
    STNERF_REQUIRE(((uintptr_t)ray_bias & 15) == 0, "mlp_stage: ray_bias must be 16-byte aligned");
    STNERF_REQUIRE(n_layers >= 1 && n_layers <= STNERF_MAX_LAYERS && n_rays >= 0 && ns >= 1, "mlp_stage: bad shape");
    STNERF_REQUIRE((raw_ray_stride & 3) == 0, "mlp_stage: raw ray stride must be a multiple of 4 floats");
    if (n_rays == 0) return STNERF_OK;
    const int deep_rgb = (flags & STNERF_STAGE_DEEP_RGB) != 0;
    StageArgs a;
    memset(&a, 0, sizeof(a));
    a.sigmoid_rgb = (flags & STNERF_STAGE_SIGMOID_RGB) != 0;
    // (the profiler's record of the stage covers the per-ray prologues too: their work is part of the networks' FLOPs)
    LaunchTimer timer(PROF_MLP_STAGE, deep_rgb, n_rays, ns, 0, as_stream(stream));
    for (int i = 0; i < n_layers; ++i) {
        const stnerf_stage_layer& s = layers[i];
        STNERF_REQUIRE(s.space && s.xyz && s.raw, "mlp_stage: layer %d: null pointer", i);
        STNERF_REQUIRE(((uintptr_t)s.space & 15) == 0 && ((uintptr_t)s.raw & 15) == 0 && (!s.motion || ((uintptr_t)s.motion & 15) == 0),
                       "mlp_stage: layer %d: packed weights / raw must be 16-byte aligned", i);
        STNERF_REQUIRE(!(s.use_time || s.motion) || s.times, "mlp_stage: layer %d needs its frame-id column", i);
        a.layer[i] = StageLayer{static_cast<const float*>(s.space), static_cast<const float*>(s.motion), s.ray_list,
                                s.ray_count, s.xyz, s.raw, s.times, s.use_time, s.motion_flags,
                                ray_bias + (int64_t)i * n_rays * 128};
        // rgb_net.1's direction / time columns once per ray of this layer (mlp_raybias.hip)
        const int kind = s.use_time ? (deep_rgb ? STNERF_NET_SPACE_TIME_DEEP : STNERF_NET_SPACE_TIME)
                                    : (deep_rgb ? STNERF_NET_SPACE_DEEP : STNERF_NET_SPACE);
        if (const int rc = launch_ray_bias(kind, static_cast<const float*>(s.space), n_rays, s.ray_list, s.ray_count, dirs,
                                           dirs_ray_stride, s.times, times_ray_stride, ray_bias + (int64_t)i * n_rays * 128,
                                           as_stream(stream)))
            return rc;
    }
    a.n_layers = n_layers;
    a.ns = ns;
    a.n_rays = n_rays;
    a.xyz_ray_stride = xyz_ray_stride;
    a.raw_ray_stride = raw_ray_stride;
    a.dirs_ray_stride = dirs_ray_stride;
    a.times_ray_stride = times_ray_stride;
    a.dirs = dirs;
    a.queue = queue;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (stage_kernel_choice() == STAGE_KERNEL_WAVE) return launch_wave_stage(a, deep_rgb != 0, cus, as_stream(stream));
    const int64_t max_items = ((n_rays * ns + ST_TM - 1) / ST_TM) * n_layers;
    const int grid = (int)(max_items < cus ? max_items : cus);  // one persistent workgroup per CU
    const void* kfn = deep_rgb ? reinterpret_cast<const void*>(mlp_stage_kernel<true>) : reinterpret_cast<const void*>(mlp_stage_kernel<false>);
    if (const int rc = reserve_dynamic_lds(kfn, ST_LDS, "mlp_stage")) return rc;
    if (deep_rgb)
        hipLaunchKernelGGL(mlp_stage_kernel<true>, dim3(grid), dim3(ST_THREADS), ST_LDS, as_stream(stream), a);
    else
        hipLaunchKernelGGL(mlp_stage_kernel<false>, dim3(grid), dim3(ST_THREADS), ST_LDS, as_stream(stream), a);
    STNERF_CHECK_LAUNCH("mlp_stage");
    return STNERF_OK;
}
