// Persistent stage kernel, second organisation: SAMPLE-split waves with the activations in registers.
//
// The round-2 stage kernel (mlp_stage.hip, removed in round 3) split a layer's OUTPUT FEATURES over the 8 waves of a
// workgroup, so every layer was an all-to-all
// through LDS: 128 KiB of activations are written (ds_write_b128: ~79 B/clk/CU, ~1.1k cycles per layer with the matrix
// pipe idle) and two workgroup barriers are taken per layer; that, the heads' LDS reductions and the issue slots of
// the activation reads are the ~9 % the kernel stands below the f32 MFMA peak.  Here a wave owns 32 SAMPLES and all
// features of a layer:
//   * v_mfma_f32_32x32x2_f32 computed transposed (A = weights, B = activations) leaves a lane (h, c) with the
//     output features 32 fb + 8 q + 4 h + r of sample c in accumulator register 4 q + r of block fb -- and with the
//     k permutation the packed weights already use (lane half h takes k = 8 s + 4 h + {0..3}) that is exactly the B
//     operand the NEXT layer needs from this lane: K step s, instruction kk reads register 4 (s & 3) + kk of block
//     s >> 2.  The activations of the whole network therefore never leave the register file: no LDS traffic, no
//     barrier, no epilogue stores between layers; a layer boundary is 128 ReLUs (v_max_i32) and nothing else.
//   * 256 features x 32 samples = 128 registers in + 128 accumulators out: one wave per SIMD with the unified
//     512-entry register file (accumulators in AGPRs), 4 waves = 128 samples per CU.  (A second wave per SIMD would
//     not buy overlap: f32 MFMAs run on the SIMD's own f32 lanes, a vector instruction costs the MFMA stream its 5 - 6
//     cycles whichever wave issues it -- tools/micro/mfma_two_waves.hip.  What counts is the vector instruction COUNT.)
//   * Weights: the A operand of a K step is 8 x 16 B per lane straight from the packed blob (buffer loads, SGPR
//     offsets, one step ahead); the four waves of a CU run the same network in step, so each line comes out of L2
//     once per CU and the other three waves hit the vector L1.
//   * rgb_net.1's direction / time columns come per RAY from mlp_raybias.hip as the layer's C operand.
//   * Encodings are staged through a wave-private 11 KiB LDS window (the two lanes of a sample split the frequencies,
//     then each lane reads its half of the feature quads back) -- ordering inside a wave only, no barrier.
//   * Heads: a lane holds every second feature quad of its sample; the partial-sum grouping of the LDS kernels
//     (4 parts x 4 interleaved chains) is kept, the two lanes of a sample swap their chains with v_permlane32_swap.
// Every output accumulates the same products in the same order as in the per-network kernels of mlp.hip (same
// instruction, same k order, bias as the C operand of the first MFMA, same head grouping): results are bit-identical to
// them (tests/test_gpu_ops.py).
//
// Reference: modeling/spacenet.py:16-160, modeling/motion_net.py:7-71, modeling/layered_rfrender.py:340-418,495-576.
#include "mlp_wave_core.h"

namespace stnerf {

// Training: the same kernel with a tap that writes every layer's input -- the activations the backward pass
// needs -- to row-major matrices the caller owns, as the item's rows pass through the registers: one launch instead of a
// chain of per-layer GEMMs through HBM for the recomputation, and the SAME arithmetic as the forward that produced the loss.
// Slot 0 of the queue only (the training entry point launches one network).  Row r of the launch <-> row r of every matrix.
struct StoreTap {
    const StoreTapArgs* a;
    uint32_t row;
    bool valid;
    template <int NBLK>
    __device__ __forceinline__ void blocks(int stage, const f32x16 (&blk)[NBLK], int nblk, int lane) const {
        if (!valid) return;
        float* base = stage == TAP_PE ? a->pe : a->buf[stage];
        const int ld = stage == TAP_PE ? a->ld_pe : a->ld[stage];
        // register 4 q + r of block fb <-> feature 32 fb + 8 q + 4 h + r: 16 bytes per (fb, q), the two lanes of a sample side by side
        float4* p = reinterpret_cast<float4*>(base + (size_t)row * (size_t)ld + 4 * (lane >> 5));
#pragma unroll
        for (int fb = 0; fb < NBLK; ++fb)
            if (fb < nblk) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    p[fb * 8 + 2 * q] = make_float4(blk[fb][4 * q + 0], blk[fb][4 * q + 1], blk[fb][4 * q + 2], blk[fb][4 * q + 3]);
            }
        // The ReLU mask of this lane's values as bits: value 16 fb + i <-> bit (16 fb + i) & 31 of word fb >> 1 -- 16 bytes per lane,
        // 32 per row, which the backward chain (csrc/train_wave.hip: the SAME lane owns the same values there) reads back instead of
        // the 1 KB of activations.  Two instructions per value: min(bits, 1) -- a post-ReLU value is +0 or positive -- and a shift-or.
        if (a->bits && stage != TAP_PE) {
            uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int wd = 0; wd < NBLK / 2; ++wd)
                if (2 * wd < nblk) {
#pragma unroll
                    for (int j = 31; j >= 0; --j) {
                        // (asm: the compiler's own choice is compare + select + shift-or, three instructions, with the 128 selects hoisted
                        // into registers the kernel does not have)
                        uint32_t t;
                        asm volatile("v_min_u32 %1, 1, %2\n\tv_lshl_or_b32 %0, %0, 1, %1" : "+v"(w[wd]), "=&v"(t) : "v"(blk[2 * wd + (j >> 4)][j & 15]));
                    }
                }
            uint4* bp = reinterpret_cast<uint4*>(a->bits + (size_t)stage * (size_t)a->bits_stride + (size_t)row * 8u + 4u * (uint32_t)(lane >> 5));
            *bp = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
};
__device__ __forceinline__ NoTap make_tap(const NoTapArgs&, uint32_t, bool) { return NoTap(); }
__device__ __forceinline__ StoreTap make_tap(const StoreTapArgs& t, uint32_t row, bool valid) { return StoreTap{&t, row, valid}; }

template <bool DEEP, class TapArgs>
__global__ __launch_bounds__(WV_THREADS, 1) void mlp_wave_stage_kernel(StageArgs a, TapArgs targs) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* encw = reinterpret_cast<float*>(smem) + wave * WV_WAVE_FLOATS;  // [encodings | bias vectors]
    uint32_t* qslot = reinterpret_cast<uint32_t*>(reinterpret_cast<float*>(smem) + WV_NW * WV_WAVE_FLOATS);
    // ---- the queue: items (128 rows) of layer slot j are [pre[j], pre[j+1]); the row counts (one global load each) are
    // kept in LDS for the per-item lookups
    int64_t* lrows = reinterpret_cast<int64_t*>(qslot + 4);
    uint32_t pre[STNERF_MAX_LAYERS + 1];
    pre[0] = 0;
#pragma unroll
    for (int j = 0; j < STNERF_MAX_LAYERS; ++j) {
        uint32_t items = 0;
        if (j < a.n_layers) {
            const int64_t rows = layer_rows(a.layer[j], a.n_rays, a.ns);
            items = (uint32_t)((rows + WV_ITEM - 1) / WV_ITEM);
            if (tid == 0) lrows[j] = rows;
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
    auto base_of = [&](uint32_t item) {
        uint32_t b = 0;
#pragma unroll
        for (int j = 1; j < STNERF_MAX_LAYERS; ++j) b = (item >= pre[j]) ? pre[j] : b;
        return b;
    };
    // row of this lane's sample in an item, and the ray it belongs to (first half of an item's fetch)
    auto row_of = [&](uint32_t item, RowRef& rr) {
        rr = RowRef{0, 0, false};
        if (item >= total) return;
        const int slot = slot_of(item);
        const int64_t rows = lrows[slot];
        const int64_t row = (int64_t)(item - base_of(item)) * WV_ITEM + wave * WV_ROWS + (lane & 31);
        rr.valid = row < rows;
        if (rr.valid) {
            int64_t rslot;
            if (rows <= 0x7fffffffll) {  // (uniform) the usual case: a 32-bit division
                const uint32_t q = (uint32_t)row / (uint32_t)a.ns;
                rslot = q;
                rr.k = (int)((uint32_t)row - q * (uint32_t)a.ns);
            } else {
                rslot = row / a.ns;
                rr.k = (int)(row - rslot * a.ns);
            }
            const int32_t* rl = a.layer[slot].ray_list;
            rr.ray = rl ? (int64_t)rl[rslot] : rslot;
        }
    };
    // second half: the sample's inputs (HBM loads)
    auto fetch = [&](uint32_t item, const RowRef& rr, WaveInputs& in) {
        in.valid = rr.valid;
        in.raw_off = 0;
        in.ray = 0;
        in.tv = 0.f;
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) in.p[c3] = 0.f;
        if (rr.valid) {
            const StageLayer& ly = a.layer[slot_of(item)];
            const float* src = ly.xyz + rr.ray * a.xyz_ray_stride + 3 * rr.k;
#pragma unroll
            for (int c3 = 0; c3 < 3; ++c3) in.p[c3] = src[c3];
            if (ly.motion) in.tv = ly.times[rr.ray * a.times_ray_stride];
            in.raw_off = rr.ray * a.raw_ray_stride + 4 * rr.k;
            in.ray = (int32_t)rr.ray;
        }
    };

    // ---- prime the pipeline: two items popped, the first one's inputs loaded
    if (tid == 0) {
        qslot[0] = atomicAdd(a.queue, 1u);
        qslot[1] = atomicAdd(a.queue, 1u);
    }
    __syncthreads();
    uint32_t it0 = __builtin_amdgcn_readfirstlane(qslot[0]);
    uint32_t it1 = __builtin_amdgcn_readfirstlane(qslot[1]);
    __syncthreads();
    WaveInputs cur, nxt;
    {
        RowRef rr;
        row_of(it0, rr);
        fetch(it0, rr, cur);
    }
    int par = 0;
    f32x16 acc[8], in[8];
    float4 wa[8], wb[8];
#ifdef STNERF_WAVE_PROF
    WaveProf wp;
    for (int i = 0; i < 16; ++i) wp.acc[i] = 0;
    wp.t = clock64();
#endif
    while (it0 < total) {
        // the item after next (consumed at the end of this one) and the ray index of the next item's sample
        uint32_t pending = 0;
        if (tid == 0) pending = atomicAdd(a.queue, 1u);
        RowRef rr_next;
        row_of(it1, rr_next);
        const StageLayer& ly = a.layer[slot_of(it0)];
        float p[3];
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) p[c3] = cur.p[c3];
#ifdef STNERF_WAVE_DEBUG
        WaveDbg dbg{a.dbg, a.dbg_stage, -1};
        if (slot_of(it0) == 0 && cur.valid) dbg.row = (int64_t)(it0 - base_of(it0)) * WV_ITEM + wave * WV_ROWS + (lane & 31);
#endif
        WP(WP_TOP);
        // (deep_rgb variant: the lane index the networks see is opaque per item -- hoisted out of the item loop, the
        // per-lane LDS addresses derived from it do not fit beside this variant's live values and go to scratch)
        int ln = lane;
        if constexpr (DEEP) asm volatile("" : "+v"(ln));
        if (ly.motion) motion_wave(ly.motion, encw, p, cur.tv, ly.motion_flags, ln, acc, in, wa, wb WV_DBG_ARG WP_ARG);
        // (training: the row of this lane's sample in the launch; items of slot 0 are rows 128 item ..)
        const auto tap = make_tap(targs, it0 * (uint32_t)WV_ITEM + (uint32_t)(wave * WV_ROWS + (lane & 31)), cur.valid);
        float4 o = space_wave<DEEP>(ly.space, ly.use_time != 0, encw, p, ly.raybias, cur.ray, ln, acc, in, wa, wb,
                                    [&]() { fetch(it1, rr_next, nxt); } WV_DBG_ARG WP_ARG, tap);
        if (cur.valid && lane < 32) {
            if (a.sigmoid_rgb) {  // torch.sigmoid(rgb): 1-ulp v_exp_f32 / v_rcp_f32, the same expression the compositor uses
                o.x = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(o.x * -1.44269504088896340736f));
                o.y = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(o.y * -1.44269504088896340736f));
                o.z = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(o.z * -1.44269504088896340736f));
            }
            *reinterpret_cast<float4*>(ly.raw + cur.raw_off) = o;
        }
        if (tid == 0) qslot[par] = pending;
        __syncthreads();
        const uint32_t it2 = __builtin_amdgcn_readfirstlane(qslot[par]);
        par ^= 1;
        it0 = it1;
        it1 = it2;
        cur = nxt;
        WP(WP_END);
#ifdef STNERF_WAVE_PROF
        wp.acc[WP_ITEMS] += 1;
#endif
    }
#ifdef STNERF_WAVE_PROF
    if (lane == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&g_wphase[i], wp.acc[i]);
#endif
}

#ifdef STNERF_WAVE_DEBUG
static float* g_dbg_buf = nullptr;
static int g_dbg_stage = -1;
#endif

int launch_wave_stage(const StageArgs& a_in, bool deep_rgb, int cus, hipStream_t stream) {
    StageArgs a = a_in;
#ifdef STNERF_WAVE_DEBUG
    a.dbg = g_dbg_buf;
    a.dbg_stage = g_dbg_stage;
#endif
    const int64_t max_items = ((a.n_rays * a.ns + WV_ITEM - 1) / WV_ITEM) * a.n_layers;
    const int grid = (int)(max_items < cus ? max_items : cus);  // one persistent workgroup per CU
    const void* kfn = deep_rgb ? reinterpret_cast<const void*>(mlp_wave_stage_kernel<true, NoTapArgs>)
                               : reinterpret_cast<const void*>(mlp_wave_stage_kernel<false, NoTapArgs>);
    if (const int rc = reserve_dynamic_lds(kfn, WV_LDS, "mlp_stage (wave)")) return rc;
    if (deep_rgb)
        hipLaunchKernelGGL((mlp_wave_stage_kernel<true, NoTapArgs>), dim3(grid), dim3(WV_THREADS), WV_LDS, stream, a, NoTapArgs());
    else
        hipLaunchKernelGGL((mlp_wave_stage_kernel<false, NoTapArgs>), dim3(grid), dim3(WV_THREADS), WV_LDS, stream, a, NoTapArgs());
    STNERF_CHECK_LAUNCH("mlp_stage (wave)");
    return STNERF_OK;
}

// One SpaceNet (queue slot 0 of `a`, no MotionNet, not deep_rgb) with every layer's input written out: see StoreTapArgs.
int launch_wave_stage_store(const StageArgs& a, float* const (&buf)[8], const int32_t (&ld)[8], float* pe, int32_t ld_pe, uint32_t* bits,
                            int64_t bits_stride, int cus, hipStream_t stream) {
    StoreTapArgs t;
    t.bits = bits;
    t.bits_stride = bits_stride;
    for (int i = 0; i < 8; ++i) {
        t.buf[i] = buf[i];
        t.ld[i] = ld[i];
    }
    t.pe = pe;
    t.ld_pe = ld_pe;
    const int64_t max_items = (a.n_rays * a.ns + WV_ITEM - 1) / WV_ITEM;
    const int grid = (int)(max_items < cus ? max_items : cus);
    if (const int rc = reserve_dynamic_lds(reinterpret_cast<const void*>(mlp_wave_stage_kernel<false, StoreTapArgs>), WV_LDS, "train_space_fwd"))
        return rc;
    hipLaunchKernelGGL((mlp_wave_stage_kernel<false, StoreTapArgs>), dim3(grid), dim3(WV_THREADS), WV_LDS, stream, a, t);
    STNERF_CHECK_LAUNCH("train_space_fwd");
    return STNERF_OK;
}

}  // namespace stnerf

#ifdef STNERF_WAVE_PROF
extern "C" int stnerf_debug_wave_phases(unsigned long long* host16, int reset) {
    if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(stnerf::g_wphase), sizeof(unsigned long long) * 16) != hipSuccess) return STNERF_ELAUNCH;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(stnerf::g_wphase), z, sizeof(z)) != hipSuccess) return STNERF_ELAUNCH;
    }
    return STNERF_OK;
}
#endif

#ifdef STNERF_WAVE_DEBUG
// development builds only: where the next stnerf_mlp_stage launches dump the activations of queue slot 0
extern "C" int stnerf_debug_wave_dump(float* buf, int stage) {
    stnerf::g_dbg_buf = buf;
    stnerf::g_dbg_stage = stage;
    return STNERF_OK;
}
#endif
