// Training: the SpaceNet's backward as two fused launches in the sample-split wave organisation of
// csrc/mlp_wave.hip instead of a chain of per-layer GEMMs through HBM.
//
//   stnerf_train_spacenet_fwd   the inference stage kernel itself (same arithmetic as the forward that produced the loss) with
//                               a tap that writes every layer's input to row-major matrices as the rows pass through the
//                               registers (mlp_wave.hip: StoreTap) -- the recomputation of the backward pass, one launch.
//   stnerf_train_spacenet_dx    the chain d act_{s-1} = (d act_s * [act_s > 0]) W_s from the heads back to stage1.2 (and on to
//                               PE(pos) when the sample points need a gradient), a wave owning 32 rows and carrying the gradient
//                               from layer to layer in its registers: the accumulator layout of one product is the B-operand
//                               layout of the next, exactly as in the forward kernel, with the TRANSPOSED weights as the A
//                               operand (blob sections [out/4][in][4], built by stnerf_amd.modeling.autograd).  The ReLU masks come
//                               as BIT PLANES the forward tap wrote (16 bytes per lane and layer: the same lane owns the same
//                               values in both kernels), fetched at the start of an item; at every layer boundary the masked
//                               gradient -- the pre-activation gradient d y_s the weight gradients need -- is written out once.
//
// (Round 6: both launches also exist in split bf16 -- csrc/mlp_bf16x3.hip: the stage kernel with a tap and train_space_dx_bx_kernel --
// and that pair is what modeling/autograd.py runs unless the model was built for exact f32; the forward's entry point is here, the chain's
// beside its kernel.)
//
// What is left to per-layer launches are the weight gradients dW_s = d y_s^T x_s (csrc/train.hip: a reduction over ALL rows that
// no 128-row item can finish on its own) and the encodings' chain rule.
//
// Reference: modeling/spacenet.py:45-86,101-160 through ATen's addmm_backward / threshold_backward, engine/layered_trainer.py:281.
#include "mlp_wave_core.h"
#include "mlp_bf16x3.h"

namespace stnerf {

struct DxArgs {
    const float* wt;          // transposed sections, see the offsets below (floats)
    const float* d_raw;       // [rows][4]: dLoss / d {r, g, b, sigma} (raw network outputs)
    int64_t bits_stride;      // uint32 words between two stages' planes of `bits`
    const uint32_t* bits;     // [8][.. bits_stride ..], rows x 8 words per stage: bit (16 fb + i) & 31 of word (fb >> 1) + 4 h of row r, stage s = [act_s > 0] for the value the lane
                              // (h, r & 31) holds in register i of block fb (stage s = 0 .. 6: stage1.0 .. stage2.4's output; 7: rgb_net.1's)
    float* dy[8];             // dy[s]: dLoss / d (pre-activation of that layer), same widths
    int32_t ld_dy[8];
    float* dpe;               // [rows][ld_dpe]: dLoss / d PE(pos) (64 wide: feature 63 is the pad) -- only read with DPOS
    int32_t ld_dpe;
    int64_t rows;
    // sections of wt: A operands [out/4][N][4] of d x = d y W  (N = inputs padded to a multiple of 32)
    uint32_t o_rgb1;          // rgb_net.1[:, :256]: out 128, N 256
    uint32_t o_l[7];          // stage1.0 (N 64), stage1.2, 1.4, 1.6 (N 256), stage2.0 (N 320: [h4 | PE]), stage2.2, 2.4 (N 256): out 256
    uint32_t o_wsigma;        // density_net.0: 256 floats
    uint32_t o_wrgb2;         // the colour head: [3][128]
};

// Layer boundary of the backward chain: in = mask ? acc : 0 for this lane's 16 x NFB values of its row (threshold_backward), written to
// dy as well (row base + 4 h floats; register 4 q + r of block fb <-> column 32 fb + 8 q + 4 h + r); acc = 0 for the next product.
// mask: the lane's 128 bits of this layer (value 16 fb + i <-> bit (16 fb + i) & 31 of component fb >> 1): one v_bfe_i32 spreads a bit
// over a word, one v_and applies it -- no load, no compare.  (Rounds' history: the masks used to be the stored fp32 activations, 1 KB
// per row and layer read back here -- 32 MB per boundary over the chip with the matrix pipes idle; profiles/r05_training_kernels.md.)
template <int NFB>
__device__ __forceinline__ void mask_boundary(f32x16 (&acc)[8], f32x16 (&in)[8], const uint4& mask, float* dy, bool valid) {
    float4* yp = reinterpret_cast<float4*>(dy);
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) {
        const uint32_t word = (fb >> 1) == 0 ? mask.x : (fb >> 1) == 1 ? mask.y : (fb >> 1) == 2 ? mask.z : mask.w;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int m;     // 0 or -1 (asm: the compiler's own choice is and + compare + select)
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(word), "n"((fb & 1) * 16 + i));
            in[fb][i] = __int_as_float(__float_as_int(acc[fb][i]) & m);
            acc[fb][i] = 0.f;
        }
        if (valid) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                yp[fb * 8 + 2 * q] = make_float4(in[fb][4 * q + 0], in[fb][4 * q + 1], in[fb][4 * q + 2], in[fb][4 * q + 3]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <bool DPOS>
__global__ __launch_bounds__(WV_THREADS, 1) void train_space_dx_kernel(DxArgs a) {
    __shared__ __attribute__((aligned(16))) float heads[256 + 3 * 128];   // density_net.0 | the colour head
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    for (int i = tid; i < 256; i += WV_THREADS) heads[i] = a.wt[a.o_wsigma + i];
    for (int i = tid; i < 384; i += WV_THREADS) heads[256 + i] = a.wt[a.o_wrgb2 + i];
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = weight_rsrc(a.wt);
    constexpr uint32_t WSTEP256 = 2u * 256u * 16u, WSTEP320 = 2u * 320u * 16u, WSTEP64 = 2u * 64u * 16u;
    const LaneOfs wl256 = lane_offsets((uint32_t)(h * 256 + c) * 16u);
    const NextOfs nx256{(uint32_t)(h * 256 + c) * 16u, false, 0u};
    const NextOfs nx320{(uint32_t)(h * 320 + c) * 16u, false, 0u};
    const NextOfs nx320b{(uint32_t)(h * 320 + c) * 16u + 8u * 512u, false, 0u};   // blocks 8, 9 of stage2.0's section: the PE columns
    const NextOfs nx64{(uint32_t)(h * 64 + c) * 16u, false, 0u};
    f32x16 acc[8], in[8];
    float4 wa[8], wb[8];
    const int64_t items = (a.rows + WV_ITEM - 1) / WV_ITEM;
    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int64_t row = item * WV_ITEM + wave * WV_ROWS + c;
        const bool valid = row < a.rows;
        const int64_t r = valid ? row : 0;
        auto yrow = [&](int s) { return a.dy[s] + r * a.ld_dy[s] + 4 * h; };
        // the first operands of the first product and the first mask, in flight behind the heads' vector work
        load_w<8>(wa, rsrc, wl256, a.o_rgb1 * 4u);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) g = *reinterpret_cast<const float4*>(a.d_raw + r * 4);
        // the item's eight ReLU masks: 16 bytes per lane and layer, the wave's 32 rows contiguous (1 KB per load)
        uint4 bw[8];
#pragma unroll
        for (int sidx = 0; sidx < 8; ++sidx)
            bw[sidx] = valid ? *reinterpret_cast<const uint4*>(a.bits + (size_t)sidx * (size_t)a.bits_stride + (size_t)r * 8u + 4u * (uint32_t)h)
                             : make_uint4(0u, 0u, 0u, 0u);
        // ---- the colour head backwards: d act7[f] = sum_o d rgb[o] W[o][f] (rgb_net.3, modeling/spacenet.py:84-85), masked by act7 > 0
        {
            const float4* w4 = reinterpret_cast<const float4*>(heads + 256) + h;   // quad 2 s + h of each of the three rows
#pragma unroll
            for (int fb = 0; fb < 4; ++fb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w0 = w4[2 * (fb * 4 + q)], w1 = w4[32 + 2 * (fb * 4 + q)], w2 = w4[64 + 2 * (fb * 4 + q)];
                    acc[fb][4 * q + 0] = g.x * w0.x + g.y * w1.x + g.z * w2.x;
                    acc[fb][4 * q + 1] = g.x * w0.y + g.y * w1.y + g.z * w2.y;
                    acc[fb][4 * q + 2] = g.x * w0.z + g.y * w1.z + g.z * w2.z;
                    acc[fb][4 * q + 3] = g.x * w0.w + g.y * w1.w + g.z * w2.w;
                }
        }
        mask_boundary<4>(acc, in, bw[7], yrow(7), valid);
        // ---- d act6 = d y7 W_rgb1[:, :256] + d sigma w_sigma: the density head's rank-1 term is the product's C operand
        {
            const float4* w4 = reinterpret_cast<const float4*>(heads) + h;
#pragma unroll
            for (int fb = 0; fb < 8; ++fb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w = w4[2 * (fb * 4 + q)];
                    acc[fb][4 * q + 0] = g.w * w.x;
                    acc[fb][4 * q + 1] = g.w * w.y;
                    acc[fb][4 * q + 2] = g.w * w.z;
                    acc[fb][4 * q + 3] = g.w * w.w;
                }
        }
        segment_r<8, 4, 16, 0, 8>(acc, reinterpret_cast<const f32x16 (&)[4]>(in[0]), wa, wb, rsrc, wl256, a.o_rgb1 * 4u, WSTEP256, nx256, a.o_l[6] * 4u);
        mask_boundary<8>(acc, in, bw[6], yrow(6), valid);
        segment_r<8, 8, 32, 0, 8>(acc, in, wa, wb, rsrc, wl256, a.o_l[6] * 4u, WSTEP256, nx256, a.o_l[5] * 4u);     // stage2.4
        mask_boundary<8>(acc, in, bw[5], yrow(5), valid);
        segment_r<8, 8, 32, 0, 8>(acc, in, wa, wb, rsrc, wl256, a.o_l[5] * 4u, WSTEP256, nx320, a.o_l[4] * 4u);     // stage2.2
        mask_boundary<8>(acc, in, bw[4], yrow(4), valid);
        // stage2.0: the h4 columns (blocks 0 .. 7), then -- only if the points need a gradient -- the PE columns (blocks 8, 9)
        f32x16 accp[8];   // (only [0], [1] are live)
        if constexpr (DPOS) {
            const LaneOfs wl320 = lane_offsets(nx320.base);
            segment_r<8, 8, 32, 0, 8>(acc, in, wa, wb, rsrc, wl320, a.o_l[4] * 4u, WSTEP320, nx320b, a.o_l[4] * 4u);
#pragma unroll
            for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                for (int i = 0; i < 16; ++i) accp[fb][i] = 0.f;
            const LaneOfs wl320b = lane_offsets(nx320b.base);
            segment_r<2, 8, 32, 0, 8>(accp, in, wa, wb, rsrc, wl320b, a.o_l[4] * 4u, WSTEP320, nx256, a.o_l[3] * 4u);
        } else {
            const LaneOfs wl320 = lane_offsets(nx320.base);
            segment_r<8, 8, 32, 0, 8>(acc, in, wa, wb, rsrc, wl320, a.o_l[4] * 4u, WSTEP320, nx256, a.o_l[3] * 4u);
        }
        mask_boundary<8>(acc, in, bw[3], yrow(3), valid);
        segment_r<8, 8, 32, 0, 8>(acc, in, wa, wb, rsrc, wl256, a.o_l[3] * 4u, WSTEP256, nx256, a.o_l[2] * 4u);     // stage1.6
        mask_boundary<8>(acc, in, bw[2], yrow(2), valid);
        segment_r<8, 8, 32, 0, 8>(acc, in, wa, wb, rsrc, wl256, a.o_l[2] * 4u, WSTEP256, nx256, a.o_l[1] * 4u);     // stage1.4
        mask_boundary<8>(acc, in, bw[1], yrow(1), valid);
        // stage1.2 (the fetch behind it: stage1.0's section when the points need a gradient, else discarded)
        segment_r<8, 8, 32, 0, 8>(acc, in, wa, wb, rsrc, wl256, a.o_l[1] * 4u, WSTEP256, DPOS ? nx64 : nx256, (DPOS ? a.o_l[0] : a.o_l[1]) * 4u);
        mask_boundary<8>(acc, in, bw[0], yrow(0), valid);
        if constexpr (DPOS) {
            // d PE = d y4 W_2.0[:, 256:] + d y0 W_1.0: stage1.0's product continues in the skip connection's accumulators
            const LaneOfs wl64 = lane_offsets(nx64.base);
            segment_r<2, 8, 32, 0, 2>(accp, in, wa, wb, rsrc, wl64, a.o_l[0] * 4u, WSTEP64, nx64, a.o_l[0] * 4u);
            if (valid) {
                float4* yp = reinterpret_cast<float4*>(a.dpe + r * a.ld_dpe + 4 * h);
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        yp[fb * 8 + 2 * q] = make_float4(accp[fb][4 * q + 0], accp[fb][4 * q + 1], accp[fb][4 * q + 2], accp[fb][4 * q + 3]);
            }
        }
    }
}

// ---- MotionNet (modeling/motion_net.py:20-71): the same two launches ---------------------------------------------------------
// Forward = motion_wave, the inference kernel's MotionNet, on rows [xyz | t], with a tap that writes the staged encoding (the right
// operand of motion_net.0's weight gradient), the five post-ReLU outputs and their masks as bit planes (8 bytes per lane and layer:
// two words for the lane's 64 values), and the flow.
struct MotionTapArgs {
    float* enc;          // [rows][ld_enc]: the 84 encoded features + 4 zeros
    int32_t ld_enc;
    float* act[5];       // the post-ReLU outputs of motion_net.0 .. .8: [rows][ld_act[s]], 128 columns
    int32_t ld_act[5];
    uint32_t* bits;      // [5][.. bits_stride ..], rows x 4 words per stage: value 16 fb + i of lane (h, row & 31) <-> bit (16 fb + i) & 31 of word (fb >> 1) + 2 h
    int64_t bits_stride;
    float* flow;         // [rows][3]
};
struct MotionStoreTap {
    const MotionTapArgs* a;
    uint32_t row;
    bool valid;
    template <int NBLK>
    __device__ __forceinline__ void blocks(int stage, const f32x16 (&blk)[NBLK], int, int lane) const {
        if (!valid) return;
        const int h = lane >> 5;
        if constexpr (NBLK == 3) {      // the encoding: K steps 0 .. 10 = quads 2 s + h
            float4* p = reinterpret_cast<float4*>(a->enc + (size_t)row * (size_t)a->ld_enc + 4 * h);
#pragma unroll
            for (int s = 0; s < 11; ++s)
                p[2 * s] = make_float4(blk[s >> 2][4 * (s & 3) + 0], blk[s >> 2][4 * (s & 3) + 1], blk[s >> 2][4 * (s & 3) + 2], blk[s >> 2][4 * (s & 3) + 3]);
        } else {
            float4* p = reinterpret_cast<float4*>(a->act[stage] + (size_t)row * (size_t)a->ld_act[stage] + 4 * h);
#pragma unroll
            for (int fb = 0; fb < 4; ++fb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    p[fb * 8 + 2 * q] = make_float4(blk[fb][4 * q + 0], blk[fb][4 * q + 1], blk[fb][4 * q + 2], blk[fb][4 * q + 3]);
            uint32_t w[2] = {0u, 0u};
#pragma unroll
            for (int wd = 0; wd < 2; ++wd)
#pragma unroll
                for (int j = 31; j >= 0; --j) {
                    uint32_t t;     // (see StoreTap in mlp_wave.hip: min(bits, 1), shift-or)
                    asm volatile("v_min_u32 %1, 1, %2\n\tv_lshl_or_b32 %0, %0, 1, %1" : "+v"(w[wd]), "=&v"(t) : "v"(blk[2 * wd + (j >> 4)][j & 15]));
                }
            *reinterpret_cast<uint2*>(a->bits + (size_t)stage * (size_t)a->bits_stride + (size_t)row * 4u + 2u * (uint32_t)h) = make_uint2(w[0], w[1]);
        }
    }
    __device__ __forceinline__ void flow(const float (&fl)[3], int lane) const {
        if (valid && lane < 32) {
            float* o = a->flow + (size_t)row * 3u;
            o[0] = fl[0];
            o[1] = fl[1];
            o[2] = fl[2];
        }
    }
};
struct MotionFwdArgs {
    const float* net;    // packed MotionNet (exact f32)
    const float* xt;     // [rows][4]: x, y, z, frame id
    int64_t rows;
    int32_t flags;       // STNERF_MOTION_PLAIN_TIME
    MotionTapArgs tap;
};
__global__ __launch_bounds__(WV_THREADS, 1) void train_motion_fwd_kernel(MotionFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 smem_mf[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* encw = reinterpret_cast<float*>(smem_mf) + wave * WV_WAVE_FLOATS;
    f32x16 acc[8], in[8];
    float4 wa[8], wb[8];
#ifdef STNERF_WAVE_DEBUG
    const WaveDbg dbg{nullptr, 0, -1};
#endif
#ifdef STNERF_WAVE_PROF
    WaveProf wp;
    for (int i = 0; i < 16; ++i) wp.acc[i] = 0;
    wp.t = clock64();
#endif
    const int64_t items = (a.rows + WV_ITEM - 1) / WV_ITEM;
    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int64_t row = item * WV_ITEM + wave * WV_ROWS + (lane & 31);
        const bool valid = row < a.rows;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) x = *reinterpret_cast<const float4*>(a.xt + row * 4);
        float p[3] = {x.x, x.y, x.z};
        const MotionStoreTap tap{&a.tap, (uint32_t)row, valid};
        motion_wave(a.net, encw, p, x.w, a.flags, lane, acc, in, wa, wb WV_DBG_ARG WP_ARG, tap);
    }
}

// Backward chain: d a_{l-1} = (d a_l * [a_l > 0]) W_l from the flow head back to motion_net.0's input, the gradient in the wave's
// registers between layers (128-wide products in K-step pairs like the forward), every masked gradient written once (the left
// operands of the weight gradients), and -- DX -- d enc = d y_0 W_0 for the encodings' chain rule.
struct MotionDxArgs {
    const float* wt;          // sections [128/4][128][4] of motion_net.0 (inputs 84 padded to 128), .2, .4, .6, .8; then the head [3][128]
    const float* d_flow;      // [rows][4]: dLoss / d flow in columns 0 .. 2
    const uint32_t* bits;
    int64_t bits_stride;
    float* dy[5];
    int32_t ld_dy[5];
    float* denc;              // [rows][ld_denc], 96 columns written (84 used) -- DX only
    int32_t ld_denc;
    int64_t rows;
    uint32_t o_l[5], o_head;
};
template <bool DX>
__global__ __launch_bounds__(WV_THREADS, 1) void train_motion_dx_kernel(MotionDxArgs a) {
    __shared__ __attribute__((aligned(16))) float heads[3 * 128];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    for (int i = tid; i < 384; i += WV_THREADS) heads[i] = a.wt[a.o_head + i];
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = weight_rsrc(a.wt);
    constexpr uint32_t WSTEP128 = 2u * 128u * 16u;
    const LaneOfs wl128p = lane_offsets_paired((uint32_t)(h * 128 + c) * 16u, WSTEP128);
    const NextOfs nx128p{(uint32_t)(h * 128 + c) * 16u, true, WSTEP128};
    f32x16 acc[8], in[8];
    float4 wa[8], wb[8];
    const int64_t items = (a.rows + WV_ITEM - 1) / WV_ITEM;
    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int64_t row = item * WV_ITEM + wave * WV_ROWS + c;
        const bool valid = row < a.rows;
        const int64_t r = valid ? row : 0;
        auto yrow = [&](int s) { return a.dy[s] + r * a.ld_dy[s] + 4 * h; };
        load_w<8>(wa, rsrc, wl128p, a.o_l[4] * 4u);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) g = *reinterpret_cast<const float4*>(a.d_flow + r * 4);
        uint4 bw[5];
#pragma unroll
        for (int sidx = 0; sidx < 5; ++sidx) {
            uint2 t = make_uint2(0u, 0u);
            if (valid) t = *reinterpret_cast<const uint2*>(a.bits + (size_t)sidx * (size_t)a.bits_stride + (size_t)r * 4u + 2u * (uint32_t)h);
            bw[sidx] = make_uint4(t.x, t.y, 0u, 0u);
        }
        // the flow head backwards: d a4[f] = sum_o d flow[o] W[o][f] (motion_net.10)
        {
            const float4* w4 = reinterpret_cast<const float4*>(heads) + h;
#pragma unroll
            for (int fb = 0; fb < 4; ++fb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w0 = w4[2 * (fb * 4 + q)], w1 = w4[32 + 2 * (fb * 4 + q)], w2 = w4[64 + 2 * (fb * 4 + q)];
                    acc[fb][4 * q + 0] = g.x * w0.x + g.y * w1.x + g.z * w2.x;
                    acc[fb][4 * q + 1] = g.x * w0.y + g.y * w1.y + g.z * w2.y;
                    acc[fb][4 * q + 2] = g.x * w0.z + g.y * w1.z + g.z * w2.z;
                    acc[fb][4 * q + 3] = g.x * w0.w + g.y * w1.w + g.z * w2.w;
                }
        }
        mask_boundary<4>(acc, in, bw[4], yrow(4), valid);
#pragma unroll
        for (int l = 4; l >= 1; --l) {     // through motion_net.8 .. .2
            const uint32_t next = l > 1 ? a.o_l[l - 1] : (DX ? a.o_l[0] : a.o_l[4]);
            segment_p<8, 16, 0, 8>(acc, in, wa, wb, rsrc, wl128p, a.o_l[l] * 4u, WSTEP128, nx128p, next * 4u);
            mask_boundary<4>(acc, in, bw[l - 1], yrow(l - 1), valid);
        }
        if constexpr (DX) {
            segment_p<8, 16, 0, 8>(acc, in, wa, wb, rsrc, wl128p, a.o_l[0] * 4u, WSTEP128, nx128p, a.o_l[4] * 4u);
            if (valid) {
                float4* yp = reinterpret_cast<float4*>(a.denc + r * a.ld_denc + 4 * h);
#pragma unroll
                for (int fb = 0; fb < 3; ++fb)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        yp[fb * 8 + 2 * q] = make_float4(acc[fb][4 * q + 0], acc[fb][4 * q + 1], acc[fb][4 * q + 2], acc[fb][4 * q + 3]);
            }
        }
    }
}

}  // namespace stnerf

using namespace stnerf;

static int train_spacenet_fwd(bool bf16x3, int kind, const void* packed, int64_t n_rays, int ns, const float* xyz, int64_t xyz_ray_stride,
                              const float* dirs, int64_t dirs_ray_stride, const float* times, int64_t times_ray_stride, float* raw,
                              int64_t raw_ray_stride, float* const* act_host, const int32_t* ld_act_host, float* pe, int32_t ld_pe,
                              uint32_t* relu_bits, int64_t relu_bits_stride, uint32_t* queue, float* ray_bias, stnerf_stream_t stream) {
    STNERF_REQUIRE(packed && xyz && dirs && raw && act_host && ld_act_host && pe && queue && ray_bias, "train_spacenet_fwd: null pointer");
    STNERF_REQUIRE(kind == STNERF_NET_SPACE || kind == STNERF_NET_SPACE_TIME, "train_spacenet_fwd: kind %d (deep_rgb networks take the per-layer path)", kind);
    STNERF_REQUIRE(n_rays >= 0 && ns >= 1 && (raw_ray_stride & 3) == 0, "train_spacenet_fwd: bad shape");
    STNERF_REQUIRE(kind == STNERF_NET_SPACE || times, "train_spacenet_fwd: this network needs its frame-id column");
    STNERF_REQUIRE(n_rays * ns <= 0x7fffff00ll, "train_spacenet_fwd: %lld rows (split the batch)", (long long)(n_rays * ns));
    if (n_rays == 0) return STNERF_OK;
    float* bufs[8];
    int32_t lds[8];
    for (int i = 0; i < 8; ++i) {
        STNERF_REQUIRE(act_host[i] && ((uintptr_t)act_host[i] & 15) == 0 && (ld_act_host[i] & 3) == 0 && ld_act_host[i] >= (i == 7 ? 128 : 256),
                       "train_spacenet_fwd: activation matrix %d must be 16-byte aligned with a row stride that is a multiple of 4 floats", i);
        bufs[i] = act_host[i];
        lds[i] = ld_act_host[i];
    }
    STNERF_REQUIRE(((uintptr_t)pe & 15) == 0 && (ld_pe & 3) == 0 && ld_pe >= 64, "train_spacenet_fwd: the PE matrix needs 64 columns, 16-byte aligned");
    // (the split-bf16 kernel fetches its weight stream by 16-byte LDS-DMA from 1 KB-aligned sections)
    STNERF_REQUIRE(!bf16x3 || ((uintptr_t)packed & 1023) == 0, "train_spacenet_fwd_bf16x3: the packed network must be 1 KB aligned");
    STNERF_REQUIRE(!relu_bits || (((uintptr_t)relu_bits & 15) == 0 && (relu_bits_stride & 3) == 0 && relu_bits_stride >= n_rays * ns * 8),
                   "train_spacenet_fwd: relu_bits must be 16-byte aligned, its stage stride a multiple of 4 words and >= 8 x rows");
    const bool timed = kind == STNERF_NET_SPACE_TIME;
    if (const int rc = launch_ray_bias(kind, static_cast<const float*>(packed), n_rays, nullptr, nullptr, dirs, dirs_ray_stride, times,
                                       times_ray_stride, ray_bias, as_stream(stream)))
        return rc;
    StageArgs a;
    memset(&a, 0, sizeof(a));
    a.layer[0] = StageLayer{static_cast<const float*>(packed), nullptr, nullptr, nullptr, xyz, raw, times, timed ? 1 : 0, 0, ray_bias};
    a.n_layers = 1;
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
    if (bf16x3) {
        StoreTapArgs t;
        t.bits = relu_bits;
        t.bits_stride = relu_bits_stride;
        for (int i = 0; i < 8; ++i) {
            t.buf[i] = bufs[i];
            t.ld[i] = lds[i];
        }
        t.pe = pe;
        t.ld_pe = ld_pe;
        return launch_bf16x3_stage_store(a, t, cus, as_stream(stream));
    }
    return launch_wave_stage_store(a, bufs, lds, pe, ld_pe, relu_bits, relu_bits_stride, cus, as_stream(stream));
}

extern "C" int stnerf_train_spacenet_fwd(int kind, const void* packed, int64_t n_rays, int ns, const float* xyz, int64_t xyz_ray_stride,
                                         const float* dirs, int64_t dirs_ray_stride, const float* times, int64_t times_ray_stride,
                                         float* raw, int64_t raw_ray_stride, float* const* act_host, const int32_t* ld_act_host, float* pe,
                                         int32_t ld_pe, uint32_t* relu_bits, int64_t relu_bits_stride, uint32_t* queue, float* ray_bias, stnerf_stream_t stream) {
    return train_spacenet_fwd(false, kind, packed, n_rays, ns, xyz, xyz_ray_stride, dirs, dirs_ray_stride, times, times_ray_stride, raw,
                              raw_ray_stride, act_host, ld_act_host, pe, ld_pe, relu_bits, relu_bits_stride, queue, ray_bias, stream);
}

extern "C" int stnerf_train_spacenet_fwd_bf16x3(int kind, const void* packed, int64_t n_rays, int ns, const float* xyz, int64_t xyz_ray_stride,
                                                const float* dirs, int64_t dirs_ray_stride, const float* times, int64_t times_ray_stride,
                                                float* raw, int64_t raw_ray_stride, float* const* act_host, const int32_t* ld_act_host,
                                                float* pe, int32_t ld_pe, uint32_t* relu_bits, int64_t relu_bits_stride, uint32_t* queue,
                                                float* ray_bias, stnerf_stream_t stream) {
    return train_spacenet_fwd(true, kind, packed, n_rays, ns, xyz, xyz_ray_stride, dirs, dirs_ray_stride, times, times_ray_stride, raw,
                              raw_ray_stride, act_host, ld_act_host, pe, ld_pe, relu_bits, relu_bits_stride, queue, ray_bias, stream);
}

extern "C" int stnerf_train_spacenet_dx(const float* wt, const uint32_t* offsets_host /* rgb1, l[0..6], wsigma, wrgb2 */, const float* d_raw,
                                        int64_t rows, const uint32_t* relu_bits, int64_t relu_bits_stride, float* const* dy_host,
                                        const int32_t* ld_dy_host, float* dpe, int32_t ld_dpe, stnerf_stream_t stream) {
    STNERF_REQUIRE(wt && offsets_host && d_raw && relu_bits && dy_host && ld_dy_host, "train_spacenet_dx: null pointer");
    STNERF_REQUIRE(rows >= 0 && rows <= 0x7fffff00ll, "train_spacenet_dx: %lld rows (split the batch)", (long long)rows);
    STNERF_REQUIRE((((uintptr_t)wt | (uintptr_t)d_raw | (uintptr_t)relu_bits) & 15) == 0 && (relu_bits_stride & 3) == 0 && relu_bits_stride >= rows * 8,
                   "train_spacenet_dx: weights, d_raw and relu_bits must be 16-byte aligned (stage stride: a multiple of 4 words, >= 8 x rows)");
    if (rows == 0) return STNERF_OK;
    DxArgs a;
    memset(&a, 0, sizeof(a));
    a.wt = wt;
    a.d_raw = d_raw;
    a.rows = rows;
    a.bits = relu_bits;
    a.bits_stride = relu_bits_stride;
    for (int i = 0; i < 8; ++i) {
        const int width = i == 7 ? 128 : 256;
        STNERF_REQUIRE(dy_host[i] && ((uintptr_t)dy_host[i] & 15) == 0 && (ld_dy_host[i] & 3) == 0 && ld_dy_host[i] >= width,
                       "train_spacenet_dx: matrix %d must be 16-byte aligned with a row stride that is a multiple of 4 floats", i);
        a.dy[i] = dy_host[i];
        a.ld_dy[i] = ld_dy_host[i];
    }
    a.o_rgb1 = offsets_host[0];
    for (int i = 0; i < 7; ++i) a.o_l[i] = offsets_host[1 + i];
    a.o_wsigma = offsets_host[8];
    a.o_wrgb2 = offsets_host[9];
    for (int i = 0; i < 10; ++i) STNERF_REQUIRE((offsets_host[i] & 3) == 0, "train_spacenet_dx: section %d is not 16-byte aligned", i);
    a.dpe = dpe;
    a.ld_dpe = ld_dpe;
    STNERF_REQUIRE(!dpe || (((uintptr_t)dpe & 15) == 0 && (ld_dpe & 3) == 0 && ld_dpe >= 64), "train_spacenet_dx: d PE needs 64 columns, 16-byte aligned");
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int64_t items = (rows + WV_ITEM - 1) / WV_ITEM;
    const int grid = (int)(items < cus ? items : cus);
    if (dpe)
        hipLaunchKernelGGL(train_space_dx_kernel<true>, dim3(grid), dim3(WV_THREADS), 0, as_stream(stream), a);
    else
        hipLaunchKernelGGL(train_space_dx_kernel<false>, dim3(grid), dim3(WV_THREADS), 0, as_stream(stream), a);
    STNERF_CHECK_LAUNCH("train_spacenet_dx");
    return STNERF_OK;
}

extern "C" int stnerf_train_motionnet_fwd(const void* packed, const float* xt, int64_t rows, int motion_flags, float* flow, float* enc,
                                          int32_t ld_enc, float* const* act_host, const int32_t* ld_act_host, uint32_t* relu_bits,
                                          int64_t relu_bits_stride, stnerf_stream_t stream) {
    STNERF_REQUIRE(packed && xt && flow && enc && act_host && ld_act_host && relu_bits, "train_motionnet_fwd: null pointer");
    STNERF_REQUIRE(rows >= 0 && rows <= 0x7fffff00ll, "train_motionnet_fwd: %lld rows (split the batch)", (long long)rows);
    STNERF_REQUIRE((((uintptr_t)packed | (uintptr_t)xt | (uintptr_t)enc) & 15) == 0 && (ld_enc & 3) == 0 && ld_enc >= 88,
                   "train_motionnet_fwd: weights, rows and the encoding matrix (>= 88 columns, a multiple of 4) must be 16-byte aligned");
    STNERF_REQUIRE(((uintptr_t)relu_bits & 7) == 0 && (relu_bits_stride & 1) == 0 && relu_bits_stride >= rows * 4,
                   "train_motionnet_fwd: relu_bits must be 8-byte aligned, its stage stride even and >= 4 x rows");
    STNERF_REQUIRE((motion_flags & ~STNERF_MOTION_PLAIN_TIME) == 0, "train_motionnet_fwd: flags %d", motion_flags);
    if (rows == 0) return STNERF_OK;
    MotionFwdArgs a;
    memset(&a, 0, sizeof(a));
    a.net = static_cast<const float*>(packed);
    a.xt = xt;
    a.rows = rows;
    a.flags = motion_flags;
    a.tap.enc = enc;
    a.tap.ld_enc = ld_enc;
    for (int i = 0; i < 5; ++i) {
        STNERF_REQUIRE(act_host[i] && ((uintptr_t)act_host[i] & 15) == 0 && (ld_act_host[i] & 3) == 0 && ld_act_host[i] >= 128,
                       "train_motionnet_fwd: activation matrix %d must be 16-byte aligned with a row stride that is a multiple of 4 floats", i);
        a.tap.act[i] = act_host[i];
        a.tap.ld_act[i] = ld_act_host[i];
    }
    a.tap.bits = relu_bits;
    a.tap.bits_stride = relu_bits_stride;
    a.tap.flow = flow;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int64_t items = (rows + WV_ITEM - 1) / WV_ITEM;
    const int grid = (int)(items < cus ? items : cus);
    if (const int rc = reserve_dynamic_lds(reinterpret_cast<const void*>(train_motion_fwd_kernel), WV_LDS, "train_motionnet_fwd")) return rc;
    hipLaunchKernelGGL(train_motion_fwd_kernel, dim3(grid), dim3(WV_THREADS), WV_LDS, as_stream(stream), a);
    STNERF_CHECK_LAUNCH("train_motionnet_fwd");
    return STNERF_OK;
}

extern "C" int stnerf_train_motionnet_dx(const float* wt, const uint32_t* offsets_host /* motion_net.0, .2, .4, .6, .8, head */, const float* d_flow,
                                         int64_t rows, const uint32_t* relu_bits, int64_t relu_bits_stride, float* const* dy_host,
                                         const int32_t* ld_dy_host, float* denc, int32_t ld_denc, stnerf_stream_t stream) {
    STNERF_REQUIRE(wt && offsets_host && d_flow && relu_bits && dy_host && ld_dy_host, "train_motionnet_dx: null pointer");
    STNERF_REQUIRE(rows >= 0 && rows <= 0x7fffff00ll, "train_motionnet_dx: %lld rows (split the batch)", (long long)rows);
    STNERF_REQUIRE((((uintptr_t)wt | (uintptr_t)d_flow) & 15) == 0 && ((uintptr_t)relu_bits & 7) == 0 && (relu_bits_stride & 1) == 0 &&
                       relu_bits_stride >= rows * 4,
                   "train_motionnet_dx: weights and d_flow must be 16-byte aligned, relu_bits 8-byte (stage stride: even, >= 4 x rows)");
    if (rows == 0) return STNERF_OK;
    MotionDxArgs a;
    memset(&a, 0, sizeof(a));
    a.wt = wt;
    a.d_flow = d_flow;
    a.bits = relu_bits;
    a.bits_stride = relu_bits_stride;
    a.rows = rows;
    for (int i = 0; i < 5; ++i) {
        STNERF_REQUIRE(dy_host[i] && ((uintptr_t)dy_host[i] & 15) == 0 && (ld_dy_host[i] & 3) == 0 && ld_dy_host[i] >= 128,
                       "train_motionnet_dx: matrix %d must be 16-byte aligned with a row stride that is a multiple of 4 floats", i);
        a.dy[i] = dy_host[i];
        a.ld_dy[i] = ld_dy_host[i];
        a.o_l[i] = offsets_host[i];
    }
    a.o_head = offsets_host[5];
    for (int i = 0; i < 6; ++i) STNERF_REQUIRE((offsets_host[i] & 3) == 0, "train_motionnet_dx: section %d is not 16-byte aligned", i);
    a.denc = denc;
    a.ld_denc = ld_denc;
    STNERF_REQUIRE(!denc || (((uintptr_t)denc & 15) == 0 && (ld_denc & 3) == 0 && ld_denc >= 96), "train_motionnet_dx: d enc needs 96 columns, 16-byte aligned");
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int64_t items = (rows + WV_ITEM - 1) / WV_ITEM;
    const int grid = (int)(items < cus ? items : cus);
    if (denc)
        hipLaunchKernelGGL(train_motion_dx_kernel<true>, dim3(grid), dim3(WV_THREADS), 0, as_stream(stream), a);
    else
        hipLaunchKernelGGL(train_motion_dx_kernel<false>, dim3(grid), dim3(WV_THREADS), 0, as_stream(stream), a);
    STNERF_CHECK_LAUNCH("train_motionnet_dx");
    return STNERF_OK;
}
