// Backward pass of the two networks (modeling/spacenet.py:101-160 and modeling/motion_net.py:34-71
// under engine/layered_trainer.py:192-217's loss.backward()): the kernels behind stnerf_amd.modeling.autograd.
//
// The inference kernels keep a network's activations in registers and never store them.  Training needs, per nn.Linear
// y = W x + b,   dx = W^T dy,   dW += dy x^T (a contraction over SAMPLES),   db += sum dy,
// i.e. the input x of every layer next to its dy.  The backward therefore RECOMPUTES the forward chunk by chunk (a chunk of
// samples at a time, so that the stored activations are a bounded workspace and not 14 KB per sample of the whole batch),
// layer by layer, with one f32 MFMA GEMM kernel in three operand flavours:
//
//   forward   Y[M,N]  = act(X[M,K] W[N,K]^T + b)        "NT"  (W in the reference's nn.Linear layout: out x in, row-major)
//   dX        dX[M,K] = (dY[M,N] W[N,K]) * (X_in > 0)    "NN"  (the ReLU of the producing layer folded into the epilogue;
//                                                               optionally accumulated into dX: skip connection / two consumers)
//   dW        dW[N,K] += dY[M,N]^T X[M,K]                "TN"  (split over the samples: every workgroup reduces its slice of M
//                                                               into a partial tile, a second kernel sums the partials in a
//                                                               fixed order -- deterministic, no atomics)
//
// v_mfma_f32_32x32x2_f32 (exact fp32 products and accumulation: gradients at the accuracy of an fp32 torch.autograd).
// A workgroup of 4 waves owns a 128 x 128 tile of the output, a wave 64 x 64 of it (2 x 2 MFMA tiles = 64 accumulator
// registers) -- 128 x 256 / 64 x 128 where the output is wider than 128 columns, so that the sample-major operand is read
// from HBM once; operands are staged through LDS in 32-deep K slices as 16-byte quads of consecutive k (swizzled [row][36]
// tiles, see lds_quad): one ds_read_b128 per operand block feeds four MFMAs, the operands of the next 8 k are fetched in
// front of the current MFMAs, the next slice's global loads are in flight while the current one is multiplied.
//
// Positional encodings (utils/dimension_kernel.py:54-73): forward into a strided destination (a column block of a layer's
// input matrix: the skip connection and rgb_net's input are built in place, no torch.cat), per-ray encodings broadcast to
// the ray's samples with rgb_net's leading ReLU applied (modeling/spacenet.py:80-86), MotionNet's fractional-time lerp
// (modeling/motion_net.py:52-60); backward by the chain rule d/dx sin(2^f x) = 2^f cos(2^f x).
#include "common.h"
#include "mlp_common.h"

namespace stnerf {
namespace {

constexpr int GBM = 128, GBK = 32, GLD = GBK + 4;   // (the N extent of a tile is a template parameter: 128 or 256)

struct GemmArgs {
    const float* A;   // forward: X [M][lda];   dX: dY [M][lda];   dW: dY [Ksamples][lda] (read transposed)
    const float* B;   // forward: W [N][ldb];   dX: W [Kred][ldb]; dW: X [Ksamples][ldb]
    float* C;         // [M][ldc] (dW: the partial tiles, [split][M][N] dense)
    int64_t lda, ldb, ldc;
    int M, N, K;      // output M x N, reduction length K
    const float* bias;   // forward: [N] or null
    const float* mask;   // dX: [M][ldmask] (the layer's stored post-ReLU INPUT: gradient passes where it is > 0) or null
    int64_t ldmask;
    int relu, accumulate;
    int k_per_split;  // dW: reduction range of one blockIdx.z
};

// LDS tile of one operand: [row][36 floats] = 8 quads of 4 consecutive k + one pad quad (rows stay 16-byte aligned), quad q
// of row r stored at quad position q ^ ((r >> 2) & 7).  Everything moves as 16-byte vectors:
//   * the MFMA operand fetch: lane (h, c) reads quad 2 s + h of row c -- ONE ds_read_b128 feeds the four K = 2 steps of
//     k = 8 s + 4 h + {0..3} (the order of k inside the reduction is free as long as both operands use the same one);
//     16 consecutive rows at one quad fall on all 32 banks twice: the minimum for 64 dwords;
//   * a k-contiguous source row (forward / dX: activations, nn.Linear weights) arrives as one global float4 = one quad;
//   * a row-contiguous source (dW's operands, dX's weights) arrives as float4s along the rows for four consecutive k: a
//     4 x 4 register transpose turns them into four quads.  Those go to rows 4 apart x 16 lanes -- a stride of 144 dwords,
//     i.e. two bank groups without the swizzle, all eight with it.
__device__ __forceinline__ int lds_quad(int row, int q) { return row * GLD + 4 * (q ^ ((row >> 2) & 7)); }

// One operand tile [ROWS][32 k]: global -> registers with BUFFER loads: a resource descriptor over the whole matrix, a
// per-lane byte offset that never changes and the slice's wave-uniform offset in a scalar register -- no vector
// instruction per load (on gfx950 every VALU instruction takes its cycles out of the f32 MFMA stream, whichever wave issues
// it: profiles/r01_dual_issue_microbench.md; with per-load 64-bit address arithmetic and edge selects this kernel ran at
// 0.45 of the MFMA peak), and the hardware's range check returns 0 beyond the end of the matrix: rows past R and -- for the
// row-contiguous flavour, whose k is the row index -- the tail of the reduction need no guard at all.
//   K_CONTIG: element (r, k) = src[(r0 + r) * ld + k0 + k];  thread -> (row t >> 3 (+ 32 i), quad t & 7)
//  !K_CONTIG: element (r, k) = src[(k0 + k) * ld + r0 + r];  thread -> unit f = t (+ 256 u): quad f / (ROWS / 4), rows 4 (f % (ROWS / 4)) ..+3
using i32x4 = __attribute__((ext_vector_type(4))) int;
struct Operand {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t step;                  // bytes per 32-k slice (wave-uniform)
};
__device__ __forceinline__ Operand make_operand(const float* src, int64_t ld, int rows_total, int cols_total, bool k_contig) {
    // bytes up to the end of the last row's round4(cols) floats (what the caller guarantees to be allocated)
    const int64_t bytes = ((int64_t)(rows_total - 1) * ld + ((cols_total + 3) & ~3)) * 4;
    Operand o;
    o.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)bytes, 0x00020000);
    o.step = k_contig ? GBK * 4u : (uint32_t)(GBK * ld * 4);
    return o;
}
// 16-byte vectors per thread and slice.  (A row-contiguous tile moves in units of 4 k x 4 rows = four vectors: 2 ROWS units, so a
// 64-row tile -- the narrow items of stnerf_train_dw_batch -- keeps only threads t < 128 busy: tile_active.)
constexpr int tile_vecs(bool k_contig, int rows) { return k_contig || rows >= 128 ? rows / 32 : 4; }
template <bool K_CONTIG, int ROWS>
__device__ __forceinline__ bool tile_active(int t) { return K_CONTIG || ROWS >= 128 || t < 2 * ROWS; }
template <bool K_CONTIG, int ROWS>
__device__ __forceinline__ void lane_offsets(uint32_t (&off)[tile_vecs(K_CONTIG, ROWS)], int64_t ld, int r0, int t) {
    if (K_CONTIG) {
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) off[i] = (uint32_t)(((int64_t)(r0 + (t >> 3) + 32 * i) * ld + 4 * (t & 7)) * 4);
    } else {
        constexpr int RQ = ROWS / 4;
#pragma unroll
        for (int u = 0; u < tile_vecs(false, ROWS) / 4; ++u) {
            const int f = t + 256 * u;
#pragma unroll
            for (int j = 0; j < 4; ++j) off[4 * u + j] = (uint32_t)(((int64_t)(4 * (f / RQ) + j) * ld + r0 + 4 * (f % RQ)) * 4);
        }
    }
}
template <int NV>
__device__ __forceinline__ void load_tile(float4 (&v)[NV], const Operand& op, const uint32_t (&off)[NV], uint32_t slice_bytes) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const i32x4 r = __builtin_amdgcn_raw_buffer_load_b128(op.rsrc, off[i], slice_bytes, 0);
        v[i] = make_float4(__int_as_float(r.x), __int_as_float(r.y), __int_as_float(r.z), __int_as_float(r.w));
    }
}
// ... and into LDS.  k_left = elements of the reduction range left from this slice's first k: only the k-contiguous flavour
// can see a range end inside its rows (K % 32 != 0: the last slice), and only then (TAIL) are the elements selected.
template <bool K_CONTIG, int ROWS, bool TAIL>
__device__ __forceinline__ void store_tile(float* lds, const float4 (&v)[tile_vecs(K_CONTIG, ROWS)], int k_left, int t) {
    if (K_CONTIG) {
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            float4 o = v[i];
            if (TAIL) {
                const int k = 4 * (t & 7);
                o.x = k < k_left ? o.x : 0.f;
                o.y = k + 1 < k_left ? o.y : 0.f;
                o.z = k + 2 < k_left ? o.z : 0.f;
                o.w = k + 3 < k_left ? o.w : 0.f;
            }
            *reinterpret_cast<float4*>(lds + lds_quad((t >> 3) + 32 * i, t & 7)) = o;
        }
    } else {
        constexpr int RQ = ROWS / 4;
#pragma unroll
        for (int u = 0; u < tile_vecs(false, ROWS) / 4; ++u) {
            const int f = t + 256 * u, q = f / RQ, row = 4 * (f % RQ);
            const float4 x0 = v[4 * u], x1 = v[4 * u + 1], x2 = v[4 * u + 2], x3 = v[4 * u + 3];
            *reinterpret_cast<float4*>(lds + lds_quad(row, q)) = make_float4(x0.x, x1.x, x2.x, x3.x);
            *reinterpret_cast<float4*>(lds + lds_quad(row + 1, q)) = make_float4(x0.y, x1.y, x2.y, x3.y);
            *reinterpret_cast<float4*>(lds + lds_quad(row + 2, q)) = make_float4(x0.z, x1.z, x2.z, x3.z);
            *reinterpret_cast<float4*>(lds + lds_quad(row + 3, q)) = make_float4(x0.w, x1.w, x2.w, x3.w);
        }
    }
}

// C = A' B' with A'[m][k], B'[k][n] read as the template flags say; MODE 0: forward epilogue (bias, ReLU), 1: dX epilogue
// (mask, accumulate), 2: dW partial tile.  BN = 256 where the output is wider than 128: the A' operand (the large,
// sample-major matrix in every flavour but dW's) is then read once instead of twice.
// One workgroup's tile: rows m0 .. m0 + 127, columns n0 .. n0 + BN - 1 of the output, reduction range [k_begin, k_end) (k_begin a
// multiple of 32), written to out[m][ldc].
template <bool A_KC, bool B_KC, int MODE, int BN>
__device__ __forceinline__ void gemm_tile(const GemmArgs& a, float* smem_gemm, int m0, int n0, int k_begin, int k_end, float* out, int64_t ldc) {
    float* As = smem_gemm;
    float* Bs = smem_gemm + GBM * GLD;
    constexpr int NJ = BN / 64;                // 32-column MFMA tiles per wave
    constexpr int NVA = tile_vecs(A_KC, GBM), NVB = tile_vecs(B_KC, BN);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, c = lane & 31;
    const bool b_active = tile_active<B_KC, BN>(t);   // (wave-uniform; constant true but for the 64-column items)
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * (BN / 2);
    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // A' = [M rows][K] (k-contiguous) or [K rows][M] (row-contiguous); B' likewise with N
    const Operand opa = make_operand(a.A, a.lda, A_KC ? a.M : a.K, A_KC ? a.K : a.M, A_KC);
    const Operand opb = make_operand(a.B, a.ldb, B_KC ? a.N : a.K, B_KC ? a.K : a.N, B_KC);
    uint32_t offa[NVA], offb[NVB];
    lane_offsets<A_KC, GBM>(offa, a.lda, m0, t);
    lane_offsets<B_KC, BN>(offb, a.ldb, n0, t);
    uint32_t sa = (uint32_t)(k_begin / GBK) * opa.step, sb = (uint32_t)(k_begin / GBK) * opb.step;   // this slice's scalar offsets
    float4 ra[NVA], rb[NVB];
    load_tile(ra, opa, offa, sa);
    if (b_active) load_tile(rb, opb, offb, sb);
    // this lane's operand rows (the swizzle term (row >> 2) & 7 = (c >> 2) & 7 is the same for every 32-row block)
    const float* arow = As + (wm + c) * GLD;
    const float* brow = Bs + (wn + c) * GLD;
    const int sw = (c >> 2) & 7;
    auto fetch = [&](int s_, float4 (&av)[2], float4 (&bv)[NJ]) {
        const int qo = 4 * ((2 * s_ + h) ^ sw);
#pragma unroll
        for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const float4*>(arow + 32 * i * GLD + qo);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bv[j] = *reinterpret_cast<const float4*>(brow + 32 * j * GLD + qo);
    };
    auto mma = [&](const float4 (&av)[2], const float4 (&bv)[NJ]) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const float x = e == 0 ? av[i].x : e == 1 ? av[i].y : e == 2 ? av[i].z : av[i].w;
                    const float y = e == 0 ? bv[j].x : e == 1 ? bv[j].y : e == 2 ? bv[j].z : bv[j].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i][j], 0, 0, 0);
                }
    };
    auto multiply = [&]() {   // 4 steps of 8 k on the slice in LDS: the operands of step s + 1 are fetched in front of the MFMAs of step s
        float4 av0[2], bv0[NJ], av1[2], bv1[NJ];
        fetch(0, av0, bv0);
        fetch(1, av1, bv1);
        __builtin_amdgcn_sched_barrier(0);
        mma(av0, bv0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(2, av0, bv0);
        __builtin_amdgcn_sched_barrier(0);
        mma(av1, bv1);
        __builtin_amdgcn_sched_barrier(0);
        fetch(3, av1, bv1);
        __builtin_amdgcn_sched_barrier(0);
        mma(av0, bv0);
        mma(av1, bv1);
    };
    for (int k0 = k_begin; k0 < k_end; k0 += GBK) {
        const bool tail = (A_KC || B_KC) && k0 + GBK > k_end;     // (uniform) the reduction range ends inside this slice
        __syncthreads();                       // the previous slice has been consumed
        if (tail) {
            store_tile<A_KC, GBM, true>(As, ra, k_end - k0, t);
            if (b_active) store_tile<B_KC, BN, true>(Bs, rb, k_end - k0, t);
        } else {
            store_tile<A_KC, GBM, false>(As, ra, 0, t);
            if (b_active) store_tile<B_KC, BN, false>(Bs, rb, 0, t);
        }
        __syncthreads();
        if (k0 + GBK < k_end) {                // the next slice's loads fly under this slice's MFMAs
            sa += opa.step;
            sb += opb.step;
            load_tile(ra, opa, offa, sa);
            if (b_active) load_tile(rb, opb, offb, sb);
        }
        multiply();
    }
    // accumulator register 4 q + r of lane (h, c): row 8 q + 4 h + r, column c of the 32 x 32 tile
    const bool inside = m0 + GBM <= a.M && n0 + BN <= a.N;       // (uniform) no edge of the matrix in this tile
    // ---- the usual case -- a tile inside the matrix, 16-byte aligned rows: the wave turns its 64 x BN/2 accumulators around
    // through LDS, 16 rows at a time (the operand tiles are dead: every wave has its own 8.5 KB there), and bias / ReLU / mask
    // / accumulate / store work on float4s along the row: a lane stores 16 bytes, 32 lanes one contiguous 512-byte piece of a
    // row (straight from the accumulators a store instruction writes 2 x 128 bytes as single dwords).
    const bool vec = inside && (ldc & 3) == 0 && ((uintptr_t)out & 15) == 0 &&
                     (MODE != 1 || !a.mask || ((a.ldmask & 3) == 0 && ((uintptr_t)a.mask & 15) == 0)) &&
                     (MODE != 0 || !a.bias || ((uintptr_t)a.bias & 15) == 0);
    if (vec) {
        constexpr int WN = BN / 2, SLD = WN + 4;                  // the wave's columns; staging row stride (floats)
        static_assert(16 * SLD * 4 * 4 <= (GBM + BN) * GLD * 4, "epilogue staging must fit the operand tiles");
        __syncthreads();                                          // every wave is done reading the operand tiles
        float* stage = smem_gemm + wave * 16 * SLD;
        const int64_t row0 = m0 + wm, col0 = n0 + wn;
#pragma unroll
        for (int R = 0; R < 4; ++R) {                             // rows 16 R .. 16 R + 15 of the wave's 64
            constexpr int dummy = 0;
            (void)dummy;
            const int i = R >> 1, qb = 2 * (R & 1);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                    for (int r = 0; r < 4; ++r) stage[(8 * qq + 4 * h + r) * SLD + 32 * j + c] = acc[i][j][4 * (qb + qq) + r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int u = 0; u < 16 * WN / 4 / 64; ++u) {          // 16 rows x WN / 4 float4s over 64 lanes
                const int idx = lane + 64 * u, rr = idx / (WN / 4), c4 = 4 * (idx % (WN / 4));
                float4 v = *reinterpret_cast<const float4*>(stage + rr * SLD + c4);
                const int64_t m = row0 + 16 * R + rr, n = col0 + c4;
                if (MODE == 0) {
                    if (a.bias) {
                        const float4 bb = *reinterpret_cast<const float4*>(a.bias + n);
                        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
                    }
                    if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                } else if (MODE == 1) {
                    if (a.mask) {
                        const float4 mk = *reinterpret_cast<const float4*>(a.mask + m * a.ldmask + n);
                        v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
                    }
                    if (a.accumulate) {
                        const float4 o = *reinterpret_cast<const float4*>(out + m * ldc + n);
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                }
                *reinterpret_cast<float4*>(out + m * ldc + n) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + wn + 32 * j + c;
            if (!inside && n >= a.N) continue;
            float bias = 0.f;
            if (MODE == 0 && a.bias) bias = a.bias[n];
            float* o = out + (int64_t)(m0 + wm + 32 * i + 4 * h) * ldc + n;
            const float* mk = MODE == 1 && a.mask ? a.mask + (int64_t)(m0 + wm + 32 * i + 4 * h) * a.ldmask + n : nullptr;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int dm = 8 * q + r;
                    if (!inside && m0 + wm + 32 * i + 4 * h + dm >= a.M) continue;
                    float v = acc[i][j][4 * q + r];
                    if (MODE == 0) {
                        v += bias;
                        if (a.relu) v = fmaxf(v, 0.f);
                    } else if (MODE == 1) {
                        if (mk && !(mk[dm * a.ldmask] > 0.f)) v = 0.f;
                        if (a.accumulate) v += o[dm * ldc];
                    }
                    o[dm * ldc] = v;
                }
        }
}

// (two workgroups per CU at least -- <= 256 registers: while one stores its tile, the other multiplies)
template <bool A_KC, bool B_KC, int MODE, int BN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void train_gemm_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem_gemm[];
    int k_begin = 0, k_end = a.K;
    if (MODE == 2) {                           // (k_per_split is a multiple of 32: a slice never straddles two splits)
        k_begin = blockIdx.z * a.k_per_split;
        k_end = min(a.K, k_begin + a.k_per_split);
    }
    // (row tiles on x: a grid's y extent ends at 65535 blocks)
    gemm_tile<A_KC, B_KC, MODE, BN>(a, smem_gemm, blockIdx.x * GBM, blockIdx.y * BN, k_begin, k_end,
                                    MODE == 2 ? a.C + (int64_t)blockIdx.z * a.M * a.N : a.C, MODE == 2 ? (int64_t)a.N : a.ldc);
}
template <bool A_KC, bool B_KC, int MODE>
int launch_gemm(const GemmArgs& a, int row_tiles, int grid_z, hipStream_t st, const char* what) {
    if (a.N > 128) {
        const dim3 grid(row_tiles, (a.N + 255) / 256, grid_z);
        hipLaunchKernelGGL((train_gemm_kernel<A_KC, B_KC, MODE, 256>), grid, dim3(256), (GBM + 256) * GLD * 4, st, a);
    } else {
        const dim3 grid(row_tiles, 1, grid_z);
        hipLaunchKernelGGL((train_gemm_kernel<A_KC, B_KC, MODE, 128>), grid, dim3(256), (GBM + 128) * GLD * 4, st, a);
    }
    STNERF_CHECK_LAUNCH(what);
    return STNERF_OK;
}

// dst[i] (+)= sum_z partial[z][i] in z order (the second half of the dW / db reductions).
__global__ void reduce_partials_kernel(const float* __restrict__ partial, int splits, int64_t count, int cols, float* __restrict__ dst,
                                       int64_t ld_dst, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float s = 0.f;
#pragma unroll 8
    for (int z = 0; z < splits; ++z) s += partial[(int64_t)z * count + i];
    float* d = dst + (i / cols) * ld_dst + (i % cols);
    *d = accumulate ? *d + s : s;
}

// Column sums of a row slice: partial[blockIdx.y][n] = sum over the block's rows of Y[m][n] (db = sum_samples dy).  A thread
// owns four consecutive columns (one 16-byte load per row: rows are 16-byte aligned and readable up to round4(N), values
// beyond N land in outputs nobody stores) and every 256 / CG-th row of the block's slice, four rows in flight; the row lanes
// are folded through LDS.  HBM-bound: the matrix is read once.
template <int CG>   // column groups (of 4) per block: 64 (N >= 256), 32, 16, ... -- 256 / CG row lanes
__device__ __forceinline__ void colsum_rows(const float* __restrict__ y, int64_t ld, int N, int col_block, int m_begin, int m_end,
                                            float* __restrict__ out, float4* red) {
    constexpr int RL = 256 / CG;
    const int cg = threadIdx.x % CG, rl = threadIdx.x / CG;
    const int col = 4 * (col_block * CG + cg);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    if (col < N) {
        const float* p = y + col;
        int m = m_begin + rl;
        for (; m + 3 * RL < m_end; m += 4 * RL) {
            const float4 a0 = *reinterpret_cast<const float4*>(p + (int64_t)m * ld), a1 = *reinterpret_cast<const float4*>(p + (int64_t)(m + RL) * ld);
            const float4 a2 = *reinterpret_cast<const float4*>(p + (int64_t)(m + 2 * RL) * ld), a3 = *reinterpret_cast<const float4*>(p + (int64_t)(m + 3 * RL) * ld);
            s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
            s1.x += a1.x; s1.y += a1.y; s1.z += a1.z; s1.w += a1.w;
            s2.x += a2.x; s2.y += a2.y; s2.z += a2.z; s2.w += a2.w;
            s3.x += a3.x; s3.y += a3.y; s3.z += a3.z; s3.w += a3.w;
        }
        for (; m < m_end; m += RL) {
            const float4 a0 = *reinterpret_cast<const float4*>(p + (int64_t)m * ld);
            s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
        }
    }
    red[rl * CG + cg] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
    __syncthreads();
    if (rl == 0 && col < N) {
        float4 t = red[cg];
        for (int r = 1; r < RL; ++r) {
            const float4 u = red[r * CG + cg];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        float* o = out + col;
        o[0] = t.x;
        if (col + 1 < N) o[1] = t.y;
        if (col + 2 < N) o[2] = t.z;
        if (col + 3 < N) o[3] = t.w;
    }
}
template <int CG>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ y, int64_t ld, int M, int N, int rows_per_block,
                                                     float* __restrict__ partial) {
    __shared__ float4 red[256];
    const int m_begin = blockIdx.y * rows_per_block;
    colsum_rows<CG>(y, ld, N, blockIdx.x, m_begin, min(M, m_begin + rows_per_block), partial + (int64_t)blockIdx.y * N, red);
}

// ---- positional encodings ------------------------------------------------------------------------------------------------
struct EncodeArgs {
    const float* x;       // [n_src][ldx]: dim input columns per source row
    int64_t ldx;
    int dim, n_freq, include_input;
    float* y;             // [rows][ldy]: the encoding goes to columns [col0, col0 + dim * (include_input + 2 n_freq))
    int64_t ldy;
    int col0;
    int64_t rows;
    int rows_per_src;     // 1: one source row per output row; ns: a ray's encoding repeated on its ns samples (:115,118)
    int relu;             // rgb_net's leading in-place ReLU on the encoded columns (modeling/spacenet.py:80)
    int lerp_col;         // >= 0: MotionNet's fractional-time lerp (motion_net.py:52-60) -- this input column is a frame id t:
                          // enc = (1 - w) PE(floor t) + w PE(floor t + 1), w = t - floor t, for that column's features
};
__global__ void train_encode_kernel(EncodeArgs a) {
    const int width = a.dim * (a.include_input + 2 * a.n_freq);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.rows * a.dim) return;
    const int64_t row = i / a.dim;
    const int j = (int)(i - row * a.dim);
    const float x = a.x[(row / a.rows_per_src) * a.ldx + j];
    float* y = a.y + row * a.ldy + a.col0;
    (void)width;
    const bool lerp = j == a.lerp_col;
    const float x0 = lerp ? floorf(x) : x, w = lerp ? x - x0 : 0.f;
    auto put = [&](int col, float v0, float v1) {
        float v = lerp ? (1.f - w) * v0 + w * v1 : v0;
        y[col] = a.relu ? fmaxf(v, 0.f) : v;
    };
    int col = 0;
    if (a.include_input) {
        put(j, x0, x0 + 1.f);
        col = a.dim;
    }
    float f = 1.f;
    for (int q = 0; q < a.n_freq; ++q, f *= 2.f) {
        float s0, c0, s1 = 0.f, c1 = 0.f;
        sincos_pe(x0 * f, s0, c0);
        if (lerp) sincos_pe((x0 + 1.f) * f, s1, c1);
        put(col + j, s0, s1);
        put(col + a.dim + j, c0, c1);
        col += 2 * a.dim;
    }
}

// dx[row][j] (+)= d_enc . d enc / d x  for the first `dim_out` input columns (MotionNet: xyz of [xyz, t]).
struct EncodeBwdArgs {
    const float* x;
    int64_t ldx;
    int dim, n_freq, include_input;
    const float* dy;      // [rows][ldy], encoding columns from col0
    int64_t ldy;
    int col0;
    int64_t rows;
    float* dx;            // [rows][lddx]
    int64_t lddx;
    int dim_out, accumulate;
};
__global__ void train_encode_bwd_kernel(EncodeBwdArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.rows * a.dim_out) return;
    const int64_t row = i / a.dim_out;
    const int j = (int)(i - row * a.dim_out);
    const float x = a.x[row * a.ldx + j];
    const float* dy = a.dy + row * a.ldy + a.col0;
    float g = 0.f;
    int col = 0;
    if (a.include_input) {
        g = dy[j];
        col = a.dim;
    }
    float f = 1.f;
    for (int q = 0; q < a.n_freq; ++q, f *= 2.f) {
        float s, c;
        sincos_pe(x * f, s, c);
        g = fmaf(f, c * dy[col + j] - s * dy[col + a.dim + j], g);
        col += 2 * a.dim;
    }
    float* d = a.dx + row * a.lddx + j;
    *d = a.accumulate ? *d + g : g;
}

}  // namespace
}  // namespace stnerf

using namespace stnerf;

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int stnerf_train_linear_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int64_t m, int n, int k,
                                       int relu, float* y, int64_t ldy, stnerf_stream_t stream) {
    STNERF_REQUIRE(x && w && y, "train_linear_fwd: null pointer");
    STNERF_REQUIRE(m >= 0 && m < (1ll << 31) && n >= 1 && k >= 1, "train_linear_fwd: bad shape m=%lld n=%d k=%d", (long long)m, n, k);
    STNERF_REQUIRE((ldx & 3) == 0 && (ldw & 3) == 0 && ldx >= k && ldw >= k && ldy >= n && aligned16(x) && aligned16(w),
                   "train_linear_fwd: x / w need 16-byte aligned rows (ld %% 4 == 0) of at least k floats");
    STNERF_REQUIRE(m * ldx < (1ll << 29) && (int64_t)n * ldw < (1ll << 29), "train_linear_fwd: operands of 2 GiB and more: split the batch");
    if (m == 0) return STNERF_OK;
    GemmArgs a{x, w, y, ldx, ldw, ldy, (int)m, n, k, bias, nullptr, 0, relu, 0, 0};
    return launch_gemm<true, true, 0>(a, (int)((m + GBM - 1) / GBM), 1, as_stream(stream), "train_linear_fwd");
}

extern "C" int stnerf_train_linear_dx(const float* dy, int64_t lddy, const float* w, int64_t ldw, int64_t m, int n, int k,
                                      const float* mask, int64_t ldmask, int accumulate, float* dx, int64_t lddx, stnerf_stream_t stream) {
    STNERF_REQUIRE(dy && w && dx, "train_linear_dx: null pointer");
    STNERF_REQUIRE(m >= 0 && m < (1ll << 31) && n >= 1 && k >= 1, "train_linear_dx: bad shape m=%lld n=%d k=%d", (long long)m, n, k);
    STNERF_REQUIRE((lddy & 3) == 0 && (ldw & 3) == 0 && lddy >= n && ldw >= ((k + 3) & ~3) && lddx >= k && aligned16(dy) && aligned16(w),
                   "train_linear_dx: dy / w need 16-byte aligned rows (ld %% 4 == 0); w rows of at least round4(k) floats");
    STNERF_REQUIRE(!mask || ldmask >= k, "train_linear_dx: mask rows shorter than k");
    STNERF_REQUIRE(m * lddy < (1ll << 29) && (int64_t)n * ldw < (1ll << 29), "train_linear_dx: operands of 2 GiB and more: split the batch");
    if (m == 0) return STNERF_OK;
    // dX[m][k] = sum_n dY[m][n] W[n][k]: output m x k, reduction over the layer's n outputs
    GemmArgs a{dy, w, dx, lddy, ldw, lddx, (int)m, k, n, nullptr, mask, ldmask, 0, accumulate, 0};
    return launch_gemm<true, false, 1>(a, (int)((m + GBM - 1) / GBM), 1, as_stream(stream), "train_linear_dx");
}

// dW's contraction runs over the SAMPLES: the launch is tiles x slices workgroups (one slice of the samples per blockIdx.z, its
// partial tile summed with the others afterwards, in slice order).  The networks' weight matrices are 1 - 3 tiles, so the slices
// are what fills the chip: about two workgroups per CU (512), slices of at least 256 samples, at most 256 of them (a 256 x 256
// layer: 2 tiles x 256 slices; the partial tiles are 64 MB written and read once -- 26 us of HBM against a 150 us GEMM).
static int dw_slices(int64_t m, int n, int k) {
    const int64_t tiles = (int64_t)((n + GBM - 1) / GBM) * (k > 128 ? (k + 255) / 256 : 1);
    int64_t want = (512 + tiles - 1) / tiles;
    const int64_t most = (m + 255) / 256;
    if (want > most) want = most;
    if (want > 256) want = 256;
    return (int)(want < 1 ? 1 : want);
}

extern "C" int64_t stnerf_train_dw_workspace_bytes(int64_t m, int n, int k) {
    if (m < 0 || n < 1 || k < 1) return STNERF_EINVAL;
    const int64_t row_blocks = m <= 0 ? 1 : (m + 1023) / 1024 > 512 ? 512 : (m + 1023) / 1024;
    return 4 * ((int64_t)dw_slices(m, n, k) * n * k + row_blocks * (int64_t)n) + 512;
}

extern "C" int stnerf_train_linear_dw(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t m, int n, int k, float* dw,
                                      int64_t lddw, float* db, int accumulate, void* workspace, int64_t workspace_bytes,
                                      stnerf_stream_t stream) {
    STNERF_REQUIRE(dy && x && dw && workspace, "train_linear_dw: null pointer");
    STNERF_REQUIRE(m >= 0 && m < (1ll << 31) && n >= 1 && k >= 1, "train_linear_dw: bad shape m=%lld n=%d k=%d", (long long)m, n, k);
    STNERF_REQUIRE((lddy & 3) == 0 && (ldx & 3) == 0 && lddy >= ((n + 3) & ~3) && ldx >= ((k + 3) & ~3) && lddw >= k && aligned16(dy) && aligned16(x),
                   "train_linear_dw: dy / x need 16-byte aligned rows (ld %% 4 == 0) of at least round4(n) / round4(k) floats");
    STNERF_REQUIRE(workspace_bytes >= stnerf_train_dw_workspace_bytes(m, n, k) && aligned16(workspace), "train_linear_dw: workspace too small");
    STNERF_REQUIRE(m * lddy < (1ll << 29) && m * ldx < (1ll << 29), "train_linear_dw: operands of 2 GiB and more: split the batch");
    if (m == 0) return STNERF_OK;
    hipStream_t st = as_stream(stream);
    const int splits = dw_slices(m, n, k);
    int kps = (int)((m + splits - 1) / splits);
    kps = (kps + GBK - 1) / GBK * GBK;
    float* partial = static_cast<float*>(workspace);
    // dW[n][k] = sum_s dY[s][n] X[s][k]: output n x k, reduction over the m samples
    GemmArgs a{dy, x, partial, lddy, ldx, 0, n, k, (int)m, nullptr, nullptr, 0, 0, 0, kps};
    if (const int rc = launch_gemm<false, false, 2>(a, (n + GBM - 1) / GBM, splits, st, "train_linear_dw")) return rc;
    const int64_t count = (int64_t)n * k;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, partial, splits, count, k, dw, lddw, accumulate);
    STNERF_CHECK_LAUNCH("train_linear_dw (reduce)");
    if (db) {
        float* bpart = partial + (int64_t)splits * count;
        const int row_blocks = (int)((m + 1023) / 1024 > 512 ? 512 : (m + 1023) / 1024);
        const int rpb = (int)((m + row_blocks - 1) / row_blocks);
        const int groups = (n + 3) / 4;
        if (groups > 32)
            hipLaunchKernelGGL(colsum_kernel<64>, dim3((groups + 63) / 64, row_blocks), dim3(256), 0, st, dy, lddy, (int)m, n, rpb, bpart);
        else if (groups > 8)
            hipLaunchKernelGGL(colsum_kernel<32>, dim3(1, row_blocks), dim3(256), 0, st, dy, lddy, (int)m, n, rpb, bpart);
        else
            hipLaunchKernelGGL(colsum_kernel<8>, dim3(1, row_blocks), dim3(256), 0, st, dy, lddy, (int)m, n, rpb, bpart);
        STNERF_CHECK_LAUNCH("train_linear_dw (bias partials)");
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((n + 255) / 256), dim3(256), 0, st, bpart, row_blocks, (int64_t)n, n, db, (int64_t)n, accumulate);
        STNERF_CHECK_LAUNCH("train_linear_dw (bias reduce)");
    }
    return STNERF_OK;
}

// (stnerf_train_dw_batch -- every weight and bias gradient of a network in one launch -- lives in train_dw.hip since round 6)

extern "C" int stnerf_train_encode(const float* x, int64_t ldx, int dim, int n_freq, int include_input, int64_t rows, int rows_per_src,
                                   int relu, int lerp_col, float* y, int64_t ldy, int col0, stnerf_stream_t stream) {
    STNERF_REQUIRE(x && y, "train_encode: null pointer");
    STNERF_REQUIRE(rows >= 0 && dim >= 1 && n_freq >= 0 && n_freq <= 30 && (include_input == 0 || include_input == 1) && rows_per_src >= 1 &&
                       ldx >= dim && col0 >= 0 && ldy >= col0 + dim * (include_input + 2 * n_freq) && lerp_col < dim,
                   "train_encode: bad shape rows=%lld dim=%d n_freq=%d", (long long)rows, dim, n_freq);
    if (rows == 0) return STNERF_OK;
    EncodeArgs a{x, ldx, dim, n_freq, include_input, y, ldy, col0, rows, rows_per_src, relu, lerp_col};
    hipLaunchKernelGGL(train_encode_kernel, dim3((unsigned)((rows * dim + 255) / 256)), dim3(256), 0, as_stream(stream), a);
    STNERF_CHECK_LAUNCH("train_encode");
    return STNERF_OK;
}

extern "C" int stnerf_train_encode_bwd(const float* x, int64_t ldx, int dim, int n_freq, int include_input, int64_t rows, const float* dy,
                                       int64_t lddy, int col0, int dim_out, int accumulate, float* dx, int64_t lddx, stnerf_stream_t stream) {
    STNERF_REQUIRE(x && dy && dx, "train_encode_bwd: null pointer");
    STNERF_REQUIRE(rows >= 0 && dim >= 1 && n_freq >= 0 && n_freq <= 30 && (include_input == 0 || include_input == 1) && dim_out >= 1 &&
                       dim_out <= dim && ldx >= dim_out && lddx >= dim_out && col0 >= 0 && lddy >= col0 + dim * (include_input + 2 * n_freq),
                   "train_encode_bwd: bad shape rows=%lld dim=%d n_freq=%d", (long long)rows, dim, n_freq);
    if (rows == 0) return STNERF_OK;
    EncodeBwdArgs a{x, ldx, dim, n_freq, include_input, dy, lddy, col0, rows, dx, lddx, dim_out, accumulate};
    hipLaunchKernelGGL(train_encode_bwd_kernel, dim3((unsigned)((rows * dim_out + 255) / 256)), dim3(256), 0, as_stream(stream), a);
    STNERF_CHECK_LAUNCH("train_encode_bwd");
    return STNERF_OK;
}
