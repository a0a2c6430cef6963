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
// registers); operands are staged through LDS in 16-deep K slices ([row][k] with a 17-word row stride: the MFMA operand
// fetch -- 32 consecutive rows at one k per half wave -- is conflict-free), the next slice's global loads are in flight
// while the current one is multiplied.  One f32 MFMA is 64 cycles of matrix pipe for 4 LDS reads: the kernel is MFMA-bound
// by construction, there is nothing to hide.
//
// Positional encodings (utils/dimension_kernel.py:54-73): forward into a strided destination (a column block of a layer's
// input matrix: the skip connection and rgb_net's input are built in place, no torch.cat), per-ray encodings broadcast to
// the ray's samples with rgb_net's leading ReLU applied (modeling/spacenet.py:80-86), MotionNet's fractional-time lerp
// (modeling/motion_net.py:52-60); backward by the chain rule d/dx sin(2^f x) = 2^f cos(2^f x).
#include "common.h"
#include "mlp_common.h"

namespace stnerf {
namespace {

constexpr int GBM = 128, GBN = 128, GBK = 16, GLD = GBK + 1;

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

// One operand tile [128 rows][16 k] from global memory into registers (2 float4 per thread), zero beyond the edges.
//   K_CONTIG: element (r, k) = src[(r0 + r) * ld + k0 + k]   (k contiguous: float4 along k)
//  !K_CONTIG: element (r, k) = src[(k0 + k) * ld + r0 + r]   (r contiguous: float4 along r)
// ld is a multiple of 4 and the allocation covers whole float4s (the Python side pads): only whole-vector guards.
template <bool K_CONTIG>
__device__ __forceinline__ void load_tile(float4 (&v)[2], const float* __restrict__ src, int64_t ld, int r0, int R, int k0, int k_end, int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (K_CONTIG) {
            const int r = r0 + (t >> 2) + 64 * i, k = k0 + 4 * (t & 3);
            if (r < R && k < k_end) {
                v[i] = *reinterpret_cast<const float4*>(src + (int64_t)r * ld + k);
                if (k + 1 >= k_end) v[i].y = 0.f;      // (a reduction range may end inside a vector: split-K slices, K % 4)
                if (k + 2 >= k_end) v[i].z = 0.f;
                if (k + 3 >= k_end) v[i].w = 0.f;
            }
        } else {
            const int k = k0 + (t >> 5) + 8 * i, r = r0 + 4 * (t & 31);
            if (k < k_end && r < R) v[i] = *reinterpret_cast<const float4*>(src + (int64_t)k * ld + r);   // (rows beyond R: unused outputs)
        }
    }
}
template <bool K_CONTIG>
__device__ __forceinline__ void store_tile(float* lds, const float4 (&v)[2], int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (K_CONTIG) {
            float* p = lds + ((t >> 2) + 64 * i) * GLD + 4 * (t & 3);
            p[0] = v[i].x; p[1] = v[i].y; p[2] = v[i].z; p[3] = v[i].w;
        } else {
            float* p = lds + (4 * (t & 31)) * GLD + (t >> 5) + 8 * i;
            p[0] = v[i].x; p[GLD] = v[i].y; p[2 * GLD] = v[i].z; p[3 * GLD] = v[i].w;
        }
    }
}

// C = A' B' with A'[m][k], B'[k][n] read as the template flags say; MODE 0: forward epilogue (bias, ReLU), 1: dX epilogue
// (mask, accumulate), 2: dW partial tile.
template <bool A_KC, bool B_KC, int MODE>
__global__ __launch_bounds__(256) void train_gemm_kernel(GemmArgs a) {
    __shared__ float As[GBM * GLD], Bs[GBN * GLD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, c = lane & 31;
    const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
    int k_begin = 0, k_end = a.K;
    if (MODE == 2) {
        k_begin = blockIdx.z * a.k_per_split;
        k_end = min(a.K, k_begin + a.k_per_split);
    }
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[2], rb[2];
    load_tile<A_KC>(ra, a.A, a.lda, m0, a.M, k_begin, k_end, t);
    load_tile<B_KC>(rb, a.B, a.ldb, n0, a.N, k_begin, k_end, t);
    for (int k0 = k_begin; k0 < k_end; k0 += GBK) {
        __syncthreads();                       // the previous slice has been consumed
        store_tile<A_KC>(As, ra, t);
        store_tile<B_KC>(Bs, rb, t);
        __syncthreads();
        if (k0 + GBK < k_end) {                // the next slice's loads fly under this slice's MFMAs
            load_tile<A_KC>(ra, a.A, a.lda, m0, a.M, k0 + GBK, k_end, t);
            load_tile<B_KC>(rb, a.B, a.ldb, n0, a.N, k0 + GBK, k_end, t);
        }
#pragma unroll
        for (int kk = 0; kk < GBK / 2; ++kk) {
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = As[(wm + 32 * i + c) * GLD + 2 * kk + h];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = Bs[(wn + 32 * j + c) * GLD + 2 * kk + h];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }
    // accumulator register 4 q + r of lane (h, c): row 8 q + 4 h + r, column c of the 32 x 32 tile
    float* out = MODE == 2 ? a.C + (int64_t)blockIdx.z * a.M * a.N : a.C;
    const int64_t ldc = MODE == 2 ? a.N : a.ldc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + 32 * j + c;
            if (n >= a.N) continue;
            float bias = 0.f;
            if (MODE == 0 && a.bias) bias = a.bias[n];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm + 32 * i + 8 * q + 4 * h + r;
                    if (m >= a.M) continue;
                    float v = acc[i][j][4 * q + r];
                    if (MODE == 0) {
                        v += bias;
                        if (a.relu) v = fmaxf(v, 0.f);
                    } else if (MODE == 1) {
                        if (a.mask && !(a.mask[(int64_t)m * a.ldmask + n] > 0.f)) v = 0.f;
                        if (a.accumulate) v += out[(int64_t)m * ldc + n];
                    }
                    out[(int64_t)m * ldc + n] = v;
                }
        }
}

// dst[i] (+)= sum_z partial[z][i] in z order (the second half of the dW / db reductions).
__global__ void reduce_partials_kernel(const float* __restrict__ partial, int splits, int64_t count, int cols, float* __restrict__ dst,
                                       int64_t ld_dst, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += partial[(int64_t)z * count + i];
    float* d = dst + (i / cols) * ld_dst + (i % cols);
    *d = accumulate ? *d + s : s;
}

// Column sums of a row slice: partial[blockIdx.y][n] = sum over the block's rows of Y[m][n] (db = sum_samples dy).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ y, int64_t ld, int M, int N, int rows_per_block,
                                                     float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    const int m_begin = blockIdx.y * rows_per_block, m_end = min(M, m_begin + rows_per_block);
    float s = 0.f;
    if (n < N)
        for (int m = m_begin + w; m < m_end; m += 4) s += y[(int64_t)m * ld + n];
    red[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && n < N) partial[(int64_t)blockIdx.y * N + n] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
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
    if (m == 0) return STNERF_OK;
    GemmArgs a{x, w, y, ldx, ldw, ldy, (int)m, n, k, bias, nullptr, 0, relu, 0, 0};
    const dim3 grid((n + GBN - 1) / GBN, (unsigned)((m + GBM - 1) / GBM), 1);
    hipLaunchKernelGGL((train_gemm_kernel<true, true, 0>), grid, dim3(256), 0, as_stream(stream), a);
    STNERF_CHECK_LAUNCH("train_linear_fwd");
    return STNERF_OK;
}

extern "C" int stnerf_train_linear_dx(const float* dy, int64_t lddy, const float* w, int64_t ldw, int64_t m, int n, int k,
                                      const float* mask, int64_t ldmask, int accumulate, float* dx, int64_t lddx, stnerf_stream_t stream) {
    STNERF_REQUIRE(dy && w && dx, "train_linear_dx: null pointer");
    STNERF_REQUIRE(m >= 0 && m < (1ll << 31) && n >= 1 && k >= 1, "train_linear_dx: bad shape m=%lld n=%d k=%d", (long long)m, n, k);
    STNERF_REQUIRE((lddy & 3) == 0 && (ldw & 3) == 0 && lddy >= n && ldw >= ((k + 3) & ~3) && lddx >= k && aligned16(dy) && aligned16(w),
                   "train_linear_dx: dy / w need 16-byte aligned rows (ld %% 4 == 0); w rows of at least round4(k) floats");
    STNERF_REQUIRE(!mask || ldmask >= k, "train_linear_dx: mask rows shorter than k");
    if (m == 0) return STNERF_OK;
    // dX[m][k] = sum_n dY[m][n] W[n][k]: output m x k, reduction over the layer's n outputs
    GemmArgs a{dy, w, dx, lddy, ldw, lddx, (int)m, k, n, nullptr, mask, ldmask, 0, accumulate, 0};
    const dim3 grid((k + GBN - 1) / GBN, (unsigned)((m + GBM - 1) / GBM), 1);
    hipLaunchKernelGGL((train_gemm_kernel<true, false, 1>), grid, dim3(256), 0, as_stream(stream), a);
    STNERF_CHECK_LAUNCH("train_linear_dx");
    return STNERF_OK;
}

extern "C" int64_t stnerf_train_dw_workspace_bytes(int64_t m, int n, int k) {
    if (m < 0 || n < 1 || k < 1) return STNERF_EINVAL;
    const int64_t splits = m <= 0 ? 1 : (m + 2047) / 2048 > 64 ? 64 : (m + 2047) / 2048;   // <= 64 slices of >= 2048 samples
    const int64_t row_blocks = m <= 0 ? 1 : (m + 4095) / 4096 > 256 ? 256 : (m + 4095) / 4096;
    return 4 * (splits * (int64_t)n * k + row_blocks * (int64_t)n) + 512;
}

extern "C" int stnerf_train_linear_dw(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t m, int n, int k, float* dw,
                                      int64_t lddw, float* db, int accumulate, void* workspace, int64_t workspace_bytes,
                                      stnerf_stream_t stream) {
    STNERF_REQUIRE(dy && x && dw && workspace, "train_linear_dw: null pointer");
    STNERF_REQUIRE(m >= 0 && m < (1ll << 31) && n >= 1 && k >= 1, "train_linear_dw: bad shape m=%lld n=%d k=%d", (long long)m, n, k);
    STNERF_REQUIRE((lddy & 3) == 0 && (ldx & 3) == 0 && lddy >= ((n + 3) & ~3) && ldx >= ((k + 3) & ~3) && lddw >= k && aligned16(dy) && aligned16(x),
                   "train_linear_dw: dy / x need 16-byte aligned rows (ld %% 4 == 0) of at least round4(n) / round4(k) floats");
    STNERF_REQUIRE(workspace_bytes >= stnerf_train_dw_workspace_bytes(m, n, k) && aligned16(workspace), "train_linear_dw: workspace too small");
    if (m == 0) return STNERF_OK;
    hipStream_t st = as_stream(stream);
    const int splits = (int)((m + 2047) / 2048 > 64 ? 64 : (m + 2047) / 2048);
    int kps = (int)((m + splits - 1) / splits);
    kps = (kps + GBK - 1) / GBK * GBK;
    float* partial = static_cast<float*>(workspace);
    // dW[n][k] = sum_s dY[s][n] X[s][k]: output n x k, reduction over the m samples
    GemmArgs a{dy, x, partial, lddy, ldx, 0, n, k, (int)m, nullptr, nullptr, 0, 0, 0, kps};
    const dim3 grid((k + GBN - 1) / GBN, (n + GBM - 1) / GBM, splits);
    hipLaunchKernelGGL((train_gemm_kernel<false, false, 2>), grid, dim3(256), 0, st, a);
    STNERF_CHECK_LAUNCH("train_linear_dw");
    const int64_t count = (int64_t)n * k;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, partial, splits, count, k, dw, lddw, accumulate);
    STNERF_CHECK_LAUNCH("train_linear_dw (reduce)");
    if (db) {
        float* bpart = partial + (int64_t)splits * count;
        const int row_blocks = (int)((m + 4095) / 4096 > 256 ? 256 : (m + 4095) / 4096);
        const int rpb = (int)((m + row_blocks - 1) / row_blocks);
        hipLaunchKernelGGL(colsum_kernel, dim3((n + 63) / 64, row_blocks), dim3(256), 0, st, dy, lddy, (int)m, n, rpb, bpart);
        STNERF_CHECK_LAUNCH("train_linear_dw (bias partials)");
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((n + 255) / 256), dim3(256), 0, st, bpart, row_blocks, (int64_t)n, n, db, (int64_t)n, accumulate);
        STNERF_CHECK_LAUNCH("train_linear_dw (bias reduce)");
    }
    return STNERF_OK;
}

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
