// Every weight and bias gradient of a network in ONE launch (stnerf_train_dw_batch; SURVEY.md section 8(f)4:
// engine/layered_trainer.py:281's loss.backward() through modeling/spacenet.py:45-86 / modeling/motion_net.py:20-32 -- what ATen
// does there per nn.Linear is dW = dY^T X (a contraction over ALL samples) and db = column sums of dY).
//
// Round 6: the OUTPUT is stationary, in registers; the operands stream past it exactly once.
//
//   * A wave owns a 128 x 128 tile of one layer's dW for a range of samples: 16 accumulator blocks of v_mfma_f32_32x32x2_f32 =
//     256 accumulator registers (the whole AGPR half of the 512-entry file, one wave per SIMD).  The contraction index of the
//     instruction (K = 2) is the SAMPLE: lane (h, c) supplies A[row c][k = h] and B[k = h][col c], i.e. values of sample
//     2 t + h.  With the tile's rows and columns dealt to the lanes as "row 4 c + r of block r" a K step is ONE 16-byte
//     load of dY[2 t + h][n0 + 4 c ..] and ONE of X[2 t + h][k0 + 4 c ..] per lane (512 contiguous bytes per half wave,
//     straight from the row-major matrices the forward tap and the gradient chain wrote) feeding 16 MFMAs = 1024 cycles:
//     no LDS, no transposes, no barrier, no vector instruction in the loop but the MFMAs themselves (on gfx950 every VALU
//     instruction takes its cycles out of the f32 MFMA stream, profiles/r01_dual_issue_microbench.md).  Operand registers are
//     refilled right behind the step that consumed them, eight steps (8 k cycles) ahead of their next use.
//   * The partial tile leaves the registers once per (tile, sample range): lanes c hold consecutive columns, a store
//     instruction writes 512 contiguous bytes per half wave.  With a 128 x 128 tile per WAVE the chip's 1024 wave slots hold
//     every tile of a SpaceNet ~36 times: 36 partial sums per output instead of round 5's 256 (60 MB instead of 476 MB through
//     HBM), summed in slice order by dw_wave_reduce_kernel: deterministic, no atomics.
//   * The four waves of a workgroup (one CU) take tiles that read the same rows: the 2 x 2 tiles of a 256 x 256 layer over
//     one sample range share each operand slab between two waves of ONE CU (vector L1 / the XCD's L2), so a slice's rows of
//     dY and X cross HBM once.
//   * Narrow remainders are narrow tiles, not padded ones: 64 columns (stage1.0's 63, the skip connection's 63, rgb_net.1's
//     48) = 8-byte loads, two column blocks; 96 (MotionNet's 84) = 12-byte loads; a layer of <= 4 outputs (the density and
//     colour heads, the flow head) one 32-row block whose lanes c >= n read nothing.  Sample ranges are sized by a tile's
//     MFMA count so that every wave of the launch finishes together.
//   * db = column sums of dY: the tile at k0 = 0 of every row block adds its A operand registers as they pass (4 vector adds
//     per step on those waves); no second pass over dY.
//
// Reading beyond a matrix: column blocks may run past n or k (into the row's padding, the next row, or -- the last row --
// past the buffer, where the descriptor's range check returns 0): such lanes feed accumulator rows / columns that are never
// stored.  ROWS past a sample range must contribute zeros: the main loop runs over whole 16-sample blocks inside the range,
// the rest goes through a rolled tail whose lanes beyond the range get an out-of-range offset.
#include "common.h"
#include "mlp_common.h"

namespace stnerf {
namespace {

constexpr int DWW_WAVES = 1024;         // wave slots of the part: 256 CUs x 4 SIMDs, one wave each (512 registers)
constexpr int DWW_U = 8;                // K steps per rotation of the operand registers
constexpr int DWW_BLOCK = 2 * DWW_U;    // samples per unrolled block
constexpr int DWW_MIN_SLICE = 64;       // samples: a shorter range is not worth its partial tile
constexpr int DWW_MAX_PROBLEMS = 16, DWW_MAX_TILES = 64, DWW_MAX_GROUPS = 64;

struct DwwProblem {
    const float* dy;
    const float* x;
    uint32_t lddy, ldx;            // floats
    uint32_t dy_bytes, x_bytes;    // readable bytes from the base pointers (the descriptors' range)
};
struct DwwTile {
    uint32_t partial_off;          // units of 4 floats into the workspace: [slice][rows][cols]
    uint32_t bias_off;             // units of 4 floats: [slice][2][rows]; only with `bias`
    uint16_t n0, k0;
    uint8_t problem, nb, kb, bias; // nb: 4 (128 rows) or 1 (<= 4 rows of a 32-row block); kb: 1 .. 4 column blocks of 32
    uint8_t rows_valid, pad_[3];   // nb == 1: the layer's n
};
struct DwwGroup {                  // tiles first .. first + count - 1 share `slices` sample ranges of `len` samples
    uint32_t first_item;
    uint16_t first_tile, count, slices;
    uint16_t pad_;
    uint32_t len;
};
struct DwwArgs {
    DwwProblem p[DWW_MAX_PROBLEMS];
    DwwTile tile[DWW_MAX_TILES];
    DwwGroup group[DWW_MAX_GROUPS];
    int32_t n_groups, m;
    uint32_t items;
    float* workspace;
};

using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x3 = __attribute__((ext_vector_type(3))) int;
using i32x2 = __attribute__((ext_vector_type(2))) int;
template <int N> struct OpVec;     // N consecutive floats of a row: one buffer load per lane
template <> struct OpVec<4> {
    i32x4 v;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) { v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0); }
    __device__ __forceinline__ float get(int i) const { return __int_as_float(v[i]); }
};
template <> struct OpVec<3> {
    i32x3 v;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) { v = __builtin_amdgcn_raw_buffer_load_b96(r, voff, soff, 0); }
    __device__ __forceinline__ float get(int i) const { return __int_as_float(v[i]); }
};
template <> struct OpVec<2> {
    i32x2 v;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) { v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0); }
    __device__ __forceinline__ float get(int i) const { return __int_as_float(v[i]); }
};
template <> struct OpVec<1> {
    int v;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) { v = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0); }
    __device__ __forceinline__ float get(int) const { return __int_as_float(v); }
};

constexpr uint32_t DWW_OOB = 0x7FFFFFFCu;   // a per-lane offset past every descriptor's range (num_records < 2^31): the load returns 0

// One wave: rows n0 .. of dW (NB blocks: row NB c' + r of the tile = lane c' of block r), columns k0 .. (KB blocks likewise),
// samples [s0, s1).
template <int NB, int KB, bool BIAS>
__device__ __forceinline__ void dww_tile(const DwwProblem& p, const DwwTile& tl, int s0, int s1, float* __restrict__ out, float* __restrict__ bias_out) {
    const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    // the lane's offsets inside a K step (sample parity h); a thin layer's lanes c >= n read nothing
    const uint32_t va_in = (uint32_t)(h * p.lddy + tl.n0 + NB * c) * 4u;
    const uint32_t va = (NB == 1 && c >= tl.rows_valid) ? DWW_OOB : va_in;
    const uint32_t vb = (uint32_t)(h * p.ldx + tl.k0 + KB * c) * 4u;
    const uint32_t step_a = 2u * p.lddy * 4u, step_b = 2u * p.ldx * 4u;    // bytes per K step (wave-uniform)
    uint32_t sa = (uint32_t)s0 * p.lddy * 4u, sb = (uint32_t)s0 * p.ldx * 4u;
    f32x16 acc[NB][KB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < KB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) bsum[i] = 0.f;
    auto mma = [&](const OpVec<NB>& a, const OpVec<KB>& b) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < KB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.get(i), b.get(j), acc[i][j], 0, 0, 0);
        if (BIAS) {   // (as volatile asm: plain adds are gathered at the end of the unrolled block, with copies of every operand
                      // register and the refills hoisted over them)
#pragma unroll
            for (int i = 0; i < NB; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(bsum[i]) : "v"(a.get(i)));
        }
    };
    const int nblk = (s1 - s0) / DWW_BLOCK;
    if (nblk > 0) {
        OpVec<NB> a[DWW_U];
        OpVec<KB> b[DWW_U];
#pragma unroll
        for (int u = 0; u < DWW_U; ++u) {      // (in step order: the loop's wait counts assume step 0's operands are the oldest loads)
            a[u].load(ra, va, sa + u * step_a);
            b[u].load(rb, vb, sb + u * step_b);
            __builtin_amdgcn_sched_barrier(0);
        }
        // every block but the last: a step's operand registers are reloaded (for the next block) right behind its MFMAs
        for (int blk = 0; blk + 1 < nblk; ++blk) {
            sa += DWW_U * step_a;
            sb += DWW_U * step_b;
#pragma unroll
            for (int u = 0; u < DWW_U; ++u) {
                mma(a[u], b[u]);
                __builtin_amdgcn_sched_barrier(0);
                a[u].load(ra, va, sa + u * step_a);
                b[u].load(rb, vb, sb + u * step_b);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int u = 0; u < DWW_U; ++u) {
            mma(a[u], b[u]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the range's last < 16 samples: lanes whose sample lies beyond it read nothing
    for (int s = s0 + nblk * DWW_BLOCK; s < s1; s += 2) {
        const bool in = s + h < s1;
        OpVec<NB> a;
        OpVec<KB> b;
        a.load(ra, in ? va : DWW_OOB, (uint32_t)s * p.lddy * 4u);
        b.load(rb, in ? vb : DWW_OOB, (uint32_t)s * p.ldx * 4u);
        mma(a, b);
    }
    // accumulator register 4 q + rr of block (i, j), lane (h, c): tile row NB (8 q + 4 h + rr) + i, column KB c + j
    constexpr int COLS = 32 * KB;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float* o = out + (NB * (8 * q + 4 * h + rr) + i) * COLS + KB * c;
                if (KB == 4) {
                    *reinterpret_cast<float4*>(o) = make_float4(acc[i][0][4 * q + rr], acc[i][1][4 * q + rr], acc[i][2][4 * q + rr], acc[i][3][4 * q + rr]);
                } else if (KB == 2) {
                    *reinterpret_cast<float2*>(o) = make_float2(acc[i][0][4 * q + rr], acc[i][1][4 * q + rr]);
                } else {
#pragma unroll
                    for (int j = 0; j < KB; ++j) o[j] = acc[i][j][4 * q + rr];
                }
            }
    if (BIAS && bias_out) {   // [parity h][row NB c + i]: the two parities are two more terms of the reduction
        float* o = bias_out + h * (32 * NB) + NB * c;
        if (NB == 4) {
            *reinterpret_cast<float4*>(o) = make_float4(bsum[0], bsum[1], bsum[2], bsum[3]);
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) o[i] = bsum[i];
        }
    }
}

__global__ __launch_bounds__(256, 1) void train_dw_wave_kernel(DwwArgs a) {
    // (the wave index through readfirstlane: everything derived from it -- tile, descriptors, sample range -- is wave-uniform and
    // stays in scalar registers; as a function of threadIdx the compiler wraps every buffer load in a waterfall loop)
    const uint32_t g = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (g >= a.items) return;
    int gi = 0;
    while (gi + 1 < a.n_groups && g >= a.group[gi + 1].first_item) ++gi;      // (wave-uniform: scalar loads of the kernel arguments)
    const DwwGroup& gr = a.group[gi];
    const uint32_t i = g - gr.first_item;
    const uint32_t slice = i / gr.count;
    const DwwTile& tl = a.tile[gr.first_tile + (i - slice * gr.count)];
    const DwwProblem& p = a.p[tl.problem];
    const int s0 = (int)(slice * gr.len), s1 = min(a.m, s0 + (int)gr.len);
    const int rows = tl.nb == 4 ? 128 : 32, cols = 32 * tl.kb;
    float* out = a.workspace + 4ull * tl.partial_off + (size_t)slice * rows * cols;
    float* bias_out = tl.bias ? a.workspace + 4ull * tl.bias_off + (size_t)slice * 2 * rows : nullptr;
    if (tl.nb == 4) {
        if (tl.kb == 4) {
            if (tl.bias) dww_tile<4, 4, true>(p, tl, s0, s1, out, bias_out);
            else dww_tile<4, 4, false>(p, tl, s0, s1, out, bias_out);
        } else if (tl.kb == 3) {
            dww_tile<4, 3, true>(p, tl, s0, s1, out, bias_out);
        } else if (tl.kb == 2) {
            dww_tile<4, 2, true>(p, tl, s0, s1, out, bias_out);
        } else {
            dww_tile<4, 1, true>(p, tl, s0, s1, out, bias_out);
        }
    } else {
        if (tl.kb == 4) dww_tile<1, 4, true>(p, tl, s0, s1, out, bias_out);
        else if (tl.kb == 3) dww_tile<1, 3, true>(p, tl, s0, s1, out, bias_out);
        else if (tl.kb == 2) dww_tile<1, 2, true>(p, tl, s0, s1, out, bias_out);
        else dww_tile<1, 1, true>(p, tl, s0, s1, out, bias_out);
    }
}

// dst[i][j] (+)= sum over the tile's sample ranges of partial[z][i][j], in z order (blockIdx.y = segment: a tile of some dW, or
// the bias sums of a row block: rows = 1, its 2 parities x slices as the terms).
struct DwwSegment {
    float* dst;
    uint32_t ld_dst, partial_off;      // partial_off in units of 4 floats
    uint32_t terms, stride;            // floats between two terms
    uint16_t rows_valid, cols_valid, cols, pad_;
};
constexpr int DWW_MAX_SEGMENTS = 100;      // (a kernel's arguments -- with the runtime's hidden ones -- stay below 4 KB)
struct DwwReduceArgs {
    DwwSegment seg[DWW_MAX_SEGMENTS];
    const float* workspace;
    int32_t accumulate;
};
__global__ void dw_wave_reduce_kernel(DwwReduceArgs a) {
    const DwwSegment& sg = a.seg[blockIdx.y];
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = e / sg.cols, j = e - i * sg.cols;
    if (i >= sg.rows_valid || j >= sg.cols_valid) return;
    const float* partial = a.workspace + 4ull * sg.partial_off + e;
    float s = 0.f;
#pragma unroll 8
    for (uint32_t z = 0; z < sg.terms; ++z) s += partial[(size_t)z * sg.stride];
    float* d = sg.dst + (size_t)i * sg.ld_dst + j;
    *d = a.accumulate ? *d + s : s;
}

// ---- the plan: tiles, groups, sample ranges, workspace layout (host; a pure function of the problems' shapes and m) ---------
struct DwwPlan {
    DwwArgs k;
    DwwReduceArgs r;
    int n_tiles, n_segments, max_elems;
    int64_t floats;
};

}  // namespace
}  // namespace stnerf

using namespace stnerf;

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int dww_check(const stnerf_dw_problem* pr, int32_t count, int64_t m) {
    STNERF_REQUIRE(pr && count >= 1 && count <= DWW_MAX_PROBLEMS, "train_dw_batch: 1 .. %d problems", DWW_MAX_PROBLEMS);
    STNERF_REQUIRE(m >= 0 && m < (1ll << 31), "train_dw_batch: bad sample count %lld", (long long)m);
    int tiles = 0, segments = 0;
    for (int i = 0; i < count; ++i) {
        const stnerf_dw_problem& q = pr[i];
        STNERF_REQUIRE(q.dy && q.x && q.dw, "train_dw_batch: null pointer in problem %d", i);
        STNERF_REQUIRE(q.n >= 1 && q.n <= 65535 && q.k >= 1 && q.k <= 65535, "train_dw_batch: bad shape n=%d k=%d in problem %d", q.n, q.k, i);
        STNERF_REQUIRE((q.lddy & 3) == 0 && (q.ldx & 3) == 0 && q.lddy >= ((q.n + 3) & ~3) && q.ldx >= ((q.k + 3) & ~3) && q.lddw >= q.k &&
                           aligned16(q.dy) && aligned16(q.x),
                       "train_dw_batch: dy / x need 16-byte aligned rows (ld %% 4 == 0) of at least round4(n) / round4(k) floats (problem %d)", i);
        STNERF_REQUIRE(m * q.lddy < (1ll << 29) && m * q.ldx < (1ll << 29), "train_dw_batch: operands of 2 GiB and more: split the batch");
        const int row_blocks = q.n <= 4 ? 1 : (q.n + 127) / 128;
        tiles += row_blocks * ((q.k + 127) / 128);
        segments += row_blocks * ((q.k + 127) / 128) + (q.db ? row_blocks : 0);
    }
    STNERF_REQUIRE(tiles <= DWW_MAX_TILES && segments <= DWW_MAX_SEGMENTS, "train_dw_batch: %d tiles (at most %d), %d reduction segments (at most %d)",
                   tiles, DWW_MAX_TILES, segments, DWW_MAX_SEGMENTS);
    return STNERF_OK;
}

// Tiles of a problem: row blocks of 128 (one block of 32 for a layer of <= 4 outputs) x column pieces of 128, the last piece as
// narrow as its columns allow (96 / 64 / 32).  Groups = a problem's tiles of one width (they cost the same and read the same
// rows); a group's sample ranges are sized by its tiles' MFMA count so that all waves of the launch finish together.
static void dww_plan(const stnerf_dw_problem* pr, int32_t count, int64_t m, DwwPlan& P) {
    P = DwwPlan{};
    DwwArgs& K = P.k;
    K.m = (int)m;
    struct G { int first, count, cost; };
    G groups[DWW_MAX_GROUPS];
    int ng = 0, nt = 0;
    for (int i = 0; i < count; ++i) {
        const stnerf_dw_problem& q = pr[i];
        K.p[i] = DwwProblem{q.dy, q.x, (uint32_t)q.lddy, (uint32_t)q.ldx,
                            (uint32_t)(m > 0 ? ((m - 1) * q.lddy + ((q.n + 3) & ~3)) * 4 : 0), (uint32_t)(m > 0 ? ((m - 1) * q.ldx + ((q.k + 3) & ~3)) * 4 : 0)};
        const int nb = q.n <= 4 ? 1 : 4, rows = 32 * nb;
        for (int kb = 4; kb >= 1; --kb) {                      // one group per piece width, the full-width pieces first
            const int first = nt;
            for (int n0 = 0; n0 < q.n; n0 += rows)
                for (int k0 = 0; k0 < q.k; k0 += 128) {
                    const int left = q.k - k0, w = left > 96 ? 4 : left > 64 ? 3 : left > 32 ? 2 : 1;
                    if (w != kb) continue;
                    DwwTile& t = K.tile[nt++];
                    t = DwwTile{};
                    t.n0 = (uint16_t)n0, t.k0 = (uint16_t)k0, t.problem = (uint8_t)i, t.nb = (uint8_t)nb, t.kb = (uint8_t)kb;
                    t.bias = (uint8_t)(q.db && k0 == 0);
                    t.rows_valid = (uint8_t)(nb == 1 ? q.n : 0);
                }
            if (nt > first) groups[ng++] = G{first, nt - first, nb * kb};
        }
    }
    P.n_tiles = nt;
    // sample ranges: slices_g ~ share of the launch's MFMA work, total <= the part's wave slots
    int64_t total_cost = 0;
    for (int g = 0; g < ng; ++g) total_cost += (int64_t)groups[g].count * groups[g].cost;
    const int64_t most = m <= 0 ? 1 : (m + DWW_MIN_SLICE - 1) / DWW_MIN_SLICE;
    for (int target = DWW_WAVES;; target -= 8) {
        int64_t items = 0;
        for (int g = 0; g < ng; ++g) {
            int64_t s = ((int64_t)target * groups[g].cost + total_cost / 2) / total_cost;
            s = s < 1 ? 1 : s > most ? most : s;
            int64_t len = m <= 0 ? DWW_BLOCK : ((m + s - 1) / s + DWW_BLOCK - 1) / DWW_BLOCK * DWW_BLOCK;
            s = m <= 0 ? 1 : (m + len - 1) / len;
            K.group[g] = DwwGroup{(uint32_t)items, (uint16_t)groups[g].first, (uint16_t)groups[g].count, (uint16_t)s, 0, (uint32_t)len};
            items += s * groups[g].count;
        }
        K.items = (uint32_t)items;
        if (items <= DWW_WAVES || target <= 8) break;
    }
    K.n_groups = ng;
    // workspace: per tile [slices][rows][cols] (+ [slices][2][rows] bias sums), 16-byte aligned pieces
    int64_t off = 0;
    int ns = 0, max_elems = 0;
    for (int g = 0; g < ng; ++g)
        for (int j = 0; j < K.group[g].count; ++j) {
            DwwTile& t = K.tile[K.group[g].first_tile + j];
            const stnerf_dw_problem& q = pr[t.problem];
            const int rows = 32 * t.nb, cols = 32 * t.kb, slices = K.group[g].slices;
            t.partial_off = (uint32_t)(off / 4);
            const int rv = q.n - t.n0 < rows ? q.n - t.n0 : rows, cv = q.k - t.k0 < cols ? q.k - t.k0 : cols;
            P.r.seg[ns++] = DwwSegment{q.dw + (int64_t)t.n0 * q.lddw + t.k0, (uint32_t)q.lddw, t.partial_off, (uint32_t)slices, (uint32_t)(rows * cols),
                                       (uint16_t)rv, (uint16_t)cv, (uint16_t)cols, 0};
            max_elems = rows * cols > max_elems ? rows * cols : max_elems;
            off += (int64_t)slices * rows * cols;
            if (t.bias) {
                t.bias_off = (uint32_t)(off / 4);
                P.r.seg[ns++] = DwwSegment{q.db + t.n0, 0, t.bias_off, (uint32_t)(2 * slices), (uint32_t)rows, 1, (uint16_t)rv, (uint16_t)rows, 0};
                off += (int64_t)slices * 2 * rows;
            }
        }
    P.n_segments = ns;
    P.max_elems = max_elems;
    P.floats = off;
}

extern "C" int64_t stnerf_train_dw_batch_workspace_bytes(const stnerf_dw_problem* problems, int32_t count, int64_t m) {
    if (dww_check(problems, count, m)) return STNERF_EINVAL;
    DwwPlan P;
    dww_plan(problems, count, m, P);
    return 4 * P.floats + 512;
}

extern "C" int stnerf_train_dw_batch(const stnerf_dw_problem* problems, int32_t count, int64_t m, int32_t accumulate, void* workspace,
                                     int64_t workspace_bytes, stnerf_stream_t stream) {
    if (const int rc = dww_check(problems, count, m)) return rc;
    DwwPlan P;
    dww_plan(problems, count, m, P);
    STNERF_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= 4 * P.floats + 512, "train_dw_batch: workspace too small");
    STNERF_REQUIRE(P.floats / 4 < (1ll << 32), "train_dw_batch: workspace beyond 64 GiB");
    if (m == 0) return STNERF_OK;
    hipStream_t st = as_stream(stream);
    P.k.workspace = static_cast<float*>(workspace);
    hipLaunchKernelGGL(train_dw_wave_kernel, dim3((P.k.items + 3) / 4), dim3(256), 0, st, P.k);
    STNERF_CHECK_LAUNCH("train_dw_batch");
    P.r.workspace = P.k.workspace;
    P.r.accumulate = accumulate;
    hipLaunchKernelGGL(dw_wave_reduce_kernel, dim3((P.max_elems + 255) / 256, P.n_segments), dim3(256), 0, st, P.r);
    STNERF_CHECK_LAUNCH("train_dw_batch (reduce)");
    return STNERF_OK;
}
