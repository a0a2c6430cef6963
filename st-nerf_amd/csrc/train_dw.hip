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
//   * Tiles that read the same rows at the same time get them through LDS (a BUNDLE, dwb_run below): the 2 x 2 tiles of a
//     256 x 256 layer, and the 64-column tiles of two layers on the same narrow input (stage1.0 on PE(pos) and stage2.0's skip
//     columns).  Streaming independently, two waves that ask for a line at the same moment BOTH go to HBM -- neither the vector
//     L1 nor the L2 merges a miss into one in flight: 7.5 GB per SpaceNet call on the counters against 4.15 GB of unique
//     operands.  A bundle's rows are copied global -> LDS once per workgroup by DMA (buffer_load_dwordx4 ... lds, a ring of four
//     16-sample slots, three blocks ahead) and read from there: 5.1 GB (1.22 x: what is left are the narrow tiles that re-read a
//     dy, and the density head's second pass over g3).  In f32 the time is the same either way (1.94 ms sustained, 0.80 of the f32
//     MFMA peak at 2.2 - 2.3 GHz): that launch is bound by its MFMAs, not by HBM.
//   * The 2 x 2 bundles -- 85 % of a SpaceNet's weight-gradient work -- run in SPLIT-BF16 (dwb_run_bx below): both operands split into
//     three bf16 pieces on the fly, six v_mfma_f32_32x32x16_bf16 per product: fp32-faithful products at 2.3 x the f32 tile's rate per
//     sample; ranges are sized by that measured cost so that f32 and bf16x3 waves finish together.  SpaceNet's ten dW + db:
//     1.96 -> 1.43 ms per 262,144 rows = 171 TF/s of fp32-faithful products (1.09 x the f32 MFMA peak).
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
constexpr int DWW_MAX_PROBLEMS = 16, DWW_MAX_TILES = 64, DWW_MAX_GROUPS = 40;
// bundles: four tiles that read the same rows get them through LDS (see dwb_run)
constexpr int DWB_RING = 4;             // ring slots of one 16-sample block each: fetched three blocks (~10 us) ahead
constexpr int DWB_MAX_PIECES = 3, DWB_MAX_BUNDLES = 12;
constexpr int DWW_BX_COST = 7;          // a split-bf16 2 x 2 bundle's time per sample in sixteenths of the f32 bundle's (measured, see dww_plan)

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
struct DwbPiece {                  // one DMA instruction per sample row: `units` 16-byte units of a matrix row -> the slot row at lds_off
    uint8_t problem, is_x;
    uint16_t col0, units, lds_off; // floats; units <= 64
};
struct DwbWave {                   // what a wave multiplies: A operand at a_off of the slot's dy rows, B operand at b_off of its x rows (floats)
    uint16_t a_off, b_off;
    uint8_t tile, kb, active, pad_;
};
struct DwbBundle {
    DwbPiece piece[DWB_MAX_PIECES];
    DwbWave wave[4];
    uint16_t n_pieces, row_floats;
    uint32_t first_wg, slices, len;
    uint32_t bx;                   // the 2 x 2 bundle in split-bf16 (dwb_run_bx)
};
struct DwwArgs {
    DwwProblem p[DWW_MAX_PROBLEMS];
    DwwTile tile[DWW_MAX_TILES];
    DwwGroup group[DWW_MAX_GROUPS];
    DwbBundle bundle[DWB_MAX_BUNDLES];
    int32_t n_groups, n_bundles, m;
    uint32_t items, bundle_wgs;    // workgroups [0, bundle_wgs) run bundles, the rest four wave items each
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

// A wave's partial tile (and bias sums) to the workspace.  Accumulator register 4 q + rr of block (i, j), lane (h, c): tile row
// NB (8 q + 4 h + rr) + i, column KB c + j: lanes c hold consecutive columns, a store writes 512 contiguous bytes per half wave.
template <int NB, int KB, bool BIAS>
__device__ __forceinline__ void dww_store(const f32x16 (&acc)[NB][KB], const float (&bsum)[NB], float* __restrict__ out, float* __restrict__ bias_out) {
    const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    constexpr int COLS = 32 * KB;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float* o = out + (NB * (8 * q + 4 * h + rr) + i) * COLS + KB * c;
                if constexpr (KB == 4) {
                    *reinterpret_cast<float4*>(o) = make_float4(acc[i][0][4 * q + rr], acc[i][1][4 * q + rr], acc[i][2][4 * q + rr], acc[i][3][4 * q + rr]);
                } else if constexpr (KB == 2) {
                    *reinterpret_cast<float2*>(o) = make_float2(acc[i][0][4 * q + rr], acc[i][1][4 * q + rr]);
                } else {
#pragma unroll
                    for (int j = 0; j < KB; ++j) o[j] = acc[i][j][4 * q + rr];
                }
            }
    if (BIAS && bias_out) {   // [parity h][row NB c + i]: the two parities are two more terms of the reduction
        float* o = bias_out + h * (32 * NB) + NB * c;
        if constexpr (NB == 4) {
            *reinterpret_cast<float4*>(o) = make_float4(bsum[0], bsum[1], bsum[2], bsum[3]);
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) o[i] = bsum[i];
        }
    }
}

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
    dww_store<NB, KB, BIAS>(acc, bsum, out, bias_out);
}

// ---- bundles: the tiles of a workgroup share their rows through LDS -------------------------------------------------------
// Two waves that ask for the same line at the same moment both go to HBM: with the 2 x 2 tiles of a 256 x 256 layer streaming
// their operands independently the launch moved 1.8 x its unique bytes (PMC), and a staggered start does not survive -- the
// wave that hits in L2 catches up with the one that waits for HBM.  So a BUNDLE's rows are fetched once per workgroup: every
// 16-sample block of the bundle's operand columns is copied global -> LDS by DMA (buffer_load_dwordx4 ... lds: one
// instruction per sample row and matrix piece, 1 KB contiguous on both sides, no registers, no vector instruction; the
// four waves take the rows of a block in turn) into a ring of four slots, three blocks ahead; a wave reads its A and B
// operands of a K step with two ds_read_b128 (the same 16 bytes per lane the direct path loads from global memory) one step
// ahead of the MFMAs.  One s_barrier per block, in the middle of it: "block g + 1 has landed everywhere, block g - 1 is free
// everywhere", then the DMA of block g + 3 goes into the freed slot.  The ds_reads are asm (the compiler would wait for every
// DMA in flight before an LDS read it cannot tell apart) with hand-counted waits.
#define DWW_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 0xf) | (((n) >> 4) << 14) | 0x0f70)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float dwb_f4 __attribute__((ext_vector_type(4)));
typedef float dwb_f2 __attribute__((ext_vector_type(2)));
// (`off` must fold to a constant: the callers sit in fully unrolled loops)
__device__ __forceinline__ void dwb_read4(dwb_f4& dst, uint32_t addr, int off) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory");
}
__device__ __forceinline__ void dwb_read2(dwb_f2& dst, uint32_t addr, int off) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory");
}
template <int KB> struct DwbB;
template <> struct DwbB<4> {
    dwb_f4 v;
    __device__ __forceinline__ void read(uint32_t addr, int off) { dwb_read4(v, addr, off); }
    __device__ __forceinline__ float get(int j) const { return v[j]; }
};
template <> struct DwbB<2> {
    dwb_f2 v;
    __device__ __forceinline__ void read(uint32_t addr, int off) { dwb_read2(v, addr, off); }
    __device__ __forceinline__ float get(int j) const { return v[j]; }
};

// wait until at most KEEP of the wave's LDS reads are outstanding; names the registers the next MFMAs read
template <int KEEP, int KB>
__device__ __forceinline__ void dwb_wait(dwb_f4& a, DwbB<KB>& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b.v) : "i"(KEEP));
}

// A ring slot = [16 rows][NA x 256 floats of dy] then [16 rows][XU x 4 floats of x] (NA = 1 or 2 matrices of 256 columns; XU = 64
// or 16 sixteen-byte units of x per row); a wave tile of 4 x KB blocks.  DMA instructions of this wave per block: rows wave, wave + 4,
// wave + 8, wave + 12 of every dy matrix (1 KB each), and of x either the same four rows (XU = 64) or -- XU = 16: 256 bytes per row,
// rows contiguous in the slot -- rows 4 wave .. 4 wave + 3 in ONE instruction (lane l: row l / 16, unit l % 16).  They go out one
// behind each of the first MFMAs after the block's barrier: issued in a clump they held the wave's issue port with the matrix pipe
// idle (measured: the 64-column bundle, whose blocks are half as long, ran 15 % behind the rest of the launch).
template <int NA, int XU, int KB, bool BIAS>
__device__ __forceinline__ void dwb_run(const DwwArgs& a, const DwbBundle& bd, const DwbWave& wv, int wave, int s0, int s1, float* lds,
                                        float* __restrict__ out, float* __restrict__ bias_out) {
    const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    constexpr int RA = NA * 256, RX = XU * 4;                 // floats per slot row of the two regions
    constexpr int XBASE = DWW_BLOCK * RA;                     // the x region's offset in a slot
    constexpr int SLOT = DWW_BLOCK * (RA + RX);               // floats per ring slot
    constexpr int IPB = 4 * NA + (XU == 64 ? 4 : 1);          // this wave's DMA instructions per block
    static_assert(IPB <= 4 * KB * (DWW_U / 2), "the DMA instructions of a block ride behind the MFMAs of its second half");
    // ---- the copy
    __amdgpu_buffer_rsrc_t rs[NA + 1];
    uint32_t voff[NA + 1], rowb[NA + 1];
#pragma unroll
    for (int q = 0; q <= NA; ++q) {
        const DwbPiece& pc = bd.piece[q];                     // pieces 0 .. NA - 1: the dy matrices; piece NA: x
        const DwwProblem& pr = a.p[pc.problem];
        rs[q] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pc.is_x ? pr.x : pr.dy), 0, (int)(pc.is_x ? pr.x_bytes : pr.dy_bytes), 0x00020000);
        rowb[q] = (pc.is_x ? pr.ldx : pr.lddy) * 4u;
        voff[q] = (q == NA && XU == 16) ? (uint32_t)(lane >> 4) * rowb[q] + (uint32_t)(pc.col0 + 4 * (lane & 15)) * 4u : (uint32_t)(pc.col0 + 4 * lane) * 4u;
    }
    auto issue_one = [&](int g, int idx) {        // instruction idx of block g -> ring slot g & 3 (rows beyond the matrix: zeros)
        float* slot = lds + (g & (DWB_RING - 1)) * SLOT;
        const uint32_t row0 = (uint32_t)(s0 + g * DWW_BLOCK);
        if (idx < 4 * NA) {
            const int j = idx / NA, q = idx % NA, r = wave + 4 * j;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[q], (lds_ptr_t)(slot + r * RA + 256 * q), 16, voff[q], (row0 + r) * rowb[q], 0, 0);
        } else if (XU == 64) {
            const int r = wave + 4 * (idx - 4 * NA);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[NA], (lds_ptr_t)(slot + XBASE + r * RX), 16, voff[NA], (row0 + r) * rowb[NA], 0, 0);
        } else {
            const int r = 4 * wave;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[NA], (lds_ptr_t)(slot + XBASE + r * RX), 16, voff[NA], (row0 + r) * rowb[NA], 0, 0);
        }
    };
    const int nblk = (s1 - s0 + DWW_BLOCK - 1) / DWW_BLOCK;
#pragma unroll
    for (int g = 0; g < DWB_RING - 1; ++g)
#pragma unroll
        for (int idx = 0; idx < IPB; ++idx) issue_one(g, idx);
    f32x16 acc[4][KB];
    float bsum[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bsum[i] = 0.f;
#pragma unroll
        for (int j = 0; j < KB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    // the lane's operand addresses in slot 0, step 0 (bytes): row h; A at a_off + 4 c of the dy region, B at b_off + KB c of the x region
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)lds;
    const uint32_t a0 = lds0 + (uint32_t)(h * RA + wv.a_off + 4 * c) * 4u, b0 = lds0 + (uint32_t)(XBASE + h * RX + wv.b_off + KB * c) * 4u;
    dwb_f4 av[2];
    DwbB<KB> bv[2];
    DWW_VMCNT(2 * IPB);                           // this wave's rows of block 0 have landed ...
    __builtin_amdgcn_s_barrier();                 // ... and everybody else's
    asm volatile("" ::: "memory");
    dwb_read4(av[0], a0, 0);
    bv[0].read(b0, 0);
    for (int g = 0; g < nblk; ++g) {
        const uint32_t so = (uint32_t)(g & (DWB_RING - 1)) * (SLOT * 4u), sn = (uint32_t)((g + 1) & (DWB_RING - 1)) * (SLOT * 4u);
        const uint32_t ac = a0 + so, bc = b0 + so, an = a0 + sn, bn = b0 + sn;
#pragma unroll
        for (int u = 0; u < DWW_U; ++u) {
            if (u == DWW_U / 2) {
                // block g + 1 has landed (this wave's rows; the barrier: everyone's), block g - 1 is free: its slot takes block g + 3
                DWW_VMCNT(IPB);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            // the next step's operands (step 0 of the next block behind step 7), then this step's MFMAs
            if (u + 1 < DWW_U) {
                dwb_read4(av[(u + 1) & 1], ac, 2 * (u + 1) * RA * 4);
                bv[(u + 1) & 1].read(bc, 2 * (u + 1) * RX * 4);
            } else {
                dwb_read4(av[(u + 1) & 1], an, 0);
                bv[(u + 1) & 1].read(bn, 0);
            }
            dwb_wait<2, KB>(av[u & 1], bv[u & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const dwb_f4 x = av[u & 1];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xa = x[i];
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, bv[u & 1].get(j), acc[i][j], 0, 0, 0);
                    const int n = (u - DWW_U / 2) * 4 * KB + i * KB + j;      // MFMAs since the barrier
                    if (u >= DWW_U / 2 && n < IPB) {
                        __builtin_amdgcn_sched_barrier(0);
                        issue_one(g + DWB_RING - 1, n);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (BIAS) asm volatile("v_add_f32 %0, %0, %1" : "+v"(bsum[i]) : "v"(xa));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    dwb_wait<0, KB>(av[0], bv[0]);                // (the read ahead of the last step)
    DWW_VMCNT(0);                                 // nothing may land in LDS after the wave has gone
    dww_store<4, KB, BIAS>(acc, bsum, out, bias_out);
}

// ---- the 2 x 2 bundle in split-bf16 ("bf16x3", round 6): fp32-faithful products at the bf16 MFMA's rate -----------------------------
// Both operands of dW = dY^T X are activations, so both are split on the fly: x = x0 + x1 + x2 (bf16 pieces, round-to-nearest of
// the remainder: exact; csrc/mlp_bf16x3.hip's split8, 5.5 vector instructions per value) and a b ~ a0 b0 + a0 b1 + a1 b0 + a1 b1 +
// a0 b2 + a2 b0 on v_mfma_f32_32x32x16_bf16 (bf16 products are exact in fp32; dropped terms <= 2^-24 |a b|), into ONE accumulator per
// block -- a dW sums thousands of samples: the f32 kernel rounds its running sum as often.  The instruction's contraction index is
// 16 SAMPLES: lane (h, c) supplies samples 8 h .. 8 h + 7, so a ring slot (16 samples) is exactly one K step, and the lane's raw
// values are 8 ds_read_b128 per operand (row 8 h + s, columns 4 c .. 4 c + 3 = one value of each of the tile's four row / column blocks).
// Registers: 256 accumulators (AGPRs); the B planes of this step and of the next (2 x 48), the A planes of two row blocks (2 x 12),
// the raw values of one step of A and one of B (2 x 32).  A step = four phases, one per A row block r: 24 MFMAs (A_r x B_0..3, six
// terms each) with, beside them, the split of A_(r+1) (of A_0 of the next step in phase 3) and of B_r of the NEXT step -- 88 vector
// instructions per 24 MFMAs; the raw B values of the next step are read at the top of the step, the raw A values in phase 3.
using bx8 = __attribute__((ext_vector_type(8))) __bf16;
using bx2 = __attribute__((ext_vector_type(2))) __bf16;
using bxu4 = __attribute__((ext_vector_type(4))) unsigned;
struct BxPlanes { bx8 p[3]; };
__device__ __forceinline__ unsigned bx_pk(float a, float b) {
    dwb_f2 v = {a, b};
    bx2 hh = __builtin_convertvector(v, bx2);              // v_cvt_pk_bf16_f32
    return *reinterpret_cast<unsigned*>(&hh);
}
__device__ __forceinline__ void bx_split8(const float (&v)[8], BxPlanes& o) {
    bxu4 w0, w1, w2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = v[2 * i], x1 = v[2 * i + 1];
        const unsigned u = bx_pk(x0, x1);
        const float r0 = x0 - __uint_as_float(u << 16), r1 = x1 - __uint_as_float(u & 0xffff0000u);
        const unsigned m = bx_pk(r0, r1);
        const float t0 = r0 - __uint_as_float(m << 16), t1 = r1 - __uint_as_float(m & 0xffff0000u);
        w0[i] = u, w1[i] = m, w2[i] = bx_pk(t0, t1);
    }
    o.p[0] = *reinterpret_cast<bx8*>(&w0);
    o.p[1] = *reinterpret_cast<bx8*>(&w1);
    o.p[2] = *reinterpret_cast<bx8*>(&w2);
}
__device__ __forceinline__ void bx_product(const BxPlanes& a, const BxPlanes& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[2], b.p[0], c, 0, 0, 0);
}
// the lane's 8 raw float4s of one operand of one slot: row 8 h + s at `addr` + s * 1024 bytes
__device__ __forceinline__ void bx_read8(dwb_f4 (&raw)[8], uint32_t addr) {
#pragma unroll
    for (int s = 0; s < 8; ++s) dwb_read4(raw[s], addr, s * 256 * 4);
}
__device__ __forceinline__ void bx_wait8(dwb_f4 (&raw)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]), "+v"(raw[7]));
}
__device__ __forceinline__ void bx_split_block(const dwb_f4 (&raw)[8], int r, BxPlanes& o) {
    float v[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) v[s] = raw[s][r];
    bx_split8(v, o);
}

template <bool BIAS>
__device__ __forceinline__ void dwb_run_bx(const DwwArgs& a, const DwbBundle& bd, const DwbWave& wv, int wave, int s0, int s1, float* lds,
                                           float* __restrict__ out, float* __restrict__ bias_out) {
    const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    constexpr int RA = 256, RX = 256, XBASE = DWW_BLOCK * RA, SLOT = DWW_BLOCK * (RA + RX), IPB = 8;
    // ---- the copy (as dwb_run<1, 64>): rows wave, wave + 4, .. of a block are this wave's, dy then x
    __amdgpu_buffer_rsrc_t rs[2];
    uint32_t voff[2], rowb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const DwbPiece& pc = bd.piece[q];
        const DwwProblem& pr = a.p[pc.problem];
        rs[q] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pc.is_x ? pr.x : pr.dy), 0, (int)(pc.is_x ? pr.x_bytes : pr.dy_bytes), 0x00020000);
        rowb[q] = (pc.is_x ? pr.ldx : pr.lddy) * 4u;
        voff[q] = (uint32_t)(pc.col0 + 4 * lane) * 4u;
    }
    auto issue_one = [&](int g, int idx) {
        float* slot = lds + (g & (DWB_RING - 1)) * SLOT;
        const int q = idx >> 2, r = wave + 4 * (idx & 3);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[q], (lds_ptr_t)(slot + q * XBASE + r * 256), 16, voff[q], (uint32_t)(s0 + g * DWW_BLOCK + r) * rowb[q], 0, 0);
    };
    const int nblk = (((s1 - s0 + DWW_BLOCK - 1) / DWW_BLOCK) + 1) & ~1;     // (pairs of steps; a bundle's range is a multiple of 32 samples,
                                                                             // the last range's extra step reads rows beyond the matrix: zeros)
#pragma unroll
    for (int g = 0; g < DWB_RING - 1; ++g)
#pragma unroll
        for (int idx = 0; idx < IPB; ++idx) issue_one(g, idx);
    f32x16 acc[4][4];
    float bsum[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bsum[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)lds;
    const uint32_t a0 = lds0 + (uint32_t)(8 * h * RA + wv.a_off + 4 * c) * 4u, b0 = lds0 + (uint32_t)(XBASE + 8 * h * RX + wv.b_off + 4 * c) * 4u;
    dwb_f4 rawA[8], rawB[8];
    BxPlanes pa[2], pb[2][4];
    auto bias_add = [&](int r) {
        if (BIAS) {
#pragma unroll
            for (int s = 0; s < 8; ++s) bsum[r] += rawA[s][r];
        }
    };
    DWW_VMCNT(2 * IPB);                           // block 0 has landed (this wave's rows; the barrier: everyone's)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    bx_read8(rawA, a0);
    bx_read8(rawB, b0);
    bx_wait8(rawA);
    bx_wait8(rawB);
#pragma unroll
    for (int j = 0; j < 4; ++j) bx_split_block(rawB, j, pb[0][j]);
    bx_split_block(rawA, 0, pa[0]);
    bias_add(0);
    for (int g = 0; g < nblk; g += 2) {
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            const int cur = gg, nxt = gg ^ 1;
            const uint32_t sn = (uint32_t)((g + gg + 1) & (DWB_RING - 1)) * (SLOT * 4u);
            // block g + gg + 1 has landed everywhere, block g + gg - 1 is free: its slot takes block g + gg + 3
            DWW_VMCNT(IPB);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            bx_read8(rawB, b0 + sn);              // the next step's raw B
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r == 3) bx_read8(rawA, a0 + sn);                     // the next step's raw A (this step's A_3 planes exist)
                // 24 MFMAs: A_r x B_0..3 of this step
#pragma unroll
                for (int j = 0; j < 4; ++j) bx_product(pa[r & 1], pb[cur][j], acc[r][j]);
                if (r == 0) {
#pragma unroll
                    for (int idx = 0; idx < IPB; ++idx) issue_one(g + gg + DWB_RING - 1, idx);
                }
                // beside them: the planes of the next A row block, and of B_r of the next step
                if (r < 3) {
                    bx_split_block(rawA, r + 1, pa[(r + 1) & 1]);
                    bias_add(r + 1);
                    if (r == 0) bx_wait8(rawB);
                    bx_split_block(rawB, r, pb[nxt][r]);
                } else {
                    bx_split_block(rawB, 3, pb[nxt][3]);
                    bx_wait8(rawA);
                    bx_split_block(rawA, 0, pa[0]);
                    bias_add(0);
                }
#pragma unroll
                for (int k = 0; k < 24; ++k) {                           // one MFMA, then four of the ~90 vector instructions
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (BIAS) {                                   // (the read-ahead's A_0 of the step behind the last was added: take it back)
#pragma unroll
        for (int s = 0; s < 8; ++s) bsum[0] -= rawA[s][0];
    }
    DWW_VMCNT(0);                                 // nothing may land in LDS after the wave has gone
    dww_store<4, 4, BIAS>(acc, bsum, out, bias_out);
}

__global__ __launch_bounds__(256, 1) void train_dw_wave_kernel(DwwArgs a) {
    // (the wave index through readfirstlane: everything derived from it -- tile, descriptors, sample range -- is wave-uniform and
    // stays in scalar registers; as a function of threadIdx the compiler wraps every buffer load in a waterfall loop)
    extern __shared__ __attribute__((aligned(16))) float dw_lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (blockIdx.x < a.bundle_wgs) {              // ---- a bundle: the workgroup's four tiles over one sample range, rows through LDS
        int bi = 0;
        while (bi + 1 < a.n_bundles && blockIdx.x >= a.bundle[bi + 1].first_wg) ++bi;
        const DwbBundle& bd = a.bundle[bi];
        const DwbWave& wv = bd.wave[wave];
        const uint32_t slice = blockIdx.x - bd.first_wg;
        const int s0 = (int)(slice * bd.len), s1 = min(a.m, s0 + (int)bd.len);
        const DwwTile& tl = a.tile[wv.tile];      // (a wave without a tile points at the bundle's first one and stores nothing)
        float* out = a.workspace + 4ull * tl.partial_off + (size_t)slice * 128 * (32 * tl.kb);
        float* bias_out = tl.bias ? a.workspace + 4ull * tl.bias_off + (size_t)slice * 2 * 128 : nullptr;
        if (bd.n_pieces == 2 && bd.bx) {          // ... in split-bf16
            if (tl.bias) dwb_run_bx<true>(a, bd, wv, wave, s0, s1, dw_lds, out, bias_out);
            else dwb_run_bx<false>(a, bd, wv, wave, s0, s1, dw_lds, out, bias_out);
        } else if (bd.n_pieces == 2) {            // dy[:, 256], x[:, 256]: the 2 x 2 tiles of a 256 x 256 block
            if (tl.bias) dwb_run<1, 64, 4, true>(a, bd, wv, wave, s0, s1, dw_lds, out, bias_out);
            else dwb_run<1, 64, 4, false>(a, bd, wv, wave, s0, s1, dw_lds, out, bias_out);
        } else {                                  // dy_a[:, 256], dy_b[:, 256], x[:, 64]: two 256-output layers on the same narrow input
            if (tl.bias) dwb_run<2, 16, 2, true>(a, bd, wv, wave, s0, s1, dw_lds, out, bias_out);
            else dwb_run<2, 16, 2, false>(a, bd, wv, wave, s0, s1, dw_lds, out, bias_out);
        }
        return;
    }
    const uint32_t g = (blockIdx.x - a.bundle_wgs) * 4u + (uint32_t)wave;
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
    int tiles = 0, segments = 0, groups = 0;
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
        groups += (q.k > 128 ? 1 : 0) + ((q.k & 127) ? 1 : 0) + ((q.k & 127) == 0 && q.k <= 128 ? 1 : 0);
    }
    STNERF_REQUIRE(groups <= DWW_MAX_GROUPS, "train_dw_batch: %d tile groups, at most %d", groups, DWW_MAX_GROUPS);
    STNERF_REQUIRE(tiles <= DWW_MAX_TILES && segments <= DWW_MAX_SEGMENTS, "train_dw_batch: %d tiles (at most %d), %d reduction segments (at most %d)",
                   tiles, DWW_MAX_TILES, segments, DWW_MAX_SEGMENTS);
    return STNERF_OK;
}

// Tiles of a problem: row blocks of 128 (one block of 32 for a layer of <= 4 outputs) x column pieces of 128, the last piece as
// narrow as its columns allow (96 / 64 / 32).  Groups = a problem's tiles of one width (they cost the same and read the same
// rows).  A group that is exactly the 2 x 2 full tiles of a 256-row, >= 256-column layer becomes a BUNDLE (one workgroup per sample
// range, rows through LDS); every other group's tiles are independent wave items.  Sample ranges are sized by MFMA count so that
// all waves of the launch finish together.
static void dww_plan(const stnerf_dw_problem* pr, int32_t count, int64_t m, DwwPlan& P) {
    P = DwwPlan{};
    DwwArgs& K = P.k;
    K.m = (int)m;
    struct G { int first, count, cost, bundle; };
    G groups[DWW_MAX_PROBLEMS * 4];
    int ng = 0, nt = 0, nbundles = 0;
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
            if (nt == first) continue;
            groups[ng++] = G{first, nt - first, nb * kb * 16, -1};      // (cost in sixteenths of a 32 x 32 block pair per two samples)

        }
    }
    // ---- bundles (rows through LDS): groups whose tiles read the same rows at the same time.  Only groups that fill all four waves
    // of a workgroup with EQUAL work: rgb_net.1's three column tiles (128 + 128 + 48 columns on the same dy) were tried as a bundle
    // with an idle fourth wave -- 4.79 GB instead of 4.93 per SpaceNet call on the counters, but 2.29 ms instead of 2.04: a bundle's
    // range is as long as its busiest wave needs, the launch is MFMA-bound, and 24 of 64 slots of that bundle did nothing.
    static const bool no_bundles = getenv("STNERF_DEV_DW_NO_BUNDLES") && getenv("STNERF_DEV_DW_NO_BUNDLES")[0] == '1';
    const bool can = !no_bundles && m >= 4 * DWW_BLOCK;
    auto tile_of = [&](int g, int j) -> DwwTile& { return K.tile[groups[g].first + j]; };
    for (int g = 0; can && g < ng && nbundles < DWB_MAX_BUNDLES; ++g) {
        const G& gr = groups[g];
        if (gr.bundle >= 0) continue;
        const DwwTile& t0 = K.tile[gr.first];
        const stnerf_dw_problem& q = pr[t0.problem];
        if (t0.nb != 4) continue;
        DwbBundle B{};
        if (gr.count == 4 && t0.kb == 4 && q.n == 256) {
            // (a) the 2 x 2 full tiles (n0, k0) = (0, 0), (0, 128), (128, 0), (128, 128) of a layer with 256 outputs
            B.n_pieces = 2, B.row_floats = 512;
            B.piece[0] = DwbPiece{t0.problem, 0, 0, 64, 0};                 // dy[:, 0:256]  -> slot row floats 0 .. 255
            B.piece[1] = DwbPiece{t0.problem, 1, 0, 64, 256};               // x[:, 0:256]   -> 256 .. 511
            for (int w = 0; w < 4; ++w) B.wave[w] = DwbWave{tile_of(g, w).n0, tile_of(g, w).k0, (uint8_t)(gr.first + w), 4, 1, 0};
            static const bool no_bx = getenv("STNERF_DEV_DW_BX") && getenv("STNERF_DEV_DW_BX")[0] == '0';   // (development: the f32 bundle)
            B.bx = no_bx ? 0u : 1u;
            // a split-bf16 wave gets through its samples faster than an f32 one: its ranges are sized by its measured cost
            // (DWW_BX_COST sixteenths of the f32 tile's time per sample), so that both kinds of wave finish together
            static const int bx_cost = getenv("STNERF_DEV_DW_BX_COST") ? atoi(getenv("STNERF_DEV_DW_BX_COST")) : DWW_BX_COST;
            if (B.bx) groups[g].cost = groups[g].cost * bx_cost / 16;
            groups[g].bundle = nbundles;
        } else if (gr.count == 2 && t0.kb == 2 && q.n == 256) {
            // (b) two layers of 256 outputs whose 64-column tiles read the SAME columns of the same matrix (stage1.0 on PE(pos) and the
            // skip connection's columns of stage2.0): one copy of x, both dy
            for (int g2 = g + 1; g2 < ng; ++g2) {
                const G& o = groups[g2];
                const DwwTile& u0 = K.tile[o.first];
                const stnerf_dw_problem& q2 = pr[u0.problem];
                if (o.bundle >= 0 || o.count != 2 || u0.nb != 4 || u0.kb != 2 || q2.n != 256 || q2.ldx != q.ldx || q2.x + u0.k0 != q.x + t0.k0) continue;
                B.n_pieces = 3, B.row_floats = 576;
                B.piece[0] = DwbPiece{t0.problem, 0, 0, 64, 0};             // dy_a[:, 0:256]
                B.piece[1] = DwbPiece{u0.problem, 0, 0, 64, 256};           // dy_b[:, 0:256]
                B.piece[2] = DwbPiece{t0.problem, 1, t0.k0, 16, 512};       // x[:, k0 : k0 + 64]
                for (int w = 0; w < 2; ++w) {
                    B.wave[w] = DwbWave{tile_of(g, w).n0, 0, (uint8_t)(gr.first + w), 2, 1, 0};
                    B.wave[2 + w] = DwbWave{(uint16_t)(256 + tile_of(g2, w).n0), 0, (uint8_t)(o.first + w), 2, 1, 0};
                }
                groups[g].bundle = groups[g2].bundle = nbundles;
                groups[g].cost = 8 * 16, groups[g2].cost = 0;
                break;
            }
        }
        if (groups[g].bundle == nbundles) K.bundle[nbundles++] = B;
    }
    P.n_tiles = nt;
    K.n_bundles = nbundles;
    // sample ranges: slices_g ~ share of the launch's MFMA work, wave slots in all <= the part's (a bundle takes four per range)
    // (a bundle occupies four wave slots per range whatever its waves do; a group that rides in another group's bundle has cost 0)
    int64_t total_cost = 0;
    for (int g = 0; g < ng; ++g) total_cost += groups[g].bundle >= 0 ? (groups[g].cost ? 4 * groups[g].cost : 0) : (int64_t)groups[g].count * groups[g].cost;
    const int64_t most = m <= 0 ? 1 : (m + DWW_MIN_SLICE - 1) / DWW_MIN_SLICE;
    uint16_t slices_of[DWW_MAX_PROBLEMS * 4];
    for (int target = DWW_WAVES;; target -= 8) {
        int64_t items = 0, wgs = 0;
        int nv = 0;
        for (int g = 0; g < ng; ++g) {
            if (groups[g].bundle >= 0 && groups[g].cost == 0) continue;      // (rides along: below)
            int64_t s = ((int64_t)target * groups[g].cost + total_cost / 2) / total_cost;
            s = s < 1 ? 1 : s > most ? most : s;
            // (a split-bf16 bundle works in pairs of 16-sample steps: ranges of a multiple of 32 samples)
            const int64_t unit = groups[g].bundle >= 0 && K.bundle[groups[g].bundle].bx ? 2 * DWW_BLOCK : DWW_BLOCK;
            int64_t len = m <= 0 ? unit : ((m + s - 1) / s + unit - 1) / unit * unit;
            s = m <= 0 ? 1 : (m + len - 1) / len;
            slices_of[g] = (uint16_t)s;
            if (groups[g].bundle >= 0) {
                for (int g2 = 0; g2 < ng; ++g2)
                    if (groups[g2].bundle == groups[g].bundle) slices_of[g2] = (uint16_t)s;
                DwbBundle& B = K.bundle[groups[g].bundle];
                B.first_wg = (uint32_t)wgs, B.slices = (uint32_t)s, B.len = (uint32_t)len;
                wgs += s;
            } else {
                K.group[nv++] = DwwGroup{(uint32_t)items, (uint16_t)groups[g].first, (uint16_t)groups[g].count, (uint16_t)s, 0, (uint32_t)len};
                items += s * groups[g].count;
            }
        }
        K.items = (uint32_t)items;
        K.bundle_wgs = (uint32_t)wgs;
        K.n_groups = nv;
        if (4 * wgs + items <= DWW_WAVES || target <= 8) break;
    }
    // workspace: per tile [slices][rows][cols] (+ [slices][2][rows] bias sums), 16-byte aligned pieces
    int64_t off = 0;
    int ns = 0, max_elems = 0;
    for (int g = 0; g < ng; ++g)
        for (int j = 0; j < groups[g].count; ++j) {
            DwwTile& t = K.tile[groups[g].first + j];
            const stnerf_dw_problem& q = pr[t.problem];
            const int rows = 32 * t.nb, cols = 32 * t.kb, slices = slices_of[g];
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
    // (a bundle's ring: four slots of 16 rows x 512 floats; the wave-item workgroups of the same launch carry the allocation along --
    // with 512 registers per wave a CU holds one workgroup either way)
    const unsigned lds_bytes = P.k.bundle_wgs ? DWB_RING * DWW_BLOCK * 576 * 4 : 0;     // (the largest slot: 16 rows x (512 + 64) floats; 144 KB of the CU's 160)
    static bool attr_set = false;
    if (lds_bytes > 65536 && !attr_set) {
        STNERF_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(train_dw_wave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           DWB_RING * DWW_BLOCK * 576 * 4) == hipSuccess, "train_dw_batch: cannot reserve the LDS ring");
        attr_set = true;
    }
    hipLaunchKernelGGL(train_dw_wave_kernel, dim3(P.k.bundle_wgs + (P.k.items + 3) / 4), dim3(256), lds_bytes, st, P.k);
    STNERF_CHECK_LAUNCH("train_dw_batch");
    P.r.workspace = P.k.workspace;
    P.r.accumulate = accumulate;
    hipLaunchKernelGGL(dw_wave_reduce_kernel, dim3((P.max_elems + 255) / 256, P.n_segments), dim3(256), 0, st, P.r);
    STNERF_CHECK_LAUNCH("train_dw_batch (reduce)");
    return STNERF_OK;
}
