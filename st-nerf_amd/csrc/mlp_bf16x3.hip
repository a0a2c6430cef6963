// Persistent stage kernel, split-bf16 arithmetic ("bf16x3"): fp32-faithful matrix products on the bf16 MFMA pipe.
//
// gfx950 has no tf32/xf32 MFMA and its f32 MFMA runs at the f32 vector rate (157 TF/s), 1/16 of the bf16 rate.  Here every
// fp32 operand is split into THREE bf16 pieces, x = x0 + x1 + x2 (8 + 8 + 8 significand bits: the split is exact and keeps
// fp32's exponent range -- no scaling, no range limit), and a product a*b is evaluated with its six leading cross terms
//     a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0)            (dropped: a1 b2, a2 b1, a2 b2 <= 2^-24 |a b|)
// on v_mfma_f32_32x32x16_bf16 (bf16 products are exact in fp32; f32 accumulate).  The a0 b0 terms go to one accumulator,
// the five small terms (<= 2^-8 of it) to a second one: the hardware rounds the running sum after every 8 products, and
// with one accumulator the 96 MFMAs of a 256-deep layer put 192 roundings at the scale of the full sum -- measured
// 1.09 x the error of an fp32 fma chain (tools/micro/bf16x3_proto.hip); split, the large accumulator sees 32 of them and
// the small one's are 2^-8 smaller: ~0.45 x the fp32 chain's error.
//
// Organisation = csrc/mlp_wave.hip's: a wave owns 32 samples and all features, the accumulator layout of a layer is the
// B-operand layout of the next one (K step t of 16 = registers 8 (t & 1) .. + 7 of block t >> 1, both lane halves), so the
// activations never leave the register file: a layer boundary is { big + small, ReLU, split into three bf16 planes }.
// What differs:
//   * 256-wide layers run as two PASSES of 128 output features (4 blocks x {big, small} = 128 accumulator registers);
//     a pass's outputs are parked in AGPRs (ReLU'd fp32) and split into the bf16 planes under the MFMAs of the NEXT pass:
//     the first pass's under the second half of the second pass, the second pass's under the first half of the next
//     layer's first pass (UnparkHook).
//   * Weights: 6 B per value and 2.7 x the f32 kernel's MFMA rate -- four waves streaming the blob through the vector L1
//     would need ~60 B/clk/CU (measured ceiling 53).  They are fetched ONCE per CU by LDS-DMA (global_load_lds_dwordx4,
//     no registers involved) into a ring of four 24 KB slots in consumption order and read by every wave with
//     ds_read_b128 straight into AGPRs (1 KB contiguous per instruction: conflict-free), one read behind each of the
//     first MFMAs of a unit; one raw s_barrier per slot (= 48 MFMAs) orders "slot landed" and "slot free" at once.
//   * Bias vectors and head weights: one copy per workgroup in LDS (LDS-DMA at the start of a work item).
//   * PE, bias, ReLU, heads, outputs: fp32, as in the exact-f32 kernels; rgb_net.1's direction / time columns come per
//     ray from mlp_raybias.hip (exact f32) as the layer's C operand.
//
// Three kernels are built from this machinery (round 6): the stage kernel of the render path (mlp_bf16x3_stage_kernel<DEEP, NoTapArgs>);
// the same kernel with the training tap (<false, StoreTapArgs>: every layer's post-ReLU output and its mask bits written out as the
// rows pass -- stnerf_train_spacenet_fwd_bf16x3); and the backward chain d x = (d y AND mask) W over the TRANSPOSED weights as a
// second bf16x3 stream (train_space_dx_bx_kernel -- stnerf_train_spacenet_dx_bf16x3).  Packers: host (stnerf_pack_net_bf16x3) and
// device (stnerf_pack_net_bf16x3_device, stnerf_pack_dx_bf16x3_device) at the end of the file.
//
// Reference: modeling/spacenet.py:16-160, modeling/motion_net.py:7-71, modeling/layered_rfrender.py:340-418,495-576; training:
// engine/layered_trainer.py:281 (loss.backward() through modeling/spacenet.py:101-160).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "mlp_wave_common.h"
#include "mlp_bf16x3.h"

namespace stnerf {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int BX_LDS_RING = BX_RING * BX_SLOT;
constexpr int BX_LDS_ENC = WV_NW * WV_ENC_FLOATS * 4;
constexpr int BX_LDS_CONST = (BX_CONST_SPACE + BX_CONST_MOTION) * 4;
constexpr int BX_LDS = BX_LDS_RING + BX_LDS_ENC + BX_LDS_CONST + 16 + STNERF_MAX_LAYERS * 8 + (STNERF_MAX_LAYERS + 1) * 4 + 12;
static_assert(BX_LDS <= 160 * 1024, "bf16x3 stage kernel: LDS budget");

#define BX_SB() __builtin_amdgcn_sched_barrier(0)

// Optional per-phase cycle accounting (development builds: -DSTNERF_BX_PROF, tools/bx_prof.py): every wave adds its clock
// deltas per phase; read back with stnerf_debug_bx_phases().
#ifdef STNERF_BX_PROF
static __device__ unsigned long long g_bxphase[16];
struct BxProf {
    unsigned long long t, acc[16];
};
#define BXP_PARAM , BxProf& bp
#define BXP_ARG , bp
#define BXP(i) do { const unsigned long long n_ = clock64(); bp.acc[i] += n_ - bp.t; bp.t = n_; } while (0)
#else
#define BXP_PARAM
#define BXP_ARG
#define BXP(i) do { } while (0)
#endif
enum { BXP_TOP = 0, BXP_MOTION_ENC = 1, BXP_MOTION_PASS = 2, BXP_MOTION_FIN = 3, BXP_PE = 4, BXP_PASS = 5, BXP_PARK = 6, BXP_ACT = 7,
       BXP_LOADC = 8, BXP_SIGMA = 9, BXP_RGB_TAIL = 10, BXP_END = 11, BXP_ITEMS = 12 };
// s_waitcnt vmcnt(n) only (gfx9 encoding: vmcnt = [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8])
#define BX_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 0xf) | (((n) >> 4) << 14) | 0x0f70)

// ---------------------------------------------------------------------------------------------
// fp32 -> three bf16 pieces (round to nearest even at every step: x = p0 + p1 + p2 exactly)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    f32x2_ v = {a, b};
    bf16x2 h = __builtin_convertvector(v, bf16x2);  // v_cvt_pk_bf16_f32
    return *reinterpret_cast<unsigned*>(&h);
}
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& p0, bf16x8& p1, bf16x8& p2) {
    // (scalar source on purpose: the compiler pairs the subtractions into v_pk_add_f32 by itself; written on explicit
    // 2-vectors the value array is not promoted to registers and the split goes through scratch memory)
    u32x4 w0, w1, w2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = v[2 * i], x1 = v[2 * i + 1];
        const unsigned u = pk_bf16(x0, x1);
        const float r0 = x0 - __uint_as_float(u << 16), r1 = x1 - __uint_as_float(u & 0xffff0000u);
        const unsigned m = pk_bf16(r0, r1);
        const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
        w0[i] = u;
        w1[i] = m;
        w2[i] = pk_bf16(s0, s1);
    }
    p0 = *reinterpret_cast<bf16x8*>(&w0);
    p1 = *reinterpret_cast<bf16x8*>(&w1);
    p2 = *reinterpret_cast<bf16x8*>(&w2);
}

// ---------------------------------------------------------------------------------------------
// Training tap (the split-bf16 twin of mlp_wave.hip's StoreTap): every layer's post-ReLU output goes to the row-major
// matrices of StoreTapArgs on its way into the park, the ReLU masks as bit planes beside them.  Plain global stores from a
// per-lane pointer (two registers, built where they are used): with BUFFER stores -- descriptor on the scalar ALU, rows past the
// end of the launch dropped by the range check, one offset register -- the four aligned scalar registers of a descriptor do not
// exist at a boundary of this kernel, and the allocator answers with ~170 spilled vector registers (round 6; a global store with
// the same data: none).  Stores count in vmcnt like the weight ring's LDS-DMA: the slot turns behind a boundary that stored wait
// for `6 + stores` (slot_turn<ST>), or they would wait for the stores too.
// ---------------------------------------------------------------------------------------------
struct BxNoTap {
    static constexpr bool on = false;
};
struct BxStoreTap {
    static constexpr bool on = true;
    const StoreTapArgs* a;   // the kernel's argument block (scalar loads)
    uint32_t row0;           // first row of the work item
    uint32_t nrows;          // rows of the launch in this item (1 .. WV_ITEM)
    int wave;
};
constexpr int BX_TAP_PARK = 18;   // VMEM stores of one park: 16 x 16 B of activations + two mask words
constexpr int BX_TAP_PE = 8;
__device__ __forceinline__ uint32_t tap_row_in_item(const BxStoreTap& tap, int lane) { return (uint32_t)(tap.wave * WV_ROWS + (lane & 31)); }
__device__ __forceinline__ bool tap_valid(const BxStoreTap& tap, int lane) {
#ifdef STNERF_DEV_TAP_ALWAYS_VALID      // (development A/B: what the per-store validity branches cost; launches of whole items only)
    return true;
#else
    return tap_row_in_item(tap, lane) < tap.nrows;
#endif
}
// this lane's 16 bytes of (stage, column col0) of its row; stage: 0 .. 7 or TAP_PE
__device__ __forceinline__ float* tap_row(const BxStoreTap& tap, int stage, int col0, int lane) {
    // (opaque here: the per-lane addresses of all fifteen boundaries are loop invariants of the item loop -- hoisted, they are live
    // through every K loop of a kernel that has no register to spare)
    asm volatile("" : "+v"(lane));
    const bool is_pe = stage == TAP_PE;
    float* base = is_pe ? tap.a->pe : tap.a->buf[is_pe ? 0 : stage];
    const uint32_t ld = (uint32_t)(is_pe ? tap.a->ld_pe : tap.a->ld[is_pe ? 0 : stage]);
    return base + (size_t)(tap.row0 + tap_row_in_item(tap, lane)) * ld + (uint32_t)(col0 + 4 * (lane >> 5));
}
// word (col0 / 128) * 2 of this lane's four mask words of its row (null: the caller wants no masks)
__device__ __forceinline__ uint32_t* tap_bits_row(const BxStoreTap& tap, int stage, int col0, int lane) {
    asm volatile("" : "+v"(lane));
    uint32_t* bw = tap.a->bits;
    if (!bw) return nullptr;
    return bw + (size_t)stage * (size_t)tap.a->bits_stride + (size_t)(tap.row0 + tap_row_in_item(tap, lane)) * 8u +
           (uint32_t)(4 * (lane >> 5) + (col0 ? 2 : 0));
}
// value i of block fb <-> bit (16 fb + i) & 31 of word fb >> 1 (mlp_wave.hip: StoreTap); a post-ReLU value is +0 or positive
__device__ __forceinline__ void tap_bit(uint32_t& w, float v, int pos /* a constant once the loops are unrolled */) {
    uint32_t t;
    asm volatile("v_min_u32 %1, 1, %2\n\tv_lshl_or_b32 %0, %1, %3, %0" : "+v"(w), "=&v"(t) : "v"(v), "i"(pos));
}

// ---------------------------------------------------------------------------------------------
// The weight stream: which global slot the workgroup fetches next (wave-uniform), the ring, the A-operand buffers.
// ---------------------------------------------------------------------------------------------
struct Seg {
    const char* p;
    uint32_t left;  // slots
};
struct ABuf {
    bf16x8 p[3];  // the three pieces of one unit's A operand (AGPRs)
};
struct Ctx {
    Seg seg[4];         // this work item's networks (deformation net, SpaceNet), then the next item's
    const char* idle;   // a valid source when nothing is left to fetch
    uint32_t gi;        // slots issued so far
    uint32_t rcur, rnext;  // this lane's LDS byte address inside the slot being consumed / the next one
    uint32_t gc;        // slots consumed so far
    char* ring;
    int wave, lane;
    bool st_on;         // training kernels: this wave issues the boundary stores its slot turns count in (slot_turn<ST>)
    ABuf A[4];
};

__device__ __forceinline__ const char* feed_next(Ctx& cx) {
    const char* p = cx.idle;
    bool done = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool take = !done && cx.seg[i].left != 0;
        p = take ? cx.seg[i].p : p;
        cx.seg[i].p += take ? BX_SLOT : 0;
        cx.seg[i].left -= take ? 1u : 0u;
        done = done || take;
    }
    return p;
}

// This wave's quarter (6 x 1 KB) of the next slot of the stream on its way into ring slot gi & 3.  Issued in a clump, the
// six LDS-DMA instructions (each with its M0 write) hold the wave's issue port for ~240 cycles with one MFMA in flight to
// cover them (tools/micro/bf16x3_proto.hip: 5 of 42 cycles per MFMA); dma_begin() does the scalar part at the slot turn,
// dma_chunk<c>() goes out one behind each of the MFMAs of the slot's last two units that carry no operand read.
struct Dma {
    const char* src;                                  // this lane's source address of chunk 0
    __attribute__((address_space(3))) char* dst;      // (wave-uniform) LDS destination of chunk 0
};
__device__ __forceinline__ void dma_begin(Ctx& cx, Dma& d) {
    d.src = feed_next(cx) + cx.wave * (6 * BX_CHUNK) + cx.lane * 16;
    d.dst = (__attribute__((address_space(3))) char*)(cx.ring) + (cx.gi & (BX_RING - 1)) * BX_SLOT + cx.wave * (6 * BX_CHUNK);
    cx.gi += 1;
}
template <int C>
__device__ __forceinline__ void dma_chunk(const Dma& d) {
    __builtin_amdgcn_global_load_lds(d.src + C * BX_CHUNK, (__attribute__((address_space(3))) void*)(d.dst + C * BX_CHUNK), 16, 0, 0);
}
__device__ __forceinline__ void dma_issue(Ctx& cx) {   // (all six at once: priming)
    Dma d;
    dma_begin(cx, d);
    dma_chunk<0>(d);
    dma_chunk<1>(d);
    dma_chunk<2>(d);
    dma_chunk<3>(d);
    dma_chunk<4>(d);
    dma_chunk<5>(d);
}

// A operands: ds_read_b128 straight into AGPRs, as asm -- the compiler does not count these reads; a_wait() is their
// s_waitcnt and names every destination, so no consumer can be scheduled above it.  `keep` = reads of LATER units that may
// stay outstanding (LDS returns in order; anything else in flight only makes the wait more conservative).
template <int OFF>
__device__ __forceinline__ void a_read(bf16x8& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst) : "v"(addr), "i"(OFF) : "memory");
}
template <int KEEP>
__device__ __forceinline__ void a_wait(ABuf& A) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+a"(A.p[0]), "+a"(A.p[1]), "+a"(A.p[2]) : "i"(KEEP));
}

__device__ __forceinline__ f32x16 mfma_bf16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// One unit = the six MFMAs of (K step, feature block): five into `small` (the first of a pass starts it from 0), a0 b0
// into `big`.  U = position in the slot (K step U >> 2, block U & 3); the reads of unit U + 2 go out one behind each of the
// first three MFMAs (four buffers; the one they fill was consumed by unit U - 2).  With the reads only one unit ahead --
// 160 cycles -- the K passes ran at 1.33 x their MFMA time (tools/bx_prof.py): the LDS round trip under four waves' load is
// longer than that.
// What may ride in the shadow of a unit's last three MFMAs (the ones without an operand read behind them): ~4 vector
// instructions each are free (tools/micro/bf16x3_proto.hip: 2 per MFMA cost 0.5 cycles of 32).
struct NoHook {
    template <int U, int I>
    __device__ __forceinline__ void at() {}
};
template <int U, bool FIRST, bool BIG0 = false, int DMA0 = -1, class Hook = NoHook>
__device__ __forceinline__ void unit(Ctx& cx, f32x16& big, f32x16& small, const bf16x8& b0, const bf16x8& b1, const bf16x8& b2,
                                     const Dma* dma = nullptr, Hook* hook = nullptr) {
    ABuf& cur = cx.A[U & 3];
    ABuf& nx = cx.A[(U + 2) & 3];
    constexpr int OFF = ((U + 2) & 7) * BX_UNIT;
    const uint32_t ra = (U + 2) < 8 ? cx.rcur : cx.rnext;
    a_wait<3>(cur);   // (the three reads of unit U + 1 may stay in flight)
    BX_SB();
    if (FIRST) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        small = mfma_bf16(cur.p[2], b0, z);
    } else {
        small = mfma_bf16(cur.p[2], b0, small);
    }
    BX_SB();
    a_read<OFF>(nx.p[0], ra);
    BX_SB();
    small = mfma_bf16(cur.p[0], b2, small);
    BX_SB();
    a_read<OFF + BX_CHUNK>(nx.p[1], ra);
    BX_SB();
    small = mfma_bf16(cur.p[1], b1, small);
    BX_SB();
    a_read<OFF + 2 * BX_CHUNK>(nx.p[2], ra);
    BX_SB();
    small = mfma_bf16(cur.p[1], b0, small);
    if (DMA0 >= 0) {
        BX_SB();
        dma_chunk<DMA0 < 0 ? 0 : DMA0>(*dma);
        BX_SB();
    }
    if constexpr (!std::is_same<Hook, NoHook>::value) {
        BX_SB();
        hook->template at<U, 0>();
        BX_SB();
    }
    small = mfma_bf16(cur.p[0], b1, small);
    if (DMA0 >= 0) {
        BX_SB();
        dma_chunk<DMA0 < 0 ? 0 : DMA0 + 1>(*dma);
        BX_SB();
    }
    if constexpr (!std::is_same<Hook, NoHook>::value) {
        BX_SB();
        hook->template at<U, 1>();
        BX_SB();
    }
    if (FIRST && BIG0) {  // (rgb_net.1: its C operand is added behind the K loop, see space_bx)
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        big = mfma_bf16(cur.p[0], b0, z);
    } else {
        big = mfma_bf16(cur.p[0], b0, big);
    }
    if (DMA0 >= 0) {
        BX_SB();
        dma_chunk<DMA0 < 0 ? 0 : DMA0 + 2>(*dma);
    }
    if constexpr (!std::is_same<Hook, NoHook>::value) {
        BX_SB();
        hook->template at<U, 2>();
    }
    BX_SB();
}

// Between units 5 and 6 of a slot: the next slot has landed everywhere and the previous one is free everywhere (every wave
// has issued -- and, to get here, completed -- its reads of it); the fetch three slots ahead goes into its place.
// ST: VMEM stores this wave issued behind the DMA of the slot it is waiting for (the training kernels' boundary stores).  The
// immediate must never exceed 6 + the stores really in flight, or the wait no longer covers the wave's share of the LDS-DMA: a
// wave whose 32 rows are all past the end of the launch branches around its conditional stores (and a launch without mask planes
// skips two per boundary) -- such a wave runs with st_on == false and waits for vmcnt(6), which is always sufficient.
template <int ST = 0>
__device__ __forceinline__ void slot_turn(Ctx& cx, Dma& d) {
    if constexpr (ST > 0) {
        if (cx.st_on)      // (wave-uniform: a scalar branch)
            BX_VMCNT(6 + ST);
        else
            BX_VMCNT(6);
    } else {
        BX_VMCNT(6);
    }
    __builtin_amdgcn_s_barrier();
    dma_begin(cx, d);
    BX_SB();
}
__device__ __forceinline__ void slot_done(Ctx& cx) {
    cx.gc += 1;
    cx.rcur = cx.rnext;
    // the slot after it, wrapping at the end of the ring (from rnext itself: a third per-lane address kept through the item
    // only to be added to here was spilled, and its reload waits behind vmcnt(0) -- the DMA queue)
    cx.rnext = cx.rnext + (((cx.gc + 1) & (BX_RING - 1)) == 0 ? BX_SLOT - BX_RING * BX_SLOT : BX_SLOT);
}

// one ring slot: K steps k0, k1 (their B operands: the three planes of the input) for the pass's four blocks
template <bool FIRST, bool BIG0 = false, class Hook = NoHook, int ST = 0>
__device__ __forceinline__ void slot(Ctx& cx, f32x16 (&big)[4], f32x16 (&small)[4], const bf16x8& k0p0, const bf16x8& k0p1,
                                     const bf16x8& k0p2, const bf16x8& k1p0, const bf16x8& k1p1, const bf16x8& k1p2, Hook* hook = nullptr) {
    unit<0, FIRST, BIG0, -1, Hook>(cx, big[0], small[0], k0p0, k0p1, k0p2, nullptr, hook);
    unit<1, FIRST, BIG0, -1, Hook>(cx, big[1], small[1], k0p0, k0p1, k0p2, nullptr, hook);
    unit<2, FIRST, BIG0, -1, Hook>(cx, big[2], small[2], k0p0, k0p1, k0p2, nullptr, hook);
    unit<3, FIRST, BIG0, -1, Hook>(cx, big[3], small[3], k0p0, k0p1, k0p2, nullptr, hook);
    unit<4, false, false, -1, Hook>(cx, big[0], small[0], k1p0, k1p1, k1p2, nullptr, hook);
    unit<5, false, false, -1, Hook>(cx, big[1], small[1], k1p0, k1p1, k1p2, nullptr, hook);
    Dma d;
    slot_turn<ST>(cx, d);
    unit<6, false, false, 0, Hook>(cx, big[2], small[2], k1p0, k1p1, k1p2, &d, hook);
    unit<7, false, false, 3, Hook>(cx, big[3], small[3], k1p0, k1p1, k1p2, &d, hook);
    slot_done(cx);
}

// K steps KS0 .. KS0 + 2 NSLOT - 1 of the activation planes
// (ST: stores issued in the boundary in front of the pass -- they are younger than the DMA its first TWO slot turns wait for)
template <int KS0, int NSLOT, bool FIRST, bool BIG0 = false, int ST = 0>
__device__ __forceinline__ void pass_act(Ctx& cx, f32x16 (&big)[4], f32x16 (&small)[4], const bf16x8 (&act)[3][16]) {
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {
        const int k = KS0 + 2 * sl;
        if (sl == 0)
            slot<FIRST, BIG0, NoHook, ST>(cx, big, small, act[0][k], act[1][k], act[2][k], act[0][k + 1], act[1][k + 1], act[2][k + 1]);
        else if (sl == 1)
            slot<false, false, NoHook, ST>(cx, big, small, act[0][k], act[1][k], act[2][k], act[0][k + 1], act[1][k + 1], act[2][k + 1]);
        else
            slot<false>(cx, big, small, act[0][k], act[1][k], act[2][k], act[0][k + 1], act[1][k + 1], act[2][k + 1]);
    }
}

// ---------------------------------------------------------------------------------------------
// Layer boundaries (vector ALU).  Register 4 q + r of block fb <-> feature 32 fb + 8 q + 4 h + r of the pass.
// ---------------------------------------------------------------------------------------------
// big = this lane's 64 values of a 128-float vector in LDS (the next pass's bias = the C operand of its a0 b0 chain)
__device__ __forceinline__ void load_c(f32x16 (&big)[4], const float* v128, int lane) {
    const float4* b4 = reinterpret_cast<const float4*>(v128) + (lane >> 5);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b = b4[fb * 8 + 2 * q];
            big[fb][4 * q + 0] = b.x;
            big[fb][4 * q + 1] = b.y;
            big[fb][4 * q + 2] = b.z;
            big[fb][4 * q + 3] = b.w;
            BX_SB();
        }
}
// block fb of load_c: issued as soon as the block's accumulators have been consumed by a boundary pass, so that the LDS
// round trip runs under the vector work of the following blocks (as relu_rebias of mlp_wave.hip)
__device__ __forceinline__ void load_c_block(f32x16& bigfb, const float* v128, int fb, int lane) {
    const float4* b4 = reinterpret_cast<const float4*>(v128) + (lane >> 5);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 b = b4[fb * 8 + 2 * q];
        bigfb[4 * q + 0] = b.x;
        bigfb[4 * q + 1] = b.y;
        bigfb[4 * q + 2] = b.z;
        bigfb[4 * q + 3] = b.w;
    }
}
__device__ __forceinline__ float out_relu(const f32x16& big, const f32x16& small, int i) { return relu_bits(big[i] + small[i]); }

// the first pass's outputs wait in AGPRs while the second pass still reads the layer's input
struct Park {
    float v[64];
};
__device__ __forceinline__ void park_put(float& dst, float v) { asm("v_accvgpr_write_b32 %0, %1" : "=a"(dst) : "v"(v)); }
__device__ __forceinline__ float park_get(const float& src) {
    float v;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(src));
    return v;
}

// The heads (density_net.0: 256 -> 1; the 128 -> 3 colour / flow layers) accumulate in FP64 on the vector ALU: 128 + 192
// v_fma_f64 per lane and work item are nothing next to 4.3 k MFMAs, and they take the heads out of the error budget -- with
// the split accumulators the backbone's h is ~2.3 x closer to fp64 than an fp32 fma chain's, and a 256-term fp32 dot
// product on top of it (however it is grouped) was most of the error left in sigma.
__device__ __forceinline__ double pair_sum_d(double x) {  // x + the other lane's (lane ^ 32) x, in every lane
    return x + __shfl_xor(x, 32, 64);
}
struct SigAcc {
    double c[4];
};
__device__ __forceinline__ void sig_take(SigAcc& sg, const float (&v)[8], const float* w, int lane, int feat0 /* of v[0], h = 0 */) {
    const float4* w4 = reinterpret_cast<const float4*>(w + feat0) + (lane >> 5);
    const float4 wa = w4[0], wb = w4[2];
    sg.c[0] = fma((double)v[0], (double)wa.x, sg.c[0]);
    sg.c[1] = fma((double)v[1], (double)wa.y, sg.c[1]);
    sg.c[2] = fma((double)v[2], (double)wa.z, sg.c[2]);
    sg.c[3] = fma((double)v[3], (double)wa.w, sg.c[3]);
    sg.c[0] = fma((double)v[4], (double)wb.x, sg.c[0]);
    sg.c[1] = fma((double)v[5], (double)wb.y, sg.c[1]);
    sg.c[2] = fma((double)v[6], (double)wb.z, sg.c[2]);
    sg.c[3] = fma((double)v[7], (double)wb.w, sg.c[3]);
}

// pass A of a 256-wide layer: relu(big + small) -> park
// next_c: the 128 C-operand values (bias) of the pass that follows
__device__ __forceinline__ void finish_park(f32x16 (&big)[4], const f32x16 (&small)[4], Park& pk, const float* next_c, int lane) {
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int j = 0; j < 8; ++j) park_put(pk.v[16 * fb + 8 * t + j], out_relu(big[fb], small[fb], 8 * t + j));
            BX_SB();  // (group by group: keeps the live values of this straight-line code bounded)
        }
        if (next_c) {  // (uniform; nullptr where the next pass starts from 0: rgb_net.1)
            load_c_block(big[fb], next_c, fb, lane);
            BX_SB();
        }
    }
}
// the same with the training tap: the pass's 128 outputs = columns col0 .. col0 + 127 of stage `stage`'s matrix
__device__ __forceinline__ void finish_park(f32x16 (&big)[4], const f32x16 (&small)[4], Park& pk, const float* next_c, int lane, const BxNoTap&,
                                            int, int) {
    finish_park(big, small, pk, next_c, lane);
}
// (Register budget: the kernel's 256 arch registers are full -- 192 of activation planes -- and the park itself runs on two
// temporaries.  Here: four values at a time (the data of one 16-byte store), one mask word, the row pointer; the kernel's tap
// variant makes room for them by recomputing per-row values it would otherwise carry through the item, see
// mlp_bf16x3_stage_kernel.)
template <bool PARK = true>
__device__ __forceinline__ void finish_park(f32x16 (&big)[4], const f32x16 (&small)[4], Park& pk, const float* next_c, int lane,
                                            const BxStoreTap& tap, int stage, int col0) {
    const bool valid = tap_valid(tap, lane);
    float4* dst = reinterpret_cast<float4*>(tap_row(tap, stage, col0, lane));
    uint32_t w = 0u;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = out_relu(big[fb], small[fb], 4 * q + r);
                if (PARK) park_put(pk.v[16 * fb + 4 * q + r], v[r]);
#ifndef STNERF_DEV_TAP_NO_BITS      // (development A/B of what the tap costs: profiles/retired_designs.md)
                tap_bit(w, v[r], (16 * fb + 4 * q + r) & 31);
#endif
            }
            // register 4 q + r of block fb <-> feature 32 fb + 8 q + 4 h + r: 16 bytes per (fb, q), the two lanes of a sample side by side
#ifndef STNERF_DEV_TAP_NO_STORES
            if (valid) dst[fb * 8 + 2 * q] = make_float4(v[0], v[1], v[2], v[3]);
#endif
            if (q & 1) BX_SB();
        }
        if (fb & 1) {
            uint32_t* bw = tap_bits_row(tap, stage, col0, lane);
            if (valid && bw) bw[fb >> 1] = w;
            if (stage == 7 && fb == 3 && valid && bw) {   // a 128-wide stage (rgb_net.1's output) owns all four words of the lane
                bw[2] = 0u;
                bw[3] = 0u;
            }
            w = 0u;
            BX_SB();
        }
        if (next_c) {
            load_c_block(big[fb], next_c, fb, lane);
            BX_SB();
        }
    }
}
// the pass's outputs -> K steps KS0 .. KS0 + 7 of the activation planes
template <int KS0>
__device__ __forceinline__ void finish_act(f32x16 (&big)[4], const f32x16 (&small)[4], bf16x8 (&act)[3][16], const float* next_c, int lane) {
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = out_relu(big[fb], small[fb], 8 * t + j);
            split8(v, act[0][KS0 + 2 * fb + t], act[1][KS0 + 2 * fb + t], act[2][KS0 + 2 * fb + t]);
            BX_SB();
        }
        if (next_c) {  // (uniform; nullptr where the next pass starts from 0: rgb_net.1)
            load_c_block(big[fb], next_c, fb, lane);
            BX_SB();
        }
    }
}
// sigma head (density_net.0, 256 -> 1) on the last backbone layer's outputs, before they are converted: features 0..127 wait
// in the park, 128..255 in the accumulators.  (A pass of its own over the values -- 2 x 128 reads once per work item --
// rather than a flavour of finish_act: the layer loop's body must define the activation planes on ONE path; with two
// the planes' 192 registers meet in phi nodes the register coalescer cannot resolve, and half of them get copied.)
__device__ __forceinline__ float sigma_head(const f32x16 (&big)[4], const f32x16 (&small)[4], const Park& pk, const float* wsig, float bias,
                                            int lane) {
    SigAcc sg;
#pragma unroll
    for (int i = 0; i < 4; ++i) sg.c[i] = 0.0;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = park_get(pk.v[16 * fb + 8 * t + j]);
            sig_take(sg, v, wsig, lane, 32 * fb + 16 * t);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = out_relu(big[fb], small[fb], 8 * t + j);
            sig_take(sg, v, wsig, lane, 128 + 32 * fb + 16 * t);
            BX_SB();
        }
    return (float)((double)bias + pair_sum_d((sg.c[0] + sg.c[1]) + (sg.c[2] + sg.c[3])));
}
template <int KS0 = 0>
__device__ __forceinline__ void unpark_act(const Park& pk, bf16x8 (&act)[3][16]) {
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = park_get(pk.v[16 * fb + 8 * t + j]);
            split8(v, act[0][KS0 + 2 * fb + t], act[1][KS0 + 2 * fb + t], act[2][KS0 + 2 * fb + t]);
            BX_SB();
        }
}

// The same conversion as unpark_act's, for ONE block, spread over the MFMA shadows of one ring slot (a pair of values per
// unit, in three steps).  BOTH passes of a 256-wide layer carry one:
//   * KS_OUT = 0, inside the SECOND pass, slots 4 .. 7: K steps 0 .. 7 of the input planes are dead by then and take the first
//     pass's outputs (round 3);
//   * KS_OUT = 8, inside the FIRST pass of the NEXT layer, slots 0 .. 3 (round 4): the previous layer's second pass parks its
//     outputs as well (finish_park, 320 cycles) instead of converting them in the open (finish_act, 1350), and the conversion
//     into K steps 8 .. 15 -- which the pass reads from slot 4 on -- rides under the MFMAs of K steps 0 .. 7.
// 832 of a layer's 1344 boundary instructions under MFMAs; what stays in the open are the two parks (ReLU + v_accvgpr_write).
template <int FB, int KS_OUT>
struct UnparkHook {
    const Park& pk;
    bf16x8 (&act)[3][16];
    float x0, x1, r0, r1;
    unsigned w0, w1;
    template <int U, int I>
    __device__ __forceinline__ void at() {
        constexpr int T = KS_OUT + 2 * FB + (U >> 2), W = U & 3;
        if (I == 0) {
            x0 = park_get(pk.v[16 * FB + 2 * U]);
            x1 = park_get(pk.v[16 * FB + 2 * U + 1]);
            w0 = pk_bf16(x0, x1);
        } else if (I == 1) {
            r0 = x0 - __uint_as_float(w0 << 16);
            r1 = x1 - __uint_as_float(w0 & 0xffff0000u);
            w1 = pk_bf16(r0, r1);
        } else {
            const float s0 = r0 - __uint_as_float(w1 << 16), s1 = r1 - __uint_as_float(w1 & 0xffff0000u);
            reinterpret_cast<u32x4&>(act[0][T])[W] = w0;
            reinterpret_cast<u32x4&>(act[1][T])[W] = w1;
            reinterpret_cast<u32x4&>(act[2][T])[W] = pk_bf16(s0, s1);
        }
    }
};
// one slot of a pass (K steps 2 FB, 2 FB + 1 of the half the pass is reading: the upper one for KS_OUT = 0, the lower one for
// KS_OUT = 8) with block FB of the park converted into the OTHER half of the planes on the way
template <int FB, int KS_OUT, bool FIRST = false, bool BIG0 = false, int ST = 0>
__device__ __forceinline__ void slot_unpark(Ctx& cx, f32x16 (&big)[4], f32x16 (&small)[4], bf16x8 (&act)[3][16], const Park& pk) {
    constexpr int k = (8 - KS_OUT) + 2 * FB;
    UnparkHook<FB, KS_OUT> hook{pk, act, 0.f, 0.f, 0.f, 0.f, 0u, 0u};
    slot<FIRST, BIG0, UnparkHook<FB, KS_OUT>, ST>(cx, big, small, act[0][k], act[1][k], act[2][k], act[0][k + 1], act[1][k + 1], act[2][k + 1], &hook);
}
// second pass of a 256-wide layer: 16 K steps, the parked first pass converted on the way
template <int ST = 0, bool BIG0 = false>
__device__ __forceinline__ void pass_b_unpark(Ctx& cx, f32x16 (&big)[4], f32x16 (&small)[4], bf16x8 (&act)[3][16], const Park& pk) {
    pass_act<0, 4, true, BIG0, ST>(cx, big, small, act);
    slot_unpark<0, 0>(cx, big, small, act, pk);
    slot_unpark<1, 0>(cx, big, small, act, pk);
    slot_unpark<2, 0>(cx, big, small, act, pk);
    slot_unpark<3, 0>(cx, big, small, act, pk);
}
// K steps 0 .. 7 of a pass whose input's upper half (features 128 .. 255 = the previous layer's second pass) still waits in
// the park: converted into K steps 8 .. 15 on the way.  The caller continues with pass_act<8, ..., false>.
template <bool BIG0 = false, int ST = 0>
__device__ __forceinline__ void pass_a_unpark(Ctx& cx, f32x16 (&big)[4], f32x16 (&small)[4], bf16x8 (&act)[3][16], const Park& pk) {
    slot_unpark<0, 8, true, BIG0, ST>(cx, big, small, act, pk);
    slot_unpark<1, 8, false, false, ST>(cx, big, small, act, pk);
    slot_unpark<2, 8>(cx, big, small, act, pk);
    slot_unpark<3, 8>(cx, big, small, act, pk);
}

// 128 -> 3 head (rgb_net's last layer, MotionNet's flow) on relu(big + small): w = [3][128] in LDS; fp64 accumulation, two
// chains per output and lane
__device__ __forceinline__ void head3(const f32x16 (&big)[4], const f32x16 (&small)[4], const float* w, const float* __restrict__ b3,
                                      int lane, float (&out)[3]) {
    const float4* w4 = reinterpret_cast<const float4*>(w) + (lane >> 5);
    double c[3][2];
#pragma unroll
    for (int o = 0; o < 3; ++o) c[o][0] = c[o][1] = 0.0;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (double)out_relu(big[fb], small[fb], 4 * q + r);
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float4 wv = w4[o * 32 + fb * 8 + 2 * q];
                double& cc = c[o][q & 1];
                cc = fma(v[0], (double)wv.x, cc);
                cc = fma(v[1], (double)wv.y, cc);
                cc = fma(v[2], (double)wv.z, cc);
                cc = fma(v[3], (double)wv.w, cc);
            }
            BX_SB();
        }
#pragma unroll
    for (int o = 0; o < 3; ++o) out[o] = (float)((double)b3[o] + pair_sum_d(c[o][0] + c[o][1]));
}

// K steps 0 .. STEPS - 1 of the activation planes from the wave's staged encoding (feature 16 t + 8 h + j; NQ quads staged)
struct NoEncTap {
    __device__ __forceinline__ void operator()(int, const float (&)[8]) const {}
};
template <int STEPS, int NQ, class EncTap = NoEncTap>
__device__ __forceinline__ void enc_to_act(const float* encw, int lane, bf16x8 (&act)[3][16], EncTap enc_tap = EncTap()) {
    const float4* e4 = reinterpret_cast<const float4*>(encw);
    const int h = lane >> 5, c = lane & 31;
#pragma unroll
    for (int t = 0; t < STEPS; ++t) {
        float v[8];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (4 * t + 2 + qq < NQ) {  // (static) both lane halves inside the staged quads
                x = e4[(4 * t + 2 * h + qq) * WV_ROWS + c];
            } else if (4 * t + qq < NQ) {  // only the lower half
                // (component by component: a select between two float4 values is lowered to a scratch array indexed by h)
                const float4 y = e4[(4 * t + qq) * WV_ROWS + c];
                x.x = h == 0 ? y.x : 0.f;
                x.y = h == 0 ? y.y : 0.f;
                x.z = h == 0 ? y.z : 0.f;
                x.w = h == 0 ? y.w : 0.f;
            }
            v[4 * qq + 0] = x.x;
            v[4 * qq + 1] = x.y;
            v[4 * qq + 2] = x.z;
            v[4 * qq + 3] = x.w;
        }
        enc_tap(t, v);
        split8(v, act[0][t], act[1][t], act[2][t]);
        BX_SB();
    }
}

// ---------------------------------------------------------------------------------------------
// MotionNet on the wave's 32 samples: p += flow (modeling/layered_rfrender.py:356,510).  19 slots.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void motion_bx(Ctx& cx, const float* net, const float* cm, float* encw, float (&p)[3], float tv, int flags,
                                          int lane, f32x16 (&big)[4], f32x16 (&small)[4], bf16x8 (&act)[3][16] BXP_PARAM) {
    const MotionLayout L = motion_layout();
    load_c(big, cm + BXM_B, lane);   // (in front of the encoding arithmetic: its LDS round trip is covered)
    encode_motion(encw, lane, p, tv, flags);
    wave_lds_sync();
    enc_to_act<6, WV_ENC_QUADS>(encw, lane, act);
    BXP(BXP_MOTION_ENC);
    pass_act<0, 3, true>(cx, big, small, act);  // motion_net.0: 84 (+4) inputs, 6 K steps
    BXP(BXP_MOTION_PASS);
#pragma unroll 1
    for (int li = 1; li <= 4; ++li) {
        finish_act<0>(big, small, act, cm + BXM_B + 128 * li, lane);
        BXP(BXP_MOTION_FIN);
        pass_act<0, 4, true>(cx, big, small, act);
        BXP(BXP_MOTION_PASS);
    }
    float fl[3];
    head3(big, small, cm + BXM_W_OUT, net + L.b_out, lane, fl);
#pragma unroll
    for (int c3 = 0; c3 < 3; ++c3) p[c3] = p[c3] + fl[c3];
    BXP(BXP_MOTION_FIN);
}

// ---------------------------------------------------------------------------------------------
// SpaceNet on the wave's 32 samples; returns {r, g, b, sigma} (raw) in every lane.  `mid` is called once, in front of the
// last backbone layer's boundary arithmetic (the caller issues the next work item's HBM loads there).
// ---------------------------------------------------------------------------------------------
// `tap` (training): BxStoreTap writes PE(pos) and every layer's post-ReLU output (stage 0 .. 6 = stage1.0 .. stage2.4, 7 = rgb_net.1)
// to the caller's matrices as they pass; BxNoTap compiles to the inference kernel.
template <bool DEEP, class Mid, class Tap>
__device__ __forceinline__ float4 space_bx(Ctx& cx, const float* net, const bool use_time, const float* cs, float* encw, const float (&p)[3],
                                           const float* __restrict__ raybias, int32_t ray, int lane, f32x16 (&big)[4], f32x16 (&small)[4],
                                           bf16x8 (&act)[3][16], Mid mid, const Tap& tap BXP_PARAM) {
    const SpaceLayout L = space_layout(use_time, DEEP);
    const int h = lane >> 5;
    constexpr int ST = Tap::on ? BX_TAP_PARK : 0;   // stores behind every park
    Park pk;
    load_c(big, cs + BXC_B, lane);   // (in front of the encoding arithmetic)
    encode_pos(encw, lane, p);
    wave_lds_sync();
    if constexpr (Tap::on) {   // feature 16 t + 8 h + j of the staged encoding: 32 bytes per lane and K step
        const bool valid = tap_valid(tap, lane);
        // (tap_row's "4 h" is 8 h here: a lane half owns 8 consecutive features of a K step)
        float4* dst = reinterpret_cast<float4*>(tap_row(tap, TAP_PE, 4 * h, lane));
        enc_to_act<4, 16>(encw, lane, act, [&](int t, const float (&v)[8]) {
            if (valid) {
                dst[4 * t] = make_float4(v[0], v[1], v[2], v[3]);
                dst[4 * t + 1] = make_float4(v[4], v[5], v[6], v[7]);
            }
        });
    } else {
        enc_to_act<4, 16>(encw, lane, act);
    }
    BXP(BXP_PE);
    // ---- stage1.0: 63 (+1) -> 256.  Every pass's C operand (its bias) is read block by block inside the boundary pass in
    // front of it, as soon as a block's accumulators have been consumed.
    pass_act<0, 2, true, false, Tap::on ? BX_TAP_PE : 0>(cx, big, small, act);
    BXP(BXP_PASS);
    finish_park(big, small, pk, cs + BXC_B + 128, lane, tap, 0, 0);
    BXP(BXP_PARK);
    pass_act<0, 2, true, false, ST>(cx, big, small, act);
    BXP(BXP_PASS);
    unpark_act(pk, act);                                     // first pass -> K steps 0 .. 7 (in the open: the only layer whose
    BXP(BXP_ACT);                                            // successor's first pass cannot start before it)
    finish_park(big, small, pk, cs + BXC_B + 256, lane, tap, 0, 128);   // second pass -> park: converted under stage1.2's first K steps
    BXP(BXP_PARK);
    // ---- stage1.2 .. stage2.4: six 256-wide layers, two passes each; stage2.0 (li == 4) takes PE(pos) again behind its 256
    // features (modeling/spacenet.py:45-57,136-138): four more K steps per pass, their B operands split on the spot from the
    // staged encoding (the activation planes are full)
    // (The three lambdas below are inlined into the inference kernel by the inliner's own choice; with the tap's code in them it
    // declines, and a real call passes the wave's 400 live registers through memory.  The tap variant forces them AT THE CALL -- an
    // attribute on the lambdas themselves changes the inlining order, and with it the code, of the inference kernel.)
#define BX_INLINED(call)                           \
    do {                                           \
        if constexpr (Tap::on) {                   \
            [[clang::always_inline]] call;         \
        } else {                                   \
            call;                                  \
        }                                          \
    } while (0)
    auto pe_slots = [&]() {
        const float4* e4 = reinterpret_cast<const float4*>(encw);
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            bf16x8 pe[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float v[8];
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const float4 x = e4[(4 * (2 * sl + t) + 2 * h + qq) * WV_ROWS + (lane & 31)];
                    v[4 * qq + 0] = x.x;
                    v[4 * qq + 1] = x.y;
                    v[4 * qq + 2] = x.z;
                    v[4 * qq + 3] = x.w;
                }
                split8(v, pe[t][0], pe[t][1], pe[t][2]);
            }
            slot<false>(cx, big, small, pe[0][0], pe[0][1], pe[0][2], pe[1][0], pe[1][1], pe[1][2]);
        }
    };
    // (stage2.0 is peeled out of the layer loop: inside it, as a conditional block, its extra K steps redefine the
    // accumulators on one of two paths and the register allocator answers with ~200 spills)
    auto layer = [&](int li, auto with_pe) {
        pass_a_unpark<false, ST>(cx, big, small, act, pk);   // K steps 0 .. 7, the previous layer's second pass -> K steps 8 .. 15
        pass_act<8, 4, false>(cx, big, small, act);
        if constexpr (decltype(with_pe)::value) BX_INLINED(pe_slots());
        BXP(BXP_PASS);
        finish_park(big, small, pk, cs + BXC_B + 256 * li + 128, lane, tap, li, 0);
        BXP(BXP_PARK);
        pass_b_unpark<ST>(cx, big, small, act, pk);
        if constexpr (decltype(with_pe)::value) BX_INLINED(pe_slots());
        BXP(BXP_PASS);
    };
    auto layer_end = [&](int li) {   // (+ the next layer's first C operand; behind stage2.4 comes rgb_net.1, which starts from 0)
        finish_park(big, small, pk, li < 6 ? cs + BXC_B + 256 * (li + 1) : nullptr, lane, tap, li, 128);
        BXP(BXP_PARK);
    };
#pragma unroll 1
    for (int li = 1; li <= 3; ++li) {
        BX_INLINED(layer(li, std::false_type{}));
        BX_INLINED(layer_end(li));
    }
    BX_INLINED(layer(4, std::true_type{}));
    BX_INLINED(layer_end(4));
    float sigma = 0.f;
#pragma unroll 1
    for (int li = 5; li <= 6; ++li) {
        BX_INLINED(layer(li, std::false_type{}));
        if (li == 6) {  // sigma = density_net(h) (:139), raw; the next work item's HBM loads go out in front of it
            mid();
            sigma = sigma_head(big, small, pk, cs + BXC_W_SIGMA, net[L.b_sigma], lane);
            BXP(BXP_SIGMA);
        }
        BX_INLINED(layer_end(li));
    }
#undef BX_INLINED
    // ---- rgb_net: relu -> Linear(283|304, 128) -> relu -> Linear(128, 3) (:80-86); the 256 backbone columns here, the
    // bias + direction / time columns = this sample's row of the ray-bias table (mlp_raybias.hip).  The exact-f32 kernels take
    // that row as the C operand; here it is added BEHIND the K loop: it is an order of magnitude larger than the backbone
    // part, and as the C operand it would put every rounding of the a0 b0 chain at its scale (measured: the colour output at
    // 1.3 x the fp32 CPU chain's error instead of 0.5 x).  Its 16 loads go out in front of the pass's last slot (14 of the 16
    // K steps' activation registers are dead by then).
    pass_a_unpark<true, ST>(cx, big, small, act, pk);        // (stage2.4's second pass is converted under its first K steps)
    pass_act<8, 3, false>(cx, big, small, act);
    BXP(BXP_PASS);
    {
        float4 crow[4][4];
        const float* row = raybias + (int64_t)ray * 128 + 4 * h;
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int q = 0; q < 4; ++q) crow[fb][q] = *reinterpret_cast<const float4*>(row + fb * 32 + 8 * q);
        pass_act<14, 1, false>(cx, big, small, act);
        BXP(BXP_PASS);
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                big[fb][4 * q + 0] = (big[fb][4 * q + 0] + small[fb][4 * q + 0]) + crow[fb][q].x;
                big[fb][4 * q + 1] = (big[fb][4 * q + 1] + small[fb][4 * q + 1]) + crow[fb][q].y;
                big[fb][4 * q + 2] = (big[fb][4 * q + 2] + small[fb][4 * q + 2]) + crow[fb][q].z;
                big[fb][4 * q + 3] = (big[fb][4 * q + 3] + small[fb][4 * q + 3]) + crow[fb][q].w;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) small[fb][i] = 0.f;
            BX_SB();
        }
    }
    if constexpr (Tap::on) finish_park<false>(big, small, pk, nullptr, lane, tap, 7, 0);   // relu(rgb_net.1): rgb_net.3's input
    if constexpr (DEEP) {  // deep_rgb (:68-79): two more 128-wide hidden layers
#pragma unroll 1
        for (int i = 0; i < 2; ++i) {
            finish_act<0>(big, small, act, cs + BXC_B_DEEP + 128 * i, lane);
            pass_act<0, 4, true>(cx, big, small, act);
        }
    }
    float rgb[3];
    head3(big, small, cs + BXC_W_RGB2, net + L.b_rgb2, lane, rgb);
    BXP(BXP_RGB_TAIL);
    return make_float4(rgb[0], rgb[1], rgb[2], sigma);
}

// =============================================================================================
// The SpaceNet's backward chain in split bf16 (the twin of csrc/train_wave.hip's train_space_dx_kernel):
// d act_{s-1} = (d act_s * [act_s > 0]) W_s from the heads back to stage1.2 (and on to PE(pos) when the sample points need a
// gradient), on the machinery of the forward kernel above -- a wave owns 32 rows, the masked gradient of a layer lives in the
// activation planes as three bf16 pieces, the TRANSPOSED weights come as a bf16x3 stream through the LDS ring
// (stnerf_pack_dx_bf16x3_device), a 256-wide product runs as two passes of 128 outputs whose results wait in the park.  A layer
// boundary is { big + small, AND with the ReLU mask, store d y_s (the left operand of the weight gradient), park }: the masks are
// the bit planes the forward tap wrote, fetched for the item's 128 rows by LDS-DMA at the start of the item (32 KB) and read
// back eight bytes at a time -- no register is held for them.
//   order of the stream:  rgb_net.1[:, :256] (K = 128: 8 K steps; 2 passes), stage2.4, stage2.2, stage2.0[:, :256] (+ DPOS: its
//   PE columns, a HALF pass: 2 blocks x 16 K steps in 4 slots), stage1.6, stage1.4, stage1.2 (+ DPOS: stage1.0, a half pass).
// d PE(pos) = d y4 W_2.0[:, 256:] + d y0 W_1.0 leaves as two matrices (dpe_skip, dpe): the f32 kernel carries the first through
// four layers in 32 accumulators this kernel does not have; the caller adds them (one elementwise launch).
// =============================================================================================
constexpr int BXD_CONST = 1024;                    // floats: density_net.0's 256 weights | the colour head [3][128] | pad
constexpr int BXD_W_SIGMA = 0, BXD_W_RGB2 = 256;
constexpr int BXD_LDS_MASK = WV_NW * 8 * 1024;     // per wave: 8 stages x 32 rows x 32 bytes
constexpr int BXD_LDS = BX_LDS_RING + BXD_CONST * 4 + BXD_LDS_MASK;
constexpr int BXD_PARK = 16;                       // VMEM stores of one boundary pass

// This lane's two mask words of a boundary pass, from the LDS copy the item's LDS-DMA filled.  As asm (as the A operands): a
// ds_read the compiler sees is preceded by s_waitcnt vmcnt(0) -- it cannot tell the weight ring's LDS-DMA in flight from the
// masks' -- which would drain the ring at every boundary.
__device__ __forceinline__ uint2 read_mask2(const uint32_t* mask2) {
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    u32x2_ m;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(m) : "v"((uint32_t)(uintptr_t)mask2) : "memory");
    return make_uint2(m[0], m[1]);
}
// The heads' weights, for the same reason as asm, issued one group ahead of their use (LDS returns in order: `keep` = reads of the
// NEXT group that may stay in flight).
typedef float bxf32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_issue4(bxf32x4& dst, uint32_t addr, int off /* a constant once the loops are unrolled */) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory");
}
__device__ __forceinline__ void lds_wait4(bxf32x4& a, int keep) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "i"(keep)); }
__device__ __forceinline__ void lds_wait4x3(bxf32x4& a, bxf32x4& b, bxf32x4& c, int keep) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "i"(keep));
}
// v = (big + small [+ gw * w_sigma]) AND mask -> d y (16 bytes per (fb, q), as the forward tap's layout) and the park
template <bool SIGMA>
__device__ __forceinline__ void finish_mask_park(const f32x16 (&big)[4], const f32x16 (&small)[4], Park& pk, const uint32_t* mask2, float* dy,
                                                 bool valid, float gw, const float* wsig, int lane) {
    const uint2 m = read_mask2(mask2);
    float4* dst = reinterpret_cast<float4*>(dy);
    const uint32_t wa = (uint32_t)(uintptr_t)wsig + 16u * (uint32_t)(lane >> 5);   // quad 2 (4 fb + q) + h of the 128 weights
    bxf32x4 wq[2];
    if (SIGMA) lds_issue4(wq[0], wa, 0);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        const uint32_t word = fb < 2 ? m.x : m.y;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[4];
            const int idx = 4 * fb + q;
            if (SIGMA) {
                if (idx < 15) lds_issue4(wq[(idx + 1) & 1], wa, 32 * (idx + 1));
                lds_wait4(wq[idx & 1], idx < 15 ? 1 : 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * q + r;
                float x = big[fb][i] + small[fb][i];
                if (SIGMA) x = fmaf(gw, wq[idx & 1][r], x);
                int mm;     // 0 or -1
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(mm) : "v"(word), "n"((fb & 1) * 16 + i));
                v[r] = __int_as_float(__float_as_int(x) & mm);
                park_put(pk.v[16 * fb + i], v[r]);
            }
            if (valid) dst[fb * 8 + 2 * q] = make_float4(v[0], v[1], v[2], v[3]);
            if (q & 1) BX_SB();
        }
    }
}
// the colour head backwards (rgb_net.3, 128 -> 3, modeling/spacenet.py:84-85): d act7[f] = sum_o d rgb[o] W[o][f], masked -> d y7 and
// K steps 0 .. 7 of the planes
__device__ __forceinline__ void dx_head_boundary(const float4 g, const float* wrgb2, const uint32_t* mask2, float* dy, bool valid,
                                                 bf16x8 (&act)[3][16], int lane) {
    const uint2 m = read_mask2(mask2);
    float4* dst = reinterpret_cast<float4*>(dy);
    const uint32_t wa = (uint32_t)(uintptr_t)wrgb2 + 16u * (uint32_t)(lane >> 5);   // quad 2 (4 fb + q) + h of each of the three rows
    bxf32x4 wq[2][3];
    lds_issue4(wq[0][0], wa, 0);
    lds_issue4(wq[0][1], wa, 512);
    lds_issue4(wq[0][2], wa, 1024);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        const uint32_t word = fb < 2 ? m.x : m.y;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v[8];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int q = 2 * t + qq;
                const int idx = 4 * fb + q;
                if (idx < 15) {
                    lds_issue4(wq[(idx + 1) & 1][0], wa, 32 * (idx + 1));
                    lds_issue4(wq[(idx + 1) & 1][1], wa, 512 + 32 * (idx + 1));
                    lds_issue4(wq[(idx + 1) & 1][2], wa, 1024 + 32 * (idx + 1));
                }
                lds_wait4x3(wq[idx & 1][0], wq[idx & 1][1], wq[idx & 1][2], idx < 15 ? 3 : 0);
                const bxf32x4 w0 = wq[idx & 1][0], w1 = wq[idx & 1][1], w2 = wq[idx & 1][2];
                const float x[4] = {g.x * w0[0] + g.y * w1[0] + g.z * w2[0], g.x * w0[1] + g.y * w1[1] + g.z * w2[1],
                                    g.x * w0[2] + g.y * w1[2] + g.z * w2[2], g.x * w0[3] + g.y * w1[3] + g.z * w2[3]};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int mm;
                    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(mm) : "v"(word), "n"((fb & 1) * 16 + 4 * q + r));
                    v[4 * qq + r] = __int_as_float(__float_as_int(x[r]) & mm);
                }
                if (valid) dst[fb * 8 + 2 * q] = make_float4(v[4 * qq], v[4 * qq + 1], v[4 * qq + 2], v[4 * qq + 3]);
            }
            split8(v, act[0][2 * fb + t], act[1][2 * fb + t], act[2][2 * fb + t]);
            BX_SB();
        }
    }
}
// one ring slot of a HALF pass (64 outputs = blocks 0, 1): K steps K0 .. K0 + 3
template <int K0, bool FIRST, int ST = 0>
__device__ __forceinline__ void slot_half(Ctx& cx, f32x16 (&big)[4], f32x16 (&small)[4], const bf16x8 (&act)[3][16]) {
    unit<0, FIRST, FIRST>(cx, big[0], small[0], act[0][K0], act[1][K0], act[2][K0]);
    unit<1, FIRST, FIRST>(cx, big[1], small[1], act[0][K0], act[1][K0], act[2][K0]);
    unit<2, false>(cx, big[0], small[0], act[0][K0 + 1], act[1][K0 + 1], act[2][K0 + 1]);
    unit<3, false>(cx, big[1], small[1], act[0][K0 + 1], act[1][K0 + 1], act[2][K0 + 1]);
    unit<4, false>(cx, big[0], small[0], act[0][K0 + 2], act[1][K0 + 2], act[2][K0 + 2]);
    unit<5, false>(cx, big[1], small[1], act[0][K0 + 2], act[1][K0 + 2], act[2][K0 + 2]);
    Dma d;
    slot_turn<ST>(cx, d);
    unit<6, false, false, 0>(cx, big[0], small[0], act[0][K0 + 3], act[1][K0 + 3], act[2][K0 + 3], &d);
    unit<7, false, false, 3>(cx, big[1], small[1], act[0][K0 + 3], act[1][K0 + 3], act[2][K0 + 3], &d);
    slot_done(cx);
}
// d y W[:, PE columns]: 16 K steps of the planes into blocks 0, 1, written out unmasked (PE(pos) has no ReLU)
template <int ST>
__device__ __forceinline__ void pe_half_pass(Ctx& cx, f32x16 (&big)[4], f32x16 (&small)[4], const bf16x8 (&act)[3][16], float* dpe_row, bool valid) {
    slot_half<0, true, ST>(cx, big, small, act);
    slot_half<4, false, ST>(cx, big, small, act);
    slot_half<8, false>(cx, big, small, act);
    slot_half<12, false>(cx, big, small, act);
    float4* dst = reinterpret_cast<float4*>(dpe_row);
#pragma unroll
    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (valid)
                dst[fb * 8 + 2 * q] = make_float4(big[fb][4 * q] + small[fb][4 * q], big[fb][4 * q + 1] + small[fb][4 * q + 1],
                                                  big[fb][4 * q + 2] + small[fb][4 * q + 2], big[fb][4 * q + 3] + small[fb][4 * q + 3]);
}

struct DxBxArgs {
    const char* blob;         // stnerf_pack_dx_bf16x3_device: [consts][stream]
    const float* d_raw;       // [rows][4]: dLoss / d {r, g, b, sigma}
    const uint32_t* bits;     // the forward tap's ReLU bit planes: [8][.. bits_stride ..], rows x 8 words per stage
    int64_t bits_stride;
    float* dy[8];             // dy[s]: dLoss / d (pre-activation of that layer): 256 wide, s = 7: 128
    int32_t ld_dy[8];
    float* dpe;               // DPOS: d y0 W_1.0 (64 wide) ...
    float* dpe_skip;          // ... and d y4 W_2.0[:, 256:]
    int32_t ld_dpe, ld_dpe_skip;
    int64_t rows;
    int32_t n_slots;          // of the stream (DPOS or not: two different blobs)
};

template <bool DPOS>
__device__ __forceinline__ void space_dx_bx(Ctx& cx, const DxBxArgs& a, const float* cst, const uint32_t* mk, const float4 g, uint32_t row,
                                            bool valid, int lane, f32x16 (&big)[4], f32x16 (&small)[4], bf16x8 (&act)[3][16]) {
    Park pk;
    // this lane's 16 bytes at column col0 of its row of d y_s (built where it is used: no register is held for it)
    auto dyrow = [&](int s_, int col0) {
        int l2 = lane;
        asm volatile("" : "+v"(l2));
        return a.dy[s_] + (size_t)row * (size_t)a.ld_dy[s_] + (uint32_t)(col0 + 4 * (l2 >> 5));
    };
    auto perow = [&](float* base, int32_t ld) {
        int l2 = lane;
        asm volatile("" : "+v"(l2));
        return base + (size_t)row * (size_t)ld + (uint32_t)(4 * (l2 >> 5));
    };
    dx_head_boundary(g, cst + BXD_W_RGB2, mk + 7 * 256, dyrow(7, 0), valid, act, lane);
    // ---- d act6 = d y7 W_rgb1[:, :256] + d sigma w_sigma (the density head's rank-1 term joins behind the K loop)
    pass_act<0, 4, true, true, BXD_PARK>(cx, big, small, act);
    finish_mask_park<true>(big, small, pk, mk + 6 * 256, dyrow(6, 0), valid, g.w, cst + BXD_W_SIGMA, lane);
    pass_act<0, 4, true, true, BXD_PARK>(cx, big, small, act);
    unpark_act(pk, act);                     // (in the open, as stage1.0's first pass in the forward kernel)
    finish_mask_park<true>(big, small, pk, mk + 6 * 256 + 2, dyrow(6, 128), valid, g.w, cst + BXD_W_SIGMA + 128, lane);
    // ---- six 256 x 256 products: d y_s = mask_s (d y_{s+1} W_{s+1}); s = 3 (stage2.0) with its PE columns between the passes
    auto layer = [&](int s_, auto with_pe) {
        pass_a_unpark<true, BXD_PARK>(cx, big, small, act, pk);
        pass_act<8, 4, false>(cx, big, small, act);
        finish_mask_park<false>(big, small, pk, mk + s_ * 256, dyrow(s_, 0), valid, 0.f, cst, lane);
        if constexpr (decltype(with_pe)::value) {
            pe_half_pass<BXD_PARK>(cx, big, small, act, perow(a.dpe_skip, a.ld_dpe_skip), valid);
            pass_b_unpark<8, true>(cx, big, small, act, pk);
        } else {
            pass_b_unpark<BXD_PARK, true>(cx, big, small, act, pk);
        }
        finish_mask_park<false>(big, small, pk, mk + s_ * 256 + 2, dyrow(s_, 128), valid, 0.f, cst, lane);
    };
#pragma unroll 1
    for (int s_ = 5; s_ >= 4; --s_) {
        [[clang::always_inline]] layer(s_, std::false_type{});
    }
    [[clang::always_inline]] layer(3, std::integral_constant<bool, DPOS>{});
#pragma unroll 1
    for (int s_ = 2; s_ >= 0; --s_) {
        [[clang::always_inline]] layer(s_, std::false_type{});
    }
    if constexpr (DPOS) {   // d y0 W_1.0: the upper half of d y0 is still parked
        unpark_act<8>(pk, act);
        pe_half_pass<BXD_PARK>(cx, big, small, act, perow(a.dpe, a.ld_dpe), valid);
    }
}

template <bool DPOS>
__global__ __launch_bounds__(WV_THREADS, 1) void train_space_dx_bx_kernel(DxBxArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_bx[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* ring = smem_bx;
    float* cst = reinterpret_cast<float*>(smem_bx + BX_LDS_RING);
    uint32_t* masks = reinterpret_cast<uint32_t*>(smem_bx + BX_LDS_RING + BXD_CONST * 4) + wave * 2048;
    const int64_t items = (a.rows + WV_ITEM - 1) / WV_ITEM;
    int64_t item = blockIdx.x;
    if (item >= items) return;  // (uniform)
    // the heads' weights: once per workgroup (4 chunks of 1 KB over the four waves)
    __builtin_amdgcn_global_load_lds(a.blob + wave * BX_CHUNK + lane * 16,
                                     (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) char*)(cst) + wave * BX_CHUNK), 16, 0, 0);
    const char* stream = a.blob + BXD_CONST * 4;
    Ctx cx;
    cx.ring = ring;
    cx.wave = wave;
    cx.lane = lane;
    cx.gi = 0;
    cx.gc = 0;
    cx.rcur = (uint32_t)(uintptr_t)ring + (uint32_t)lane * 16u;
    cx.rnext = cx.rcur + BX_SLOT;
    cx.seg[0] = Seg{stream, (uint32_t)a.n_slots};
    cx.seg[1] = Seg{nullptr, 0u};
    cx.seg[2] = Seg{stream, item + gridDim.x < items ? (uint32_t)a.n_slots : 0u};
    cx.seg[3] = Seg{nullptr, 0u};
    cx.idle = stream;
    dma_issue(cx);
    dma_issue(cx);
    dma_issue(cx);
    BX_VMCNT(12);
    __builtin_amdgcn_s_barrier();
    a_read<0>(cx.A[0].p[0], cx.rcur);
    a_read<BX_CHUNK>(cx.A[0].p[1], cx.rcur);
    a_read<2 * BX_CHUNK>(cx.A[0].p[2], cx.rcur);
    a_read<BX_UNIT>(cx.A[1].p[0], cx.rcur);
    a_read<BX_UNIT + BX_CHUNK>(cx.A[1].p[1], cx.rcur);
    a_read<BX_UNIT + 2 * BX_CHUNK>(cx.A[1].p[2], cx.rcur);
    f32x16 big[4], small[4];
    bf16x8 act[3][16];
    for (; item < items; item += gridDim.x) {
        const int64_t row64 = item * WV_ITEM + wave * WV_ROWS + (lane & 31);
#ifdef STNERF_DEV_TAP_ALWAYS_VALID
        const bool valid = true;
#else
        const bool valid = row64 < a.rows;
#endif
        const uint32_t row = (uint32_t)(valid ? row64 : a.rows - 1);
        cx.st_on = item * WV_ITEM + wave * WV_ROWS < a.rows;     // (some row of the wave's 32 inside the launch: its stores are issued)
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) g = *reinterpret_cast<const float4*>(a.d_raw + (size_t)row * 4);
        // ---- the item's ReLU masks: per stage the wave's 32 rows are 1 KB contiguous; lane i fetches half-row i & 1 of row i >> 1
        {
            int64_t mrow = item * WV_ITEM + wave * WV_ROWS + (lane >> 1);
            mrow = mrow < a.rows ? mrow : a.rows - 1;
            const uint32_t* src = a.bits + (size_t)mrow * 8u + 4u * (uint32_t)(lane & 1);
            auto dst = (__attribute__((address_space(3))) char*)(masks);
#pragma unroll
            for (int s_ = 0; s_ < 8; ++s_)
                __builtin_amdgcn_global_load_lds(src + (size_t)s_ * (size_t)a.bits_stride, (__attribute__((address_space(3))) void*)(dst + s_ * 1024), 16, 0, 0);
        }
        BX_VMCNT(0);
        int ln = lane;
        asm volatile("" : "+v"(ln));
        // this lane's words of (stage s, pass hf): mk[s * 256 + 2 hf + {0, 1}]
        const uint32_t* mk = masks + (ln & 31) * 8 + 4 * (ln >> 5);
        space_dx_bx<DPOS>(cx, a, cst, mk, g, row, valid, ln, big, small, act);
        // the stream: the next item's moves up, the one after it joins
        cx.seg[0] = cx.seg[2];
        cx.seg[2] = Seg{stream, item + 2 * (int64_t)gridDim.x < items ? (uint32_t)a.n_slots : 0u};
    }
    BX_VMCNT(0);  // (no LDS-DMA may outlive the workgroup)
}

int launch_bf16x3_dx(const DxBxArgs& a, bool dpos, int cus, hipStream_t stream) {
    const int64_t items = (a.rows + WV_ITEM - 1) / WV_ITEM;
    const int grid = (int)(items < cus ? items : cus);
    const void* kfn = dpos ? reinterpret_cast<const void*>(train_space_dx_bx_kernel<true>) : reinterpret_cast<const void*>(train_space_dx_bx_kernel<false>);
    if (const int rc = reserve_dynamic_lds(kfn, BXD_LDS, "train_spacenet_dx (bf16x3)")) return rc;
    if (dpos)
        hipLaunchKernelGGL(train_space_dx_bx_kernel<true>, dim3(grid), dim3(WV_THREADS), BXD_LDS, stream, a);
    else
        hipLaunchKernelGGL(train_space_dx_bx_kernel<false>, dim3(grid), dim3(WV_THREADS), BXD_LDS, stream, a);
    STNERF_CHECK_LAUNCH("train_spacenet_dx (bf16x3)");
    return STNERF_OK;
}

// does this wave issue every store of a boundary?  (some row of its 32 inside the launch, mask planes wanted)
__device__ __forceinline__ bool tap_wave_stores(const BxStoreTap& tap, const StoreTapArgs& t) {
    return (uint32_t)(tap.wave * WV_ROWS) < tap.nrows && t.bits != nullptr;
}
__device__ __forceinline__ BxNoTap make_bx_tap(const NoTapArgs&, uint32_t, int64_t, int) { return BxNoTap(); }
__device__ __forceinline__ BxStoreTap make_bx_tap(const StoreTapArgs& t, uint32_t item, int64_t rows, int wave) {
    // (training launches one network: items of queue slot 0 are rows 128 item ..)
    const int64_t left = rows - (int64_t)item * WV_ITEM;
    return BxStoreTap{&t, item * (uint32_t)WV_ITEM, (uint32_t)(left < WV_ITEM ? left : WV_ITEM), wave};
}

// The tap variant (training: ONE SpaceNet on every ray, no MotionNet, no ray list) gives the tap's stores the registers they need
// by not carrying per-row values through the item: the next item's (ray, sample) is located where it is fetched, not at the
// top of the item; this item's output offset is recomputed from its ray at the end; the MotionNet path is compiled out.
template <bool DEEP, class TapArgs>
__global__ __launch_bounds__(WV_THREADS, 1) void mlp_bf16x3_stage_kernel(StageArgs a, TapArgs targs) {
    constexpr bool TAP = !std::is_same<TapArgs, NoTapArgs>::value;
    extern __shared__ __attribute__((aligned(16))) char smem_bx[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* ring = smem_bx;
    float* encw = reinterpret_cast<float*>(smem_bx + BX_LDS_RING) + wave * WV_ENC_FLOATS;
    float* cs = reinterpret_cast<float*>(smem_bx + BX_LDS_RING + BX_LDS_ENC);
    float* cm = cs + BX_CONST_SPACE;
    uint32_t* qslot = reinterpret_cast<uint32_t*>(smem_bx + BX_LDS_RING + BX_LDS_ENC + BX_LDS_CONST);
    int64_t* lrows = reinterpret_cast<int64_t*>(qslot + 4);
    // ---- the queue (as in mlp_wave.hip): items (128 rows) of layer slot j are [pre[j], pre[j+1]).  The prefix table lives in
    // LDS, not in 17 SGPRs: this kernel's scalar registers are short (the stream state, the kernel arguments), and what does
    // not fit is kept in VGPR lanes -- of which it has none to spare either.
    uint32_t* lpre = reinterpret_cast<uint32_t*>(lrows + STNERF_MAX_LAYERS);
    uint32_t total = 0;
#pragma unroll 1
    for (int j = 0; j < a.n_layers; ++j) {
        const int64_t rows = layer_rows(a.layer[j], a.n_rays, a.ns);
        if (tid == 0) {
            lrows[j] = rows;
            lpre[j] = total;
        }
        total += (uint32_t)((rows + WV_ITEM - 1) / WV_ITEM);
    }
    // (layer slot, first item of that slot) of an item: wave-uniform
    auto locate = [&](uint32_t item, int& slot, uint32_t& base) {
        slot = 0;
        base = 0;
#pragma unroll 1
        for (int j = 1; j < a.n_layers; ++j) {
            const uint32_t pj = (uint32_t)__builtin_amdgcn_readfirstlane((int)lpre[j]);
            if (item >= pj) {
                slot = j;
                base = pj;
            }
        }
    };
    auto slot_of = [&](uint32_t item) {
        int s_;
        uint32_t b_;
        locate(item, s_, b_);
        return s_;
    };
    auto base_of = [&](uint32_t item) {
        int s_;
        uint32_t b_;
        locate(item, s_, b_);
        return b_;
    };
    auto row_of = [&](uint32_t item, RowRef& rr) {
        rr = RowRef{0, 0, false};
        if (item >= total) return;
        const int s = slot_of(item);
        const int64_t rows = lrows[s];
        const int64_t row = (int64_t)(item - base_of(item)) * WV_ITEM + wave * WV_ROWS + (lane & 31);
        rr.valid = row < rows;
        if (rr.valid) {
            int64_t rslot;
            if (rows <= 0x7fffffffll) {
                const uint32_t q = (uint32_t)row / (uint32_t)a.ns;
                rslot = q;
                rr.k = (int)((uint32_t)row - q * (uint32_t)a.ns);
            } else {
                rslot = row / a.ns;
                rr.k = (int)(row - rslot * a.ns);
            }
            const int32_t* rl = a.layer[s].ray_list;
            rr.ray = rl ? (int64_t)rl[rslot] : rslot;
        }
    };
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
    // the weight streams of an item's networks
    const int64_t sp_kind_stream[2] = {bx_layout(DEEP ? STNERF_NET_SPACE_DEEP : STNERF_NET_SPACE).stream_off,
                                       bx_layout(DEEP ? STNERF_NET_SPACE_TIME_DEEP : STNERF_NET_SPACE_TIME).stream_off};
    const int64_t mo_stream = bx_layout(STNERF_NET_MOTION).stream_off;
    auto segs_of = [&](uint32_t item, Seg& m, Seg& s) {
        m = Seg{nullptr, 0u};
        s = Seg{nullptr, 0u};
        if (item >= total) return;
        const StageLayer& ly = a.layer[slot_of(item)];
        s.p = reinterpret_cast<const char*>(ly.space) + sp_kind_stream[ly.use_time ? 1 : 0];
        s.left = (uint32_t)bx_space_slots(DEEP);
        if (ly.motion) {
            m.p = reinterpret_cast<const char*>(ly.motion) + mo_stream;
            m.left = (uint32_t)bx_motion_slots();
        }
    };

    // ---- prime the pipeline: two items popped, the first one's inputs loaded, three slots of its stream in flight
    if (tid == 0) {
        qslot[0] = atomicAdd(a.queue, 1u);
        qslot[1] = atomicAdd(a.queue, 1u);
    }
    __syncthreads();
    uint32_t it0 = __builtin_amdgcn_readfirstlane(qslot[0]);
    uint32_t it1 = __builtin_amdgcn_readfirstlane(qslot[1]);
    __syncthreads();
    if (it0 >= total) return;  // (uniform)
    WaveInputs cur, nxt;
    {
        RowRef rr;
        row_of(it0, rr);
        fetch(it0, rr, cur);
    }
    Ctx cx;
    cx.ring = ring;
    cx.wave = wave;
    cx.lane = lane;
    cx.gi = 0;
    cx.gc = 0;
    cx.st_on = false;
    cx.rcur = (uint32_t)(uintptr_t)ring + (uint32_t)lane * 16u;   // LDS byte address of the ring + lane * 16
    cx.rnext = cx.rcur + BX_SLOT;
    segs_of(it0, cx.seg[0], cx.seg[1]);
    segs_of(it1, cx.seg[2], cx.seg[3]);
    cx.idle = cx.seg[1].p;
    dma_issue(cx);
    dma_issue(cx);
    dma_issue(cx);
    BX_VMCNT(12);
    __builtin_amdgcn_s_barrier();
    a_read<0>(cx.A[0].p[0], cx.rcur);
    a_read<BX_CHUNK>(cx.A[0].p[1], cx.rcur);
    a_read<2 * BX_CHUNK>(cx.A[0].p[2], cx.rcur);
    a_read<BX_UNIT>(cx.A[1].p[0], cx.rcur);
    a_read<BX_UNIT + BX_CHUNK>(cx.A[1].p[1], cx.rcur);
    a_read<BX_UNIT + 2 * BX_CHUNK>(cx.A[1].p[2], cx.rcur);
    int par = 0;
    f32x16 big[4], small[4];
    bf16x8 act[3][16];
#ifdef STNERF_BX_PROF
    BxProf bp;
    for (int i = 0; i < 16; ++i) bp.acc[i] = 0;
    bp.t = clock64();
#endif
    while (it0 < total) {
        // the item after next (consumed at the end of this one) and the ray index of the next item's sample
        uint32_t pending = 0;
        if (tid == 0) pending = atomicAdd(a.queue, 1u);
        RowRef rr_next;
        uint32_t next_ray = 0;   // (tap variant: all that is kept of the next item's row until its inputs are fetched)
        if constexpr (TAP) {
            const int64_t row = (int64_t)it1 * WV_ITEM + wave * WV_ROWS + (lane & 31);
            if (row < a.n_rays * a.ns) next_ray = (uint32_t)row / (uint32_t)a.ns;   // (rows <= 0x7fffff00: stnerf_train_spacenet_fwd)
        } else {
            row_of(it1, rr_next);
        }
        const StageLayer& ly = a.layer[slot_of(it0)];
        // ---- this item's bias vectors / head weights: blob consts -> LDS (12 + 4 chunks of 1 KB over the four waves).  The
        // previous item's last reads of the region are behind the barrier that ended it.
        {
            const char* sc = reinterpret_cast<const char*>(ly.space) + (sp_kind_stream[ly.use_time ? 1 : 0] - BX_CONST_SPACE * 4) + lane * 16;
            auto d = (__attribute__((address_space(3))) char*)(cs);
#pragma unroll
            for (int i = 0; i < 3; ++i)
                __builtin_amdgcn_global_load_lds(sc + (wave + 4 * i) * BX_CHUNK, (__attribute__((address_space(3))) void*)(d + (wave + 4 * i) * BX_CHUNK), 16, 0, 0);
            if (ly.motion) {
                const char* mc = reinterpret_cast<const char*>(ly.motion) + (mo_stream - BX_CONST_MOTION * 4) + lane * 16;
                auto dm = (__attribute__((address_space(3))) char*)(cm);
                __builtin_amdgcn_global_load_lds(mc + wave * BX_CHUNK, (__attribute__((address_space(3))) void*)(dm + wave * BX_CHUNK), 16, 0, 0);
            }
        }
        float p[3];
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) p[c3] = cur.p[c3];
        BX_VMCNT(0);
        __builtin_amdgcn_s_barrier();
        BXP(BXP_TOP);
        // (the lane index the networks see is opaque per item: hoisted out of the item loop, the per-lane LDS addresses and
        // constants derived from it -- ~40 registers of the encodings alone -- do not fit beside the loop's live values and
        // come back from scratch, each reload behind a vmcnt(0) that also drains the weight ring's DMA queue)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        if constexpr (!TAP) {
            if (ly.motion) motion_bx(cx, ly.motion, cm, encw, p, cur.tv, ly.motion_flags, ln, big, small, act BXP_ARG);
        }
        const auto tap = make_bx_tap(targs, it0, a.n_rays * a.ns, wave);
        if constexpr (TAP) cx.st_on = tap_wave_stores(tap, targs);
        float4 o = space_bx<DEEP>(cx, ly.space, ly.use_time != 0, cs, encw, p, ly.raybias, cur.ray, ln, big, small, act,
                                  [&]() {
                                      if constexpr (TAP) {
                                          const int64_t row = (int64_t)it1 * WV_ITEM + wave * WV_ROWS + (lane & 31);
                                          rr_next.valid = row < a.n_rays * a.ns;
                                          rr_next.ray = next_ray;
                                          rr_next.k = (int)((uint32_t)row - next_ray * (uint32_t)a.ns);
                                      }
                                      fetch(it1, rr_next, nxt);
                                  }, tap BXP_ARG);
        if constexpr (TAP) {   // (queue slot 0, every ray: row = 128 item + ..., sample k = row - ray * ns)
            const int64_t row = (int64_t)it0 * WV_ITEM + wave * WV_ROWS + (lane & 31);
            cur.raw_off = (int64_t)cur.ray * a.raw_ray_stride + 4 * (row - (int64_t)cur.ray * a.ns);
        }
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
        // the stream: the next item's networks move up, the one after it joins
        cx.seg[0] = cx.seg[2];
        cx.seg[1] = cx.seg[3];
        segs_of(it1, cx.seg[2], cx.seg[3]);
        BXP(BXP_END);
#ifdef STNERF_BX_PROF
        bp.acc[BXP_ITEMS] += 1;
#endif
    }
    BX_VMCNT(0);  // (no LDS-DMA may outlive the workgroup)
#ifdef STNERF_BX_PROF
    if (lane == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&g_bxphase[i], bp.acc[i]);
#endif
}

int launch_bf16x3_stage(const StageArgs& a, bool deep_rgb, int cus, hipStream_t stream) {
    const int64_t max_items = ((a.n_rays * a.ns + WV_ITEM - 1) / WV_ITEM) * a.n_layers;
    const int grid = (int)(max_items < cus ? max_items : cus);  // one persistent workgroup per CU
    const void* kfn = deep_rgb ? reinterpret_cast<const void*>(mlp_bf16x3_stage_kernel<true, NoTapArgs>)
                               : reinterpret_cast<const void*>(mlp_bf16x3_stage_kernel<false, NoTapArgs>);
    if (const int rc = reserve_dynamic_lds(kfn, BX_LDS, "mlp_stage (bf16x3)")) return rc;
    if (deep_rgb)
        hipLaunchKernelGGL((mlp_bf16x3_stage_kernel<true, NoTapArgs>), dim3(grid), dim3(WV_THREADS), BX_LDS, stream, a, NoTapArgs());
    else
        hipLaunchKernelGGL((mlp_bf16x3_stage_kernel<false, NoTapArgs>), dim3(grid), dim3(WV_THREADS), BX_LDS, stream, a, NoTapArgs());
    STNERF_CHECK_LAUNCH("mlp_stage (bf16x3)");
    return STNERF_OK;
}

// One SpaceNet (queue slot 0 of `a`, every ray, no MotionNet, not deep_rgb) with every layer's input written out: see StoreTapArgs.
int launch_bf16x3_stage_store(const StageArgs& a, const StoreTapArgs& t, int cus, hipStream_t stream) {
    const int64_t max_items = (a.n_rays * a.ns + WV_ITEM - 1) / WV_ITEM;
    const int grid = (int)(max_items < cus ? max_items : cus);
    if (const int rc = reserve_dynamic_lds(reinterpret_cast<const void*>(mlp_bf16x3_stage_kernel<false, StoreTapArgs>), BX_LDS,
                                           "train_space_fwd (bf16x3)"))
        return rc;
    hipLaunchKernelGGL((mlp_bf16x3_stage_kernel<false, StoreTapArgs>), dim3(grid), dim3(WV_THREADS), BX_LDS, stream, a, t);
    STNERF_CHECK_LAUNCH("train_space_fwd (bf16x3)");
    return STNERF_OK;
}

}  // namespace stnerf

#ifdef STNERF_BX_PROF
extern "C" int stnerf_debug_bx_phases(unsigned long long* host16, int reset) {
    if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(stnerf::g_bxphase), sizeof(unsigned long long) * 16) != hipSuccess) return STNERF_ELAUNCH;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(stnerf::g_bxphase), z, sizeof(z)) != hipSuccess) return STNERF_ELAUNCH;
    }
    return STNERF_OK;
}
#endif

// ---------------------------------------------------------------------------------------------
// Host: the packer
// ---------------------------------------------------------------------------------------------
using namespace stnerf;

namespace {
inline uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);  // inf / nan: as is
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
inline float bf16_f32(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline void split3(float w, uint16_t (&p)[3]) {
    p[0] = bf16_rne(w);
    const float r1 = w - bf16_f32(p[0]);
    p[1] = bf16_rne(r1);
    const float r2 = r1 - bf16_f32(p[1]);
    p[2] = bf16_rne(r2);
}

// One pass = 128 output rows n0 .. n0 + 127 of W[out][in] (row-major, `in` columns), `ksteps` K steps; col(t, h, j) = input
// column of B position (t, h, j) or -1 (zero).  Appends ksteps * 4 units of 3 KB at dst; returns the bytes written.
template <class ColFn>
int64_t emit_pass(uint16_t* dst, const float* W, int in, int n0, int ksteps, ColFn col) {
    uint16_t* d = dst;
    for (int t = 0; t < ksteps; ++t)
        for (int fb = 0; fb < 4; ++fb) {
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int h = lane >> 5, c = lane & 31;
                    const int k = col(t, h, j);
                    uint16_t p[3] = {0, 0, 0};
                    if (k >= 0) split3(W[(int64_t)(n0 + 32 * fb + c) * in + k], p);
                    for (int pc = 0; pc < 3; ++pc) d[pc * (BX_CHUNK / 2) + lane * 8 + j] = p[pc];
                }
            d += BX_UNIT / 2;
        }
    return (int64_t)(d - dst) * 2;
}
}  // namespace

extern "C" int64_t stnerf_packed_bytes_bf16x3(int kind) {
    if (!STNERF_NET_IS_SPACE(kind) && kind != STNERF_NET_MOTION) {
        set_error("packed_bytes_bf16x3: unknown net kind %d", kind);
        return STNERF_EINVAL;
    }
    return bx_layout(kind).total_bytes;
}

extern "C" int stnerf_pack_net_bf16x3(int kind, const float* const* W, const float* const* B, int n_tensors, void* dst_host,
                                      int64_t dst_bytes) {
    STNERF_REQUIRE(W && B && dst_host, "pack_net_bf16x3: null pointer");
    STNERF_REQUIRE(STNERF_NET_IS_SPACE(kind) || kind == STNERF_NET_MOTION, "pack_net_bf16x3: unknown net kind %d", kind);
    const BxLayout X = bx_layout(kind);
    STNERF_REQUIRE(dst_bytes >= X.total_bytes, "pack_net_bf16x3: dst too small (%lld < %lld)", (long long)dst_bytes, (long long)X.total_bytes);
    memset(dst_host, 0, (size_t)X.total_bytes);
    // ---- the f32 section = the exact-f32 blob (also validates the tensor count)
    if (const int rc = stnerf_pack_net(kind, W, B, n_tensors, dst_host, X.f32_floats * 4)) return rc;
    // The split x = x0 + x1 + x2 is exact for finite values up to bf16's largest finite number, 0x7f7f = 3.3895e38; above it (fp32's
    // last 0.4 %) the leading piece rounds to inf and x0 + x1 is inf - inf; an inf or NaN weight would likewise turn every output
    // it touches into NaN where ATen propagates the inf.  Refused here rather than silently different: the f32 section holds
    // every weight and bias of the network.
    {
        const float* f = static_cast<const float*>(dst_host);
        for (int64_t i = 0; i < X.f32_floats; ++i)
            STNERF_REQUIRE(fabsf(f[i]) <= 3.3895313892515355e38f, "pack_net_bf16x3: a weight or bias is not finite or exceeds bf16's range "
                           "(|w| <= 3.3895e38): %g -- use the exact-f32 packing (stnerf_pack_net) for such a network", (double)f[i]);
    }
    char* base = static_cast<char*>(dst_host);
    float* cst = reinterpret_cast<float*>(base + X.consts_off);
    uint16_t* st = reinterpret_cast<uint16_t*>(base + X.stream_off);
    int64_t off = 0;  // bytes into the stream
    auto hidden = [](int t, int h, int j) { return bx_kmap_hidden(t, h, j); };
    if (STNERF_NET_IS_SPACE(kind)) {
        const bool deep = STNERF_NET_IS_DEEP(kind);
        const int nt = deep ? 12 : 10;
        const int in_f[7] = {63, 256, 256, 256, 319, 256, 256};
        for (int i = 0; i < 7; ++i) memcpy(cst + BXC_B + 256 * i, B[i], 256 * sizeof(float));
        for (int i = 0; i < 2 && deep; ++i) memcpy(cst + BXC_B_DEEP + 128 * i, B[9 + i], 128 * sizeof(float));
        memcpy(cst + BXC_W_SIGMA, W[7], 256 * sizeof(float));
        memcpy(cst + BXC_W_RGB2, W[nt - 1], 3 * 128 * sizeof(float));
        for (int i = 0; i < 7; ++i)
            for (int half = 0; half < 2; ++half) {
                if (i == 0) {
                    off += emit_pass(st + off / 2, W[0], 63, 128 * half, 4, [](int t, int h, int j) {
                        const int k = bx_kmap_enc(t, h, j);
                        return k < 63 ? k : -1;
                    });
                } else if (i == 4) {  // stage2.0: the 256 features, then PE(pos)
                    off += emit_pass(st + off / 2, W[4], 319, 128 * half, 20, [](int t, int h, int j) {
                        if (t < 16) return bx_kmap_hidden(t, h, j);
                        const int k = bx_kmap_enc(t - 16, h, j);
                        return k < 63 ? 256 + k : -1;
                    });
                } else {
                    off += emit_pass(st + off / 2, W[i], in_f[i], 128 * half, 16, hidden);
                }
            }
        // rgb_net.1: the 256 backbone columns (direction / time columns: mlp_raybias.hip)
        off += emit_pass(st + off / 2, W[8], 256 + 27 + (STNERF_NET_USES_TIME(kind) ? 21 : 0), 0, 16, hidden);
        for (int i = 0; i < 2 && deep; ++i) off += emit_pass(st + off / 2, W[9 + i], 128, 0, 8, hidden);
    } else {
        for (int i = 0; i < 5; ++i) memcpy(cst + BXM_B + 128 * i, B[i], 128 * sizeof(float));
        memcpy(cst + BXM_W_OUT, W[5], 3 * 128 * sizeof(float));
        off += emit_pass(st + off / 2, W[0], 84, 0, 6, [](int t, int h, int j) {
            const int k = bx_kmap_enc(t, h, j);
            return k < 84 ? k : -1;
        });
        for (int i = 1; i < 5; ++i) off += emit_pass(st + off / 2, W[i], 128, 0, 8, hidden);
    }
    STNERF_REQUIRE(off == (int64_t)X.n_slots * BX_SLOT, "pack_net_bf16x3: internal: stream of %lld B, expected %lld", (long long)off,
                   (long long)X.n_slots * BX_SLOT);
    return STNERF_OK;
}

// ---------------------------------------------------------------------------------------------
// The same blob from tensors in DEVICE memory (training: the weights change with every optimizer step): the f32 section by
// stnerf_pack_net_device, consts and stream by one kernel -- same split (integer round-to-nearest-even), same order, bit for bit
// (tests/test_gpu_ops.py).  No finiteness check here (no host round trip): an inf / NaN / > 3.39e38 weight gives NaN pieces and NaN
// outputs, which a training loop notices; stnerf_pack_net_bf16x3 is the one that refuses.
// ---------------------------------------------------------------------------------------------
namespace stnerf {
struct BxPackPass {
    const float* src;
    int64_t dst;        // stream passes: byte offset of the pass in the blob; copies: float offset in the blob
    int32_t in, n0, ksteps;
    int32_t mode;       // 0 hidden, 1 staged encoding (limit `lim`), 2 stage2.0 (256 hidden columns, then PE(pos)), 3 plain copy of `ksteps` floats,
                        // 4 TRANSPOSED (the backward chain's A operands): output n = n0 + 32 fb + c is COLUMN n of src (valid below `lim`),
                        //   K index k = bx_kmap_hidden(..) is its ROW (valid below `klim`); `nblk` blocks per K step (4, a half pass: 2)
    int32_t lim;
    int32_t klim, nblk;
};
struct BxPackTable {
    BxPackPass pass[32];
};
__device__ __forceinline__ uint32_t bx_bf16_rne_dev(float f) {
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7f800000u) == 0x7f800000u) return u >> 16;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__global__ void pack_bf16x3_device_kernel(BxPackTable t, char* blob) {
    const BxPackPass ps = t.pass[blockIdx.y];
    if (ps.mode == 3) {
        float* d = reinterpret_cast<float*>(blob) + ps.dst;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ps.ksteps; i += gridDim.x * blockDim.x) d[i] = ps.src[i];
        return;
    }
    uint16_t* d = reinterpret_cast<uint16_t*>(blob + ps.dst);
    const int nblk = ps.mode == 4 ? ps.nblk : 4;
    const int total = ps.ksteps * nblk * 512;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int j = e & 7, lane = (e >> 3) & 63, fb = (e >> 9) % nblk, tt = (e >> 9) / nblk;
        const int h = lane >> 5, c = lane & 31;
        int k;
        if (ps.mode == 4) {
            k = bx_kmap_hidden(tt, h, j);
            const int n = ps.n0 + 32 * fb + c;
            uint32_t q0 = 0, q1 = 0, q2 = 0;
            if (k < ps.klim && n < ps.lim) {
                const float w = ps.src[(int64_t)k * ps.in + n];
                q0 = bx_bf16_rne_dev(w);
                const float r1 = w - __uint_as_float(q0 << 16);
                q1 = bx_bf16_rne_dev(r1);
                const float r2 = r1 - __uint_as_float(q1 << 16);
                q2 = bx_bf16_rne_dev(r2);
            }
            uint16_t* u = d + (size_t)(tt * nblk + fb) * (BX_UNIT / 2) + lane * 8 + j;
            u[0] = (uint16_t)q0;
            u[BX_CHUNK / 2] = (uint16_t)q1;
            u[BX_CHUNK] = (uint16_t)q2;
            continue;
        }
        if (ps.mode == 0) {
            k = bx_kmap_hidden(tt, h, j);
        } else if (ps.mode == 1) {
            k = bx_kmap_enc(tt, h, j);
            k = k < ps.lim ? k : -1;
        } else if (tt < 16) {
            k = bx_kmap_hidden(tt, h, j);
        } else {
            k = bx_kmap_enc(tt - 16, h, j);
            k = k < ps.lim ? 256 + k : -1;
        }
        uint32_t p0 = 0, p1 = 0, p2 = 0;
        if (k >= 0) {
            const float w = ps.src[(int64_t)(ps.n0 + 32 * fb + c) * ps.in + k];
            p0 = bx_bf16_rne_dev(w);
            const float r1 = w - __uint_as_float(p0 << 16);
            p1 = bx_bf16_rne_dev(r1);
            const float r2 = r1 - __uint_as_float(p1 << 16);
            p2 = bx_bf16_rne_dev(r2);
        }
        uint16_t* u = d + (size_t)(tt * 4 + fb) * (BX_UNIT / 2) + lane * 8 + j;
        u[0] = (uint16_t)p0;
        u[BX_CHUNK / 2] = (uint16_t)p1;
        u[BX_CHUNK] = (uint16_t)p2;
    }
}
}  // namespace stnerf

extern "C" int stnerf_pack_net_bf16x3_device(int kind, const float* const* W, const float* const* B, int n_tensors, void* dst_dev,
                                             int64_t dst_bytes, stnerf_stream_t stream) {
    STNERF_REQUIRE(W && B && dst_dev, "pack_net_bf16x3_device: null pointer");
    STNERF_REQUIRE(STNERF_NET_IS_SPACE(kind) || kind == STNERF_NET_MOTION, "pack_net_bf16x3_device: unknown net kind %d", kind);
    const BxLayout X = bx_layout(kind);
    STNERF_REQUIRE(dst_bytes >= X.total_bytes, "pack_net_bf16x3_device: dst too small (%lld < %lld)", (long long)dst_bytes, (long long)X.total_bytes);
    STNERF_REQUIRE(((uintptr_t)dst_dev & 1023) == 0, "pack_net_bf16x3_device: the blob must be 1 KB aligned");
    // ---- the f32 section (also validates the tensor count), then the pad and the consts cleared
    if (const int rc = stnerf_pack_net_device(kind, W, B, n_tensors, dst_dev, X.f32_floats * 4, stream)) return rc;
    if (hipMemsetAsync(static_cast<char*>(dst_dev) + X.f32_floats * 4, 0, (size_t)(X.stream_off - X.f32_floats * 4), as_stream(stream)) != hipSuccess)
        return STNERF_ELAUNCH;
    BxPackTable t;
    memset(&t, 0, sizeof(t));
    int n = 0;
    int64_t off = X.stream_off;
    auto pass = [&](const float* w, int in, int n0, int ksteps, int mode, int lim) {
        t.pass[n++] = BxPackPass{w, off, in, n0, ksteps, mode, lim, 0, 4};
        off += (int64_t)ksteps * 4 * BX_UNIT;
    };
    auto cpy = [&](const float* src, int count, int cst_off) { t.pass[n++] = BxPackPass{src, X.consts_off / 4 + cst_off, 0, 0, count, 3, 0, 0, 4}; };
    if (STNERF_NET_IS_SPACE(kind)) {
        const bool deep = STNERF_NET_IS_DEEP(kind);
        const int nt = deep ? 12 : 10;
        const int in_f[7] = {63, 256, 256, 256, 319, 256, 256};
        for (int i = 0; i < 7; ++i) cpy(B[i], 256, BXC_B + 256 * i);
        for (int i = 0; i < 2 && deep; ++i) cpy(B[9 + i], 128, BXC_B_DEEP + 128 * i);
        cpy(W[7], 256, BXC_W_SIGMA);
        cpy(W[nt - 1], 3 * 128, BXC_W_RGB2);
        for (int i = 0; i < 7; ++i)
            for (int half = 0; half < 2; ++half) {
                if (i == 0)
                    pass(W[0], 63, 128 * half, 4, 1, 63);
                else if (i == 4)
                    pass(W[4], 319, 128 * half, 20, 2, 63);
                else
                    pass(W[i], in_f[i], 128 * half, 16, 0, 0);
            }
        pass(W[8], 256 + 27 + (STNERF_NET_USES_TIME(kind) ? 21 : 0), 0, 16, 0, 0);
        for (int i = 0; i < 2 && deep; ++i) pass(W[9 + i], 128, 0, 8, 0, 0);
    } else {
        for (int i = 0; i < 5; ++i) cpy(B[i], 128, BXM_B + 128 * i);
        cpy(W[5], 3 * 128, BXM_W_OUT);
        pass(W[0], 84, 0, 6, 1, 84);
        for (int i = 1; i < 5; ++i) pass(W[i], 128, 0, 8, 0, 0);
    }
    STNERF_REQUIRE(off == X.total_bytes && n <= 32, "pack_net_bf16x3_device: internal: stream of %lld B, expected %lld", (long long)(off - X.stream_off),
                   (long long)X.n_slots * BX_SLOT);
    hipLaunchKernelGGL(pack_bf16x3_device_kernel, dim3(32, n), dim3(256), 0, as_stream(stream), t, static_cast<char*>(dst_dev));
    STNERF_CHECK_LAUNCH("pack_net_bf16x3_device");
    return STNERF_OK;
}

// The backward chain's blob (train_space_dx_bx_kernel): [consts: density_net.0's 256 weights | the colour head [3][128] | pad to 4 KB]
// [stream: the transposed weights as bf16 triples in consumption order, 24 KB slots].  with_dpos: with the two half passes that
// carry the gradient on to PE(pos) (stage2.0's PE columns behind its first pass, stage1.0 at the end) -- a stream without them for
// networks whose sample points need no gradient.  Tensors as for stnerf_pack_net_device (reference layout, fused-path networks:
// TKERNEL_INC_RAW, USE_DIR, no deep_rgb).
extern "C" int64_t stnerf_packed_bytes_dx_bf16x3(int kind, int with_dpos) {
    if (kind != STNERF_NET_SPACE && kind != STNERF_NET_SPACE_TIME) {
        set_error("packed_bytes_dx_bf16x3: kind %d (SpaceNets without deep_rgb)", kind);
        return STNERF_EINVAL;
    }
    const int slots = 8 + 6 * 16 + (with_dpos ? 8 : 0);
    return (int64_t)BXD_CONST * 4 + (int64_t)slots * BX_SLOT;
}

extern "C" int stnerf_pack_dx_bf16x3_device(int kind, const float* const* W, int n_tensors, int with_dpos, void* dst_dev, int64_t dst_bytes,
                                            stnerf_stream_t stream) {
    STNERF_REQUIRE(W && dst_dev, "pack_dx_bf16x3_device: null pointer");
    const int64_t total = stnerf_packed_bytes_dx_bf16x3(kind, with_dpos);
    if (total < 0) return (int)total;
    STNERF_REQUIRE(n_tensors == 10, "pack_dx_bf16x3_device: a SpaceNet without deep_rgb has 10 weight tensors, got %d", n_tensors);
    for (int i = 0; i < 10; ++i) STNERF_REQUIRE(W[i], "pack_dx_bf16x3_device: tensor %d is null", i);
    STNERF_REQUIRE(dst_bytes >= total && ((uintptr_t)dst_dev & 1023) == 0, "pack_dx_bf16x3_device: the destination needs %lld bytes, 1 KB aligned",
                   (long long)total);
    if (hipMemsetAsync(dst_dev, 0, (size_t)BXD_CONST * 4, as_stream(stream)) != hipSuccess) return STNERF_ELAUNCH;
    BxPackTable t;
    memset(&t, 0, sizeof(t));
    int n = 0;
    int64_t off = (int64_t)BXD_CONST * 4;
    auto pass = [&](const float* w, int in, int n0, int ksteps, int nlim, int klim, int nblk) {
        t.pass[n++] = BxPackPass{w, off, in, n0, ksteps, 4, nlim, klim, nblk};
        off += (int64_t)ksteps * nblk * BX_UNIT;
    };
    t.pass[n++] = BxPackPass{W[7], BXD_W_SIGMA, 0, 0, 256, 3, 0, 0, 4};
    t.pass[n++] = BxPackPass{W[9], BXD_W_RGB2, 0, 0, 3 * 128, 3, 0, 0, 4};
    const int in8 = 256 + 27 + (kind == STNERF_NET_SPACE_TIME ? 21 : 0);
    for (int half = 0; half < 2; ++half) pass(W[8], in8, 128 * half, 8, 256, 128, 4);          // rgb_net.1[:, :256]
    for (int l = 6; l >= 1; --l) {
        const int in = l == 4 ? 319 : 256;
        pass(W[l], in, 0, 16, 256, 256, 4);
        if (l == 4 && with_dpos) pass(W[4], 319, 256, 16, 319, 256, 2);                        // stage2.0's PE(pos) columns: a half pass
        pass(W[l], in, 128, 16, 256, 256, 4);
    }
    if (with_dpos) pass(W[0], 63, 0, 16, 63, 256, 2);                                          // stage1.0: a half pass
    STNERF_REQUIRE(off == total && n <= 32, "pack_dx_bf16x3_device: internal: %lld B, expected %lld", (long long)off, (long long)total);
    hipLaunchKernelGGL(pack_bf16x3_device_kernel, dim3(32, n), dim3(256), 0, as_stream(stream), t, static_cast<char*>(dst_dev));
    STNERF_CHECK_LAUNCH("pack_dx_bf16x3_device");
    return STNERF_OK;
}

// The backward chain in split bf16: see stnerf_train_spacenet_dx (include/stnerf.h); dpe / dpe_skip both or neither.
extern "C" int stnerf_train_spacenet_dx_bf16x3(const void* packed_dx, int with_dpos, const float* d_raw, int64_t rows, const uint32_t* relu_bits,
                                               int64_t relu_bits_stride, float* const* dy_host, const int32_t* ld_dy_host, float* dpe,
                                               int32_t ld_dpe, float* dpe_skip, int32_t ld_dpe_skip, stnerf_stream_t stream) {
    STNERF_REQUIRE(packed_dx && d_raw && relu_bits && dy_host && ld_dy_host, "train_spacenet_dx_bf16x3: null pointer");
    STNERF_REQUIRE(rows >= 0 && rows <= 0x7fffff00ll, "train_spacenet_dx_bf16x3: %lld rows (split the batch)", (long long)rows);
    STNERF_REQUIRE(((uintptr_t)packed_dx & 1023) == 0 && (((uintptr_t)d_raw | (uintptr_t)relu_bits) & 15) == 0 && (relu_bits_stride & 3) == 0 &&
                       relu_bits_stride >= rows * 8,
                   "train_spacenet_dx_bf16x3: the packed weights must be 1 KB aligned, d_raw and relu_bits 16-byte (stage stride: a multiple of 4 words, >= 8 x rows)");
    STNERF_REQUIRE((with_dpos != 0) == (dpe != nullptr) && (dpe != nullptr) == (dpe_skip != nullptr),
                   "train_spacenet_dx_bf16x3: dpe and dpe_skip come with a with_dpos stream, and only with one");
    if (rows == 0) return STNERF_OK;
    DxBxArgs a;
    memset(&a, 0, sizeof(a));
    a.blob = static_cast<const char*>(packed_dx);
    a.d_raw = d_raw;
    a.bits = relu_bits;
    a.bits_stride = relu_bits_stride;
    a.rows = rows;
    a.n_slots = 8 + 6 * 16 + (with_dpos ? 8 : 0);
    for (int i = 0; i < 8; ++i) {
        const int width = i == 7 ? 128 : 256;
        STNERF_REQUIRE(dy_host[i] && ((uintptr_t)dy_host[i] & 15) == 0 && (ld_dy_host[i] & 3) == 0 && ld_dy_host[i] >= width,
                       "train_spacenet_dx_bf16x3: matrix %d must be 16-byte aligned with a row stride that is a multiple of 4 floats", i);
        a.dy[i] = dy_host[i];
        a.ld_dy[i] = ld_dy_host[i];
    }
    if (dpe) {
        STNERF_REQUIRE((((uintptr_t)dpe | (uintptr_t)dpe_skip) & 15) == 0 && (ld_dpe & 3) == 0 && ld_dpe >= 64 && (ld_dpe_skip & 3) == 0 && ld_dpe_skip >= 64,
                       "train_spacenet_dx_bf16x3: d PE needs 64 columns, 16-byte aligned");
        a.dpe = dpe;
        a.ld_dpe = ld_dpe;
        a.dpe_skip = dpe_skip;
        a.ld_dpe_skip = ld_dpe_skip;
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return launch_bf16x3_dx(a, with_dpos != 0, cus, as_stream(stream));
}
