// Prototype + microbenchmarks for the split-bf16 ("bf16x3") wave kernel (csrc/mlp_bf16x3.hip).
//
//   every fp32 operand = three bf16 pieces (8 + 8 + 8 significand bits: exact), a*b = the six leading cross terms on
//   v_mfma_f32_32x32x16_bf16, f32 accumulate.  A wave owns 32 samples and all features (the organisation of
//   csrc/mlp_wave.hip: accumulator layout = next layer's B operand layout); the weights (6 B per value) are streamed ONCE
//   per CU by LDS-DMA (global_load_lds_dwordx4) into a ring of 24 KB slots and read by the four waves with ds_read_b128.
//
// What this file measures (one run, MI355X):
//   1. numerics of a stack of 256x256 layers (ReLU between) against fp64, next to an fp32 fma chain
//   2. the layer loop's rate with its parts switched off one at a time (conversion pass, barrier, ds_read, DMA)
//   3. vector instructions in the shadow of bf16 MFMAs (same wave), dependent-accumulator distance
//   4. weights straight through the vector L1 (four waves, same addresses) instead of the LDS ring
//   5. (round 5, `--sustain SEC`) joules per item at the socket's power limit: every variant run for SEC seconds with the
//      socket's energy counter read on both sides (rocm_smi), next to an UPPER BOUND for a wave that owns 64 samples
//      (proto64_kernel: every weight operand read from the ring feeds two sample tiles' MFMAs, so the L2 -> LDS stream, the
//      LDS -> AGPR reads and the barriers per MFMA all halve -- with accumulators that a real kernel has no registers for)
//
// hipcc --offload-arch=gfx950 -O3 -o bf16x3_proto bf16x3_proto.hip -L/opt/rocm/lib -lrocm_smi64
#include <hip/hip_runtime.h>
#include <rocm_smi/rocm_smi.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int SLOT = 24576;   // one K step of 16 for 256 outputs: 8 feature blocks x 3 pieces x 1 KB
constexpr int RING = 4;
constexpr int NLMAX = 8;

enum { F_CONV = 1, F_BAR = 2, F_LDS = 4, F_DMA = 8, F_ALL = 15, F_NOWAITV = 16, F_SPREAD = 32, F_NOISSUE = 64, F_FLAGS = 128 };

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 h = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<unsigned*>(&h);
}

// 8 fp32 values -> three bf16x8 (x = p0 + p1 + p2 exactly)
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& p0, bf16x8& p1, bf16x8& p2) {
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

__device__ __forceinline__ float relu_bits(float x) {
    const int b = __float_as_int(x);
    return __int_as_float(b > 0 ? b : 0);
}

#define VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 0xf) | (((n) >> 4) << 14) | 0x0f70)

// the wave's quarter of ring slot `g` (6 x 1 KB) on its way
template <int FLAGS, int C0 = 0, int C1 = 6>
__device__ __forceinline__ void dma_slot(const char* blob, unsigned g, unsigned nslots, char* ring, int wave, int lane) {
    if (!(FLAGS & F_DMA) || (FLAGS & F_NOISSUE)) return;
    const unsigned src = g % nslots;
    const char* s = blob + (size_t)src * SLOT + wave * 6144 + lane * 16;
    auto d = (__attribute__((address_space(3))) char*)(ring) + (g % RING) * SLOT + wave * 6144;
#pragma unroll
    for (int c = C0; c < C1; ++c)
        __builtin_amdgcn_global_load_lds(s + c * 1024, (__attribute__((address_space(3))) void*)(d + c * 1024), 16, 0, 0);
}

struct Apair {
    bf16x8 a[2][3];
};
// The A operands live in AGPRs (the arch VGPRs hold the 192 activation registers): ds_read_b128 straight into them, as
// asm -- the compiler does not count these reads, wait_pair() is their s_waitcnt (it names every destination, so no
// consumer can be scheduled above it).
template <int FLAGS>
__device__ __forceinline__ void read_pair(Apair& A, const char* slot, int pr, int lane) {
    if (!(FLAGS & F_LDS)) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int p = 0; p < 3; ++p) asm volatile("" : "+a"(A.a[f][p]));
        return;
    }
    const unsigned addr = (unsigned)(uintptr_t)(slot + pr * 6144 + lane * 16);
    asm volatile(
        "ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:1024\n\tds_read_b128 %2, %6 offset:2048\n\t"
        "ds_read_b128 %3, %6 offset:3072\n\tds_read_b128 %4, %6 offset:4096\n\tds_read_b128 %5, %6 offset:5120"
        : "=a"(A.a[0][0]), "=a"(A.a[0][1]), "=a"(A.a[0][2]), "=a"(A.a[1][0]), "=a"(A.a[1][1]), "=a"(A.a[1][2])
        : "v"(addr)
        : "memory");
}
__device__ __forceinline__ void wait_pair(Apair& A) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+a"(A.a[0][0]), "+a"(A.a[0][1]), "+a"(A.a[0][2]), "+a"(A.a[1][0]), "+a"(A.a[1][1]), "+a"(A.a[1][2]));
}

__device__ __forceinline__ void mma_pair(f32x16& c0, f32x16& c1, const Apair& A, const bf16x8& b0, const bf16x8& b1, const bf16x8& b2) {
    // smallest terms first: (a2 b0), (a0 b2), (a1 b1), (a1 b0), (a0 b1), (a0 b0)
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[0][2], b0, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[1][2], b0, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[0][0], b2, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[1][0], b2, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[0][1], b1, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[1][1], b1, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[0][1], b0, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[1][1], b0, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[0][0], b1, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[1][0], b1, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[0][0], b0, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[1][0], b0, c1, 0, 0, 0);
}

// One 256 -> 256 layer.  On entry: slot g0 is readable (its barrier passed), Ax holds pair 0 of slot g0, slots g0+1, g0+2
// are in flight.  On exit the same holds for g0 + 16.
template <int FLAGS>
__device__ __forceinline__ void layer256(f32x16 (&acc)[8], const bf16x8 (&act)[3][16], Apair& Ax, Apair& Ay, const char* blob,
                                         unsigned& g, unsigned nslots, char* ring, int wave, int lane) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const char* slot = ring + (s % RING) * SLOT;
        const char* nslot = ring + ((s + 1) % RING) * SLOT;
        __builtin_amdgcn_sched_barrier(0);
        read_pair<FLAGS>(Ay, slot, 1, lane);
        __builtin_amdgcn_sched_barrier(0);
        mma_pair(acc[0], acc[1], Ax, act[0][s], act[1][s], act[2][s]);
        __builtin_amdgcn_sched_barrier(0);
        wait_pair(Ay);
        read_pair<FLAGS>(Ax, slot, 2, lane);
        __builtin_amdgcn_sched_barrier(0);
        mma_pair(acc[2], acc[3], Ay, act[0][s], act[1][s], act[2][s]);
        __builtin_amdgcn_sched_barrier(0);
        wait_pair(Ax);
        read_pair<FLAGS>(Ay, slot, 3, lane);
        __builtin_amdgcn_sched_barrier(0);
        mma_pair(acc[4], acc[5], Ax, act[0][s], act[1][s], act[2][s]);
        __builtin_amdgcn_sched_barrier(0);
        // slot g + 1 landed everywhere, slot g - 1 is free everywhere
        if ((FLAGS & F_DMA) && !(FLAGS & F_NOWAITV)) VMCNT(6);
        if (FLAGS & F_BAR) __builtin_amdgcn_s_barrier();
        if (!(FLAGS & F_SPREAD)) dma_slot<FLAGS>(blob, g + 3, nslots, ring, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        wait_pair(Ay);
        read_pair<FLAGS>(Ax, nslot, 0, lane);
        __builtin_amdgcn_sched_barrier(0);
        if (FLAGS & F_SPREAD) {
            // the six DMA instructions one behind each of the first MFMAs of the pair
            const bf16x8 &b0 = act[0][s], &b1 = act[1][s], &b2 = act[2][s];
            f32x16 &c0 = acc[6], &c1 = acc[7];
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ay.a[0][2], b0, c0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0); dma_slot<FLAGS, 0, 1>(blob, g + 3, nslots, ring, wave, lane); __builtin_amdgcn_sched_barrier(0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ay.a[1][2], b0, c1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0); dma_slot<FLAGS, 1, 2>(blob, g + 3, nslots, ring, wave, lane); __builtin_amdgcn_sched_barrier(0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ay.a[0][0], b2, c0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0); dma_slot<FLAGS, 2, 3>(blob, g + 3, nslots, ring, wave, lane); __builtin_amdgcn_sched_barrier(0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ay.a[1][0], b2, c1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0); dma_slot<FLAGS, 3, 4>(blob, g + 3, nslots, ring, wave, lane); __builtin_amdgcn_sched_barrier(0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ay.a[0][1], b1, c0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0); dma_slot<FLAGS, 4, 5>(blob, g + 3, nslots, ring, wave, lane); __builtin_amdgcn_sched_barrier(0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ay.a[1][1], b1, c1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0); dma_slot<FLAGS, 5, 6>(blob, g + 3, nslots, ring, wave, lane); __builtin_amdgcn_sched_barrier(0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ay.a[0][1], b0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ay.a[1][1], b0, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ay.a[0][0], b1, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ay.a[1][0], b1, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ay.a[0][0], b0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ay.a[1][0], b0, c1, 0, 0, 0);
        } else
        mma_pair(acc[6], acc[7], Ay, act[0][s], act[1][s], act[2][s]);
        g += 1;
        __builtin_amdgcn_sched_barrier(0);
        wait_pair(Ax);
    }
    __builtin_amdgcn_sched_barrier(0);
}


// ---------------------------------------------------------------------------------------------------------------------
// Ring synchronisation WITHOUT the per-slot workgroup barrier: ring buffer b belongs to wave b, which fetches every slot
// g = b (mod 4) whole (24 x 1 KB).  ready[b] = (slot landed in buffer b) + 1, done[w] = slots wave w has read completely.
//   slot g, wave (g + 3) % 4: once min(done) >= g (everyone is through slot g - 1), refill buffer (g - 1) % 4 with slot g + 3
//   slot g, wave (g + 1) % 4: its fetch of slot g + 1 (issued during slot g - 2) has landed: vmcnt(0), ready[(g+1)%4] = g + 2
//   every wave, before its first read of slot g + 1: ready[(g + 1) % 4] >= g + 2;  after its last read of slot g: done[w] = g + 1
// ---------------------------------------------------------------------------------------------------------------------
// (asm: a compiler-visible LDS access behind an LDS-DMA in flight is given an s_waitcnt vmcnt(0) -- the DMA might alias it --,
// which would make every poll wait for the poller's own fetches)
__device__ __forceinline__ unsigned lds_u32(const unsigned* p) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(uintptr_t)p) : "memory");
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void lds_put(unsigned* p, unsigned v) {
    asm volatile("ds_write_b32 %0, %1" : : "v"((unsigned)(uintptr_t)p), "v"(v) : "memory");
}
template <int C0, int NC>
__device__ __forceinline__ void mma_pair_dma(f32x16& c0, f32x16& c1, const Apair& A, const bf16x8& b0, const bf16x8& b1, const bf16x8& b2,
                                             bool issue, const char* src, __attribute__((address_space(3))) char* dst) {
#define PROTO_DMA(c) do { if ((c) < NC) { __builtin_amdgcn_sched_barrier(0); if (issue) __builtin_amdgcn_global_load_lds(src + (C0 + c) * 1024, (__attribute__((address_space(3))) void*)(dst + (C0 + c) * 1024), 16, 0, 0); __builtin_amdgcn_sched_barrier(0); } } while (0)
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[0][2], b0, c0, 0, 0, 0);
    PROTO_DMA(0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[1][2], b0, c1, 0, 0, 0);
    PROTO_DMA(1);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[0][0], b2, c0, 0, 0, 0);
    PROTO_DMA(2);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[1][0], b2, c1, 0, 0, 0);
    PROTO_DMA(3);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[0][1], b1, c0, 0, 0, 0);
    PROTO_DMA(4);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[1][1], b1, c1, 0, 0, 0);
    PROTO_DMA(5);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[0][1], b0, c0, 0, 0, 0);
    PROTO_DMA(6);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[1][1], b0, c1, 0, 0, 0);
    PROTO_DMA(7);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[0][0], b1, c0, 0, 0, 0);
    PROTO_DMA(8);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[1][0], b1, c1, 0, 0, 0);
    PROTO_DMA(9);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[0][0], b0, c0, 0, 0, 0);
    PROTO_DMA(10);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[1][0], b0, c1, 0, 0, 0);
    PROTO_DMA(11);
#undef PROTO_DMA
}

// Every wave fetches its quarter of every slot (as in the barrier version: a single wave's LDS-DMAs complete one after the
// other, 24 from one wave take ~6 slot times); ready[w] = slots of which wave w's quarter has landed, done[w] = slots wave w
// has read completely; both polled one phase after they are read (the read is issued early, its value used later).
using Quad = u32x4;
__device__ __forceinline__ void lds_get4(Quad& q, const unsigned* p) {   // issue only
    asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"((unsigned)(uintptr_t)p) : "memory");
}
__device__ __forceinline__ unsigned quad_min(Quad& q) {                  // after its wait
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q));
    const unsigned a = (unsigned)__builtin_amdgcn_readfirstlane((int)q[0]), b = (unsigned)__builtin_amdgcn_readfirstlane((int)q[1]);
    const unsigned c = (unsigned)__builtin_amdgcn_readfirstlane((int)q[2]), d = (unsigned)__builtin_amdgcn_readfirstlane((int)q[3]);
    const unsigned ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}
__device__ __forceinline__ void poll4(const unsigned* p, unsigned want, Quad& early, unsigned* spins) {
    if (quad_min(early) >= want) return;   // the early read already shows it (the usual case)
    unsigned tries = 0;
    Quad q;
    do {
        lds_get4(q, p);
    } while (quad_min(q) < want && ++tries <= (1u << 20));
    if ((threadIdx.x & 63) == 0) atomicAdd(spins, tries + 1);
}

template <int VAR>
__device__ __forceinline__ void layer256_flags(f32x16 (&acc)[8], const bf16x8 (&act)[3][16], Apair& Ax, Apair& Ay, const char* blob,
                                               unsigned& g, unsigned nslots, char* ring, unsigned* ready, unsigned* done, int wave, int lane,
                                               unsigned* spins) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const char* slot = ring + (s % RING) * SLOT;
        const char* nslot = ring + ((s + 1) % RING) * SLOT;
        __builtin_amdgcn_sched_barrier(0);
        Quad qd, qr;
        lds_get4(qd, done);                         // (used behind pair 0: everyone through slot g - 1?)
        // my quarter of slot g + 1 (issued at the start of slot g - 2) has landed: in flight behind it at most slot g + 2's six
        if (!(VAR & 1)) VMCNT(6);
        lds_put(ready + wave, g + 2);
        const unsigned src_slot = (g + 3) % nslots;
        const char* src = blob + (size_t)src_slot * SLOT + wave * 6144 + lane * 16;
        auto dst = (__attribute__((address_space(3))) char*)(ring) + ((s + 3) & 3) * SLOT + wave * 6144;
        __builtin_amdgcn_sched_barrier(0);
        read_pair<F_ALL>(Ay, slot, 1, lane);
        __builtin_amdgcn_sched_barrier(0);
        mma_pair_dma<0, 0>(acc[0], acc[1], Ax, act[0][s], act[1][s], act[2][s], false, src, dst);
        __builtin_amdgcn_sched_barrier(0);
        wait_pair(Ay);
        if (!(VAR & 4)) poll4(done, g, qd, spins);                  // buffer (g - 1) % 4 is free everywhere: my six chunks of slot g + 3 go out
        read_pair<F_ALL>(Ax, slot, 2, lane);
        lds_get4(qr, ready);                        // (used at the turn: slot g + 1 landed everywhere?)
        __builtin_amdgcn_sched_barrier(0);
        mma_pair_dma<0, 6>(acc[2], acc[3], Ay, act[0][s], act[1][s], act[2][s], true, src, dst);
        __builtin_amdgcn_sched_barrier(0);
        wait_pair(Ax);
        read_pair<F_ALL>(Ay, slot, 3, lane);
        __builtin_amdgcn_sched_barrier(0);
        mma_pair_dma<0, 0>(acc[4], acc[5], Ax, act[0][s], act[1][s], act[2][s], false, src, dst);
        __builtin_amdgcn_sched_barrier(0);
        wait_pair(Ay);   // my last read of slot g is complete
        lds_put(done + wave, g + 1);
        if (!(VAR & 2)) poll4(ready, g + 2, qr, spins + 1);         // slot g + 1 landed everywhere
        __builtin_amdgcn_sched_barrier(0);
        read_pair<F_ALL>(Ax, nslot, 0, lane);
        __builtin_amdgcn_sched_barrier(0);
        mma_pair_dma<0, 0>(acc[6], acc[7], Ay, act[0][s], act[1][s], act[2][s], false, src, dst);
        g += 1;
        __builtin_amdgcn_sched_barrier(0);
        wait_pair(Ax);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// in place: act = split(relu(acc)), acc = bias of the next layer (lbias: 256 floats in LDS)
template <int FLAGS>
__device__ __forceinline__ void convert(f32x16 (&acc)[8], bf16x8 (&act)[3][16], const float* lbias, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int fb = 0; fb < 8; ++fb) {
        if (FLAGS & F_CONV) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = relu_bits(acc[fb][8 * t + j]);
                split8(v, act[0][2 * fb + t], act[1][2 * fb + t], act[2][2 * fb + t]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b = *reinterpret_cast<const float4*>(lbias + 32 * fb + 8 * q + 4 * h);
            acc[fb][4 * q + 0] = b.x;
            acc[fb][4 * q + 1] = b.y;
            acc[fb][4 * q + 2] = b.z;
            acc[fb][4 * q + 3] = b.w;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// x: [128][256] fp32 (>= 0), feature-major per sample; out: [blocks][128][256] pre-activation of the last layer
template <int FLAGS>
__global__ __launch_bounds__(256, 1) void proto_kernel(const char* blob, const float* bias, int nl, const float* x, float* out,
                                                       int items, unsigned long long* cycles, unsigned* spins = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    float* lbias = reinterpret_cast<float*>(smem + RING * SLOT);
    unsigned* ready = reinterpret_cast<unsigned*>(smem + RING * SLOT + NLMAX * 1024);
    unsigned* done = ready + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    for (int i = tid; i < nl * 256; i += 256) lbias[i] = bias[i];
    __syncthreads();
    const unsigned nslots = nl * 16;
    unsigned g = 0;
    if (FLAGS & F_FLAGS) {
        // prime: every wave its quarter of slots 0, 1, 2; slot 0 landed everywhere, nothing read yet
        dma_slot<F_ALL>(blob, 0, nslots, ring, wave, lane);
        dma_slot<F_ALL>(blob, 1, nslots, ring, wave, lane);
        dma_slot<F_ALL>(blob, 2, nslots, ring, wave, lane);
        if (tid < 4) {
            ready[tid] = 1;
            done[tid] = 0;
        }
        VMCNT(12);
        __syncthreads();
    } else {
    // prime: slots 0, 1, 2
    dma_slot<FLAGS>(blob, 0, nslots, ring, wave, lane);
    dma_slot<FLAGS>(blob, 1, nslots, ring, wave, lane);
    dma_slot<FLAGS>(blob, 2, nslots, ring, wave, lane);
    if (FLAGS & F_DMA) VMCNT(12);
    __builtin_amdgcn_s_barrier();
    }
    f32x16 acc[8];
    bf16x8 act[3][16];
    Apair Ax, Ay;
    read_pair<F_ALL>(Ax, ring, 0, lane);
    wait_pair(Ax);
    const unsigned long long t0 = clock64();
    for (int it = 0; it < items; ++it) {
        // this wave's 32 samples: lane (h, c) holds features 32 fb + 8 q + 4 h + r of sample 32 wave + c
        const float* xs = x + (size_t)(32 * wave + c) * 256;
#pragma unroll
        for (int fb = 0; fb < 8; ++fb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(xs + 32 * fb + 8 * q + 4 * h);
                acc[fb][4 * q + 0] = v.x;
                acc[fb][4 * q + 1] = v.y;
                acc[fb][4 * q + 2] = v.z;
                acc[fb][4 * q + 3] = v.w;
            }
        convert<F_ALL>(acc, act, lbias, lane);
#pragma unroll 1
        for (int l = 0; l < nl; ++l) {
            if (l > 0) convert<FLAGS>(acc, act, lbias + l * 256, lane);
            if (FLAGS & F_FLAGS)
                layer256_flags<(FLAGS >> 8) & 7>(acc, act, Ax, Ay, blob, g, nslots, ring, ready, done, wave, lane, spins);
            else
                layer256<FLAGS>(acc, act, Ax, Ay, blob, g, nslots, ring, wave, lane);
        }
    }
    const unsigned long long t1 = clock64();
    VMCNT(0);
    float* o = out + ((size_t)blockIdx.x * 128 + 32 * wave + c) * 256;
#pragma unroll
    for (int fb = 0; fb < 8; ++fb)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[32 * fb + 8 * (i >> 2) + 4 * h + (i & 3)] = acc[fb][i];
    if (tid == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------------
// 5. A wave that owns 64 samples -- an UPPER BOUND, not a kernel: each weight operand pair read from the ring is used for
// the MFMAs of TWO sample tiles (act0, act1: 2 x 192 registers).  That leaves 80 registers for everything else, so all eight
// feature blocks of a tile accumulate into ONE accumulator (a real kernel needs 8 x 16 x {big, small} per tile
// and pass: the organisation cannot exist in a 512-register wave).  What it has in common with a real one is what costs
// energy: per slot 96 MFMAs on live random operands, 24 operand reads, 6 DMA instructions, one barrier.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mma_quad(f32x16& c, f32x16& d, const Apair& A, const bf16x8& b0,
                                         const bf16x8& b1, const bf16x8& b2, const bf16x8& e0, const bf16x8& e1, const bf16x8& e2) {
    // (back-to-back dependent 32x32x16 MFMAs issue at the full rate: shadow test, DIST = 1: 32.01 cycles)
#define Q4(pa, pb, pe) \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[0][pa], pb, c, 0, 0, 0); \
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[0][pa], pe, d, 0, 0, 0); \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[1][pa], pb, c, 0, 0, 0); \
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.a[1][pa], pe, d, 0, 0, 0);
    Q4(2, b0, e0) Q4(0, b2, e2) Q4(1, b1, e1) Q4(1, b0, e0) Q4(0, b1, e1) Q4(0, b0, e0)
#undef Q4
}

template <int FLAGS>
__global__ __launch_bounds__(256, 1) void proto64_kernel(const char* blob, int nl, const float* x, float* out, int items,
                                                         unsigned long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    const unsigned nslots = nl * 16;
    unsigned g = 0;
    dma_slot<FLAGS>(blob, 0, nslots, ring, wave, lane);
    dma_slot<FLAGS>(blob, 1, nslots, ring, wave, lane);
    dma_slot<FLAGS>(blob, 2, nslots, ring, wave, lane);
    if (FLAGS & F_DMA) VMCNT(12);
    __builtin_amdgcn_s_barrier();
    bf16x8 act0[3][16], act1[3][16];
    // tile 0 = samples 32 wave + c, tile 1 = samples 128 + 32 wave + c of x[256][256]
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        float v[8], w[8];
        const float* xs = x + (size_t)(32 * wave + c) * 256 + 32 * (t >> 1) + 16 * (t & 1) + 4 * h;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = xs[8 * (j >> 2) + (j & 3)];
            w[j] = xs[(size_t)128 * 256 + 8 * (j >> 2) + (j & 3)];
        }
        split8(v, act0[0][t], act0[1][t], act0[2][t]);
        split8(w, act1[0][t], act1[1][t], act1[2][t]);
    }
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    Apair Ax, Ay;
    read_pair<F_ALL>(Ax, ring, 0, lane);
    wait_pair(Ax);
    const unsigned long long t0 = clock64();
    for (int it = 0; it < items; ++it) {
#pragma unroll 1
        for (int l = 0; l < nl; ++l) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const char* slot = ring + (s % RING) * SLOT;
                const char* nslot = ring + ((s + 1) % RING) * SLOT;
                __builtin_amdgcn_sched_barrier(0);
                read_pair<FLAGS>(Ay, slot, 1, lane);
                __builtin_amdgcn_sched_barrier(0);
                mma_quad(acc[0], acc[1], Ax, act0[0][s], act0[1][s], act0[2][s], act1[0][s], act1[1][s], act1[2][s]);
                __builtin_amdgcn_sched_barrier(0);
                wait_pair(Ay);
                read_pair<FLAGS>(Ax, slot, 2, lane);
                __builtin_amdgcn_sched_barrier(0);
                mma_quad(acc[0], acc[1], Ay, act0[0][s], act0[1][s], act0[2][s], act1[0][s], act1[1][s], act1[2][s]);
                __builtin_amdgcn_sched_barrier(0);
                wait_pair(Ax);
                read_pair<FLAGS>(Ay, slot, 3, lane);
                __builtin_amdgcn_sched_barrier(0);
                mma_quad(acc[0], acc[1], Ax, act0[0][s], act0[1][s], act0[2][s], act1[0][s], act1[1][s], act1[2][s]);
                __builtin_amdgcn_sched_barrier(0);
                if ((FLAGS & F_DMA) && !(FLAGS & F_NOWAITV)) VMCNT(6);
                if (FLAGS & F_BAR) __builtin_amdgcn_s_barrier();
                dma_slot<FLAGS>(blob, g + 3, nslots, ring, wave, lane);
                __builtin_amdgcn_sched_barrier(0);
                wait_pair(Ay);
                read_pair<FLAGS>(Ax, nslot, 0, lane);
                __builtin_amdgcn_sched_barrier(0);
                mma_quad(acc[0], acc[1], Ay, act0[0][s], act0[1][s], act0[2][s], act1[0][s], act1[1][s], act1[2][s]);
                g += 1;
                __builtin_amdgcn_sched_barrier(0);
                wait_pair(Ax);
            }
            // keep the running sums bounded (they are not a layer's outputs): halve them once per layer
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] *= 0.5f;
        }
    }
    const unsigned long long t1 = clock64();
    VMCNT(0);
    float* o = out + ((size_t)blockIdx.x * 256 + tid) * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[16 * i + r] = acc[i][r];
    if (tid == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------------
// 6. (round 5, `--reuse SEC`) does the ORDER of a unit's MFMAs matter for energy?  MFMAs only, random operands in registers,
// 12 MFMAs per group = two feature blocks x the six cross terms (A: 2 blocks x 3 weight pieces, B: 3 activation pieces,
// accumulators small0, big0, small1, big1), in different orders.  Per-accumulator term order is the product's in patterns
// 0 - 2 (bit-identical results); 3 - 5 are bounds.
//   0 product (csrc/mlp_bf16x3.hip unit<>): block 0's five small terms back to back, its big term, then block 1
//   1 the two blocks interleaved term by term (B reused by consecutive MFMAs, accumulators alternate)
//   2 B-major: every MFMA that reads b0, then b1, then b2 (per-accumulator order NOT the product's)
//   3 A-major: each weight piece with all its activation pieces back to back
//   4 one A, one B for all twelve (floor)      5 A and B both change on every MFMA (ceiling)
// ---------------------------------------------------------------------------------------------------------------------
template <int PATTERN>
__global__ __launch_bounds__(256, 1) void reuse_kernel(float* out, int iters, unsigned long long* cycles) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.001f * ((threadIdx.x * 7 + i * 3 + r) & 63);
    bf16x8 a[2][3], b[3];
    {
        unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
        for (int i = 0; i < 9; ++i) {
            u32x4 w;
            for (int j = 0; j < 4; ++j) {
                h = h * 1664525u + 1013904223u;
                // two bf16 in [0.5, 2) with random significands and signs
                w[j] = ((h & 0x807f807fu) | 0x3f003f00u) ^ ((h >> 9) & 0x00800080u);
            }
            if (i < 6) a[i / 3][i % 3] = *reinterpret_cast<bf16x8*>(&w);
            else b[i - 6] = *reinterpret_cast<bf16x8*>(&w);
        }
    }
#define MM(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[ACC]) : "v"(A), "v"(B))
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (PATTERN == 0) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                MM(2 * f, a[f][2], b[0]); MM(2 * f, a[f][0], b[2]); MM(2 * f, a[f][1], b[1]); MM(2 * f, a[f][1], b[0]); MM(2 * f, a[f][0], b[1]);
                MM(2 * f + 1, a[f][0], b[0]);
            }
        } else if (PATTERN == 1) {
            MM(0, a[0][2], b[0]); MM(2, a[1][2], b[0]); MM(0, a[0][0], b[2]); MM(2, a[1][0], b[2]); MM(0, a[0][1], b[1]); MM(2, a[1][1], b[1]);
            MM(0, a[0][1], b[0]); MM(2, a[1][1], b[0]); MM(0, a[0][0], b[1]); MM(2, a[1][0], b[1]); MM(1, a[0][0], b[0]); MM(3, a[1][0], b[0]);
        } else if (PATTERN == 2) {
            MM(0, a[0][2], b[0]); MM(0, a[0][1], b[0]); MM(1, a[0][0], b[0]); MM(2, a[1][2], b[0]); MM(2, a[1][1], b[0]); MM(3, a[1][0], b[0]);
            MM(0, a[0][1], b[1]); MM(0, a[0][0], b[1]); MM(2, a[1][1], b[1]); MM(2, a[1][0], b[1]); MM(0, a[0][0], b[2]); MM(2, a[1][0], b[2]);
        } else if (PATTERN == 3) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                MM(2 * f + 1, a[f][0], b[0]); MM(2 * f, a[f][0], b[1]); MM(2 * f, a[f][0], b[2]); MM(2 * f, a[f][1], b[0]); MM(2 * f, a[f][1], b[1]);
                MM(2 * f, a[f][2], b[0]);
            }
        } else if (PATTERN == 4) {
#pragma unroll
            for (int i = 0; i < 12; ++i) MM(i & 3, a[0][0], b[0]);
        } else {
            MM(0, a[0][0], b[0]); MM(2, a[1][1], b[1]); MM(0, a[0][2], b[2]); MM(2, a[1][0], b[0]); MM(0, a[0][1], b[1]); MM(2, a[1][2], b[2]);
            MM(0, a[0][0], b[1]); MM(2, a[1][1], b[2]); MM(1, a[0][2], b[0]); MM(3, a[1][0], b[1]); MM(0, a[0][1], b[2]); MM(2, a[1][2], b[0]);
        }
        if ((it & 63) == 63) {   // keep the running sums finite
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] *= 0.001f;
        }
    }
#undef MM
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------------
// 3. vector instructions in the shadow of bf16 MFMAs (one wave per SIMD): V instructions behind each MFMA; DIST = number
// of independent accumulators the MFMAs rotate over (1 = every MFMA depends on the previous one)
// ---------------------------------------------------------------------------------------------------------------------
template <int V, int DIST>
__global__ __launch_bounds__(256, 1) void shadow_kernel(float* out, int iters, unsigned long long* cycles) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.001f * (threadIdx.x + i + r);
    u32x4 aw, bw;
    for (int i = 0; i < 4; ++i) {
        aw[i] = 0x3f803f80u + threadIdx.x * 0x10001u * (i + 1);
        bw[i] = 0x3e003f00u + threadIdx.x * 0x00030005u * (i + 3);
    }
    bf16x8 a = *reinterpret_cast<bf16x8*>(&aw), b = *reinterpret_cast<bf16x8*>(&bw);
    float x[4] = {1.f, 2.f, 3.f, 4.f}, y = 0.5f;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i % DIST]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < V; ++j) {
                if ((j & 3) == 0) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[j & 3]) : "v"(y));
                if ((j & 3) == 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x[j & 3]));
                if ((j & 3) == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[j & 3]) : "v"(y));
                if ((j & 3) == 3) asm volatile("v_max_i32 %0, 0, %0" : "+v"(x[j & 3]));
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = clock64();
    float s = x[0] + x[1] + x[2] + x[3];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------------
// 4. weights through the vector L1: every wave of the CU streams the whole blob (12 x 16 B per lane per 6 MFMAs... here per
// K step: 24 loads of 16 B per lane, 48 MFMAs), one K step ahead; B operands constant
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void l1_kernel(const char* blob, unsigned nslots, float* out, int steps, unsigned long long* cycles) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 bw = {0x3f803f80u, 0x3f003f00u, 0x3e803e80u, 0x3f803f00u};
    bf16x8 b = *reinterpret_cast<bf16x8*>(&bw);
    bf16x8 wa[24], wb[24];
    const char* base = blob + lane * 16;
#pragma unroll
    for (int i = 0; i < 24; ++i) wa[i] = *reinterpret_cast<const bf16x8*>(base + i * 1024);
    const unsigned long long t0 = clock64();
    unsigned g = 1;
    for (int it = 0; it < steps; it += 2) {
        {
            const char* s = base + (size_t)(g % nslots) * SLOT;
#pragma unroll
            for (int i = 0; i < 24; ++i) wb[i] = *reinterpret_cast<const bf16x8*>(s + i * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int fb = 0; fb < 8; ++fb) acc[fb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[fb * 3 + (t % 3)], b, acc[fb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            g += 1;
        }
        {
            const char* s = base + (size_t)(g % nslots) * SLOT;
#pragma unroll
            for (int i = 0; i < 24; ++i) wa[i] = *reinterpret_cast<const bf16x8*>(s + i * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int fb = 0; fb < 8; ++fb) acc[fb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[fb * 3 + (t % 3)], b, acc[fb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            g += 1;
        }
    }
    const unsigned long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------------------
static unsigned short bf16_rne(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    const unsigned r = u + 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(r >> 16);
}
static float bf16_f(unsigned short h) {
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static void split3(float w, unsigned short (&p)[3]) {
    p[0] = bf16_rne(w);
    const float r1 = w - bf16_f(p[0]);
    p[1] = bf16_rne(r1);
    const float r2 = r1 - bf16_f(p[1]);
    p[2] = bf16_rne(r2);
}
// input feature of B position (K step s, lane half h, element j)
static int kperm(int s, int h, int j) { return 32 * (s >> 1) + 8 * (2 * (s & 1) + (j >> 2)) + 4 * h + (j & 3); }

static double frand() { return (double)rand() / RAND_MAX; }

static unsigned* g_spins = nullptr;
template <int FLAGS>
static void run_proto(const char* name, const char* d_blob, const float* d_bias, int nl, const float* d_x, float* d_out, int items,
                      unsigned long long* d_cyc, int blocks) {
    const int lds = RING * SLOT + NLMAX * 1024 + 64;
    if (!g_spins) { CK(hipMalloc(&g_spins, 8)); }
    CK(hipMemset(g_spins, 0, 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(proto_kernel<FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    proto_kernel<FLAGS><<<blocks, 256, lds>>>(d_blob, d_bias, nl, d_x, d_out, 4, d_cyc, g_spins);
    CK(hipDeviceSynchronize());
    CK(hipMemset(g_spins, 0, 8));
    CK(hipEventRecord(e0));
    proto_kernel<FLAGS><<<blocks, 256, lds>>>(d_blob, d_bias, nl, d_x, d_out, items, d_cyc, g_spins);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long cyc;
    CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
    const double mfma = (double)items * nl * 768;
    const double alg = (double)blocks * items * 128.0 * nl * 2.0 * 256 * 256;
    unsigned sp[2];
    CK(hipMemcpy(sp, g_spins, 8, hipMemcpyDeviceToHost));
    printf("%-34s %8.3f ms  %7.1f alg TF/s  %7.1f exec TF/s  %6.2f cyc/MFMA (ideal 32)  %6.1f MHz(clock64)", name, ms, alg / ms * 1e-9,
           6.0 * alg / ms * 1e-9, (double)cyc / mfma, (double)cyc / (ms * 1e3));
    if (FLAGS & F_FLAGS) {
        // correctness of the flag protocol: the outputs must be those of the barrier version (checked by the caller through d_out)
        printf("  spins: refill %u, ready %u", sp[0], sp[1]);
    }
    printf("\n");
    fflush(stdout);
}

template <int V, int DIST>
static void run_shadow(float* d_out, unsigned long long* d_cyc) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    shadow_kernel<V, DIST><<<256, 256>>>(d_out, 100, d_cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    shadow_kernel<V, DIST><<<256, 256>>>(d_out, iters, d_cyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long cyc;
    CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
    printf("shadow V=%2d DIST=%d : %6.2f cycles per MFMA, %7.1f TF/s executed\n", V, DIST, (double)cyc / (iters * 8.0),
           256.0 * 4 * iters * 8.0 * 32768.0 / ms * 1e-9);
    fflush(stdout);
}

// ---- 5. sustained, energy-metered runs -------------------------------------------------------------------------------
struct Meter {
    bool ok = false;
    float res = 0.f;   // micro joules per count
    Meter() { ok = rsmi_init(0) == RSMI_STATUS_SUCCESS; }
    double joules() {
        uint64_t e = 0, ts = 0;
        if (!ok || rsmi_dev_energy_count_get(0, &e, &res, &ts) != RSMI_STATUS_SUCCESS) return -1.0;
        return (double)e * res * 1e-6;
    }
    double watts() {
        uint64_t p = 0;
        if (!ok || rsmi_dev_current_socket_power_get(0, &p) != RSMI_STATUS_SUCCESS) return -1.0;
        return p * 1e-6;
    }
    double sclk_mhz() {
        rsmi_frequencies_t f;
        if (!ok || rsmi_dev_gpu_clk_freq_get(0, RSMI_CLK_TYPE_SYS, &f) != RSMI_STATUS_SUCCESS) return -1.0;
        return f.frequency[f.current] * 1e-6;
    }
};

// `launch(items)` enqueues one kernel; tiles = sample tiles of 32 per wave (1 or 2): MFMAs per wave and launch = items * nl * 768 * tiles
template <typename L>
static void sustain(Meter& m, const char* name, double seconds, int items, int nl, int tiles, unsigned long long* d_cyc, int blocks, L launch) {
    launch(4);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // a second of load first: the clock settles at the power limit within ~0.3 s
    for (double t = 0; t < 1.0;) {
        CK(hipEventRecord(e0)); launch(items); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t += ms * 1e-3;
    }
    const double j0 = m.joules();
    double busy = 0, cyc_sum = 0, w_sum = 0, f_sum = 0;
    int launches = 0, samples = 0;
    while (busy < seconds) {
        CK(hipEventRecord(e0));
        launch(items);
        CK(hipEventRecord(e1));
        // sample instantaneous power / sclk while the launch runs
        while (hipEventQuery(e1) == hipErrorNotReady) {
            const double w = m.watts(), f = m.sclk_mhz();
            if (w > 0) { w_sum += w; f_sum += f; ++samples; }
        }
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long cyc;
        CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
        busy += ms * 1e-3;
        cyc_sum += (double)cyc;
        ++launches;
    }
    const double j1 = m.joules();
    const double mfma_wave = (double)launches * items * nl * 768.0 * tiles;      // per wave
    const double mfma_all = mfma_wave * blocks * 4;
    const double exec_tf = mfma_all * 32768.0 / busy * 1e-12;
    const double items32 = (double)launches * items * blocks * 4 * tiles;         // 32-sample x nl-layer units
    // (the energy window includes the host gaps between launches: a few % idle at ~200 W; power from the samples is launch-only)
    printf("%-44s %7.1f exec TF/s  %6.2f cyc/MFMA  %6.0f MHz(clock64)  %6.0f MHz(smi)  %6.0f W(smi, %d samples)  %7.1f W(energy ctr)  %8.3f mJ per 32-sample x %d-layer unit  %6.2f pJ/FLOP exec\n",
           name, exec_tf, cyc_sum / mfma_wave, cyc_sum / (busy * 1e6), samples ? f_sum / samples : -1.0, samples ? w_sum / samples : -1.0, samples,
           (j1 - j0) / busy, (j1 - j0) / items32 * 1e3, nl, (j1 - j0) / (mfma_all * 32768.0) * 1e12);
    fflush(stdout);
}

template <int FLAGS>
static void sustain32(Meter& m, const char* name, double seconds, const char* d_blob, const float* d_bias, int nl, const float* d_x, float* d_out,
                      unsigned long long* d_cyc, int blocks) {
    const int lds = RING * SLOT + NLMAX * 1024 + 64;
    if (!g_spins) { CK(hipMalloc(&g_spins, 8)); }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(proto_kernel<FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    sustain(m, name, seconds, 400, nl, 1, d_cyc, blocks,
            [&](int items) { proto_kernel<FLAGS><<<blocks, 256, lds>>>(d_blob, d_bias, nl, d_x, d_out, items, d_cyc, g_spins); });
}
template <int FLAGS>
static void sustain64(Meter& m, const char* name, double seconds, const char* d_blob, int nl, const float* d_x2, float* d_out,
                      unsigned long long* d_cyc, int blocks) {
    const int lds = RING * SLOT + 64;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(proto64_kernel<FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    sustain(m, name, seconds, 200, nl, 2, d_cyc, blocks,
            [&](int items) { proto64_kernel<FLAGS><<<blocks, 256, lds>>>(d_blob, nl, d_x2, d_out, items, d_cyc); });
}

int main(int argc, char** argv) {
    const int nl = 7, blocks = 256;
    srand(1234);
    // ---- weights (nn.Linear-like scale, x4 on odd layers to keep the activations alive), inputs
    std::vector<float> W((size_t)nl * 256 * 256), B((size_t)nl * 256), X(128 * 256);
    for (int l = 0; l < nl; ++l) {
        const double sc = 1.0 / 16.0 * 1.8;
        for (int i = 0; i < 256 * 256; ++i) W[(size_t)l * 65536 + i] = (float)((2 * frand() - 1) * sc);
        for (int i = 0; i < 256; ++i) B[l * 256 + i] = (float)((2 * frand() - 1) * sc);
    }
    for (auto& v : X) v = (float)(frand() * 2.0);
    // ---- blob: [layer][s][fb][piece][lane][8]
    std::vector<unsigned short> blob((size_t)nl * 16 * SLOT / 2);
    for (int l = 0; l < nl; ++l)
        for (int s = 0; s < 16; ++s)
            for (int fb = 0; fb < 8; ++fb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int hh = lane >> 5, cc = lane & 31;
                        unsigned short p[3];
                        split3(W[(size_t)l * 65536 + (size_t)(32 * fb + cc) * 256 + kperm(s, hh, j)], p);
                        for (int pc = 0; pc < 3; ++pc)
                            blob[((((size_t)(l * 16 + s) * 8 + fb) * 3 + pc) * 64 + lane) * 8 + j] = p[pc];
                    }
    char* d_blob;
    float *d_bias, *d_x, *d_out;
    unsigned long long* d_cyc;
    CK(hipMalloc(&d_blob, blob.size() * 2));
    CK(hipMalloc(&d_bias, B.size() * 4));
    CK(hipMalloc(&d_x, X.size() * 4));
    CK(hipMalloc(&d_out, (size_t)blocks * 128 * 256 * 4));
    CK(hipMalloc(&d_cyc, 8));
    CK(hipMemcpy(d_blob, blob.data(), blob.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bias, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_x, X.data(), X.size() * 4, hipMemcpyHostToDevice));

    if (argc > 2 && !strcmp(argv[1], "--reuse")) {
        const double sec = atof(argv[2]);
        Meter m;
        printf("MFMA order / operand reuse, %.1f s each after 1 s, 256 x 4 waves, 12 MFMAs per group; a 'unit' below = 768 MFMAs of one wave\n", sec);
        const int items = 470;     // one "item" = 64 iterations x 12 MFMAs = 768 MFMAs per wave (sustain()'s unit): ~6 ms per launch
        for (int rep = 0; rep < 2; ++rep) {
#define RUN_REUSE(P, NAME) sustain(m, NAME, sec, items, 1, 1, d_cyc, blocks, [&](int it) { reuse_kernel<P><<<blocks, 256>>>(d_out, it * 64, d_cyc); })
            RUN_REUSE(0, "order 0: product (5 small back to back, big)");
            RUN_REUSE(1, "order 1: two blocks interleaved (B reuse x2)");
            RUN_REUSE(2, "order 2: B-major (b0 x6, b1 x4, b2 x2)");
            RUN_REUSE(3, "order 3: A-major (each piece's terms in a row)");
            RUN_REUSE(4, "order 4: one A, one B (floor)");
            RUN_REUSE(5, "order 5: A and B change every MFMA (ceiling)");
#undef RUN_REUSE
        }
        return 0;
    }
    if (argc > 2 && !strcmp(argv[1], "--sustain")) {
        const double sec = atof(argv[2]);
        std::vector<float> X2(256 * 256);
        for (auto& v : X2) v = (float)(frand() * 2.0);
        float* d_x2;
        CK(hipMalloc(&d_x2, X2.size() * 4));
        CK(hipMemcpy(d_x2, X2.data(), X2.size() * 4, hipMemcpyHostToDevice));
        Meter m;
        printf("sustained runs, %.1f s each after 1 s of the same load, 256 workgroups of 4 waves (one per SIMD); energy counter %s\n", sec,
               m.ok ? "rocm_smi rsmi_dev_energy_count_get" : "UNAVAILABLE");
        for (int rep = 0; rep < 2; ++rep) {
            sustain32<F_ALL>(m, "M=32 full (ring, reads, barrier, conversion)", sec, d_blob, d_bias, nl, d_x, d_out, d_cyc, blocks);
            sustain32<F_ALL & ~F_CONV>(m, "M=32 no conversion", sec, d_blob, d_bias, nl, d_x, d_out, d_cyc, blocks);
            sustain64<F_ALL & ~F_CONV>(m, "M=64 upper bound, no conversion", sec, d_blob, nl, d_x2, d_out, d_cyc, blocks);
            sustain32<(F_ALL & ~F_CONV) | F_NOISSUE>(m, "M=32 no conv, NO L2->LDS stream (stale ring)", sec, d_blob, d_bias, nl, d_x, d_out, d_cyc, blocks);
            sustain32<(F_ALL & ~F_CONV & ~F_LDS)>(m, "M=32 no conv, NO operand reads", sec, d_blob, d_bias, nl, d_x, d_out, d_cyc, blocks);
            sustain64<(F_ALL & ~F_CONV) | F_NOISSUE>(m, "M=64 upper bound, NO L2->LDS stream", sec, d_blob, nl, d_x2, d_out, d_cyc, blocks);
            sustain32<0>(m, "M=32 MFMAs only", sec, d_blob, d_bias, nl, d_x, d_out, d_cyc, blocks);
            sustain64<0>(m, "M=64 MFMAs only", sec, d_blob, nl, d_x2, d_out, d_cyc, blocks);
        }
        return 0;
    }
    // ---- 1. numerics: one item
    {
        const int lds = RING * SLOT + NLMAX * 1024;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(proto_kernel<F_ALL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        proto_kernel<F_ALL><<<blocks, 256, lds>>>(d_blob, d_bias, nl, d_x, d_out, 1, d_cyc);
        CK(hipDeviceSynchronize());
        std::vector<float> O((size_t)blocks * 128 * 256);
        CK(hipMemcpy(O.data(), d_out, O.size() * 4, hipMemcpyDeviceToHost));
        // references: fp64 and an fp32 fma chain (k ascending), ReLU between layers, the prototype's bias convention: layer l
        // adds bias[l] (the convert in front of layer l loads lbias + l * 256; layer 0 uses bias[0])
        std::vector<double> a64(128 * 256), n64(128 * 256);
        std::vector<float> a32(128 * 256), n32(128 * 256);
        for (int i = 0; i < 128 * 256; ++i) { a64[i] = X[i]; a32[i] = X[i]; }
        for (int l = 0; l < nl; ++l) {
            for (int smp = 0; smp < 128; ++smp)
                for (int o = 0; o < 256; ++o) {
                    double s64 = B[l * 256 + o];
                    float s32 = B[l * 256 + o];
                    const float* w = &W[(size_t)l * 65536 + (size_t)o * 256];
                    for (int k = 0; k < 256; ++k) {
                        s64 += (double)w[k] * a64[smp * 256 + k];
                        s32 = fmaf(w[k], a32[smp * 256 + k], s32);
                    }
                    n64[smp * 256 + o] = s64;
                    n32[smp * 256 + o] = s32;
                }
            if (l + 1 < nl)
                for (int i = 0; i < 128 * 256; ++i) { a64[i] = n64[i] > 0 ? n64[i] : 0; a32[i] = n32[i] > 0 ? n32[i] : 0; }
        }
        double e_g = 0, e_c = 0, m_g = 0, m_c = 0, scale = 0;
        int bad_blocks = 0;
        for (int i = 0; i < 128 * 256; ++i) {
            const double dg = fabs((double)O[i] - n64[i]), dc = fabs((double)n32[i] - n64[i]);
            e_g += dg * dg; e_c += dc * dc;
            m_g = dg > m_g ? dg : m_g; m_c = dc > m_c ? dc : m_c;
            scale += n64[i] * n64[i];
        }
        for (int b = 1; b < blocks; ++b)
            if (memcmp(&O[(size_t)b * 128 * 256], &O[0], 128 * 256 * 4)) ++bad_blocks;
        printf("numerics after %d layers (rms of outputs %.4g):\n  bf16x3 MFMA  rms err %.3e  max %.3e\n  fp32 chain   rms err %.3e  max %.3e\n"
               "  ratio rms %.2f max %.2f; workgroups that differ from workgroup 0: %d\n",
               nl, sqrt(scale / (128 * 256)), sqrt(e_g / (128 * 256)), m_g, sqrt(e_c / (128 * 256)), m_c,
               sqrt(e_g / e_c), m_g / m_c, bad_blocks);
        fflush(stdout);
    }
    // ---- 2. rate, parts switched off
    const int items = 120;
    run_proto<F_ALL>("full", d_blob, d_bias, nl, d_x, d_out, items, d_cyc, blocks);
    run_proto<F_ALL>("full (again)", d_blob, d_bias, nl, d_x, d_out, items, d_cyc, blocks);
    run_proto<F_ALL & ~F_CONV>("no conversion pass", d_blob, d_bias, nl, d_x, d_out, items, d_cyc, blocks);
    run_proto<F_ALL & ~F_BAR>("no barrier (racy)", d_blob, d_bias, nl, d_x, d_out, items, d_cyc, blocks);
    run_proto<F_ALL & ~F_DMA>("no DMA", d_blob, d_bias, nl, d_x, d_out, items, d_cyc, blocks);
    run_proto<F_ALL & ~F_DMA & ~F_BAR>("no DMA, no barrier", d_blob, d_bias, nl, d_x, d_out, items, d_cyc, blocks);
    run_proto<F_CONV>("conversion only (no LDS/DMA/barrier)", d_blob, d_bias, nl, d_x, d_out, items, d_cyc, blocks);
    run_proto<0>("MFMAs only", d_blob, d_bias, nl, d_x, d_out, items, d_cyc, blocks);
    run_proto<F_ALL>("full, 64 workgroups", d_blob, d_bias, nl, d_x, d_out, items, d_cyc, 64);
    run_proto<(F_ALL & ~F_CONV)>("no conv", d_blob, d_bias, nl, d_x, d_out, items, d_cyc, blocks);
    run_proto<(F_ALL & ~F_CONV) | F_NOWAITV>("no conv, DMA issued, no vmcnt wait (racy)", d_blob, d_bias, nl, d_x, d_out, items, d_cyc, blocks);
    run_proto<(F_ALL & ~F_CONV) | F_NOISSUE>("no conv, vmcnt wait + barrier, DMA not issued", d_blob, d_bias, nl, d_x, d_out, items, d_cyc, blocks);
    run_proto<(F_ALL & ~F_CONV) | F_SPREAD>("no conv, DMA spread behind MFMAs", d_blob, d_bias, nl, d_x, d_out, items, d_cyc, blocks);
    run_proto<(F_ALL & ~F_CONV & ~F_BAR) | F_SPREAD>("no conv, DMA spread, no barrier (racy)", d_blob, d_bias, nl, d_x, d_out, items, d_cyc, blocks);
    run_proto<(F_ALL & ~F_CONV & ~F_BAR) | F_SPREAD | F_NOWAITV>("no conv, DMA spread, no barrier, no wait (racy)", d_blob, d_bias, nl, d_x, d_out, items, d_cyc, blocks);
    run_proto<(F_ALL & ~F_CONV & ~F_LDS)>("no conv, no ds_read", d_blob, d_bias, nl, d_x, d_out, items, d_cyc, blocks);
    // ---- the flag protocol instead of the barrier; its outputs against the barrier version's
    {
        std::vector<float> A((size_t)128 * 256), Bv((size_t)128 * 256);
        run_proto<F_ALL | F_SPREAD>("full, DMA spread (barrier)", d_blob, d_bias, nl, d_x, d_out, items, d_cyc, blocks);
        CK(hipMemcpy(A.data(), d_out + (size_t)17 * 128 * 256, A.size() * 4, hipMemcpyDeviceToHost));
        run_proto<F_ALL | F_FLAGS>("full, ownership + flags (no barrier)", d_blob, d_bias, nl, d_x, d_out, items, d_cyc, blocks);
        CK(hipMemcpy(Bv.data(), d_out + (size_t)17 * 128 * 256, Bv.size() * 4, hipMemcpyDeviceToHost));
        printf("flag protocol outputs == barrier outputs (workgroup 17): %s\n", memcmp(A.data(), Bv.data(), A.size() * 4) ? "NO" : "yes");
        run_proto<(F_ALL & ~F_CONV) | F_FLAGS>("no conv, ownership + flags", d_blob, d_bias, nl, d_x, d_out, items, d_cyc, blocks);
        // (measured: correct, and 2 - 3 x SLOWER than the barrier -- 93 .. 120 cycles per MFMA against 45, with the fetch
        // spread over all four waves as here, 140 .. 260 with one wave fetching whole slots: a wave's LDS-DMAs complete one
        // after the other, and every poll's wait sits behind them.  The barrier stays.)
    }
    if (argc > 1) return 0;
    // ---- 3. shadow
    run_shadow<0, 8>(d_out, d_cyc);
    run_shadow<0, 4>(d_out, d_cyc);
    run_shadow<0, 2>(d_out, d_cyc);
    run_shadow<0, 1>(d_out, d_cyc);
    run_shadow<2, 8>(d_out, d_cyc);
    run_shadow<4, 8>(d_out, d_cyc);
    run_shadow<6, 8>(d_out, d_cyc);
    run_shadow<8, 8>(d_out, d_cyc);
    run_shadow<12, 8>(d_out, d_cyc);
    run_shadow<4, 2>(d_out, d_cyc);
    // ---- 4. L1 streaming
    {
        const int steps = 120 * nl * 16;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        l1_kernel<<<blocks, 256>>>(d_blob, nl * 16, d_out, 64, d_cyc);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        l1_kernel<<<blocks, 256>>>(d_blob, nl * 16, d_out, steps, d_cyc);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long cyc;
        CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
        printf("L1 streaming (4 waves x whole blob): %8.3f ms, %6.2f cyc/MFMA, %7.1f exec TF/s, %5.1f B/clk/CU from L1\n", ms,
               (double)cyc / (steps * 48.0), 256.0 * 4 * steps * 48.0 * 32768.0 / ms * 1e-9, 4.0 * steps * 24576.0 / (double)cyc);
    }
    return 0;
}
