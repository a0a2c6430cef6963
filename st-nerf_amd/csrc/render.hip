// Compositor (density edits + per-layer composite + cross-layer depth merge + merged composite)
// and inverse-CDF resampler.  One WAVE (64 lanes) owns one ray (compositor) or one (ray, layer)
// pair (resampler); the ray's samples are staged once through LDS with coalesced loads and every
// cumulative quantity (transmittance product, cdf) is a wavefront prefix scan.
//
// HBM-bound: compositor reads 20 B/sample (+1 B/layer mask) and writes 20 B per (ray, output);
// resampler reads 8 B per coarse sample and writes 16 B per fine sample (t + xyz).
//
// Reference: layers/render_layer.py:8-58, utils/sample_pdf.py:18-63,
//            modeling/layered_rfrender.py:414-475, :538-606.
#include "common.h"
#include <cstdlib>
#include <cstring>

// Occupancy targets (waves per SIMD) of the three kernels: the VGPR budget follows from them (512 / waves).
#ifndef STNERF_WAVES_COMPOSITE
#define STNERF_WAVES_COMPOSITE 6
#endif
#ifndef STNERF_WAVES_SINGLE
#define STNERF_WAVES_SINGLE 6
#endif
#ifndef STNERF_WAVES_RESAMPLE
#define STNERF_WAVES_RESAMPLE 8
#endif

namespace stnerf {

// Optional per-phase cycle accounting of composite_kernel (development builds: -DSTNERF_COMP_PROF): every wave adds its
// s_memtime deltas per phase; read back with stnerf_debug_composite_phases().
#ifdef STNERF_COMP_PROF
static __device__ unsigned long long g_cphase[8];
#define CP_DECL unsigned long long cp_t = clock64(), cp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define CP(i) do { const unsigned long long n_ = clock64(); cp_acc[i] += n_ - cp_t; cp_t = n_; } while (0)
#define CP_FLUSH do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_cphase[i_], cp_acc[i_]); } while (0)
#else
#define CP_DECL
#define CP(i) do { } while (0)
#define CP_FLUSH do { } while (0)
#endif

// ---- wave64 cross-lane primitives on DPP (gfx9 row_shr / row_bcast / wave_shr controls: one VALU op per
// scan step, no LDS crossbar traffic; ds_bpermute-based __shfl_up costs ~5 instructions per step).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_move(float identity, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143, DPP_WAVE_SHR1 = 0x138;

// The fp32 scans run IN PLACE: `v_mul_f32_dpp v, v, v row_shr:1` multiplies every lane whose source lane exists by that lane's
// value and leaves the others untouched (bound_ctrl off: a lane without a source is not executed) -- one vector
// instruction per step.  Written through the update_dpp builtin the same step is three (identity into a scratch
// register, DPP move over it, multiply), and the compositor is bound by its vector-instruction count.
// The two wait states a DPP read needs behind the write of its source are the s_nop 1 in front of every step.  The FIRST
// step of a block waits five: LLVM's hazard recognizer does not look inside inline asm, so a VALU write of EXEC (v_cmpx of
// the predicated code these scans are called behind) directly in front of the block would otherwise leave the
// "VALU writes EXEC -> DPP" hazard (5 wait states) uncovered.  tests/test_kernel_resources.py scans the ISA of every
// other DPP instruction of this file for the same hazard.
#define STNERF_DPP_STEP(op, ctrl) "s_nop 1\n\t" op " %0, %0, %0 " ctrl "\n\t"
#define STNERF_DPP_FIRST(op, ctrl) "s_nop 4\n\t" op " %0, %0, %0 " ctrl "\n\t"
__device__ __forceinline__ float wave_scan_mul(float v) {  // inclusive
    asm(STNERF_DPP_FIRST("v_mul_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf")
        STNERF_DPP_STEP("v_mul_f32_dpp", "row_shr:2 row_mask:0xf bank_mask:0xf")
        STNERF_DPP_STEP("v_mul_f32_dpp", "row_shr:4 row_mask:0xf bank_mask:0xf")
        STNERF_DPP_STEP("v_mul_f32_dpp", "row_shr:8 row_mask:0xf bank_mask:0xf")
        STNERF_DPP_STEP("v_mul_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf")
        STNERF_DPP_STEP("v_mul_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "+v"(v));
    return v;
}

__device__ __forceinline__ float wave_scan_add(float v) {  // inclusive
    asm(STNERF_DPP_FIRST("v_add_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf")
        STNERF_DPP_STEP("v_add_f32_dpp", "row_shr:2 row_mask:0xf bank_mask:0xf")
        STNERF_DPP_STEP("v_add_f32_dpp", "row_shr:4 row_mask:0xf bank_mask:0xf")
        STNERF_DPP_STEP("v_add_f32_dpp", "row_shr:8 row_mask:0xf bank_mask:0xf")
        STNERF_DPP_STEP("v_add_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf")
        STNERF_DPP_STEP("v_add_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "+v"(v));
    return v;
}

// Inclusive add-scan in fp64 (cdf accumulation of the resampler, see resample_kernel): the same DPP ladder as the fp32
// scans, moving the two halves of the double separately (2 DPP moves + one v_add_f64 per step; a __shfl_up of a double
// is two ds_bpermute round trips per step).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_move_f64(double v) {  // lanes the control leaves out receive +0.0
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_scan_add_f64(double v, int lane) {
    (void)lane;
    v += dpp_move_f64<DPP_ROW_SHR1>(v);
    v += dpp_move_f64<DPP_ROW_SHR2>(v);
    v += dpp_move_f64<DPP_ROW_SHR4>(v);
    v += dpp_move_f64<DPP_ROW_SHR8>(v);
    v += dpp_move_f64<DPP_ROW_BCAST15, 0xa>(v);
    v += dpp_move_f64<DPP_ROW_BCAST31, 0xc>(v);
    return v;
}

// torch.sum(x, -1) of one contiguous fp32 row, bit for bit as ATen computes it on a CPU: the reference's
// `torch.sum(weights, -1, keepdim=True)` (utils/sample_pdf.py:22) is an fp32 reduction whose rounding depends on the
// order, so "the reference's value" is defined by ATen's kernel (aten/src/ATen/native/cpu/SumKernel.cpp, dispatched
// with 8-float vectors on every x86 capability level -- DEFAULT, AVX2 and AVX512 alike, checked against torch 2.10 in
// tests/test_oracle_golden.py::test_aten_sum_order): rows of >= 8 elements go through vectorized_inner_sum = 8 vector
// lanes x 4 interleaved accumulators over the full vectors (row_sum / multi_row_sum, with its 16-row cascade level),
// the accumulators folded ((a0+a1)+a2)+a3, then a scalar chain over the < 8 leftover elements followed by the 8
// lanes in order; shorter rows take the scalar row_sum.  x lives in (wave-private) LDS; every lane returns the sum.
__device__ __forceinline__ float aten_cpu_row_sum(const float* x, int n, int lane) {
    if (n < 8) {
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
        const int rows = n >> 2;
        for (int i = 0; i < rows; ++i) {
            p0 += x[4 * i + 0];
            p1 += x[4 * i + 1];
            p2 += x[4 * i + 2];
            p3 += x[4 * i + 3];
        }
        for (int j = rows * 4; j < n; ++j) p0 += x[j];
        return ((p0 + p1) + p2) + p3;
    }
    const int nv = n >> 3, rows = nv >> 2, j = lane & 7;  // lanes 8.. repeat column lane & 7 (uniform control flow)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    int i = 0;
#pragma unroll 1
    for (; i + 16 <= rows;) {  // cascade level 1: every 16 rows the running accumulators are folded away
#pragma unroll 1
        for (int r = 0; r < 16; ++r, ++i) {
            a0 += x[(4 * i + 0) * 8 + j];
            a1 += x[(4 * i + 1) * 8 + j];
            a2 += x[(4 * i + 2) * 8 + j];
            a3 += x[(4 * i + 3) * 8 + j];
        }
        b0 += a0; b1 += a1; b2 += a2; b3 += a3;
        a0 = a1 = a2 = a3 = 0.f;
    }
#pragma unroll 1
    for (; i < rows; ++i) {
        a0 += x[(4 * i + 0) * 8 + j];
        a1 += x[(4 * i + 1) * 8 + j];
        a2 += x[(4 * i + 2) * 8 + j];
        a3 += x[(4 * i + 3) * 8 + j];
    }
    a0 += b0; a1 += b1; a2 += b2; a3 += b3;  // (levels 2 and 3 stay zero below 256 rows = 8192 elements)
#pragma unroll 1
    for (int v = rows * 4; v < nv; ++v) a0 += x[v * 8 + j];
    const float col = ((a0 + a1) + a2) + a3;
    float fin = 0.f;
#pragma unroll 1
    for (int k = nv * 8; k < n; ++k) fin += x[k];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) fin += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(col), jj));
    return fin;
}

// value of the previous lane (lane 0 gets `first`)
__device__ __forceinline__ float wave_prev(float v, float first) { return dpp_move<DPP_WAVE_SHR1>(first, v); }
__device__ __forceinline__ float wave_last(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }

// Branch-free binary lifting over an ascending LDS array: #{x in a[0..n) : x < v} / #{x <= v}.
// `p2` = largest power of two <= n (wave-uniform).
__device__ __forceinline__ int lower_bound_lds(const float* a, int n, int p2, float v) {
    int pos = 0;
    for (int step = p2; step > 0; step >>= 1) {
        const int np = pos + step;
        const float x = a[(np < n ? np : n) - 1];
        pos = (np <= n && x < v) ? np : pos;
    }
    return pos;
}
__device__ __forceinline__ int upper_bound_lds(const float* a, int n, int p2, float v) {
    int pos = 0;
    for (int step = p2; step > 0; step >>= 1) {
        const int np = pos + step;
        const float x = a[(np < n ? np : n) - 1];
        pos = (np <= n && x <= v) ? np : pos;
    }
    return pos;
}
// The same two counts over a strictly DESCENDING array, read back to front (a[n-1-i] is ascending).
__device__ __forceinline__ int lower_bound_lds_rev(const float* a, int n, int p2, float v) {
    int pos = 0;
    for (int step = p2; step > 0; step >>= 1) {
        const int np = pos + step;
        const float x = a[n - (np < n ? np : n)];
        pos = (np <= n && x < v) ? np : pos;
    }
    return pos;
}
__device__ __forceinline__ int upper_bound_lds_rev(const float* a, int n, int p2, float v) {
    int pos = 0;
    for (int step = p2; step > 0; step >>= 1) {
        const int np = pos + step;
        const float x = a[n - (np < n ? np : n)];
        pos = (np <= n && x <= v) ? np : pos;
    }
    return pos;
}
__host__ __device__ __forceinline__ int floor_pow2(int n) {
    int p = 1;
    while (p * 2 <= n) p *= 2;
    return p;
}

// Every LDS region below is private to one wave, so phases only need ordering inside the wave: LDS operations of
// a wave execute in issue order, the fences stop the compiler from moving accesses across the phase boundary.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// exp(-x) on the hardware exponential: v_exp_f32 is 2^y to 1 ulp; the scaling by log2(e) adds |x| * 2^-24 of relative
// error, which only matters where exp(-x) has already left the fp32 range of 1 - exp(-x).  ocml's expf costs ~3x the
// issue slots and this kernel is bound by them.
__device__ __forceinline__ float exp_neg(float x) { return __builtin_amdgcn_exp2f(x * -1.44269504088896340736f); }
// torch.sigmoid: 1/(1+exp(-x)), 1-ulp exponential and 1-ulp reciprocal
__device__ __forceinline__ float sigmoidf(float x) { return __builtin_amdgcn_rcpf(1.f + exp_neg(x)); }

// ---------------------------------------------------------------------------------------------
// Alpha-composite `count` samples read through accessor functors, in index order.
// gen_weight + VolumeRenderer.forward: layers/render_layer.py:8-17, :37-49.
//   delta_k = t_{k+1}-t_k, last = border;  alpha = 1-exp(-relu(sigma) delta);
//   T_k = prod_{j<k} (1-alpha_j+1e-10);  w = alpha T;  color = sum w sigmoid(rgb); depth = sum w t.
// raw_at(k) returns {sigmoid(r), sigmoid(g), sigmoid(b), sigma}: the sigmoid is evaluated once per sample when
// the ray is staged and shared by the per-layer and the merged composite (3 of the 8 expf per sample saved).
// ---------------------------------------------------------------------------------------------
// Returns bit 0: some t_{k+1} < t_k (the list is not ascending); bit 1: some t_{k+1} >= t_k (it is not strictly
// descending) -- per lane, the caller reduces over the wave.
template <class TAt, class RawAt, class WOut>
__device__ __forceinline__ int composite_run(int count, float border, int lane, TAt t_at, RawAt raw_at, WOut w_out,
                                             float (&out)[5]) {
    float carry = 1.f;
    float cr = 0.f, cg = 0.f, cb = 0.f, cd = 0.f, ca = 0.f;
    bool descending = false, not_descending = false;
    for (int base = 0; base < count; base += 64) {
        const int k = base + lane;
        const bool ok = k < count;
        float tr = 1.f, alpha = 0.f, tk = 0.f;
        float4 rw = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            tk = t_at(k);
            rw = raw_at(k);
            float delta = border;
            if (k + 1 < count) {
                const float tn = t_at(k + 1);
                descending = descending || (tn < tk);
                not_descending = not_descending || !(tn < tk);
                delta = tn - tk;
            }
            alpha = 1.f - exp_neg(fmaxf(rw.w, 0.f) * delta);
            tr = (1.f - alpha) + 1e-10f;
        }
        const float incl = wave_scan_mul(tr);
        const float excl = wave_prev(incl, 1.f);
        const float w = alpha * (carry * excl);
        carry = carry * wave_last(incl);
        if (ok) {
            w_out(k, w);
            cr += w * rw.x;  // rw.xyz = sigmoid(raw rgb), applied once when the ray is staged
            cg += w * rw.y;
            cb += w * rw.z;
            cd += w * tk;
            ca += w;
        }
    }
    out[0] = wave_last(wave_scan_add(cr));
    out[1] = wave_last(wave_scan_add(cg));
    out[2] = wave_last(wave_scan_add(cb));
    out[3] = wave_last(wave_scan_add(cd));
    out[4] = wave_last(wave_scan_add(ca));
    return (descending ? 1 : 0) | (not_descending ? 2 : 0);
}

// The same composite with the whole layer in registers (block b of lane i = sample 64 b + i): the path of a ray with
// ONE live layer, which needs neither the LDS staging nor the merge.  cut_near: the merged stream's `t < near` cut of
// the fine stage (modeling/layered_rfrender.py:605).
template <int MAXB, class WOut>
__device__ __forceinline__ void composite_regs(int S, float border, int lane, const float (&tk)[MAXB], const float (&tn)[MAXB],
                                               const float4 (&rw)[MAXB], bool cut_near, float nearv, WOut w_out,
                                               float (&out)[5]) {
    float carry = 1.f;
    float cr = 0.f, cg = 0.f, cb = 0.f, cd = 0.f, ca = 0.f;
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
        if (b * 64 < S) {  // (uniform)
            const int k = b * 64 + lane;
            const bool ok = k < S;
            float tr = 1.f, alpha = 0.f;
            if (ok) {
                const float delta = (k + 1 < S) ? tn[b] - tk[b] : border;
                float sg = rw[b].w;
                if (cut_near && tk[b] < nearv) sg = 0.f;
                alpha = 1.f - exp_neg(fmaxf(sg, 0.f) * delta);
                tr = (1.f - alpha) + 1e-10f;
            }
            const float incl = wave_scan_mul(tr);
            const float excl = wave_prev(incl, 1.f);
            const float w = alpha * (carry * excl);
            carry = carry * wave_last(incl);
            if (ok) {
                w_out(k, w);
                cr += w * rw[b].x;
                cg += w * rw[b].y;
                cb += w * rw[b].z;
                cd += w * tk[b];
                ca += w;
            }
        }
    }
    out[0] = wave_last(wave_scan_add(cr));
    out[1] = wave_last(wave_scan_add(cg));
    out[2] = wave_last(wave_scan_add(cb));
    out[3] = wave_last(wave_scan_add(cd));
    out[4] = wave_last(wave_scan_add(ca));
}

// gen_weight stand-alone: one wave per row.
__global__ void gen_weight_kernel(const float* __restrict__ sigma, const float* __restrict__ delta, int64_t n, int S,
                                  float* __restrict__ weights) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    float carry = 1.f;
    for (int base = 0; base < S; base += 64) {
        const int k = base + lane;
        float tr = 1.f, alpha = 0.f;
        if (k < S) {
            alpha = 1.f - exp_neg(fmaxf(sigma[row * S + k], 0.f) * delta[row * S + k]);
            tr = (1.f - alpha) + 1e-10f;
        }
        const float incl = wave_scan_mul(tr);
        const float excl = wave_prev(incl, 1.f);
        if (k < S) weights[row * S + k] = alpha * (carry * excl);
        carry = carry * wave_last(incl);
    }
}

struct CompositeArgs {
    const float* t;
    const float4* raw;
    const uint8_t* mask;
    int64_t n;
    int l, S;
    stnerf_composite_params p;
    float* layer_out;
    float* mixed_out;
    float* weights;
    int32_t* order;
    int waves_per_block;
    int p2;  // floor_pow2(S)
    uint8_t* handled;  // [n] or nullptr: rays composite_single_kernel has already finished (it writes 0 / 1 for every ray)
    int lds_layers;    // composite_merge_kernel: layers its merged list holds (rays with more live layers are left to the next launch)
};

// The LDS-staged compositor: every ray on its own, the whole ray (all l * S samples) in LDS, rank merge by binary
// searches.  Since round 3 it serves the `order` parity output and layers of more than 192 samples only; production
// calls take composite_single_kernel + composite_merge_kernel below (same numbers, bit for bit).
__global__ void __attribute__((amdgpu_waves_per_eu(STNERF_WAVES_COMPOSITE, 8))) composite_kernel(CompositeArgs a) {
    constexpr bool FASTPATH = true;  // rays with one live layer are composited from registers
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (uniform: ray index and addresses on the scalar unit)
    const int LS = a.l * a.S;
    // per-wave LDS: raws[LS] float4 | ts[LS] float | mord[LS] u16 (merged position -> source sample), 16-B rounded
    const int per_wave = ((LS * 22 + 15) / 16) * 16;
    unsigned char* mine = smem_raw + (size_t)wave * per_wave;
    float4* raws = reinterpret_cast<float4*>(mine);
    float* ts = reinterpret_cast<float*>(mine + (size_t)LS * 16);
    unsigned short* mord = reinterpret_cast<unsigned short*>(ts + LS);

    const int64_t rays_per_iter = (int64_t)gridDim.x * a.waves_per_block;
    // hit mask (lanes 0 .. l-1) and `handled` flag (lane 63) of a ray in ONE load each, fetched one ray ahead: the
    // chain "flags -> which layers -> their samples" is otherwise two or more dependent HBM round trips per ray
    auto ray_flags = [&](int64_t ray) -> int {
        int v = 0;
        if (ray < a.n) {
            if (lane < a.l && a.mask) v = a.mask[ray * a.l + lane];
            if (lane == 63 && a.handled) v = a.handled[ray];
        }
        return v;
    };
    int flags_next = ray_flags((int64_t)blockIdx.x * a.waves_per_block + wave);
    CP_DECL
    for (int64_t ray0 = (int64_t)blockIdx.x * a.waves_per_block; ray0 < a.n; ray0 += rays_per_iter) {
        const int64_t ray = ray0 + wave;
        const unsigned long long fb = __ballot((flags_next & 1) != 0 || (lane == 63u && flags_next != 0));
        const unsigned miss_bits = (unsigned)__ballot((flags_next & 2) != 0) & 0xffffu;   // (hint of the sampler: every depth is -1000)
        flags_next = ray_flags(ray + rays_per_iter);
        const unsigned mask_bits = (unsigned)fb;
        const bool active = ray < a.n && !(fb >> 63 & 1ull);
        int n_merged = 0;
        CP(0);
        // ---- stage the ray, applying the post-network density edits (a10); layer-major so every edit
        // switch is wave-uniform.
        // A layer the ray misses altogether (not evaluated, every t == -1000: bin width 0 from start = end = -1000,
        // layers/RaySamplePoint.py:53-62,98-102) is dropped from everything below: its samples have sigma = 0, so
        // alpha = 0, w = 0 and the transmittance factor fl(1 - 0 + 1e-10) is exactly 1; they sort in front of every
        // real sample, so they are nobody's successor and change no delta.  The composites are the same numbers (the
        // dropped factors are exact ones; only the association order of the parallel transmittance scan moves with the
        // lane a sample lands in, i.e. fp32 rounding), and with performers covering a fraction of the image most rays
        // carry one or two live layers instead of l.  The `order` parity output does not change what is composited
        // (tests/test_gpu_ops.py::test_composite_production_shortcuts_are_bitwise_neutral).
        // (A not-evaluated layer with real depths -- hidden, or a grazing hit -- still takes part: its depths
        // shape its neighbours' deltas.)
        unsigned live = 0, have_m = 0;  // bit i: layer i takes part / has network output on this ray
        if (active) {
            const float* tsrc = a.t + ray * LS;
            for (int layer = 0; layer < a.l; ++layer) {
                // evaluated: 0 = no network output for this layer (hidden), 1 = on the rays its hit mask marks,
                // 2 = on every ray whatever the mask says -- the background: bkgd_spacenet runs on all rays and its
                // output is composited even where ray_mask[0] is False (a ray through an edge of the background
                // box: start == end, bin width 0; layered_rfrender.py:382-392,435-444, fixture fwd_grazing)
                const int ev = a.p.evaluated[layer];
                const bool have = ev == 2 || (ev != 0 && (!a.mask || (mask_bits >> layer & 1u)));
                bool lv = have;
                if (!have && !(miss_bits >> layer & 1u)) {  // without output a layer still takes part if it has real depths (hidden, or a grazing hit)
                    bool missed = true;
                    for (int k = lane; k < a.S; k += 64) missed = missed && tsrc[layer * a.S + k] == -1000.f;
                    lv = !__all(missed);
                }
                if (lv) live |= 1u << layer;
                if (have) have_m |= 1u << layer;
            }
        }
        CP(1);
        // ---- ONE live layer (about half the rays of a view: the background alone): composite it straight from
        // registers -- no LDS staging, no merge; the mix is that layer's composite (same samples, deltas, arithmetic)
        // unless the fine stage's `t < near` cut bites, which costs a second pass over the registers.
        bool done = false;
        constexpr int MAXB = 3;
        if (FASTPATH && active && __popc(live) == 1 && a.S <= 64 * MAXB) {
            const int layer = __ffs(live) - 1;
            const bool have = (have_m >> layer & 1u) != 0;
            const bool cut_neg = !a.p.fine && a.p.cut_negative_t && layer > 0;
            const bool cut_near = !a.p.fine && layer == 0;
            const bool use_thr = a.p.use_threshold[layer] != 0;
            const float thr = a.p.threshold[layer], sscale = a.p.sigma_scale[layer], nearv = a.p.near;
            const float* tl = a.t + ray * LS + layer * a.S;
            const float4* rl = a.raw + ray * LS + layer * a.S;
            float tk[MAXB], tn[MAXB];
            float4 rw[MAXB];
            bool desc = false;
#pragma unroll
            for (int b = 0; b < MAXB; ++b) {
                const int k = b * 64 + lane;
                tk[b] = tn[b] = 0.f;
                rw[b] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < a.S) {
                    tk[b] = tl[k];
                    if (k + 1 < a.S) {
                        tn[b] = tl[k + 1];
                        desc = desc || (tn[b] < tk[b]);
                    }
                    if (have) {
                        float4 v = rl[k];
                        if (cut_neg && tk[b] < 0.f) v.w = 0.f;
                        if (use_thr && v.w < thr) v.w = 0.f;
                        v.w = v.w * sscale;
                        if (cut_near && tk[b] < nearv) v.w = 0.f;
                        if (!a.p.rgb_activated) {
                            v.x = sigmoidf(v.x);
                            v.y = sigmoidf(v.y);
                            v.z = sigmoidf(v.z);
                        }
                        rw[b] = v;
                    }
                }
            }
            if (!__any(desc)) {  // (a descending list needs the merge to turn it round: general path)
                float* wdst = a.weights ? a.weights + (ray * a.l + layer) * a.S : nullptr;
                float o5[5];
                composite_regs<MAXB>(a.S, a.p.border, lane, tk, tn, rw, false, 0.f, [&](int k, float w) { if (wdst) wdst[k] = w; }, o5);
                for (int other = 0; other < a.l; ++other) {  // the layers the ray misses: zero weights and outputs
                    if (other == layer) continue;
                    if (a.weights)
                        for (int k = lane; k < a.S; k += 64) a.weights[(ray * a.l + other) * a.S + k] = 0.f;
                    if (a.layer_out && lane < 5) a.layer_out[(ray * a.l + other) * 5 + lane] = 0.f;
                }
                auto pick = [&](const float (&o)[5]) { return lane == 0 ? o[0] : lane == 1 ? o[1] : lane == 2 ? o[2] : lane == 3 ? o[3] : o[4]; };
                if (a.layer_out && lane < 5) a.layer_out[(ray * a.l + layer) * 5 + lane] = pick(o5);
                if (a.mixed_out) {
                    const float t_first = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tk[0]), 0));
                    if (a.p.fine && t_first < nearv) {
                        float m5[5];
                        composite_regs<MAXB>(a.S, a.p.border, lane, tk, tn, rw, true, nearv, [&](int, float) {}, m5);
                        if (lane < 5) a.mixed_out[ray * 5 + lane] = pick(m5);
                    } else if (lane < 5) {
                        a.mixed_out[ray * 5 + lane] = pick(o5);
                    }
                }
                if (a.order) {  // ascending single layer + leading -1000 samples of the others: computed below
                    const float* tsrc = a.t + ray * LS;
                    for (int e = lane; e < LS; e += 64) ts[e] = tsrc[e];
                }
                done = true;
            }
        }
        CP(2);
        // ---- general path: stage the ray in LDS, applying the post-network density edits (a10); layer-major so
        // every edit switch is wave-uniform.
        if (active && !done) {
            const float* tsrc = a.t + ray * LS;
            const float4* rsrc = a.raw + ray * LS;
            for (int layer = 0; layer < a.l; ++layer) {
                const bool have = (have_m >> layer & 1u) != 0;
                const bool cut_neg = !a.p.fine && a.p.cut_negative_t && layer > 0;             // :414
                const bool cut_near = !a.p.fine && layer == 0;                                 // :422
                const bool use_thr = a.p.use_threshold[layer] != 0;                            // :416-418, :538-547, :564-566
                const float thr = a.p.threshold[layer], sscale = a.p.sigma_scale[layer], nearv = a.p.near;
                // every load of the layer is issued before the first one is consumed: written as a plain
                // load -> edit -> store loop each 64-sample block costs its own HBM round trip
                constexpr int SB = 3;
                for (int k0 = 0; k0 < a.S; k0 += 64 * SB) {
                    float tv[SB];
                    float4 rv[SB];
#pragma unroll
                    for (int b = 0; b < SB; ++b) {
                        const int k = k0 + b * 64 + lane;
                        tv[b] = 0.f;
                        rv[b] = make_float4(0.f, 0.f, 0.f, 0.f);  // zero tensors (:398-399); sigma = 0 makes the colour moot
                        if (k < a.S) {
                            tv[b] = tsrc[layer * a.S + k];
                            if (have) rv[b] = rsrc[layer * a.S + k];
                        }
                    }
#pragma unroll
                    for (int b = 0; b < SB; ++b) {
                        const int k = k0 + b * 64 + lane;
                        if (k < a.S) {
                            const int e = layer * a.S + k;
                            float4 rw = rv[b];
                            ts[e] = tv[b];
                            if (have) {
                                if (cut_neg && tv[b] < 0.f) rw.w = 0.f;
                                if (use_thr && rw.w < thr) rw.w = 0.f;
                                rw.w = rw.w * sscale;                                      // :575-576
                                if (cut_near && tv[b] < nearv) rw.w = 0.f;
                                if (!a.p.rgb_activated) {
                                    rw.x = sigmoidf(rw.x);  // torch.sigmoid(rgb), render_layer.py:47
                                    rw.y = sigmoidf(rw.y);
                                    rw.z = sigmoidf(rw.z);
                                }
                            }
                            raws[e] = rw;
                        }
                    }
                }
            }
        }
        wave_sync();
        CP(3);
        // ---- per-layer composites (:435-444 / :598-603)
        bool merged_done = false, unsorted_any = false;
        unsigned reversed_all = 0;
        if (active && !done) {
            // a layer's list is ascending, unless its bin width is negative: a box edit, or a ray that misses the
            // background box (far = -1000, start clamped to 0: depths run from 0 down to -1000).  Such a list is strictly
            // descending and is merged through a reversed view; anything else (ties inside a descending list) takes
            // the general rank.
            bool unsorted = false;
            unsigned reversed = 0;  // bit i: layer i is strictly descending
            const bool single = __popc(live) == 1;  // one live layer: the union IS that layer
            float single5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            for (int layer = 0; layer < a.l; ++layer) {
                float* wdst = a.weights ? a.weights + (ray * a.l + layer) * a.S : nullptr;
                if (!(live >> layer & 1u)) {  // missed: every weight and every composite output is zero
                    if (wdst)
                        for (int k = lane; k < a.S; k += 64) wdst[k] = 0.f;
                    if (a.layer_out && lane < 5) a.layer_out[(ray * a.l + layer) * 5 + lane] = 0.f;
                    continue;
                }
                const float* tl = ts + layer * a.S;
                const float4* rl = raws + layer * a.S;
                float o5[5];
                const int dir = composite_run(a.S, a.p.border, lane, [&](int k) { return tl[k]; },
                                              [&](int k) { return rl[k]; },
                                              [&](int k, float w) { if (wdst) wdst[k] = w; }, o5);
                const bool some_desc = __any(dir & 1), some_asc = __any(dir & 2);
                if (some_desc && !some_asc) reversed |= 1u << layer;
                unsorted = unsorted || (some_desc && some_asc);
                if (a.layer_out && lane < 5) {
                    const float v = lane == 0 ? o5[0] : lane == 1 ? o5[1] : lane == 2 ? o5[2] : lane == 3 ? o5[3] : o5[4];
                    a.layer_out[(ray * a.l + layer) * 5 + lane] = v;
                }
                if (single) {
#pragma unroll
                    for (int c = 0; c < 5; ++c) single5[c] = o5[c];
                    // the merged composite differs from the layer's own only by the fine stage's `t < near` cut (:605)
                    merged_done = !a.p.fine || !(tl[0] < a.p.near);
                }
            }
            CP(4);
            const bool sorted_ok = !unsorted;  // (wave-uniform)
            unsorted_any = unsorted;
            reversed_all = reversed;
            merged_done = merged_done && sorted_ok && reversed == 0;
            if (merged_done) {  // same samples, same deltas, same arithmetic: the layer's composite is the mix
                if (a.mixed_out && lane < 5) {
                    const float v = lane == 0 ? single5[0] : lane == 1 ? single5[1] : lane == 2 ? single5[2]
                                  : lane == 3 ? single5[3] : single5[4];
                    a.mixed_out[ray * 5 + lane] = v;
                }
            } else if (sorted_ok) {
                // ---- cross-layer merge by depth (:425-429 / :587-592): rank of every live sample in the union.
                // Stable: ties resolve by source index (layer-major), the order a stable sort of the
                // concatenation gives.
                int before = 0;  // live samples of the layers in front of `la`
                for (int la = 0; la < a.l; ++la) {
                    if (!(live >> la & 1u)) continue;
                    for (int k = lane; k < a.S; k += 64) {
                        const int e = la * a.S + k;
                        const float v = ts[e];
                        int rank = (reversed >> la & 1u) ? a.S - 1 - k : k;
                        for (int lb = 0; lb < la; ++lb)
                            if (live >> lb & 1u)
                                rank += (reversed >> lb & 1u) ? upper_bound_lds_rev(ts + lb * a.S, a.S, a.p2, v)
                                                              : upper_bound_lds(ts + lb * a.S, a.S, a.p2, v);
                        for (int lb = la + 1; lb < a.l; ++lb)
                            if (live >> lb & 1u)
                                rank += (reversed >> lb & 1u) ? lower_bound_lds_rev(ts + lb * a.S, a.S, a.p2, v)
                                                              : lower_bound_lds(ts + lb * a.S, a.S, a.p2, v);
                        mord[rank] = (unsigned short)e;
                    }
                    before += a.S;
                }
                n_merged = before;
            } else {  // general O(n^2) fallback over every sample (missed layers included: they sort first, weight 0)
                for (int e = lane; e < LS; e += 64) {
                    const float v = ts[e];
                    int rank = 0;
                    for (int x = 0; x < LS; ++x) {
                        const float xv = ts[x];
                        rank += (xv < v || (xv == v && x < e)) ? 1 : 0;
                    }
                    mord[rank] = (unsigned short)e;
                }
                n_merged = LS;
            }
        }
        wave_sync();
        CP(5);
        // ---- merged composite (:448 / :605-606)
        if (active && !done && !merged_done && a.mixed_out) {
            float o5[5];
            const bool cut_near = a.p.fine != 0;
            const float nearv = a.p.near;
            composite_run(n_merged, a.p.border, lane, [&](int m) { return ts[mord[m]]; },
                          [&](int m) {
                              const int src = mord[m];
                              float4 rw = raws[src];
                              if (cut_near && ts[src] < nearv) rw.w = 0.f;  // :605
                              return rw;
                          },
                          [&](int, float) {}, o5);
            if (a.mixed_out && lane < 5) {
                const float v = lane == 0 ? o5[0] : lane == 1 ? o5[1] : lane == 2 ? o5[2] : lane == 3 ? o5[3] : o5[4];
                a.mixed_out[ray * 5 + lane] = v;
            }
        }
        CP(6);
        // ---- optional parity output: torch.sort's index over ALL l * S samples (the composites above leave the
        // layers a ray misses out; their samples, t = -1000, sort in front of everything and carry no weight)
        if (active && a.order) {
            int32_t* od = a.order + ray * LS;
            if (!unsorted_any) {
                for (int la = 0; la < a.l; ++la) {
                    for (int k = lane; k < a.S; k += 64) {
                        const int e = la * a.S + k;
                        const float v = ts[e];
                        int rank = (reversed_all >> la & 1u) ? a.S - 1 - k : k;
                        for (int lb = 0; lb < la; ++lb)
                            rank += (reversed_all >> lb & 1u) ? upper_bound_lds_rev(ts + lb * a.S, a.S, a.p2, v)
                                                              : upper_bound_lds(ts + lb * a.S, a.S, a.p2, v);
                        for (int lb = la + 1; lb < a.l; ++lb)
                            rank += (reversed_all >> lb & 1u) ? lower_bound_lds_rev(ts + lb * a.S, a.S, a.p2, v)
                                                              : lower_bound_lds(ts + lb * a.S, a.S, a.p2, v);
                        od[rank] = e;
                    }
                }
            } else {
                for (int e = lane; e < LS; e += 64) {
                    const float v = ts[e];
                    int rank = 0;
                    for (int x = 0; x < LS; ++x) {
                        const float xv = ts[x];
                        rank += (xv < v || (xv == v && x < e)) ? 1 : 0;
                    }
                    od[rank] = e;
                }
            }
        }
        wave_sync();
    }
    CP_FLUSH;
}

// ---------------------------------------------------------------------------------------------
// Rays with ONE live layer (about half the rays of a view: the background alone), software pipelined.
// composite_kernel spends ~10 us per ray on such a ray although it needs ~150 instructions: the chain
// "hit mask -> (which layer?) -> its samples -> composite -> store" is two dependent HBM round trips per ray with
// nothing else for the wave to do, and 8 waves per SIMD cannot hide that.  This kernel needs no LDS and runs the
// chain as a three-stage pipeline over the rays of a wave: while ray j is composited from registers, the samples of ray
// j+1 (whose mask arrived one iteration earlier) and the mask of ray j+2 are in flight.  It writes handled[ray] = 1
// for the rays it finishes and 0 for the others (several live layers, a descending list, a masked-out layer with real
// depths, ...), which composite_kernel then takes.  Same arithmetic, same lanes as composite_kernel: bit-identical.
// ---------------------------------------------------------------------------------------------
template <int MAXB, int MAXCHK>
struct SingleBuf {
    float tk[MAXB], tn[MAXB], chk[MAXCHK];
    float4 rw[MAXB];
    int layer;
    bool eligible;
};

template <int MAXB, int MAXCHK>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MAXB > 2 ? 4 : STNERF_WAVES_SINGLE, 8))) composite_single_kernel(CompositeArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t first = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int LS = a.l * a.S, B = (a.S + 63) >> 6;
    unsigned ev1 = 0, ev2 = 0;
    for (int i = 0; i < a.l; ++i) {
        if (a.p.evaluated[i] == 2) ev2 |= 1u << i;
        else if (a.p.evaluated[i] != 0) ev1 |= 1u << i;
    }
    using Buf = SingleBuf<MAXB, MAXCHK>;
    auto mask_lane = [&](int64_t ray) -> int { return (a.mask && ray < a.n && lane < a.l) ? (int)a.mask[ray * a.l + lane] : 0; };
    auto have_of = [&](int mv) -> unsigned {
        const unsigned mb = (unsigned)__ballot((mv & 1) != 0);
        return ev2 | (ev1 & (a.mask ? mb : ~0u));
    };
    auto miss_of = [&](int mv) -> unsigned { return (unsigned)__ballot((mv & 2) != 0); };   // (the sampler's hint: every depth -1000)
    const unsigned all_layers = a.l >= 32 ? ~0u : (1u << a.l) - 1u;
    auto issue = [&](Buf& b, int64_t ray, unsigned have, unsigned miss) {
        b.eligible = ray < a.n && __popc(have) == 1;
        b.layer = b.eligible ? __ffs(have) - 1 : 0;
#pragma unroll
        for (int i = 0; i < MAXB; ++i) {
            b.tk[i] = b.tn[i] = 0.f;
            b.rw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int c = 0; c < MAXCHK; ++c) b.chk[c] = -1000.f;
        if (b.eligible) {
            const float* tsrc = a.t + ray * LS;
            const float* tl = tsrc + b.layer * a.S;
            const float4* rl = a.raw + ray * LS + b.layer * a.S;
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                const int k = i * 64 + lane;
                if (k < a.S) {
                    b.tk[i] = tl[k];
                    if (k + 1 < a.S) b.tn[i] = tl[k + 1];
                    b.rw[i] = rl[k];
                }
            }
            const unsigned others = all_layers & ~(1u << b.layer);
            if ((miss & others) != others) {   // (uniform) not every other layer carries the sampler's "missed" hint: look
#pragma unroll
                for (int c = 0; c < MAXCHK; ++c) {  // depths of the other layers: "missed" means every one of them is -1000
                    const int oi = c / B, blk = c - oi * B;
                    const int x = oi < b.layer ? oi : oi + 1;
                    const int k = blk * 64 + lane;
                    if (oi < a.l - 1 && k < a.S) b.chk[c] = tsrc[x * a.S + k];
                }
            }
        }
    };
    int64_t r0 = first, r1 = first + stride, r2 = first + 2 * stride;
    Buf cur, nxt;
    int m1;
    {
        const int m0 = mask_lane(r0);
        m1 = mask_lane(r1);
        issue(cur, r0, have_of(m0), miss_of(m0));
    }
    for (; r0 < a.n; r0 = r1, r1 = r2, r2 += stride) {
        const unsigned have1 = have_of(m1), miss1 = miss_of(m1);
        const int m2 = mask_lane(r2);
        issue(nxt, r1, have1, miss1);
        __builtin_amdgcn_sched_barrier(0);  // keep the loads of the next ray ahead of this ray's arithmetic
        // ---- ray r0 from `cur`
        bool ok = cur.eligible;
        if (ok) {
            bool others_missed = true, desc = false;
#pragma unroll
            for (int c = 0; c < MAXCHK; ++c) others_missed = others_missed && cur.chk[c] == -1000.f;
#pragma unroll
            for (int i = 0; i < MAXB; ++i) desc = desc || (i * 64 + lane + 1 < a.S && cur.tn[i] < cur.tk[i]);
            ok = __all(others_missed) && !__any(desc);
        }
        if (ok) {
            const int layer = cur.layer;
            const bool cut_neg = !a.p.fine && a.p.cut_negative_t && layer > 0;
            const bool cut_near = !a.p.fine && layer == 0;
            const bool use_thr = a.p.use_threshold[layer] != 0;
            const float thr = a.p.threshold[layer], sscale = a.p.sigma_scale[layer], nearv = a.p.near;
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                float4 v = cur.rw[i];
                if (cut_neg && cur.tk[i] < 0.f) v.w = 0.f;
                if (use_thr && v.w < thr) v.w = 0.f;
                v.w = v.w * sscale;
                if (cut_near && cur.tk[i] < nearv) v.w = 0.f;
                if (!a.p.rgb_activated) {
                    v.x = sigmoidf(v.x);
                    v.y = sigmoidf(v.y);
                    v.z = sigmoidf(v.z);
                }
                cur.rw[i] = v;
            }
            float* wdst = a.weights ? a.weights + (r0 * a.l + layer) * a.S : nullptr;
            float o5[5];
            composite_regs<MAXB>(a.S, a.p.border, lane, cur.tk, cur.tn, cur.rw, false, 0.f,
                                 [&](int k, float w) { if (wdst) wdst[k] = w; }, o5);
            for (int other = 0; other < a.l; ++other) {  // the layers the ray misses: zero weights and outputs
                if (other == layer) continue;
                if (a.weights)
                    for (int k = lane; k < a.S; k += 64) a.weights[(r0 * a.l + other) * a.S + k] = 0.f;
                if (a.layer_out && lane < 5) a.layer_out[(r0 * a.l + other) * 5 + lane] = 0.f;
            }
            auto pick = [&](const float (&o)[5]) { return lane == 0 ? o[0] : lane == 1 ? o[1] : lane == 2 ? o[2] : lane == 3 ? o[3] : o[4]; };
            if (a.layer_out && lane < 5) a.layer_out[(r0 * a.l + layer) * 5 + lane] = pick(o5);
            if (a.mixed_out) {
                const float t_first = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur.tk[0]), 0));
                if (a.p.fine && t_first < nearv) {
                    float m5[5];
                    composite_regs<MAXB>(a.S, a.p.border, lane, cur.tk, cur.tn, cur.rw, true, nearv, [&](int, float) {}, m5);
                    if (lane < 5) a.mixed_out[r0 * 5 + lane] = pick(m5);
                } else if (lane < 5) {
                    a.mixed_out[r0 * 5 + lane] = pick(o5);
                }
            }
        }
        if (lane == 0) a.handled[r0] = ok ? 1 : 0;
        cur = nxt;
        m1 = m2;
    }
}

// ---------------------------------------------------------------------------------------------
// Rays with SEVERAL live layers (round 3): layers in registers, merge by insertion.
//
// composite_kernel stages the whole ray in LDS (22 B per sample: 8.4 KB per wave at 3 x 128 samples, 38 KB at 9 x 192 --
// one wave per SIMD) and ranks every sample against every other layer (log2 S dependent LDS reads per sample and layer:
// cost ~ l^2).  This kernel keeps a layer in registers while it is composited (the arithmetic of composite_regs, as the
// single-layer kernel above), and builds the merged order one layer at a time in a 6 B / sample LDS list
// (depth + source index):
//   * the samples of the NEW layer search the list merged so far (one upper_bound each: ties go behind the earlier
//     layers, the order of a stable sort of the concatenation) and mark their output slots in a bit mask (ds_or);
//   * every output slot then knows from the mask alone what it receives: bit set -> the next sample of the new layer,
//     clear -> list element (slot - set bits below it); the prefix count is s_bcnt1 + v_mbcnt on the mask words, no scan.
//     The list is rewritten in place from the top block down (a list element only ever moves up).
//   Search steps per ray: S log2(m) per inserted layer instead of S (l-1) log2(S) per layer -- 2.7 x fewer at l = 3,
//   10 x at l = 9 -- and the LDS footprint is 2.9 KB (3 x 128) / 11.4 KB (9 x 192) per wave.
//   * the merged composite gathers each sample's float4 from global memory by source index (the rows were read by this
//     wave a moment ago: L2 / L1 hits) and re-applies the layer's density edits per lane (two LDS table reads).
// Same merged order, same lanes, same arithmetic as composite_kernel: bit-identical outputs (the `order` parity call
// still takes composite_kernel; tests/test_gpu_ops.py::test_composite_production_shortcuts_are_bitwise_neutral compares
// the two).  A layer that is neither ascending nor strictly descending sends the ray to a brute-force rank (tests only).
// ---------------------------------------------------------------------------------------------
constexpr int DPP_WAVE_SHL1 = 0x130;
constexpr int MERGE_TAB_BYTES = 128;  // per workgroup: effective threshold[16] | sigma_scale[16]

__host__ __device__ __forceinline__ int64_t merge_lds_per_wave(int l, int S) {   // l: layers the merged list holds
    const int64_t LS = (int64_t)l * S, SP = (S + 63) / 64 * 64, words = (LS + 63) / 64 * 2;
    return ((4 * LS + 4 * SP + 4 * words + 2 * LS + 15) / 16) * 16;
}

// occupancy target of composite_merge_kernel (waves per SIMD -> 512 / n registers); the host sizes the LDS tiers with it
__host__ __device__ constexpr int merge_waves_per_simd(int maxb, bool full) { return !full ? 5 : maxb > 2 ? 6 : 7; }

template <int MAXB>
struct LayerRegs {
    float tk[MAXB];
    float4 rw[MAXB];
};

// Running state of one alpha-composite (gen_weight + VolumeRenderer.forward, see composite_run): the transmittance
// carried from block to block and the five weighted sums.
struct CompositeAcc {
    float carry = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, cd = 0.f, ca = 0.f;
};

// One block of 64 samples (lane = sample): weight of the lane's sample, sums updated.  The arithmetic, operation for
// operation, of composite_run / composite_regs (the kernels are compared bit for bit).  `ok`: the lane holds a sample.
template <bool ALL>
__device__ __forceinline__ float composite_block(CompositeAcc& A, float sigma, float delta, float r, float g, float b, float t, bool ok) {
    float alpha = 1.f - exp_neg(fmaxf(sigma, 0.f) * delta);
    float tr = (1.f - alpha) + 1e-10f;
    if (!ALL) {
        alpha = ok ? alpha : 0.f;
        tr = ok ? tr : 1.f;
    }
    const float incl = wave_scan_mul(tr);
    const float excl = wave_prev(incl, 1.f);
    float w = alpha * (A.carry * excl);
    A.carry = A.carry * wave_last(incl);
    if (!ALL) w = ok ? w : 0.f;   // (alpha = 0 does not make it zero when the transmittance has overflowed)
    A.cr += w * r;
    A.cg += w * g;
    A.cb += w * b;
    A.cd += w * t;
    A.ca += w;
    return w;
}

// the five sums over the wave, stored by lane 63 (which holds the totals of the in-place scans)
__device__ __forceinline__ void composite_store5(const CompositeAcc& A, float* dst, float* dst2, unsigned lane) {
    const float o0 = wave_scan_add(A.cr), o1 = wave_scan_add(A.cg), o2 = wave_scan_add(A.cb), o3 = wave_scan_add(A.cd),
                o4 = wave_scan_add(A.ca);
    if (lane == 63u) {
        if (dst) { dst[0] = o0; dst[1] = o1; dst[2] = o2; dst[3] = o3; dst[4] = o4; }
        if (dst2) { dst2[0] = o0; dst2[1] = o1; dst2[2] = o2; dst2[3] = o3; dst2[4] = o4; }
    }
}

// FULL: S == 64 * MAXB (64 / 128 / 192 samples per layer: every BASELINE configuration) -- no lane is ever idle, the
// `k < S` predicates and their exec-mask bookkeeping disappear.
template <int MAXB, bool FULL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(merge_waves_per_simd(MAXB, FULL), 8))) composite_merge_kernel(CompositeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned S = FULL ? 64u * MAXB : (unsigned)a.S, L = (unsigned)a.l, LS = L * S;
    const unsigned LC = (unsigned)a.lds_layers * S;   // capacity of the merged list (samples)
    float* tab = reinterpret_cast<float*>(smem_raw);
    unsigned ev1 = 0, ev2 = 0;
#pragma unroll
    for (int i = 0; i < STNERF_MAX_LAYERS; ++i) {
        if (i < a.l) {
            if (a.p.evaluated[i] == 2) ev2 |= 1u << i;
            else if (a.p.evaluated[i] != 0) ev1 |= 1u << i;
            if ((int)threadIdx.x == i) {
                tab[i] = a.p.use_threshold[i] != 0 ? a.p.threshold[i] : -INFINITY;   // (sigma < -inf never holds)
                tab[16 + i] = a.p.sigma_scale[i];
            }
        }
    }
    __syncthreads();
    unsigned char* mine = smem_raw + MERGE_TAB_BYTES + (size_t)wave * merge_lds_per_wave(a.lds_layers, (int)S);
    float* mkey = reinterpret_cast<float*>(mine);            // [LC] merged depths, ascending
    float* ckey = mkey + LC;                                  // [SP] the layer being inserted, ascending (ckey = mkey + LC is used below)
    unsigned* bits = reinterpret_cast<unsigned*>(ckey + (S + 63u) / 64u * 64u);   // [2 ceil(LC / 64)] output slots of the new layer
    unsigned short* mpay = reinterpret_cast<unsigned short*>(bits + (LC + 63u) / 64u * 2u);   // [LC] source sample of a list entry
    const unsigned lmask = L >= 32u ? ~0u : (1u << L) - 1u;
    const float invS = 1.f / (float)S, nearv = a.p.near, border = a.p.border;
    const bool fine = a.p.fine != 0, activated = a.p.rgb_activated != 0;
    auto ok_lane = [&](int b) -> bool { return FULL || (unsigned)b * 64u + lane < S; };

    // A wave takes the rays wave_id + k * (number of waves), k = 0, 1, ... (neighbouring rays -- same performers, same cost --
    // go to different waves), GR of them at a time: one round trip fetches the `handled` bytes and hit masks of the whole
    // group (lane i < GR: the group's ray i; its mask packed into one word), one group ahead of the one being worked on.
    // The rays composite_single_kernel has finished (about 60 % of a view) then cost nothing here -- taken one at a time,
    // every one of them is a dependent HBM round trip with nothing behind it.
    constexpr unsigned GR = 16;
    const int64_t nwaves = (int64_t)gridDim.x * a.waves_per_block, wave_id = (int64_t)blockIdx.x * a.waves_per_block + wave;
    auto group_flags = [&](int64_t k0, unsigned& mk) -> int {
        int hnd = 1;
        mk = 0u;
        const int64_t ray = wave_id + (k0 + lane) * nwaves;
        if (lane < GR && ray < a.n) {
            hnd = a.handled ? (int)a.handled[ray] : 0;
            if (a.mask) {
#pragma unroll
                for (int i = 0; i < STNERF_MAX_LAYERS; ++i)
                    if (i < a.l) {   // bit i: hit (the reference's ray_mask); bit 16 + i: the sampler's "missed" hint
                        const unsigned mv = a.mask[ray * a.l + i];
                        mk |= (mv & 1u) << i | (mv >> 1 & 1u) << (16 + i);
                    }
            }
        }
        return hnd;
    };
    using Regs = LayerRegs<MAXB>;
    unsigned mk_next = 0u;
    int hnd_next = group_flags(0, mk_next);
    CP_DECL
    for (int64_t k0 = 0; wave_id + k0 * nwaves < a.n; k0 += GR) {
      const unsigned mk = mk_next;
      unsigned todo_rays = (unsigned)__ballot(hnd_next == 0);
      hnd_next = group_flags(k0 + GR, mk_next);
      CP(0);
      while (todo_rays) {  // (wave-uniform; no workgroup barrier inside the loops)
        const int jr = __ffs(todo_rays) - 1;
        todo_rays &= todo_rays - 1u;
        const int64_t ray = wave_id + (k0 + jr) * nwaves;
        const unsigned mask_word = (unsigned)__builtin_amdgcn_readlane((int)mk, jr);
        const unsigned mask_bits = mask_word & 0xffffu, miss_bits = mask_word >> 16;
        const unsigned have_m = (ev2 | (ev1 & (a.mask ? mask_bits : ~0u))) & lmask;
        const float* __restrict__ tsrc = a.t + ray * LS;
        const float4* __restrict__ rsrc = a.raw + ray * LS;
        // (idle lanes of a ragged last block read the layer's last sample: no branch around the loads, nothing uses the value)
        auto sample_of = [&](int b) -> unsigned { const unsigned k = (unsigned)b * 64u + lane; return FULL ? k : (k < S ? k : S - 1u); };
        auto load = [&](Regs& r, unsigned layer) {   // raw is read for a layer without output as well (zeroed when used)
#pragma unroll
            for (int b = 0; b < MAXB; ++b) {
                r.tk[b] = tsrc[layer * S + sample_of(b)];
                r.rw[b] = rsrc[layer * S + sample_of(b)];
            }
        };
        // the first layer with network output is (almost always) the first live layer: its loads go out together with the
        // depth checks below instead of one round trip behind them
        const int first_have = have_m ? __ffs(have_m) - 1 : -1;
        Regs cur;
        load(cur, first_have >= 0 ? (unsigned)first_have : 0u);
        // ---- which layers take part: those with network output, and those without whose depths are real (a hidden layer,
        // a grazing hit: they shape their neighbours' deltas; see composite_kernel).  Eight layers' depths per round trip.
        unsigned live = have_m;
        constexpr int CB = 8;
        for (unsigned cand = ~have_m & ~miss_bits & lmask; cand;) {   // (layers the sampler flagged as missed are not even looked at)
            int ly[CB];
            float v[CB][MAXB];
#pragma unroll
            for (int j = 0; j < CB; ++j) {
                ly[j] = cand ? __ffs(cand) - 1 : -1;
                cand &= cand - 1u;   // (0 stays 0)
#pragma unroll
                for (int b = 0; b < MAXB; ++b) v[j][b] = ly[j] >= 0 ? tsrc[(unsigned)ly[j] * S + sample_of(b)] : -1000.f;
            }
#pragma unroll
            for (int j = 0; j < CB; ++j) {
                bool missed = true;
#pragma unroll
                for (int b = 0; b < MAXB; ++b) missed = missed && v[j][b] == -1000.f;
                if (ly[j] >= 0 && !__all(missed)) live |= 1u << ly[j];
            }
        }
        const unsigned nlive = (unsigned)__popc(live);
        CP(1);
        if (nlive * S > LC) continue;   // more live layers than this launch's list holds: the next launch takes the ray
        // ---- the layers the ray misses: zero weights and outputs
        for (unsigned dead = ~live & lmask; dead; dead &= dead - 1u) {
            const unsigned other = (unsigned)__ffs(dead) - 1u;
            if (a.weights) {
                float* wz = a.weights + (ray * L + other) * S;
#pragma unroll
                for (int b = 0; b < MAXB; ++b)
                    if (ok_lane(b)) wz[(unsigned)b * 64u + lane] = 0.f;
            }
            if (a.layer_out && lane < 5u) a.layer_out[(ray * L + other) * 5 + lane] = 0.f;
        }
        if (a.handled && lane == 0u) a.handled[ray] = 1;
        if (nlive == 0u) {
            if (a.mixed_out && lane < 5u) a.mixed_out[ray * 5 + lane] = 0.f;
            continue;
        }
        unsigned m = 0;                    // length of the merged list
        bool any_unsorted = false, merged_done = false;
        unsigned todo = live;
        int layer = __ffs(todo) - 1;
        todo &= todo - 1u;
        if (layer != first_have) load(cur, (unsigned)layer);   // (a layer without output in front of it takes part: hidden / grazing)
        Regs nxt = cur;
        while (layer >= 0) {
            const int nlayer = todo ? __ffs(todo) - 1 : -1;
            todo &= todo - 1u;
            if (nlayer >= 0) load(nxt, (unsigned)nlayer);
            __builtin_amdgcn_sched_barrier(0);  // the next layer's loads go out ahead of this layer's arithmetic
            // ---- density edits (a10), as composite_kernel's staging
            {
                const bool have = (have_m >> layer & 1u) != 0;
                const bool cut_neg = !fine && a.p.cut_negative_t && layer > 0;             // :414
                const bool cut_near = !fine && layer == 0;                                 // :422
                const float thr = tab[layer], sscale = tab[16 + layer];
#pragma unroll
                for (int b = 0; b < MAXB; ++b) {
                    float4 v = cur.rw[b];
                    if (!have) v = make_float4(0.f, 0.f, 0.f, 0.f);     // zero tensors (:398-399); sigma = 0 makes the colour moot
                    else {
                        if (cut_neg && cur.tk[b] < 0.f) v.w = 0.f;
                        if (v.w < thr) v.w = 0.f;                                           // :416-418, :538-547, :564-566
                        v.w = v.w * sscale;                                                 // :575-576
                        if (cut_near && cur.tk[b] < nearv) v.w = 0.f;
                        if (!activated) {
                            v.x = sigmoidf(v.x);
                            v.y = sigmoidf(v.y);
                            v.z = sigmoidf(v.z);
                        }
                    }
                    cur.rw[b] = v;
                }
            }
            // ---- successor depths inside the layer (lane + 1; lane 63 takes the next block's lane 0) and the list's direction
            float tn[MAXB];
            bool desc = false, not_desc = false;
#pragma unroll
            for (int b = 0; b < MAXB; ++b) {
                float first_next = 0.f;
                if (b + 1 < MAXB) first_next = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cur.tk[b + 1 < MAXB ? b + 1 : b])));
                const float shifted = dpp_move<DPP_WAVE_SHL1>(0.f, cur.tk[b]);
                tn[b] = lane == 63u ? first_next : shifted;
                const bool has_next = FULL ? (b + 1 < MAXB || lane != 63u) : ((unsigned)b * 64u + lane + 1u < S);
                desc = desc || (has_next && tn[b] < cur.tk[b]);
                not_desc = not_desc || (has_next && !(tn[b] < cur.tk[b]));
            }
            const bool some_desc = __any(desc), some_asc = __any(not_desc);
            const bool rev = some_desc && !some_asc, unsorted = some_desc && some_asc;
            // ---- the layer's own composite (:435-444 / :598-603), from registers
            float* wdst = a.weights ? a.weights + (ray * L + (unsigned)layer) * S : nullptr;
            const bool single_asc = a.mixed_out && nlive == 1u && !some_desc;
            {
                CompositeAcc A;
#pragma unroll
                for (int b = 0; b < MAXB; ++b) {
                    if (FULL || (unsigned)b * 64u < S) {  // (uniform)
                        const unsigned k = (unsigned)b * 64u + lane;
                        const bool last = FULL ? (b + 1 == MAXB && lane == 63u) : (k + 1u >= S);
                        const float delta = last ? border : tn[b] - cur.tk[b];
                        const float w = composite_block<FULL>(A, cur.rw[b].w, delta, cur.rw[b].x, cur.rw[b].y, cur.rw[b].z, cur.tk[b], ok_lane(b));
                        if (wdst && ok_lane(b)) wdst[k] = w;
                    }
                }
                const float t_first = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cur.tk[0])));
                // one live, ascending layer: the union IS the layer; the mix differs from its composite only by the fine
                // stage's `t < near` cut (:605)
                const bool mix_is_layer = single_asc && !(fine && t_first < nearv);
                composite_store5(A, a.layer_out ? a.layer_out + (ray * L + (unsigned)layer) * 5 : nullptr,
                                 mix_is_layer ? a.mixed_out + ray * 5 : nullptr, lane);
                if (single_asc && !mix_is_layer) {
                    CompositeAcc M;
#pragma unroll
                    for (int b = 0; b < MAXB; ++b) {
                        if (FULL || (unsigned)b * 64u < S) {
                            const unsigned k = (unsigned)b * 64u + lane;
                            const bool last = FULL ? (b + 1 == MAXB && lane == 63u) : (k + 1u >= S);
                            const float delta = last ? border : tn[b] - cur.tk[b];
                            const float sg = cur.tk[b] < nearv ? 0.f : cur.rw[b].w;
                            composite_block<FULL>(M, sg, delta, cur.rw[b].x, cur.rw[b].y, cur.rw[b].z, cur.tk[b], ok_lane(b));
                        }
                    }
                    composite_store5(M, a.mixed_out + ray * 5, nullptr, lane);
                }
                merged_done = single_asc;
            }
            CP(2);
            if (a.mixed_out && !single_asc) {
                if (unsorted) {
                    any_unsorted = true;
                } else if (!any_unsorted) {
                    // ---- insert the layer into the merged list (:425-429 / :587-592)
                    const unsigned base_e = (unsigned)layer * S;
                    if (m == 0u) {
#pragma unroll
                        for (int b = 0; b < MAXB; ++b) {
                            const unsigned k = (unsigned)b * 64u + lane;
                            if (ok_lane(b)) {
                                const unsigned r = rev ? S - 1u - k : k;
                                mkey[r] = cur.tk[b];
                                mpay[r] = (unsigned short)(base_e + k);
                            }
                        }
                    } else {
                        const unsigned tot = m + S, nblk = (tot + 63u) >> 6;
                        for (unsigned w = lane; w < 2u * nblk; w += 64u) bits[w] = 0u;
#pragma unroll
                        for (int b = 0; b < MAXB; ++b) {
                            const unsigned k = (unsigned)b * 64u + lane;
                            if (ok_lane(b)) ckey[rev ? S - 1u - k : k] = cur.tk[b];
                        }
                        wave_sync();
                        // #{list entries <= v} for the MAXB samples of a lane in lockstep (independent LDS chains): a lower
                        // bound whose interval LENGTH is the same for every lane (a scalar), so that a probe is an add, a
                        // ds_read, a compare and a select -- no clamp against the list's end, no per-lane bound test.
                        // `at[b]` points at the entry in front of the interval (never read before it has moved).
                        const float* at[MAXB];
#pragma unroll
                        for (int b = 0; b < MAXB; ++b) at[b] = mkey - 1;
                        for (unsigned len = (unsigned)__builtin_amdgcn_readfirstlane((int)m); len > 1u;) {
                            const unsigned half = (unsigned)__builtin_amdgcn_readfirstlane((int)(len >> 1));
#pragma unroll
                            for (int b = 0; b < MAXB; ++b) {
                                const float* probe = at[b] + half;
                                at[b] = (*probe <= cur.tk[b]) ? probe : at[b];
                            }
                            len = (unsigned)__builtin_amdgcn_readfirstlane((int)(len - half));
                        }
                        unsigned pos[MAXB];
#pragma unroll
                        for (int b = 0; b < MAXB; ++b)   // (32-bit LDS addresses: `at` is one entry in front of the interval)
                            pos[b] = (((unsigned)(uintptr_t)at[b] - (unsigned)(uintptr_t)mkey + 4u) >> 2) + ((at[b][1] <= cur.tk[b]) ? 1u : 0u);
#pragma unroll
                        for (int b = 0; b < MAXB; ++b) {
                            const unsigned k = (unsigned)b * 64u + lane;
                            if (ok_lane(b)) {
                                const unsigned p = (rev ? S - 1u - k : k) + pos[b];
                                __hip_atomic_fetch_or(&bits[p >> 5], 1u << (p & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                        wave_sync();
                        unsigned above = 0;  // samples of the new layer in the blocks already written (higher slots)
                        for (int B = (int)nblk - 1; B >= 0; --B) {
                            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)bits[2 * B]);
                            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)bits[2 * B + 1]);
                            const unsigned cnt_in = (unsigned)(__popc(lo) + __popc(hi));
                            const unsigned below = S - above - cnt_in + __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
                            const bool is_new = ((lane < 32u ? lo >> lane : hi >> (lane - 32u)) & 1u) != 0u;
                            const unsigned p = (unsigned)B * 64u + lane;
                            if (p < tot) {
                                const unsigned j = p - below;                        // list entries in front of slot p
                                const float key = mkey[is_new ? LC + below : j];     // (ckey = mkey + LC)
                                const unsigned short old = mpay[j];
                                const unsigned short pay = is_new ? (unsigned short)(base_e + (rev ? S - 1u - below : below)) : old;
                                mkey[p] = key;
                                mpay[p] = pay;
                            }
                            above += cnt_in;
                        }
                    }
                    m = (unsigned)__builtin_amdgcn_readfirstlane((int)(m + S));
                    wave_sync();
                }
            }
            CP(3);
            cur = nxt;
            layer = nlayer;
        }
        if (!a.mixed_out || merged_done) continue;
        if (any_unsorted) {  // general rank over the live samples: (depth, source index) lexicographic, O(n^2) (tests only)
            for (unsigned la_m = live; la_m; la_m &= la_m - 1u) {
                const unsigned la = (unsigned)__ffs(la_m) - 1u;
                for (unsigned k = lane; k < S; k += 64u) {
                    const unsigned e = la * S + k;
                    const float v = tsrc[e];
                    unsigned rank = 0;
                    for (unsigned lb_m = live; lb_m; lb_m &= lb_m - 1u) {
                        const unsigned lb = (unsigned)__ffs(lb_m) - 1u;
                        for (unsigned x = 0; x < S; ++x) {
                            const float xv = tsrc[lb * S + x];
                            rank += (xv < v || (xv == v && lb * S + x < e)) ? 1u : 0u;
                        }
                    }
                    mkey[rank] = v;
                    mpay[rank] = (unsigned short)e;
                }
            }
            m = nlive * S;
            wave_sync();
        }
        // ---- merged composite (:448 / :605-606): two blocks of the list per round trip of the float4 gather (four cost 13 registers = one wave per SIMD)
        {
            constexpr int G = 2;
            const bool cut_neg_on = !fine && a.p.cut_negative_t;
            CompositeAcc A;
            for (unsigned base = 0; base < m; base += 64u * G) {
                float key[G], keyn[G];
                float4 rw[G];
                unsigned src[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const unsigned mm = base + (unsigned)g * 64u + lane;
                    const unsigned mc = FULL ? mm : (mm < m ? mm : m - 1u);
                    if (base + (unsigned)g * 64u < m) {  // (uniform)
                        key[g] = mkey[mc];
                        keyn[g] = mkey[mc + 1u];          // (one past the list's end for its last sample: inside the LDS window, unused)
                        src[g] = mpay[mc];
                        rw[g] = rsrc[src[g]];
                    }
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (base + (unsigned)g * 64u < m) {  // (uniform)
                        const unsigned mm = base + (unsigned)g * 64u + lane;
                        const bool ok = FULL || mm < m;
                        float4 v = rw[g];
                        const unsigned ly = (unsigned)(((float)src[g] + 0.5f) * invS);
                        const float tk = key[g];
                        if (!(have_m >> ly & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                        else {
                            if (cut_neg_on && ly > 0u && tk < 0.f) v.w = 0.f;
                            if (v.w < tab[ly]) v.w = 0.f;
                            v.w = v.w * tab[16 + ly];
                            if (!fine && ly == 0u && tk < nearv) v.w = 0.f;
                            if (!activated) {
                                v.x = sigmoidf(v.x);
                                v.y = sigmoidf(v.y);
                                v.z = sigmoidf(v.z);
                            }
                        }
                        if (fine && tk < nearv) v.w = 0.f;                        // :605
                        const float delta = (mm + 1u < m) ? keyn[g] - tk : border;
                        composite_block<FULL>(A, v.w, delta, v.x, v.y, v.z, tk, ok);
                    }
                }
            }
            composite_store5(A, a.mixed_out + ray * 5, nullptr, lane);
        }
        CP(4);
        wave_sync();
      }
    }
    CP_FLUSH;
}

// ---- searches over an ascending LDS array padded with +inf up to (a power of two) - 1 entries: no bounds logic, the running
// position is a byte address -- add, ds_read, compare, select per step (the bounded form above: eight instructions).
// `half_bytes` = 4 * P / 2 for P = the smallest power of two > n (wave-uniform).  Returns #{x <= v} / #{x < v}.
__host__ __device__ __forceinline__ int pow2_above(int n) {
    int p = 1;
    while (p <= n) p *= 2;
    return p;
}
__device__ __forceinline__ int upper_bound_padded(const float* a, int half_bytes, float v) {
    const char* base = reinterpret_cast<const char*>(a) - 4;
    const char* q = base;
    for (int step = half_bytes; step >= 4; step >>= 1) {
        const char* c = q + step;
        const float x = *reinterpret_cast<const float*>(c);
        q = (x <= v) ? c : q;
    }
    return (int)(q - base) >> 2;
}
__device__ __forceinline__ int lower_bound_padded(const float* a, int half_bytes, float v) {
    const char* base = reinterpret_cast<const char*>(a) - 4;
    const char* q = base;
    for (int step = half_bytes; step >= 4; step >>= 1) {
        const char* c = q + step;
        const float x = *reinterpret_cast<const float*>(c);
        q = (x < v) ? c : q;
    }
    return (int)(q - base) >> 2;
}

// ---- ascending bitonic sort of one value per lane (64 lanes).  The partner of a compare-exchange at distance 1, 2 and 8 is
// a DPP operand of the min / max themselves (quad_perm / row_ror:8), at distance 4 two bank-masked DPP moves, at 16 a
// ds_swizzle, at 32 a ds_bpermute; which lanes keep the minimum is a lane pattern, i.e. a 64-bit constant per stage fed to
// v_cndmask as a scalar mask.  3 - 5 vector instructions per stage (21 stages) against ~ 7 with __shfl_xor and a computed
// direction.
constexpr unsigned long long bitonic_keep_min_mask(int kk, int j) {
    unsigned long long m = 0;
    for (int lane = 0; lane < 64; ++lane)
        if (((lane & j) == 0) == ((lane & kk) == 0)) m |= 1ull << lane;
    return m;
}
template <int KK, int J>
__device__ __forceinline__ float bitonic_stage(float v, int lane) {
    constexpr unsigned long long KEEP_MIN = bitonic_keep_min_mask(KK, J);
    float lo, hi;
    if constexpr (J == 1) {
        asm("s_nop 1\n\tv_min_f32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_max_f32_dpp %1, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(lo), "=&v"(hi) : "v"(v));
    } else if constexpr (J == 2) {
        asm("s_nop 1\n\tv_min_f32_dpp %0, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_max_f32_dpp %1, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=&v"(lo), "=&v"(hi) : "v"(v));
    } else if constexpr (J == 8) {
        asm("s_nop 1\n\tv_min_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_max_f32_dpp %1, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf" : "=&v"(lo), "=&v"(hi) : "v"(v));
    } else {
        float pv;
        if constexpr (J == 4) {   // lanes 0-3 / 8-11 of a row take lane + 4, lanes 4-7 / 12-15 lane - 4
            pv = v;
            asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                "v_mov_b32_dpp %0, %1 row_shr:4 row_mask:0xf bank_mask:0xa" : "+&v"(pv) : "v"(v));   // early clobber: %0 is
            // written by the first move before the second reads %1 -- the two must never share a register
        } else if constexpr (J == 16) {
            pv = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401f));   // bit mode: and 0x1f, or 0, xor 0x10
        } else {
            pv = __shfl_xor(v, J);
        }
        lo = fminf(v, pv);
        hi = fmaxf(v, pv);
    }
    // (the mask is materialised next to its use: as an "s" operand the 21 constants are hoisted out of the pair loop -- 42 scalar
    // registers, which the allocator then parks in vector lanes and fetches back with two v_readlane per stage)
    float r;
    asm("s_mov_b32 vcc_lo, %3\n\ts_mov_b32 vcc_hi, %4\n\tv_cndmask_b32_e32 %0, %1, %2, vcc"
        : "=v"(r) : "v"(hi), "v"(lo), "n"((unsigned)(KEEP_MIN & 0xffffffffull)), "n"((unsigned)(KEEP_MIN >> 32)) : "vcc");
    (void)lane;
    return r;
}
__device__ __forceinline__ float bitonic_sort64(float v, int lane) {
    v = bitonic_stage<2, 1>(v, lane);
    v = bitonic_stage<4, 2>(v, lane);   v = bitonic_stage<4, 1>(v, lane);
    v = bitonic_stage<8, 4>(v, lane);   v = bitonic_stage<8, 2>(v, lane);   v = bitonic_stage<8, 1>(v, lane);
    v = bitonic_stage<16, 8>(v, lane);  v = bitonic_stage<16, 4>(v, lane);  v = bitonic_stage<16, 2>(v, lane);  v = bitonic_stage<16, 1>(v, lane);
    v = bitonic_stage<32, 16>(v, lane); v = bitonic_stage<32, 8>(v, lane);  v = bitonic_stage<32, 4>(v, lane);  v = bitonic_stage<32, 2>(v, lane);
    v = bitonic_stage<32, 1>(v, lane);
    v = bitonic_stage<64, 32>(v, lane); v = bitonic_stage<64, 16>(v, lane); v = bitonic_stage<64, 8>(v, lane);  v = bitonic_stage<64, 4>(v, lane);
    v = bitonic_stage<64, 2>(v, lane);  v = bitonic_stage<64, 1>(v, lane);
    return v;
}

// ---------------------------------------------------------------------------------------------
// Resampler: one wave per (ray, layer).
// ---------------------------------------------------------------------------------------------
// floats of LDS per wave of resample_kernel
__host__ __device__ __forceinline__ int resample_lds_floats(int n1, int n2) {
    return pow2_above(n1) + pow2_above(n1 - 1) + pow2_above(n2) + 2 * n1 + n1 + n2;
}

struct ResampleArgs {
    const float* t;
    const float* weights;
    int64_t n;
    int l, n1, n2;
    const float* u;
    uint64_t seed;
    RayWindow win;
    const float* rays;
    int ray_stride;
    EditArgs ed;
    const uint8_t* mask;   // [n][l] or null: bit 1 = the sampler's "every depth of this pair is -1000" hint -> the pair is skipped
    float* t_fine;
    float* xyz_fine;
    float* z_new;
    int32_t* inds;
    float* cdf_out;
};

// NB1 > 0: the coarse list fits NB1 blocks of 64 lanes and is software pipelined -- the depths, weights and ray of pair
// i + 1 are in flight (in registers) while pair i is worked on.  The kernel is otherwise a chain of dependent round
// trips per pair ("are all depths -1000?" -> "stage depths and weights" -> compute -> stores) with ~ 150 instructions
// between them: measured, 80 % of a pair's cycles were waits for the first two (tools/resample_phase_prof.py).
// NB1 = 0: lists of any length, loads where they are needed.
// EXACT: n1 = 64 * NB1 and n2 = 64 (every BASELINE configuration: 64+64, 128+64) are compile-time constants -- the lane
// predicates (k < n1, k < n1 - 2, lane < n2, ...) fold away instead of living as 64-bit masks in spilled scalar registers,
// the searches unroll onto immediate LDS offsets: -30 % vector instructions (rocprofv3 SQ_INSTS_VALU), -25 % time.
// PLAIN: the production call -- device draws, no debug outputs, no box edits.  The arguments of the other flavours (u, z_new,
// inds, cdf, the edit table) are then dead: a third of this kernel's vector instructions were v_readlane / v_writelane
// traffic of scalar registers spilled into vector lanes, most of it kernel arguments it never uses on this path.
template <int NB1, bool PLAIN, bool EXACT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NB1 >= 4 ? 6 : NB1 >= 2 ? 7 : STNERF_WAVES_RESAMPLE, 8))) resample_kernel(ResampleArgs a) {
    const float* const u_in = PLAIN ? nullptr : a.u;
    float* const z_out = PLAIN ? nullptr : a.z_new;
    int32_t* const inds_out = PLAIN ? nullptr : a.inds;
    float* const cdf_dbg = PLAIN ? nullptr : a.cdf_out;
    const bool edited = PLAIN ? false : a.ed.any != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform: the pair index, its 64-bit divisions and
                                                                        // the RNG key of the pair stay on the scalar unit
    const int n1 = EXACT ? 64 * NB1 : a.n1, n2 = EXACT ? 64 : a.n2, S = n1 + n2, nb = n1 - 1;  // nb = #bins = len(cdf)
    // the three searched arrays are padded with +inf to (a power of two) - 1 entries, once (upper_bound_padded)
    const int P1 = pow2_above(n1), PC = pow2_above(nb), P2 = pow2_above(n2);
    float* mine = reinterpret_cast<float*>(smem_raw) + (size_t)wave * resample_lds_floats(n1, n2);
    float* tc = mine;          // [n1 | pad to P1]  coarse depths
    float* cdf = tc + P1;      // [n1-1 | pad to PC]
    float* zs = cdf + PC;      // [n2 | pad to P2]
    float* bins = zs + P2;     // [n1-1]
    float* wv = bins + n1;     // [n1-2] pdf numerators w + 1e-5
    float* tf = wv + n1;       // [S]
    for (int k = n1 + lane; k < P1; k += 64) tc[k] = __builtin_inff();
    for (int k = nb + lane; k < PC; k += 64) cdf[k] = __builtin_inff();
    for (int k = n2 + lane; k < P2; k += 64) zs[k] = __builtin_inff();
    const int p2_n1 = floor_pow2(n1);
    const int64_t pairs = a.n * a.l;
    const int64_t per_iter = (int64_t)gridDim.x * 4;
    constexpr int NBR = NB1 > 0 ? NB1 : 1;
    struct Pre {
        float t[NBR], w[NBR], r;   // r: lane i < 6 holds component i of the ray (origin, direction)
        int m;                     // the pair's mask byte (every lane), fetched with the rest: no round trip of its own
    };
    auto issue = [&](Pre& q, int64_t pr, int64_t ray_of_pr) {
        q.m = (a.mask && pr < pairs) ? (int)a.mask[pr] : 0;
        if (NB1 > 0 && pr < pairs) {
            const float* tsrc = a.t + pr * n1;
            const float* wsrc = a.weights + pr * n1;
#pragma unroll
            for (int b = 0; b < NBR; ++b) {
                const int k = b * 64 + lane;
                q.t[b] = tsrc[k < n1 ? k : n1 - 1];
                q.w[b] = wsrc[k + 1 < n1 ? k + 1 : n1 - 1];   // (the caller's w[..., 1:-1]: numerator k is weight k + 1)
            }
            q.r = a.rays[ray_of_pr * a.ray_stride + (lane < 6 ? lane : 5)];
        }
    };
    Pre nxt_in;
#pragma unroll
    for (int b = 0; b < NBR; ++b) nxt_in.t[b] = nxt_in.w[b] = 0.f;
    nxt_in.r = 0.f;
    nxt_in.m = 0;
    // (ray, layer) of the wave's pair are carried along instead of divided out of the pair index every iteration: a 64-bit
    // division is ~ 80 scalar + vector instructions, and there were two per pair
    const int64_t dray = per_iter / a.l;
    const int dlayer = (int)(per_iter - dray * a.l);
    int64_t ray_n = ((int64_t)blockIdx.x * 4 + wave) / a.l;
    int layer_n = (int)((int64_t)blockIdx.x * 4 + wave - ray_n * a.l);
    issue(nxt_in, (int64_t)blockIdx.x * 4 + wave, ray_n);
    CP_DECL
    for (int64_t p0 = (int64_t)blockIdx.x * 4; p0 < pairs; p0 += per_iter) {
        const int64_t pr = p0 + wave;
        CP(7);
        const Pre in = nxt_in;
        const int64_t ray = ray_n;
        const int layer = layer_n;
        ray_n += dray;
        layer_n += dlayer;
        if (layer_n >= a.l) {
            layer_n -= a.l;
            ++ray_n;
        }
        issue(nxt_in, pr + per_iter, ray_n);
        __builtin_amdgcn_sched_barrier(0);  // the next pair's loads go out ahead of this pair's arithmetic
        const bool active = pr < pairs;
        // the sampler flagged the pair as missed (include/stnerf.h): nothing of it is read downstream -- nothing is written
        if (active && !z_out && !inds_out && !cdf_dbg && (__builtin_amdgcn_readfirstlane(in.m) & 2)) continue;
        bool sorted_z = false;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f;
        if (active) {
            if (NB1 > 0) {
                o0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(in.r), 0));
                o1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(in.r), 1));
                o2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(in.r), 2));
                d0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(in.r), 3));
                d1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(in.r), 4));
                d2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(in.r), 5));
            } else {
                const float* r = a.rays + ray * a.ray_stride;
                o0 = r[0], o1 = r[1], o2 = r[2], d0 = r[3], d1 = r[4], d2 = r[5];
            }
        }
        // ---- a layer the ray misses altogether: every coarse depth is -1000 (bin width 0), so every bin edge and every
        // resampled depth is exactly -1000 whatever the draws; write that and skip the work (60 % of the performer
        // pairs of a typical view).  The optional debug outputs take the general path.
        if (active && !z_out && !inds_out && !cdf_dbg) {
            const float* tsrc = a.t + pr * n1;
            bool missed = true;
            if (NB1 > 0) {
#pragma unroll
                for (int b = 0; b < NBR; ++b) missed = missed && in.t[b] == -1000.f;   // (idle lanes hold the last depth)
            } else {
                for (int k = lane; k < n1; k += 64) missed = missed && tsrc[k] == -1000.f;
            }
            if (__all(missed)) {
                float x = -1000.f * d0 + o0, y = -1000.f * d1 + o1, w = -1000.f * d2 + o2;  // :465
                if (edited) unedit_point(x, y, w, a.ed.e[layer], a.ed.pivot);
                for (int m = lane; m < S; m += 64) {
                    a.t_fine[pr * S + m] = -1000.f;
                    if (a.xyz_fine) {
                        float* dst = a.xyz_fine + (pr * S + m) * 3;
                        dst[0] = x;
                        dst[1] = y;
                        dst[2] = w;
                    }
                }
                CP(0);
                continue;
            }
        }
        CP(0);
        // ---- pdf / cdf / bins   (sample_pdf.py:20-24; the caller passes w[..., 1:-1], layered_rfrender.py:460)
        if (active) {
            if (NB1 > 0) {
#pragma unroll
                for (int b = 0; b < NBR; ++b) {
                    const int k = b * 64 + lane;
                    if (k < n1) tc[k] = in.t[b];
                    if (k < n1 - 2) wv[k] = in.w[b] + 1e-5f;  // weights + 1e-5 (sample_pdf.py:21)
                }
            } else {
                const float* tsrc = a.t + pr * n1;
                const float* wsrc = a.weights + pr * n1;
                for (int k = lane; k < n1; k += 64) tc[k] = tsrc[k];
                for (int k = lane; k < n1 - 2; k += 64) wv[k] = wsrc[k + 1] + 1e-5f;  // weights + 1e-5 (sample_pdf.py:21)
            }
        }
        wave_sync();
        CP(1);
        if (active) {
            // pdf = w / torch.sum(w) in ATen's CPU summation order; cdf = torch.cumsum(pdf): ATen's CPU cumsum
            // accumulates fp32 rows in DOUBLE and rounds every prefix to fp32 (cumsum_cpu_kernel: at::acc_type<float,
            // false>).  The pdf values are fp32 numbers in [2^-17, 1], so every fp64 partial sum is exact and the
            // parallel scan below yields the sequential loop's bits: cdf, and with it inds and z, are bit-equal to the
            // reference's CPU evaluation for the same (t, w, u).
            const float total = aten_cpu_row_sum(wv, n1 - 2, lane);
            double carry = 0.0;
            if (lane == 0) cdf[0] = 0.f;
            for (int base = 0; base < n1 - 2; base += 64) {
                const int k = base + lane;
                const double pdf = (k < n1 - 2) ? (double)(wv[k] / total) : 0.0;
                const double incl = wave_scan_add_f64(pdf, lane);
                if (k < n1 - 2) cdf[k + 1] = (float)(carry + incl);
                carry = carry + __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(__double_as_longlong(incl) >> 32), 63) << 32) |
                                                     (unsigned int)__builtin_amdgcn_readlane((int)(__double_as_longlong(incl) & 0xffffffffll), 63));
            }
        }
        wave_sync();
        CP(2);
        if (active) {
            for (int k = lane; k < nb; k += 64) bins[k] = 0.5f * (tc[k + 1] + tc[k]);
            if (cdf_dbg)
                for (int k = lane; k < nb; k += 64) cdf_dbg[pr * nb + k] = cdf[k];
        }
        wave_sync();
        CP(3);
        // ---- invert the cdf (sample_pdf.py:44-61)
        if (active) {
            const uint64_t gray = u_in ? 0ull : (uint64_t)global_ray(a.win, ray);  // (wave-uniform)
            for (int j = lane; j < n2; j += 64) {
                const float u = u_in ? u_in[((int64_t)layer * a.n + ray) * n2 + j]
                                    : philox_uniform(a.seed, gray, (uint32_t)layer, 1u, (uint32_t)j);
                const int ind = upper_bound_padded(cdf, 2 * PC, u);   // searchsorted(right=True)
                const int below = ind - 1 > 0 ? ind - 1 : 0;
                const int above = ind < nb - 1 ? ind : nb - 1;
                float den = cdf[above] - cdf[below];
                if (den < 1e-5f) den = 1.f;
                const float frac = (u - cdf[below]) / den;
                const float z = bins[below] + frac * (bins[above] - bins[below]);
                zs[j] = z;
                if (z_out) z_out[pr * n2 + j] = z;
                if (inds_out) inds_out[pr * n2 + j] = ind;
            }
        }
        wave_sync();
        CP(4);
        // ---- sort(cat[t, z])  (layered_rfrender.py:462) by ranks == a stable sort with t before z on ties.
        // The coarse list is ascending (unless a box edit made the bin width negative), so a t keeps its index
        // plus the number of smaller z, and a z its index among the sorted z plus the number of t <= z.  Up to 64
        // new samples are sorted in registers (bitonic network over the wave's lanes); equal z are interchangeable
        // because only values leave this kernel.
        if (active) {
            bool desc = false, ndesc = false;
            for (int k = lane; k + 1 < n1; k += 64) {
                desc = desc || (tc[k + 1] < tc[k]);
                ndesc = ndesc || !(tc[k + 1] < tc[k]);
            }
            bool asc = !__any(desc);
            if (!asc && !__any(ndesc)) {
                // strictly descending coarse list (negative bin width: a ray that misses the background box, or an
                // edited box): only the sorted VALUES leave this kernel and bins / cdf are done with, so turn the
                // list round in place and take the sorted path
                for (int k = lane; k < n1 / 2; k += 64) {
                    const float lo = tc[k], hi = tc[n1 - 1 - k];
                    tc[k] = hi;
                    tc[n1 - 1 - k] = lo;
                }
                wave_sync();
                asc = true;
            }
            if (asc && n2 <= 64) {
                const float v = bitonic_sort64(lane < n2 ? zs[lane] : __builtin_inff(), lane);
                if (lane < n2) {
                    zs[lane] = v;  // now ascending
                    tf[lane + upper_bound_padded(tc, 2 * P1, v)] = v;
                }
            }
            sorted_z = asc && n2 <= 64;
        }
        wave_sync();
        if (active) {
            if (sorted_z) {
                for (int k = lane; k < n1; k += 64) {
                    const float v = tc[k];
                    tf[k + lower_bound_padded(zs, 2 * P2, v)] = v;
                }
            } else {
                bool desc = false;
                for (int k = lane; k + 1 < n1; k += 64) desc = desc || (tc[k + 1] < tc[k]);
                const bool asc = !__any(desc);
                for (int e = lane; e < S; e += 64) {
                    const bool is_t = e < n1;
                    const float v = is_t ? tc[e] : zs[e - n1];
                    int rank;
                    if (asc) {
                        rank = is_t ? e : upper_bound_lds(tc, n1, p2_n1, v);
                    } else {
                        rank = 0;
                        for (int x = 0; x < n1; ++x) {
                            const float xv = tc[x];
                            rank += (xv < v || (xv == v && (!is_t || x < e))) ? 1 : 0;
                        }
                    }
                    for (int x = 0; x < n2; ++x) {
                        const float xv = zs[x];
                        rank += (xv < v || (xv == v && !is_t && x < e - n1)) ? 1 : 0;
                    }
                    tf[rank] = v;
                }
            }
        }
        wave_sync();
        CP(5);
        if (active) {
            for (int m = lane; m < S; m += 64) {
                const float z = tf[m];
                a.t_fine[pr * S + m] = z;
                if (a.xyz_fine) {
                    float x = z * d0 + o0, y = z * d1 + o1, w = z * d2 + o2;  // :465
                    if (edited) unedit_point(x, y, w, a.ed.e[layer], a.ed.pivot);
                    float* dst = a.xyz_fine + (pr * S + m) * 3;  // (staging these through LDS for 16-B stores measured slower)
                    dst[0] = x;
                    dst[1] = y;
                    dst[2] = w;
                }
            }
        }
        wave_sync();
        CP(6);
    }
    CP_FLUSH;
}

}  // namespace stnerf

using namespace stnerf;

#ifdef STNERF_COMP_PROF
extern "C" int stnerf_debug_composite_phases(unsigned long long* host8, int reset) {
    if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_cphase), sizeof(unsigned long long) * 8) != hipSuccess) return STNERF_ELAUNCH;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_cphase), z, sizeof(z)) != hipSuccess) return STNERF_ELAUNCH;
    }
    return STNERF_OK;
}
#endif

extern "C" int stnerf_gen_weight(const float* sigma, const float* delta, int64_t n, int S, float* weights,
                                 stnerf_stream_t stream) {
    STNERF_REQUIRE(sigma && delta && weights, "gen_weight: null pointer");
    STNERF_REQUIRE(n >= 0 && S >= 1, "gen_weight: bad shape");
    if (n == 0) return STNERF_OK;
    hipLaunchKernelGGL(gen_weight_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), sigma, delta, n,
                       S, weights);
    STNERF_CHECK_LAUNCH("gen_weight");
    return STNERF_OK;
}

// Development switch: STNERF_COMPOSITE_KERNEL=staged sends every call to the LDS-staged kernel (A/B timing, bitwise checks).
static bool legacy_composite() {
    static const bool v = [] {
        const char* e = getenv("STNERF_COMPOSITE_KERNEL");
        return e && !strcmp(e, "staged");
    }();
    return v;
}

// Which kernels a stnerf_composite call launches, and with how much LDS (host arithmetic only; also exported as
// stnerf_composite_plan so that the sizing is testable without a GPU).
struct CompositePlan {
    int staged;      // 1: the LDS-staged kernel alone (`order` output, more than 192 samples per layer, development switch)
    int single;      // single-layer pre-pass: 0 none, 1 composite_single_kernel<2, 6>, 2 composite_single_kernel<3, 24>
    int tiers;       // launches of composite_merge_kernel: 1, or 2 when a list of all l layers would cost occupancy
    int cap;         // layers the first launch's merged list holds (= l with one launch)
    int clear;       // 1: scratch is cleared first (two launches, no pre-pass to write it)
    int wpb[2];      // waves per workgroup of the launch(es) (for the staged kernel: [0])
    int64_t lds[2];  // dynamic LDS bytes per workgroup
    int64_t need;    // LDS bytes per wave of the launch that needs most
};
constexpr int64_t COMPOSITE_LDS_BUDGET = 150 * 1024;   // of the CU's 160 KiB: the rest stays with the kernels' static LDS
constexpr int MERGE_MAXB = 3;

static bool plan_composite(int l, int S, bool scratch, bool order, bool any_output, bool staged_switch, CompositePlan& p) {
    p = CompositePlan{};
    const int nblk = (S + 63) / 64;
    if (order || nblk > MERGE_MAXB || staged_switch) {
        p.staged = 1;
        p.need = (((int64_t)l * S * 22 + 15) / 16) * 16;
        p.wpb[0] = (int)(COMPOSITE_LDS_BUDGET / p.need);
        if (p.wpb[0] > 4) p.wpb[0] = 4;
        p.lds[0] = p.need * p.wpb[0];
        return p.wpb[0] >= 1;
    }
    if (scratch && any_output) p.single = (nblk <= 2 && (l - 1) * nblk <= 6) ? 1 : ((l - 1) * nblk <= 24) ? 2 : 0;
    // The merged list lives in LDS, 6 B per sample and layer: 11.4 KB per wave at 9 x 192 samples -- three waves per SIMD,
    // where the registers allow six.  Few rays of such a scene cross every box, so with scratch the rays are served in two
    // launches: first with lists of as many layers as full occupancy leaves room for (a ray with more live layers is left
    // unmarked), then the rest with lists of l layers.
    const bool full = S == 64 * nblk;
    const int waves_per_simd = merge_waves_per_simd(nblk, full);   // (the kernels' amdgpu_waves_per_eu)
    int cap = l;
    while (cap > 2 && MERGE_TAB_BYTES + 4 * merge_lds_per_wave(cap, S) > COMPOSITE_LDS_BUDGET / waves_per_simd) --cap;
    const bool two = cap < l && scratch;
    p.tiers = two ? 2 : 1;
    p.cap = two ? cap : l;
    p.clear = two && !p.single;
    for (int tier = 0; tier < p.tiers; ++tier) {
        const int64_t per_wave = merge_lds_per_wave(tier == p.tiers - 1 ? l : cap, S);
        int wpb = (int)((COMPOSITE_LDS_BUDGET - MERGE_TAB_BYTES) / per_wave);
        if (wpb > 4) wpb = 4;
        p.wpb[tier] = wpb;
        p.lds[tier] = MERGE_TAB_BYTES + per_wave * wpb;
        p.need = per_wave;
        if (wpb < 1) return false;
    }
    return true;
}

extern "C" int stnerf_composite_plan(int l, int S, int with_scratch, int with_order, int64_t* plan) {
    STNERF_REQUIRE(plan, "composite_plan: null pointer");
    STNERF_REQUIRE(l >= 1 && l <= STNERF_MAX_LAYERS && S >= 1 && (int64_t)l * S <= 65535, "composite_plan: bad shape l=%d S=%d", l, S);
    CompositePlan p;
    const bool ok = plan_composite(l, S, with_scratch != 0, with_order != 0, true, false, p);
    const int64_t v[9] = {p.staged, p.single, p.tiers, p.cap, p.clear, p.wpb[0], p.lds[0], p.tiers == 2 ? p.wpb[1] : 0, p.tiers == 2 ? p.lds[1] : 0};
    for (int i = 0; i < 9; ++i) plan[i] = v[i];
    STNERF_REQUIRE(ok, "composite: %d samples per ray need %lld B of LDS per wave, more than the %lld B this kernel may use", l * S,
                   (long long)p.need, (long long)COMPOSITE_LDS_BUDGET);
    return STNERF_OK;
}

extern "C" int stnerf_composite(const float* t, const float* raw, const uint8_t* mask, int64_t n, int l, int S,
                                const stnerf_composite_params* params_host, float* layer_out, float* mixed_out,
                                float* weights, int32_t* order, uint8_t* scratch, stnerf_stream_t stream) {
    STNERF_REQUIRE(t && raw && params_host, "composite: null pointer");
    STNERF_REQUIRE(n >= 0 && l >= 1 && l <= STNERF_MAX_LAYERS && S >= 1, "composite: bad shape n=%lld l=%d S=%d",
                   (long long)n, l, S);
    STNERF_REQUIRE(((uintptr_t)raw & 15) == 0, "composite: raw must be 16-byte aligned");
    if (n == 0) return STNERF_OK;
    STNERF_REQUIRE((int64_t)l * S <= 65535, "composite: more than 65535 samples per ray");
    CompositePlan plan;
    const bool fits = plan_composite(l, S, scratch != nullptr, order != nullptr, layer_out || mixed_out || weights, legacy_composite(), plan);
    STNERF_REQUIRE(fits, "composite: %d samples per ray need %lld B of LDS per wave, more than the %lld B this kernel may use", l * S,
                   (long long)plan.need, (long long)COMPOSITE_LDS_BUDGET);
    LaunchTimer timer(PROF_COMPOSITE, 0, n, S,
                      20ll * l * S + l + 20ll * (l + 1) + (weights ? 4ll * l * S : 0) + (order ? 4ll * l * S : 0),
                      as_stream(stream));
    const int nblk = (S + 63) / 64;
    if (!plan.staged) {
        // ---- production path: rays with one live layer first when the caller lends n bytes of scratch (pipelined over
        // the rays of a wave, no LDS), the others -- or all of them -- in the register / insertion-merge kernel
        CompositeArgs a{t, reinterpret_cast<const float4*>(raw), mask, n, l, S, *params_host, layer_out, mixed_out,
                        weights, nullptr, 4, floor_pow2(S), nullptr};
        if (plan.single) {
            int64_t waves = n < 256 * 32 ? n : 256 * 32;  // 8 waves per SIMD, every wave strides over the rays
            const dim3 grid((unsigned)((waves + 3) / 4));
            a.handled = scratch;
            if (plan.single == 1) hipLaunchKernelGGL((composite_single_kernel<2, 6>), grid, dim3(256), 0, as_stream(stream), a);
            else hipLaunchKernelGGL((composite_single_kernel<3, 24>), grid, dim3(256), 0, as_stream(stream), a);
            STNERF_CHECK_LAUNCH("composite (single-layer rays)");
        }
        if (plan.clear) {
            if (hipMemsetAsync(scratch, 0, (size_t)n, as_stream(stream)) != hipSuccess) return STNERF_ELAUNCH;
            a.handled = scratch;
        }
        const bool full = S == 64 * nblk;
        for (int tier = 0; tier < plan.tiers; ++tier) {
            a.lds_layers = tier == plan.tiers - 1 ? l : plan.cap;
            const int wpb = plan.wpb[tier], lds = (int)plan.lds[tier];
            a.waves_per_block = wpb;
            int64_t blocks = (n + wpb - 1) / wpb;
            if (blocks > 256 * 16) blocks = 256 * 16;
            const dim3 grid((unsigned)blocks), block(wpb * 64);
            auto launch = [&](auto kernel) -> int {
                if (lds > 64 * 1024)
                    if (const int rc = reserve_dynamic_lds(reinterpret_cast<const void*>(kernel), lds, "composite")) return rc;
                hipLaunchKernelGGL(kernel, grid, block, lds, as_stream(stream), a);
                return STNERF_OK;
            };
            const int rc = nblk == 1 ? (full ? launch(composite_merge_kernel<1, true>) : launch(composite_merge_kernel<1, false>))
                         : nblk == 2 ? (full ? launch(composite_merge_kernel<2, true>) : launch(composite_merge_kernel<2, false>))
                                     : (full ? launch(composite_merge_kernel<3, true>) : launch(composite_merge_kernel<3, false>));
            if (rc) return rc;
            STNERF_CHECK_LAUNCH("composite");
        }
        return STNERF_OK;
    }
    // ---- the `order` parity output and layers of more than 192 samples: the LDS-staged kernel (every ray on its own)
    const int wpb = plan.wpb[0], lds = (int)plan.lds[0];
    if (lds > 64 * 1024)
        if (const int rc = reserve_dynamic_lds(reinterpret_cast<const void*>(composite_kernel), lds, "composite")) return rc;
    CompositeArgs a{t, reinterpret_cast<const float4*>(raw), mask, n, l, S, *params_host, layer_out, mixed_out,
                    weights, order, wpb, floor_pow2(S), nullptr};
    int64_t blocks = (n + wpb - 1) / wpb;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(composite_kernel, dim3((unsigned)blocks), dim3(wpb * 64), lds, as_stream(stream), a);
    STNERF_CHECK_LAUNCH("composite");
    return STNERF_OK;
}

extern "C" int stnerf_resample(const float* t, const float* weights, int64_t n, int l, int n1, int n2, const float* u,
                               uint64_t seed, int64_t ray_index_base, int64_t ray_index_stripe, int64_t ray_index_period,
                               const float* rays, int ray_stride,
                               const stnerf_layer_edit* edits_host, const float* pivot_host, const uint8_t* mask, float* t_fine,
                               float* xyz_fine, float* z_new, int32_t* inds, float* cdf, stnerf_stream_t stream) {
    STNERF_REQUIRE(t && weights && rays && t_fine, "resample: null pointer");
    STNERF_REQUIRE(n >= 0 && l >= 1 && l <= STNERF_MAX_LAYERS && n1 >= 3 && n2 >= 0 && ray_stride >= 6,
                   "resample: bad shape n=%lld l=%d n1=%d n2=%d", (long long)n, l, n1, n2);
    STNERF_REQUIRE_WINDOW("resample", ray_index_stripe, ray_index_period);
    if (n == 0) return STNERF_OK;
    ResampleArgs a;
    a.t = t; a.weights = weights; a.n = n; a.l = l; a.n1 = n1; a.n2 = n2; a.u = u; a.seed = seed;
    a.win = RayWindow{ray_index_base, ray_index_stripe, ray_index_period}; a.rays = rays; a.ray_stride = ray_stride;
    fill_edit_args(a.ed, edits_host, pivot_host, l);
    a.mask = mask;
    a.t_fine = t_fine; a.xyz_fine = xyz_fine; a.z_new = z_new; a.inds = inds; a.cdf_out = cdf;
    const int lds = 4 * resample_lds_floats(n1, n2) * (int)sizeof(float);
    // (the +inf padding of the branch-free searches rounds three of the arrays up to powers of two: 512+512 samples need
    // 73.7 KB for the block's four waves -- above the 64 KB a kernel gets without asking, inside the CU's 160 KB)
    STNERF_REQUIRE(lds <= 160 * 1024, "resample: %d+%d samples per ray exceed the LDS budget", n1, n2);
    int64_t blocks = (n * l + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    LaunchTimer timer(PROF_RESAMPLE, 0, n, n1 + n2, (int64_t)l * (8ll * n1 + (xyz_fine ? 16ll : 4ll) * (n1 + n2)) + 24,
                      as_stream(stream));
    const bool plain = !u && !z_new && !inds && !cdf && !a.ed.any;
    const dim3 grid((unsigned)blocks), block(256);
    int reserve_rc = STNERF_OK;
    auto launch = [&](auto kernel) {
        if (lds > 64 * 1024) reserve_rc = reserve_dynamic_lds(reinterpret_cast<const void*>(kernel), lds, "resample");
        if (reserve_rc == STNERF_OK) hipLaunchKernelGGL(kernel, grid, block, lds, as_stream(stream), a);
    };
    const bool exact = plain && n2 == 64 && (n1 == 64 || n1 == 128);
    if (exact) n1 == 64 ? launch(resample_kernel<1, true, true>) : launch(resample_kernel<2, true, true>);
    else if (n1 <= 64) plain ? launch(resample_kernel<1, true, false>) : launch(resample_kernel<1, false, false>);
    else if (n1 <= 128) plain ? launch(resample_kernel<2, true, false>) : launch(resample_kernel<2, false, false>);
    else if (n1 <= 256) plain ? launch(resample_kernel<4, true, false>) : launch(resample_kernel<4, false, false>);
    else launch(resample_kernel<0, false, false>);
    if (reserve_rc) return reserve_rc;
    STNERF_CHECK_LAUNCH("resample");
    return STNERF_OK;
}
