// Backward of the compositor: d(colour, depth, acc) of every per-layer composite and of the depth-merged
// one -> d(rgb, sigma) of the networks' raw outputs.  What loss.backward() of engine/layered_trainer.py:192-282 does to
// layers/render_layer.py:8-58 (gen_weight, VolumeRenderer.forward) and to the merge gather of
// modeling/layered_rfrender.py:425-429 (coarse) / :587-592 (fine) through ATen, as ONE kernel: a wave owns a ray, recomputes
// each stream's alpha / transmittance front to back (inclusive-product scan across the lanes, carried from 64-sample block to
// block), then walks it back to front with the suffix sum the transmittance product's gradient needs.  The sampler and
// sample_pdf are detached in the reference (:314-315, :460-461): depths get no gradient.
//
// Per stream of n samples in composite order (a layer's own list, or the merged list through `order`):
//   s = relu(sigma'), e = exp(-s delta), alpha = 1 - e, om = (1 - alpha) + 1e-10, T_k = prod_{j<k} om_j, w = alpha T
//   C = sum w sigmoid(c), D = sum w t, A = sum w                                  (render_layer.py:11-15, :47-49)
//   gw_k      = gC . sigmoid(c_k) + gD t_k + gA
//   d c_k     = gC * w_k sigmoid(c_k) (1 - sigmoid(c_k))
//   d alpha_k = gw_k T_k - (sum_{j>k} gw_j w_j) / om_k          (torch.cumprod's backward without zeros: reversed cumsum / input)
//   d sigma_k = d alpha_k * delta_k e_k * [sigma' > 0] * f_k    (f: what the in-place density edits did to sigma: 0 where it
//                                                                 was overwritten, sigma_scale where it was multiplied)
// sigma' and f follow the forward's edit rules (csrc/render.hip, a10): modeling/layered_rfrender.py:414-422, :538-547, :564-566,
// :575-576, and the fine stage's merged-only `t < near` cut (:605).
#include <math.h>

#include "common.h"

namespace stnerf {
namespace {

__device__ __forceinline__ void bwd_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// the forward's exponential and sigmoid (csrc/render.hip): the recomputed alpha / sigmoid are the forward's own
__device__ __forceinline__ float bwd_exp_neg(float x) { return __builtin_amdgcn_exp2f(x * -1.44269504088896340736f); }
__device__ __forceinline__ float bwd_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + bwd_exp_neg(x)); }

// inclusive scans over the 64 lanes (ds_bpermute shuffles: this kernel runs once per training step on a few thousand rays)
__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_up(v, d, 64);
        if (lane >= d) v *= o;
    }
    return v;
}
__device__ __forceinline__ float wave_incl_suffix_sum(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_down(v, d, 64);
        if (lane + d < 64) v += o;
    }
    return v;
}

struct CompositeBwdArgs {
    const float* t;        // [n][l][S]
    const float4* raw;     // [n][l][S]
    const uint8_t* mask;   // [n][l] or null
    const int32_t* order;  // [n][l*S] merged position -> source index (stnerf_composite's `order`), null: no merged stream
    const float* g_layer;  // [n][l][5] or null
    const float* g_mixed;  // [n][5] or null
    float4* d_raw;         // [n][l][S]
    int64_t n;
    int l, S;
    int waves_per_block;
    stnerf_composite_params p;
};

__global__ void composite_bwd_kernel(CompositeBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int LS = a.l * a.S;
    // per workgroup: the per-layer edit table; per wave: dacc[LS] float4 | T, om, gww, q, gw [LS] floats each
    float* tab = reinterpret_cast<float*>(smem_raw);   // thr[16] | scale[16]
    const int per_wave = (LS * 36 + 15) & ~15;          // (16-byte aligned regions: dacc is read and written as float4)
    unsigned char* mine = smem_raw + 128 + (size_t)wave * per_wave;
    float4* dacc = reinterpret_cast<float4*>(mine);
    float* sT = reinterpret_cast<float*>(mine + (size_t)LS * 16);
    float* sOm = sT + LS;
    float* sGww = sOm + LS;
    float* sQ = sGww + LS;
    float* sGw = sQ + LS;
    if (threadIdx.x < STNERF_MAX_LAYERS) {
        tab[threadIdx.x] = a.p.use_threshold[threadIdx.x] != 0 ? a.p.threshold[threadIdx.x] : -INFINITY;
        tab[16 + threadIdx.x] = a.p.sigma_scale[threadIdx.x];
    }
    __syncthreads();
    const bool fine = a.p.fine != 0;
    const bool cut_neg_on = !fine && a.p.cut_negative_t != 0;
    const float nearv = a.p.near, border = a.p.border;
    unsigned ev1 = 0, ev2 = 0;
    for (int i = 0; i < a.l; ++i) {
        if (a.p.evaluated[i] == 2) ev2 |= 1u << i;
        else if (a.p.evaluated[i] != 0) ev1 |= 1u << i;
    }
    const int64_t rays_per_iter = (int64_t)gridDim.x * a.waves_per_block;
    for (int64_t ray = (int64_t)blockIdx.x * a.waves_per_block + wave; ray < a.n; ray += rays_per_iter) {
        unsigned mask_bits = 0xffffu;
        if (a.mask) {
            const int v = lane < a.l ? a.mask[ray * a.l + lane] : 0;
            mask_bits = (unsigned)__ballot((v & 1) != 0);
        }
        const unsigned have_m = ev2 | (ev1 & mask_bits);   // layers whose raw outputs this ray's composites read
        const float* tr = a.t + ray * LS;
        const float4* rr = a.raw + ray * LS;
        for (int e = lane; e < LS; e += 64) dacc[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        bwd_wave_sync();
        // streams 0 .. l-1: the layers' own composites; stream l: the merged one
        for (int stream = 0; stream <= a.l; ++stream) {
            const bool merged = stream == a.l;
            if (merged ? (a.g_mixed == nullptr || a.order == nullptr) : (a.g_layer == nullptr || !(have_m >> stream & 1u))) continue;
            const float* g = merged ? a.g_mixed + ray * 5 : a.g_layer + (ray * a.l + stream) * 5;
            const float gC0 = g[0], gC1 = g[1], gC2 = g[2], gD = g[3], gA = g[4];
            if (gC0 == 0.f && gC1 == 0.f && gC2 == 0.f && gD == 0.f && gA == 0.f) continue;
            const int ns = merged ? LS : a.S;
            const int32_t* ord = merged ? a.order + ray * LS : nullptr;
            // ---- front to back: alpha, transmittance, the colour gradient
            float carry = 1.f;
            for (int m0 = 0; m0 < ns; m0 += 64) {
                const int m = m0 + lane;
                const bool in = m < ns;
                int src = 0;
                float tm = 0.f, delta = 0.f, om = 1.f, alpha = 0.f, q = 0.f, gw = 0.f;
                float4 sg = make_float4(0.f, 0.f, 0.f, 0.f);
                bool have = false;
                if (in) {
                    src = merged ? ord[m] : stream * a.S + m;
                    tm = tr[src];
                    if (m + 1 < ns) delta = tr[merged ? ord[m + 1] : src + 1] - tm;
                    else delta = border;
                    const int layer = merged ? src / a.S : stream;
                    have = (have_m >> layer & 1u) != 0;
                    float v = 0.f, f = 0.f;
                    if (have) {
                        const float4 r = rr[src];
                        v = r.w;
                        f = 1.f;
                        if (cut_neg_on && layer > 0 && tm < 0.f) { v = 0.f; f = 0.f; }              // :414
                        if (v < tab[layer]) { v = 0.f; f = 0.f; }                                   // :416-418, :538-547, :564-566
                        v = v * tab[16 + layer];                                                    // :575-576
                        f = f * tab[16 + layer];
                        if (!fine && layer == 0 && tm < nearv) { v = 0.f; f = 0.f; }                // :422
                        if (merged && fine && tm < nearv) { v = 0.f; f = 0.f; }                     // :605
                        if (a.p.rgb_activated) sg = make_float4(r.x, r.y, r.z, 0.f);
                        else sg = make_float4(bwd_sigmoid(r.x), bwd_sigmoid(r.y), bwd_sigmoid(r.z), 0.f);
                    }
                    const float s = fmaxf(v, 0.f);
                    const float ex = bwd_exp_neg(s * delta);
                    alpha = 1.f - ex;
                    om = (1.f - alpha) + 1e-10f;
                    q = v > 0.f ? delta * ex * f : 0.f;
                    gw = gC0 * sg.x + gC1 * sg.y + gC2 * sg.z + gD * tm + gA;
                }
                const float incl = wave_incl_prod(om, lane);
                float excl = __shfl_up(incl, 1, 64);
                if (lane == 0) excl = 1.f;
                const float T = carry * excl;
                carry = carry * __shfl(incl, 63, 64);
                if (in) {
                    const float w = alpha * T;
                    sT[m] = T;
                    sOm[m] = om;
                    sGww[m] = gw * w;
                    sQ[m] = q;
                    sGw[m] = gw;
                    if (have) {
                        float4 d = dacc[src];
                        if (a.p.rgb_activated) {   // (raw already holds sigmoid(rgb): the gradient is with respect to it)
                            d.x += gC0 * w;
                            d.y += gC1 * w;
                            d.z += gC2 * w;
                        } else {
                            d.x += gC0 * (w * sg.x * (1.f - sg.x));
                            d.y += gC1 * (w * sg.y * (1.f - sg.y));
                            d.z += gC2 * (w * sg.z * (1.f - sg.z));
                        }
                        dacc[src] = d;
                    }
                }
            }
            bwd_wave_sync();
            // ---- back to front: the suffix sum of gw w behind every sample, d alpha, d sigma
            float tail = 0.f;
            for (int m0 = ((ns - 1) / 64) * 64; m0 >= 0; m0 -= 64) {
                const int m = m0 + lane;
                const bool in = m < ns;
                const float x = in ? sGww[m] : 0.f;
                const float incl = wave_incl_suffix_sum(x, lane);
                const float behind = (incl - x) + tail;
                tail += __shfl(incl, 0, 64);
                if (in) {
                    const float q = sQ[m];
                    if (q != 0.f) {
                        const float dalpha = sGw[m] * sT[m] - behind / sOm[m];
                        const int src = merged ? ord[m] : stream * a.S + m;
                        dacc[src].w += dalpha * q;
                    }
                }
            }
            bwd_wave_sync();
        }
        float4* dr = a.d_raw + ray * LS;
        for (int e = lane; e < LS; e += 64) dr[e] = dacc[e];
        bwd_wave_sync();
    }
}

}  // namespace
}  // namespace stnerf

using namespace stnerf;

extern "C" int stnerf_composite_bwd(const float* t, const float* raw, const uint8_t* mask, const int32_t* order, int64_t n, int l,
                                    int S, const stnerf_composite_params* params_host, const float* g_layer, const float* g_mixed,
                                    float* d_raw, stnerf_stream_t stream) {
    STNERF_REQUIRE(t && raw && params_host && d_raw, "composite_bwd: null pointer");
    STNERF_REQUIRE(n >= 0 && l >= 1 && l <= STNERF_MAX_LAYERS && S >= 1, "composite_bwd: bad shape n=%lld l=%d S=%d", (long long)n, l, S);
    STNERF_REQUIRE((((uintptr_t)raw | (uintptr_t)d_raw) & 15) == 0, "composite_bwd: raw and d_raw must be 16-byte aligned");
    STNERF_REQUIRE(!g_mixed || order, "composite_bwd: the merged composite's gradient needs the forward's `order`");
    if (n == 0) return STNERF_OK;
    const int64_t per_wave = ((int64_t)l * S * 36 + 15) & ~(int64_t)15;   // (as in the kernel: rounded up to 16 bytes)
    STNERF_REQUIRE(per_wave + 128 <= 160 * 1024 - 1024, "composite_bwd: %d samples per ray need %lld B of LDS per wave", l * S,
                   (long long)per_wave);
    int wpb = (int)((64 * 1024 - 128) / per_wave);
    wpb = wpb < 1 ? 1 : (wpb > 4 ? 4 : wpb);
    const int lds = (int)(128 + wpb * per_wave);
    if (lds > 64 * 1024)
        if (const int rc = reserve_dynamic_lds(reinterpret_cast<const void*>(composite_bwd_kernel), lds, "composite_bwd")) return rc;
    CompositeBwdArgs a{t, reinterpret_cast<const float4*>(raw), mask, order, g_layer, g_mixed, reinterpret_cast<float4*>(d_raw), n, l, S, wpb,
                       *params_host};
    int64_t blocks = (n + wpb - 1) / wpb;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)blocks), dim3(wpb * 64), lds, as_stream(stream), a);
    STNERF_CHECK_LAUNCH("composite_bwd");
    return STNERF_OK;
}
