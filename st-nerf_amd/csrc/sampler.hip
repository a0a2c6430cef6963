// Ray generation, ray/box slab test, stratified coarse sampler, mask compaction.
//
// These are HBM-bound elementwise kernels (16 B written per sample).  They are compiled with
// -ffp-contract=off and written as separate IEEE mul/add/div so that, given the same jitter, the
// sample depths, points and hit masks are BIT-identical to the reference's ATen CPU arithmetic
// (SURVEY.md section 8c "exactness classes").
//
// Reference: utils/render_helpers.py:42-128, layers/RaySamplePoint.py:8-107.
#include "common.h"

namespace stnerf {

// ------------------------------------------------------------------------------------- a1 + a2
struct RayGenArgs {
    float kinv[9];
    float T[16];
    float frame_ids[STNERF_MAX_LAYERS + 1];
};

__global__ void generate_rays_kernel(RayGenArgs a, int w, RayWindow win, int64_t n, int n_frame_cols,
                                     float* __restrict__ rays, int ray_stride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t pix = global_ray(win, i);
    const float u = (float)(pix % w);  // column: integer pixel centre, render_helpers.py:96-102
    const float v = (float)(pix / w);  // row
    // K^-1 [u, v, 1]  (:105), normalised (:108)
    float cx = a.kinv[0] * u + a.kinv[1] * v + a.kinv[2];
    float cy = a.kinv[3] * u + a.kinv[4] * v + a.kinv[5];
    float cz = a.kinv[6] * u + a.kinv[7] * v + a.kinv[8];
    const float nrm = sqrtf(cx * cx + cy * cy + cz * cz);
    cx = cx / nrm;
    cy = cy / nrm;
    cz = cz / nrm;
    // rotate by T[:3,:3] (:109-114); origin = T[:3,3] (:116-117)
    float* r = rays + i * ray_stride;
    r[0] = a.T[3];
    r[1] = a.T[7];
    r[2] = a.T[11];
    r[3] = a.T[0] * cx + a.T[1] * cy + a.T[2] * cz;
    r[4] = a.T[4] * cx + a.T[5] * cy + a.T[6] * cz;
    r[5] = a.T[8] * cx + a.T[9] * cy + a.T[10] * cz;
    for (int c = 0; c < n_frame_cols; ++c) r[6 + c] = a.frame_ids[c];  // ray_dataset.py:276-281
}

// ------------------------------------------------------------------------------------- a5
// (far, near) of one ray against one 8-corner box.  b = 24 floats, corner-major.
__device__ __forceinline__ void intersect_box(const float o[3], const float d[3], const float* __restrict__ b,
                                              float& far_t, float& near_t) {
    constexpr float kEps = 2.220446049250313e-16f;  // np.finfo(float).eps in fp32, RaySamplePoint.py:17-22
    // plane coordinate + axis per face, in the reference's column order (left,right,front,back,bottom,up)
    const float plane[6] = {b[0 * 3 + 0], b[6 * 3 + 0], b[0 * 3 + 1], b[6 * 3 + 1], b[0 * 3 + 2], b[6 * 3 + 2]};
    constexpr int axis[6] = {0, 0, 1, 1, 2, 2};
    // rectangle test: (lo corner, hi corner, two axes), RaySamplePoint.py:34-51
    constexpr int lo[6] = {0, 1, 0, 3, 0, 4};
    constexpr int hi[6] = {7, 6, 5, 6, 2, 6};
    constexpr int ax0[6] = {1, 1, 0, 0, 0, 0};
    constexpr int ax1[6] = {2, 2, 2, 2, 1, 1};
    float best = 0.f, second = 0.f;
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        const float t = (plane[f] - o[axis[f]]) / (d[axis[f]] + kEps);
        const float p0 = t * d[ax0[f]] + o[ax0[f]];
        const float p1 = t * d[ax1[f]] + o[ax1[f]];
        const bool inside = (p0 >= b[lo[f] * 3 + ax0[f]]) && (p0 <= b[hi[f] * 3 + ax0[f]]) &&
                            (p1 >= b[lo[f] * 3 + ax1[f]]) && (p1 <= b[hi[f] * 3 + ax1[f]]);
        const float v = inside ? t : -1000.0f;  // :53-59
        if (f == 0) {
            best = v;
            second = -3.0e38f;
        } else if (v > best) {
            second = best;
            best = v;
        } else if (v > second) {
            second = v;
        }
    }
    far_t = best;     // topk(2)[:,0], :60-62
    near_t = second;  // topk(2)[:,1]
}

__global__ void intersect_kernel(const float* __restrict__ rays, int64_t n, int ray_stride,
                                 const float* __restrict__ boxes, int64_t box_ray_stride, int l,
                                 float* __restrict__ far_near) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * l) return;
    const int64_t ray = e / l;
    const int layer = (int)(e - ray * l);
    const float* r = rays + ray * ray_stride;
    const float o[3] = {r[0], r[1], r[2]};
    const float d[3] = {r[3], r[4], r[5]};
    float f, nr;
    intersect_box(o, d, boxes + ray * box_ray_stride + layer * 24, f, nr);
    far_near[e * 2 + 0] = f;
    far_near[e * 2 + 1] = nr;
}

// ------------------------------------------------------------------------------------- a6
// One thread per (ray, layer, sample): consecutive threads write consecutive t / xyz elements.
__global__ void sample_coarse_kernel(const float* __restrict__ rays, int64_t n, int ray_stride,
                                     const float* __restrict__ boxes, int64_t box_ray_stride, int l, int n1,
                                     const float* __restrict__ jitter, uint64_t seed, RayWindow win,
                                     EditArgs ed, float* __restrict__ t_out, float* __restrict__ xyz_out,
                                     uint8_t* __restrict__ mask_out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per_ray = (int64_t)l * n1;
    if (e >= n * per_ray) return;
    const int64_t ray = e / per_ray;
    const int rem = (int)(e - ray * per_ray);
    const int layer = rem / n1;
    const int k = rem - layer * n1;
    const float* r = rays + ray * ray_stride;
    const float o[3] = {r[0], r[1], r[2]};
    const float d[3] = {r[3], r[4], r[5]};
    float far_t, near_t;
    intersect_box(o, d, boxes + ray * box_ray_stride + layer * 24, far_t, near_t);
    float start = near_t;
    if (layer == 0 && start <= 0.f) start = 0.f;  // RaySamplePoint.py:93-95
    const float width = (far_t - start) / (float)n1;  // :100
    const float xi = jitter ? jitter[((int64_t)layer * n + ray) * n1 + k]
                            : philox_uniform(seed, (uint64_t)global_ray(win, ray), (uint32_t)layer, 0u, (uint32_t)k);
    const float t = ((float)k + xi) * width + start;  // :102
    t_out[e] = t;
    if (xyz_out) {
        float x = t * d[0] + o[0], y = t * d[1] + o[1], z = t * d[2] + o[2];  // :103
        if (ed.any) unedit_point(x, y, z, ed.e[layer], ed.pivot);
        xyz_out[e * 3 + 0] = x;
        xyz_out[e * 3 + 1] = y;
        xyz_out[e * 3 + 2] = z;
    }
    // bit 0: the reference's ray_mask (:105).  Bit 1 (a hint for the compositor, see include/stnerf.h): the ray misses the box
    // altogether -- start = end = -1000 (:53-62), bin width 0, every depth of the layer is exactly -1000
    if (k == 0) mask_out[ray * l + layer] = (fabsf(width) > 1e-5f ? 1 : 0) | ((width == 0.f && start == -1000.0f) ? 2 : 0);
}

// Same arithmetic, G = 4 or 2 consecutive samples of one (ray, layer) per thread: one slab test per group, vector
// stores for t and points that leave as contiguous 16-byte stores.  Used when n1 % G == 0 (every group is then
// 4G-byte aligned): G = 4 for the usual 64 / 128 samples, G = 2 for the 90 of configs/config_taekwondo.yml.
template <int G>
__global__ void sample_coarse_kernel_xg(const float* __restrict__ rays, int64_t n, int ray_stride,
                                        const float* __restrict__ boxes, int64_t box_ray_stride, int l, int n1,
                                        const float* __restrict__ jitter, uint64_t seed, RayWindow win,
                                        EditArgs ed, float* __restrict__ t_out, float* __restrict__ xyz_out,
                                        uint8_t* __restrict__ mask_out) {
    __shared__ __attribute__((aligned(16))) float xyz_stage[4 * 64 * 3 * G];  // 4 waves x 64 groups x 3G floats
    const int64_t g_raw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // group of G samples
    const int gpl = n1 / G;                                            // groups per (ray, layer)
    const int64_t per_ray = (int64_t)l * gpl;
    const bool valid = g_raw < n * per_ray;      // lanes past the end keep running: they help with the stores below
    const int64_t g = valid ? g_raw : n * per_ray - 1;
    const int64_t ray = g / per_ray;
    const int rem = (int)(g - ray * per_ray);
    const int layer = rem / gpl;
    const int k0 = (rem - layer * gpl) * G;
    const float* r = rays + ray * ray_stride;
    const float o[3] = {r[0], r[1], r[2]};
    const float d[3] = {r[3], r[4], r[5]};
    float far_t, near_t;
    intersect_box(o, d, boxes + ray * box_ray_stride + layer * 24, far_t, near_t);
    float start = near_t;
    if (layer == 0 && start <= 0.f) start = 0.f;
    const float width = (far_t - start) / (float)n1;
    float xi[G];
    if (jitter) {
        const float* jp = jitter + ((int64_t)layer * n + ray) * n1 + k0;
        if (G == 4) {
            const float4 j4 = *reinterpret_cast<const float4*>(jp);
            xi[0] = j4.x; xi[1] = j4.y; xi[G - 2] = j4.z; xi[G - 1] = j4.w;
        } else {
            const float2 j2 = *reinterpret_cast<const float2*>(jp);
            xi[0] = j2.x; xi[1] = j2.y;
        }
    } else {
        const uint64_t gray = (uint64_t)global_ray(win, ray);
#pragma unroll
        for (int j = 0; j < G; ++j) xi[j] = philox_uniform(seed, gray, (uint32_t)layer, 0u, (uint32_t)(k0 + j));
    }
    float tv[G], px[3 * G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const float t = ((float)(k0 + j) + xi[j]) * width + start;
        tv[j] = t;
        float x = t * d[0] + o[0], y = t * d[1] + o[1], z = t * d[2] + o[2];
        if (ed.any) unedit_point(x, y, z, ed.e[layer], ed.pivot);
        px[3 * j + 0] = x;
        px[3 * j + 1] = y;
        px[3 * j + 2] = z;
    }
    const int64_t e = (ray * l + layer) * n1 + k0;
    if (valid) {
        if (G == 4) *reinterpret_cast<float4*>(t_out + e) = make_float4(tv[0], tv[1], tv[G - 2], tv[G - 1]);
        else *reinterpret_cast<float2*>(t_out + e) = make_float2(tv[0], tv[1]);
        if (k0 == 0) mask_out[ray * l + layer] = (fabsf(width) > 1e-5f ? 1 : 0) | ((width == 0.f && start == -1000.0f) ? 2 : 0);   // (bit 1: see sample_coarse_kernel)
    }
    if (xyz_out) {
        // A wave's 64 groups are consecutive, so its points are 768 G bytes of contiguous output.  Written straight from
        // the registers every store instruction would touch 64 pieces at a 12 G-byte stride; transposed through LDS every
        // store instruction writes 1 KB contiguous.
        const int lane = threadIdx.x & 63;
        float* stage = xyz_stage + (threadIdx.x >> 6) * (64 * 3 * G);
#pragma unroll
        for (int j = 0; j < 3 * G; j += 2)
            *reinterpret_cast<float2*>(stage + lane * 3 * G + j) = make_float2(px[j], px[j + 1]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int64_t g0 = g_raw - lane;                           // first group of this wave
        const int64_t live = n * per_ray - g0;                     // groups of this wave that exist (<= 0: none)
        const int nf = (int)(live < 64 ? (live > 0 ? live : 0) : 64) * 3 * G;   // floats to write
        float* dstf = xyz_out + g0 * 3 * G;                        // 16-byte aligned: g0 is a multiple of 64
        float4* dst = reinterpret_cast<float4*>(dstf);
        const float4* src = reinterpret_cast<const float4*>(stage);
        for (int q = lane; q < nf / 4; q += 64) dst[q] = src[q];
        if (lane < (nf & 3)) dstf[(nf & ~3) + lane] = stage[(nf & ~3) + lane];   // odd tail of the very last wave
    }
}

// ------------------------------------------------------------------------------------- compaction
// grid.y = layer.  Order inside a block is preserved; blocks append in arrival order.
__global__ void compact_rays_kernel(const uint8_t* __restrict__ mask, int64_t n, int l, int32_t* __restrict__ ray_list,
                                    int32_t* __restrict__ ray_count) {
    __shared__ int wave_base[4];
    __shared__ int block_base;
    const int layer = blockIdx.y;
    const int64_t ray = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool hit = ray < n && (mask[ray * l + layer] & 1) != 0;
    const unsigned long long ball = __ballot(hit);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int before = __popcll(ball & ((1ull << lane) - 1ull));
    if (lane == 0) wave_base[wave] = __popcll(ball);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int i = 0; i < 4; ++i) {
            const int c = wave_base[i];
            wave_base[i] = tot;
            tot += c;
        }
        block_base = tot ? atomicAdd(&ray_count[layer], tot) : 0;
    }
    __syncthreads();
    if (hit) ray_list[(int64_t)layer * n + block_base + wave_base[wave] + before] = (int32_t)ray;
}

}  // namespace stnerf

using namespace stnerf;

extern "C" int stnerf_generate_rays(const float* Kinv_host, const float* T_host, int h, int w, int64_t first_ray,
                                    int64_t ray_index_stripe, int64_t ray_index_period, int64_t n,
                                    const float* frame_ids_host, int n_frame_cols, float* rays, int ray_stride,
                                    stnerf_stream_t stream) {
    STNERF_REQUIRE(Kinv_host && T_host && rays, "generate_rays: null pointer");
    STNERF_REQUIRE_WINDOW("generate_rays", ray_index_stripe, ray_index_period);
    const int64_t last = n > 0 ? global_ray(RayWindow{first_ray, ray_index_stripe, ray_index_period}, n - 1) : first_ray;
    STNERF_REQUIRE(h > 0 && w > 0 && n >= 0 && first_ray >= 0 && last < (int64_t)h * w,
                   "generate_rays: rays [%lld .. %lld] (%lld of them) outside a %dx%d view", (long long)first_ray,
                   (long long)last, (long long)n, h, w);
    STNERF_REQUIRE(n_frame_cols >= 0 && n_frame_cols <= STNERF_MAX_LAYERS + 1 && ray_stride >= 6 + n_frame_cols,
                   "generate_rays: bad frame-id columns %d / stride %d", n_frame_cols, ray_stride);
    STNERF_REQUIRE(n_frame_cols == 0 || frame_ids_host, "generate_rays: frame_ids is null");
    if (n == 0) return STNERF_OK;
    RayGenArgs a;
    for (int i = 0; i < 9; ++i) a.kinv[i] = Kinv_host[i];
    for (int i = 0; i < 16; ++i) a.T[i] = T_host[i];
    for (int i = 0; i < STNERF_MAX_LAYERS + 1; ++i) a.frame_ids[i] = i < n_frame_cols ? frame_ids_host[i] : 0.f;
    const int bs = 256;
    hipLaunchKernelGGL(generate_rays_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, as_stream(stream), a, w,
                       RayWindow{first_ray, ray_index_stripe, ray_index_period}, n, n_frame_cols, rays, ray_stride);
    STNERF_CHECK_LAUNCH("generate_rays");
    return STNERF_OK;
}

extern "C" int stnerf_intersect(const float* rays, int64_t n, int ray_stride, const float* boxes,
                                int64_t box_ray_stride, int l, float* far_near, stnerf_stream_t stream) {
    STNERF_REQUIRE(rays && boxes && far_near, "intersect: null pointer");
    STNERF_REQUIRE(n >= 0 && ray_stride >= 6 && l >= 1 && l <= STNERF_MAX_LAYERS, "intersect: bad shape");
    STNERF_REQUIRE(box_ray_stride == 0 || box_ray_stride >= (int64_t)l * 24, "intersect: bad box stride");
    if (n == 0) return STNERF_OK;
    const int bs = 256;
    const int64_t tot = n * l;
    hipLaunchKernelGGL(intersect_kernel, dim3((unsigned)((tot + bs - 1) / bs)), dim3(bs), 0, as_stream(stream), rays, n,
                       ray_stride, boxes, box_ray_stride, l, far_near);
    STNERF_CHECK_LAUNCH("intersect");
    return STNERF_OK;
}

extern "C" int stnerf_sample_coarse(const float* rays, int64_t n, int ray_stride, const float* boxes,
                                    int64_t box_ray_stride, int l, int n1, const float* jitter, uint64_t seed,
                                    int64_t ray_index_base, int64_t ray_index_stripe, int64_t ray_index_period,
                                    const stnerf_layer_edit* edits_host,
                                    const float* pivot_host, float* t, float* xyz, uint8_t* mask,
                                    stnerf_stream_t stream) {
    STNERF_REQUIRE(rays && boxes && t && mask, "sample_coarse: null pointer");
    STNERF_REQUIRE(n >= 0 && ray_stride >= 6 && l >= 1 && l <= STNERF_MAX_LAYERS && n1 >= 1,
                   "sample_coarse: bad shape n=%lld stride=%d l=%d n1=%d", (long long)n, ray_stride, l, n1);
    STNERF_REQUIRE(box_ray_stride == 0 || box_ray_stride >= (int64_t)l * 24, "sample_coarse: bad box stride");
    STNERF_REQUIRE_WINDOW("sample_coarse", ray_index_stripe, ray_index_period);
    if (n == 0) return STNERF_OK;
    const RayWindow win{ray_index_base, ray_index_stripe, ray_index_period};
    EditArgs ed;
    fill_edit_args(ed, edits_host, pivot_host, l);
    const int bs = 256;
    const int64_t tot = n * l * n1;
    STNERF_REQUIRE((tot + bs - 1) / bs < (1ll << 31), "sample_coarse: chunk too large");
    LaunchTimer timer(PROF_SAMPLE_COARSE, 0, n, n1, (xyz ? 16ll : 4ll) * l * n1 + 4ll * ray_stride + l, as_stream(stream));
    const int G = n1 % 4 == 0 ? 4 : n1 % 2 == 0 ? 2 : 1;
    const uintptr_t am = (uintptr_t)(4 * G - 1);   // vector loads / stores of one group
    const bool aligned = G > 1 && ((uintptr_t)t & am) == 0 && (!xyz || ((uintptr_t)xyz & 15) == 0) &&
                         (!jitter || ((uintptr_t)jitter & am) == 0);
    if (aligned) {
        const int64_t groups = tot / G;
        const dim3 grid((unsigned)((groups + bs - 1) / bs));
        if (G == 4)
            hipLaunchKernelGGL(sample_coarse_kernel_xg<4>, grid, dim3(bs), 0, as_stream(stream), rays, n, ray_stride, boxes,
                               box_ray_stride, l, n1, jitter, seed, win, ed, t, xyz, mask);
        else
            hipLaunchKernelGGL(sample_coarse_kernel_xg<2>, grid, dim3(bs), 0, as_stream(stream), rays, n, ray_stride, boxes,
                               box_ray_stride, l, n1, jitter, seed, win, ed, t, xyz, mask);
        STNERF_CHECK_LAUNCH("sample_coarse");
        return STNERF_OK;
    }
    hipLaunchKernelGGL(sample_coarse_kernel, dim3((unsigned)((tot + bs - 1) / bs)), dim3(bs), 0, as_stream(stream), rays,
                       n, ray_stride, boxes, box_ray_stride, l, n1, jitter, seed, win, ed, t, xyz, mask);
    STNERF_CHECK_LAUNCH("sample_coarse");
    return STNERF_OK;
}

extern "C" int stnerf_compact_rays(const uint8_t* mask, int64_t n, int l, int32_t* ray_list, int32_t* ray_count,
                                   stnerf_stream_t stream) {
    STNERF_REQUIRE(mask && ray_list && ray_count, "compact_rays: null pointer");
    STNERF_REQUIRE(n >= 0 && n < (1ll << 31) && l >= 1 && l <= STNERF_MAX_LAYERS, "compact_rays: bad shape");
    if (n == 0) return STNERF_OK;
    const int bs = 256;
    hipLaunchKernelGGL(compact_rays_kernel, dim3((unsigned)((n + bs - 1) / bs), (unsigned)l), dim3(bs), 0,
                       as_stream(stream), mask, n, l, ray_list, ray_count);
    STNERF_CHECK_LAUNCH("compact_rays");
    return STNERF_OK;
}
