// rgb_net.1 once per RAY.  The first layer of SpaceNet's colour branch, Linear(256 + 27 [+ 21] -> 128)
// (modeling/spacenet.py:80-86,141-151), reads the 256 backbone features of a sample and the encodings of the ray's
// direction and frame id -- the same 27 (+ 21) numbers for every sample of the ray (:115,118 repeat them).  Their part
// of the layer,
//     c[f] = bias[f] + sum_k W[f][256 + k] * relu(enc(dir, time))[k],
// is therefore evaluated here, once per (layer, hit ray), and the MLP kernels take row c[ray] as the C operand of the
// layer's first MFMA instead of the bias: 16 % of the layer's multiply-adds (1.3 % of the network's) and the 22 sin/cos
// pairs per SAMPLE of the fused kernels go away.
// This kernel defines the arithmetic for every exact-f32 MLP kernel (mlp.hip, mlp_wave.hip): c starts from
// the bias and takes the encoded features in index order with one fmaf each; the layer then adds the 256 backbone
// features in the kernels' usual k order.
#include "mlp_common.h"

namespace stnerf {

constexpr int RB_RAYS = 16;     // rays per workgroup
constexpr int RB_THREADS = 256;  // two groups of 128 threads (= 128 outputs), 8 rays each

__global__ __launch_bounds__(RB_THREADS) void ray_bias_kernel(const float* __restrict__ net, int use_time, int deep,
                                                              int64_t n_rays, const int32_t* __restrict__ ray_list,
                                                              const int32_t* __restrict__ ray_count,
                                                              const float* __restrict__ dirs, int64_t dirs_ray_stride,
                                                              const float* __restrict__ times, int64_t times_ray_stride,
                                                              float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float enc[RB_RAYS][48];
    __shared__ int64_t ray_of[RB_RAYS];
    const SpaceLayout L = space_layout(use_time != 0, deep != 0);
    int64_t cnt = n_rays;
    if (ray_count) {
        const int64_t c = *ray_count;
        cnt = c < cnt ? c : cnt;
    }
    const int64_t slot0 = (int64_t)blockIdx.x * RB_RAYS;
    if (slot0 >= cnt) return;  // (uniform)
    const int K = use_time ? 48 : 27;
    // ---- relu(PE_4(dir)) (27), relu(PE_10(time)) (21) of the workgroup's rays: the values the fused kernels used to
    // write per sample; one (ray, feature) per thread and pass
    for (int e = threadIdx.x; e < RB_RAYS * 48; e += RB_THREADS) {
        const int r = e / 48, f = e - r * 48;
        const int64_t slot = slot0 + r;
        if (slot >= cnt) continue;
        const int64_t ray = ray_list ? (int64_t)ray_list[slot] : slot;
        if (f == 0) ray_of[r] = ray;
        float v = 0.f;
        if (f < 27) {
            const float* d = dirs + ray * dirs_ray_stride;
            if (f < 3) {
                v = fmaxf(d[f], 0.f);
            } else {
                const int gq = f - 3, fq = gq / 6, w = gq - fq * 6, dmn = w < 3 ? w : w - 3;
                float sn, cs;
                sincos_pe(d[dmn] * (float)(1 << fq), sn, cs);
                v = relu_bits(w < 3 ? sn : cs);
            }
        } else if (f < K) {
            const float tv = times[ray * times_ray_stride];
            if (f == 27) {
                v = fmaxf(tv, 0.f);
            } else {
                const int gq = f - 28, fq = gq >> 1;
                float sn, cs;
                sincos_pe(tv * (float)(1 << fq), sn, cs);
                v = relu_bits((gq & 1) ? cs : sn);
            }
        }
        enc[r][f] = v;
    }
    __syncthreads();
    // ---- c[f] = bias[f], then one fmaf per encoded feature, in index order; a thread keeps its output's 48 weights
    const int g = threadIdx.x >> 7, f = threadIdx.x & 127;
    const float4* w4 = reinterpret_cast<const float4*>(net + L.w_rgb1) + (int64_t)64 * 128 + f;  // quad row 64 + q, column f
    float4 w[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) w[q] = 4 * q < K ? w4[q * 128] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float bias = net[L.b_rgb1 + f];
    for (int r = g; r < RB_RAYS; r += 2) {
        if (slot0 + r >= cnt) break;
        const float4* e4 = reinterpret_cast<const float4*>(enc[r]);
        float c = bias;
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            if (4 * q < K) {
                const float4 e = e4[q];
                c = fmaf(e.x, w[q].x, c);
                if (4 * q + 1 < K) c = fmaf(e.y, w[q].y, c);
                if (4 * q + 2 < K) c = fmaf(e.z, w[q].z, c);
                if (4 * q + 3 < K) c = fmaf(e.w, w[q].w, c);
            }
        }
        out[ray_of[r] * 128 + f] = c;
    }
}

int launch_ray_bias(int kind, const float* net, int64_t n_rays, const int32_t* ray_list, const int32_t* ray_count,
                    const float* dirs, int64_t dirs_ray_stride, const float* times, int64_t times_ray_stride, float* out,
                    hipStream_t stream) {
    if (n_rays == 0) return STNERF_OK;
    const int64_t blocks = (n_rays + RB_RAYS - 1) / RB_RAYS;
    hipLaunchKernelGGL(ray_bias_kernel, dim3((unsigned)blocks), dim3(RB_THREADS), 0, stream, net,
                       STNERF_NET_USES_TIME(kind) ? 1 : 0, STNERF_NET_IS_DEEP(kind) ? 1 : 0, n_rays, ray_list, ray_count, dirs,
                       dirs_ray_stride, times, times_ray_stride, out);
    STNERF_CHECK_LAUNCH("rgb_ray_bias");
    return STNERF_OK;
}

}  // namespace stnerf

using namespace stnerf;

extern "C" int stnerf_rgb_ray_bias(int kind, const void* packed, int64_t n_rays, const int32_t* ray_list,
                                   const int32_t* ray_count, const float* dirs, int64_t dirs_ray_stride,
                                   const float* times, int64_t times_ray_stride, float* out, stnerf_stream_t stream) {
    STNERF_REQUIRE(STNERF_NET_IS_SPACE(kind), "rgb_ray_bias: bad kind %d", kind);
    STNERF_REQUIRE(packed && dirs && out, "rgb_ray_bias: null pointer");
    STNERF_REQUIRE(!STNERF_NET_USES_TIME(kind) || times, "rgb_ray_bias: net takes time but times is null");
    STNERF_REQUIRE(n_rays >= 0 && ((uintptr_t)out & 15) == 0, "rgb_ray_bias: bad shape / out must be 16-byte aligned");
    return launch_ray_bias(kind, static_cast<const float*>(packed), n_rays, ray_list, ray_count, dirs, dirs_ray_stride, times,
                           times_ray_stride, out, as_stream(stream));
}
