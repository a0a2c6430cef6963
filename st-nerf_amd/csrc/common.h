// Shared host/device helpers for libstnerf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "stnerf.h"

namespace stnerf {

void set_error(const char* fmt, ...);

static inline hipStream_t as_stream(stnerf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Opt a kernel into `bytes` of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize; needed above 64 KiB).  The
// attribute belongs to the (device, function) pair: the cache is keyed by the calling thread's current device and is
// guarded by a mutex, so a process that renders on several GPUs (or from several host threads) opts in on each.
int reserve_dynamic_lds(const void* kernel, int bytes, const char* what);

// Optional launch profiler (stnerf_profile_begin / _end): while enabled, every kernel launch of the op-level
// entry points is bracketed by a HIP event pair recorded on the launch stream.
enum ProfKernel { PROF_SPACENET = 0, PROF_MOTIONNET = 1, PROF_COMPOSITE = 2, PROF_RESAMPLE = 3, PROF_SAMPLE_COARSE = 4, PROF_MLP_STAGE = 5 };
bool profiling_enabled();
void set_launch_tag(int tag);  // e.g. the layer a pipeline launch works on; -1 = none
struct LaunchTimer {            // RAII: records start on construction, stop + bookkeeping on destruction
    LaunchTimer(int kernel, int kind, int64_t n_rays, int ns, int64_t bytes_per_ray, hipStream_t stream);
    ~LaunchTimer();
    void* rec_;
    hipStream_t stream_;
};

#define STNERF_REQUIRE(cond, ...)                \
    do {                                         \
        if (!(cond)) {                           \
            ::stnerf::set_error(__VA_ARGS__);    \
            return STNERF_EINVAL;                \
        }                                        \
    } while (0)

#define STNERF_CHECK_LAUNCH(what)                                                        \
    do {                                                                                 \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess) {                                                          \
            ::stnerf::set_error("%s: %s", what, hipGetErrorString(e_));                  \
            return STNERF_ELAUNCH;                                                       \
        }                                                                                \
    } while (0)

// Ray window (include/stnerf.h): local ray i of a call -> global ray index of the view.
struct RayWindow {
    int64_t first, stripe, period;
};
__host__ __device__ inline int64_t global_ray(const RayWindow& w, int64_t i) {
    if (w.stripe <= 0) return w.first + i;
    const int64_t s = i / w.stripe;
    return w.first + s * w.period + (i - s * w.stripe);
}
#define STNERF_REQUIRE_WINDOW(what, stripe, period)                                                              \
    STNERF_REQUIRE((stripe) >= 0 && ((stripe) == 0 || (period) >= (stripe)), what ": bad ray window (stripe %lld, period %lld)", \
                   (long long)(stripe), (long long)(period))

// Scene constants passed by value to kernels (small, wave-uniform -> SGPRs).
struct EditArgs {
    stnerf_layer_edit e[STNERF_MAX_LAYERS];
    float pivot[3];
    int32_t any;  // 0: no edit at all
};

static inline void fill_edit_args(EditArgs& a, const stnerf_layer_edit* edits, const float* pivot, int l) {
    a.any = 0;
    for (int i = 0; i < STNERF_MAX_LAYERS; ++i) {
        a.e[i].shift[0] = a.e[i].shift[1] = a.e[i].shift[2] = 0.f;
        a.e[i].scale = 1.f;
        a.e[i].has_shift = a.e[i].has_scale = 0;
    }
    a.pivot[0] = a.pivot[1] = a.pivot[2] = 0.f;
    if (edits) {
        for (int i = 0; i < l; ++i) {
            a.e[i] = edits[i];
            if (edits[i].has_shift || edits[i].has_scale) a.any = 1;
        }
    }
    if (pivot) { a.pivot[0] = pivot[0]; a.pivot[1] = pivot[1]; a.pivot[2] = pivot[2]; }
}

#if defined(__HIPCC__)
// ---- Philox4x32-10 counter-based RNG (Salmon et al. 2011): stateless, so the draw of
// (ray, layer, sample) does not depend on how rays are chunked or sharded over GPUs.
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
        const uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += W0;
        k.y += W1;
    }
    return c;
}

// Uniform in [0,1) on the 2^-24 grid (the grid torch.rand uses for fp32).
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t ray, uint32_t layer, uint32_t stream,
                                                uint32_t sample) {
    const uint4 c = make_uint4((uint32_t)ray, (uint32_t)(ray >> 32), layer | (stream << 16), sample >> 2);
    const uint4 r = philox4x32_10(c, make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    const uint32_t w = (sample & 3) == 0 ? r.x : (sample & 3) == 1 ? r.y : (sample & 3) == 2 ? r.z : r.w;
    return (float)(w >> 8) * 5.9604644775390625e-8f;  // 2^-24
}

// Inverse of the box edit, applied to a sample point.  modeling/layered_rfrender.py:293-303 / :467-475.
// Arithmetic kept as separate IEEE ops (this file set is compiled with -ffp-contract=off).
__device__ __forceinline__ void unedit_point(float& x, float& y, float& z, const stnerf_layer_edit& e,
                                             const float* pivot) {
    if (e.has_shift) {
        x = x - e.shift[0];
        y = y - e.shift[1];
        z = z - e.shift[2];
    }
    if (e.has_scale) {
        x = (x - pivot[0]) / e.scale + pivot[0];
        y = (y - pivot[1]) / e.scale + pivot[1];
        z = (z - pivot[2]) / e.scale + pivot[2];
    }
}
#endif

}  // namespace stnerf
