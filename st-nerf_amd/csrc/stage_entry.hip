// stnerf_mlp_stage: one network stage of the pipeline as ONE persistent launch (a device-side queue of 128-row work
// items over every listed layer; the MotionNet of a deformed layer runs in front of its SpaceNet on the same rows).
// Two arithmetics behind it, both in the sample-split organisation (a wave owns 32 samples, activations in registers):
//   csrc/mlp_wave.hip    exact f32 (v_mfma_f32_32x32x2_f32)                       -- the default
//   csrc/mlp_bf16x3.hip  split-bf16: three bf16 pieces per fp32 operand, six MFMAs -- STNERF_STAGE_BF16X3
// Reference: modeling/layered_rfrender.py:340-418 (coarse), :495-576 (fine).
#include <string.h>

#include "mlp_stage.h"
#include "mlp_bf16x3.h"

using namespace stnerf;

// layers[i] describes slot i of the queue (heavier, deformed layers first); `queue` is a zeroed uint32 on the device.
extern "C" int stnerf_mlp_stage(const stnerf_stage_layer* layers, int n_layers, int64_t n_rays, int ns, const float* dirs,
                                int64_t dirs_ray_stride, int64_t times_ray_stride, int64_t xyz_ray_stride,
                                int64_t raw_ray_stride, int flags, uint32_t* queue, float* ray_bias, stnerf_stream_t stream) {
    STNERF_REQUIRE(layers && dirs && queue && ray_bias, "mlp_stage: null pointer");
    STNERF_REQUIRE(((uintptr_t)ray_bias & 15) == 0, "mlp_stage: ray_bias must be 16-byte aligned");
    STNERF_REQUIRE(n_layers >= 1 && n_layers <= STNERF_MAX_LAYERS && n_rays >= 0 && ns >= 1, "mlp_stage: bad shape");
    STNERF_REQUIRE((raw_ray_stride & 3) == 0, "mlp_stage: raw ray stride must be a multiple of 4 floats");
    STNERF_REQUIRE((flags & ~(STNERF_STAGE_DEEP_RGB | STNERF_STAGE_SIGMOID_RGB | STNERF_STAGE_BF16X3)) == 0, "mlp_stage: unknown flags %d", flags);
    if (n_rays == 0) return STNERF_OK;
    const int deep_rgb = (flags & STNERF_STAGE_DEEP_RGB) != 0;
    const bool bf16x3 = (flags & STNERF_STAGE_BF16X3) != 0;
    StageArgs a;
    memset(&a, 0, sizeof(a));
    a.sigmoid_rgb = (flags & STNERF_STAGE_SIGMOID_RGB) != 0;
    // (the profiler's record of the stage covers the per-ray prologues too: their work is part of the networks' FLOPs)
    LaunchTimer timer(PROF_MLP_STAGE, deep_rgb | (bf16x3 ? 2 : 0), n_rays, ns, 0, as_stream(stream));
    for (int i = 0; i < n_layers; ++i) {
        const stnerf_stage_layer& s = layers[i];
        STNERF_REQUIRE(s.space && s.xyz && s.raw, "mlp_stage: layer %d: null pointer", i);
        // (a bf16x3 blob is streamed by 16-byte LDS-DMA from 1 KB-aligned sections: the blob itself must be 1 KB aligned)
        const uintptr_t amask = bf16x3 ? 1023 : 15;
        STNERF_REQUIRE(((uintptr_t)s.space & amask) == 0 && ((uintptr_t)s.raw & 15) == 0 && (!s.motion || ((uintptr_t)s.motion & amask) == 0),
                       "mlp_stage: layer %d: packed weights must be %d-byte aligned, raw 16-byte aligned", i, (int)amask + 1);
        STNERF_REQUIRE(!(s.use_time || s.motion) || s.times, "mlp_stage: layer %d needs its frame-id column", i);
        a.layer[i] = StageLayer{static_cast<const float*>(s.space), static_cast<const float*>(s.motion), s.ray_list,
                                s.ray_count, s.xyz, s.raw, s.times, s.use_time, s.motion_flags,
                                ray_bias + (int64_t)i * n_rays * 128};
        // rgb_net.1's direction / time columns once per ray of this layer (mlp_raybias.hip; exact f32 for both arithmetics:
        // a bf16x3 blob starts with the network's exact-f32 blob)
        const int kind = s.use_time ? (deep_rgb ? STNERF_NET_SPACE_TIME_DEEP : STNERF_NET_SPACE_TIME)
                                    : (deep_rgb ? STNERF_NET_SPACE_DEEP : STNERF_NET_SPACE);
        if (const int rc = launch_ray_bias(kind, static_cast<const float*>(s.space), n_rays, s.ray_list, s.ray_count, dirs,
                                           dirs_ray_stride, s.times, times_ray_stride, ray_bias + (int64_t)i * n_rays * 128,
                                           as_stream(stream)))
            return rc;
    }
    a.n_layers = n_layers;
    a.ns = ns;
    a.n_rays = n_rays;
    a.xyz_ray_stride = xyz_ray_stride;
    a.raw_ray_stride = raw_ray_stride;
    a.dirs_ray_stride = dirs_ray_stride;
    a.times_ray_stride = times_ray_stride;
    a.dirs = dirs;
    a.queue = queue;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return bf16x3 ? launch_bf16x3_stage(a, deep_rgb != 0, cus, as_stream(stream)) : launch_wave_stage(a, deep_rgb != 0, cus, as_stream(stream));
}
