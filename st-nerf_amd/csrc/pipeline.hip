// stnerf_render_rays: the whole chunk pipeline of LayeredRFRender.forward (modeling/layered_rfrender.py:141-734)
// behind ONE C-ABI call -- coarse sampler -> mask compaction -> [MotionNet] -> SpaceNets -> density edits +
// per-layer composite + depth merge + merged composite -> inverse-CDF resample -> [MotionNet] -> fine SpaceNets ->
// composite.  Host-side sequencing only: every stage is one of the kernels behind the op-level entry points,
// enqueued on the caller's stream into a caller-provided workspace (no allocation, no synchronisation).
#include <string.h>

#include "common.h"

using namespace stnerf;

namespace {

struct Carve {
    char* base;
    int64_t off, cap;
    template <class T>
    T* take(int64_t count) {
        off = (off + 255) & ~int64_t(255);
        T* p = reinterpret_cast<T*>(base + off);
        off += count * (int64_t)sizeof(T);
        return p;
    }
};

// sizes of the workspace regions for n rays (shared by the size query and the carve)
struct Plan {
    int64_t t_c, xyz_c, raw_c, w_c, t_f, xyz_f, raw_f, list, count, flags, raybias;
    // Once the resampler has read t_c / w_c the whole coarse block [t_c | xyz_c | raw_c | w_c] is dead: the fine network outputs
    // (raw_f, the largest fine buffer) are written over it.  shared = the larger of the two.  64 + 64 samples: 9 n1 = 576 floats
    // of coarse block per (ray, layer) against 4 S = 512 of raw_f -- the workspace goes from 1728 to 1216 floats (- 30 %).
    static int64_t pad(int64_t floats) { return (floats + 63) & ~int64_t(63); }   // 256-byte granules, as Carve::take
    int64_t coarse_block() const { return pad(t_c) + pad(xyz_c) + pad(raw_c) + pad(w_c); }
    int64_t shared() const { return coarse_block() > pad(raw_f) ? coarse_block() : pad(raw_f); }
};
Plan make_plan(int64_t n, int l, int n1, int n2, int only_coarse) {
    Plan p;
    const int64_t S = n1 + n2;
    p.t_c = n * l * n1;
    p.xyz_c = n * l * n1 * 3;
    p.raw_c = n * l * n1 * 4;
    p.w_c = only_coarse ? 0 : n * l * n1;
    p.t_f = only_coarse ? 0 : n * l * S;
    p.xyz_f = only_coarse ? 0 : n * l * S * 3;
    p.raw_f = only_coarse ? 0 : n * l * S * 4;
    p.list = (int64_t)l * n;
    p.flags = n;                      // one byte per ray: the compositor's scratch
    p.count = STNERF_MAX_LAYERS + 2;  // hit-ray counts + the work-queue heads of the two network stages
    p.raybias = (int64_t)l * n * 128;  // per (layer, ray): the C operands of rgb_net.1 (mlp_raybias.hip), reused by both stages
    return p;
}

}  // namespace

extern "C" int64_t stnerf_render_workspace_bytes(int64_t n, int l, int n1, int n2, int only_coarse) {
    if (n < 0 || l < 1 || l > STNERF_MAX_LAYERS || n1 < 3 || n2 < 0) {
        set_error("render_workspace_bytes: bad shape");
        return STNERF_EINVAL;
    }
    const Plan p = make_plan(n, l, n1, n2, only_coarse);
    const int64_t floats = p.shared() + p.t_f + p.xyz_f + p.raybias;
    return floats * 4 + (p.list + p.count) * 4 + p.flags + 16 * 256;
}

// The mask the caller gets back is the reference's ray_mask (0 / 1): the sampler's "missed" hint (bit 1) served the compositor and
// the resampler inside the call and is cleared on the way out (a C caller testing `mask != 0` must not see it).
static __global__ void clear_mask_hints_kernel(uint8_t* mask, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) mask[i] &= 1;
}
static int clear_mask_hints(uint8_t* mask, int64_t count, hipStream_t stream) {
    hipLaunchKernelGGL(clear_mask_hints_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, mask, count);
    STNERF_CHECK_LAUNCH("render_rays (mask)");
    return STNERF_OK;
}

extern "C" int stnerf_render_rays(const float* rays, int64_t n, const float* boxes, int64_t box_ray_stride,
                                  const stnerf_nets* nets, const stnerf_render_params* p, const float* jitter,
                                  const float* u, void* workspace, int64_t workspace_bytes, float* mixed_fine,
                                  float* mixed_coarse, float* layer_fine, float* layer_coarse, uint8_t* mask,
                                  stnerf_stream_t stream) {
    STNERF_REQUIRE(rays && boxes && nets && p && workspace && mask, "render_rays: null pointer");
    STNERF_REQUIRE(mixed_coarse && layer_coarse, "render_rays: coarse outputs are required");
    STNERF_REQUIRE(p->only_coarse || (mixed_fine && layer_fine), "render_rays: fine outputs are required");
    const int l = p->l, n1 = p->n1, n2 = p->n2, S = n1 + n2, rs = p->ray_stride;
    STNERF_REQUIRE(n >= 0 && l >= 1 && l <= STNERF_MAX_LAYERS && n1 >= 3 && n2 >= 0, "render_rays: bad shape");
    STNERF_REQUIRE(rs >= (p->retiming ? 6 + l : 7), "render_rays: ray stride %d too small for the frame-id columns", rs);
    STNERF_REQUIRE(p->precision == 0 || p->precision == 2 || p->precision == 3, "render_rays: unknown precision %d", p->precision);
    STNERF_REQUIRE(nets->bkgd && (p->only_coarse || nets->bkgd_fine), "render_rays: background network missing");
    STNERF_REQUIRE(!p->bkgd_use_deform_time || nets->motion[0], "render_rays: bkgd_time_deform_net missing");
    STNERF_REQUIRE(!p->bkgd_use_space_time || p->use_space_time,
                   "render_rays: BKGD_USE_SPACE_TIME needs USE_SPACE_TIME (the reference passes the background its "
                   "frame id only then, layered_rfrender.py:382-390)");
    for (int i = 1; i < l; ++i) {
        if (!p->shown[i]) continue;
        STNERF_REQUIRE(nets->space[i] && (p->only_coarse || nets->space_fine[i]), "render_rays: SpaceNet of layer %d missing", i);
        STNERF_REQUIRE(!p->use_deform_time || nets->motion[i], "render_rays: MotionNet of layer %d missing", i);
    }
    const int64_t need = stnerf_render_workspace_bytes(n, l, n1, n2, p->only_coarse);
    STNERF_REQUIRE(workspace_bytes >= need, "render_rays: workspace of %lld B, need %lld", (long long)workspace_bytes,
                   (long long)need);
    if (n == 0) return STNERF_OK;

    const Plan pl = make_plan(n, l, n1, n2, p->only_coarse);
    Carve ws{static_cast<char*>(workspace), 0, workspace_bytes};
    float* shared = ws.take<float>(pl.shared());
    Carve cb{reinterpret_cast<char*>(shared), 0, pl.shared() * 4};   // the coarse block inside the shared region
    float* t_c = cb.take<float>(pl.t_c);
    float* xyz_c = cb.take<float>(pl.xyz_c);
    float* raw_c = cb.take<float>(pl.raw_c);
    float* w_c = cb.take<float>(pl.w_c);
    float* raw_f = shared;            // over the coarse block: first written by the fine stage, after the resampler read t_c / w_c
    float* t_f = ws.take<float>(pl.t_f);
    float* xyz_f = ws.take<float>(pl.xyz_f);
    int32_t* ray_list = ws.take<int32_t>(pl.list);
    int32_t* ray_count = ws.take<int32_t>(pl.count);
    uint8_t* ray_flags = ws.take<uint8_t>(pl.flags);
    float* ray_bias = ws.take<float>(pl.raybias);
    hipStream_t st = as_stream(stream);
    int rc;

    // ---- coarse samples + hit masks (+ un-edit), then the per-layer lists of hit rays
    rc = stnerf_sample_coarse(rays, n, rs, boxes, box_ray_stride, l, n1, jitter, p->seed, p->ray_index_base,
                              p->ray_index_stripe, p->ray_index_period,
                              p->has_edits ? p->edits_coarse : nullptr, p->pivot, t_c, xyz_c, mask, stream);
    if (rc) return rc;
    if (hipMemsetAsync(ray_count, 0, sizeof(int32_t) * (STNERF_MAX_LAYERS + 2), st) != hipSuccess) {
        set_error("render_rays: hipMemsetAsync failed");
        return STNERF_ELAUNCH;
    }
    rc = stnerf_compact_rays(mask, n, l, ray_list, ray_count, stream);
    if (rc) return rc;

    // ---- one network stage: deform + evaluate every shown layer on its hit rays (:340-418 / :495-576)
    auto stage = [&](float* xyz, float* raw, int ns, bool fine) -> int {
        const int64_t xs = (int64_t)l * ns * 3, ws_ = (int64_t)l * ns * 4;
        if (p->precision == 0 || p->precision == 3) {
            // exact f32 / bf16x3: ONE persistent launch over every shown layer (csrc/stage_entry.hip); deformed performers
            // first, the background last
            stnerf_stage_layer sl[STNERF_MAX_LAYERS];
            int ns_l = 0;
            for (int pass = 0; pass < 2; ++pass) {
                for (int i = 0; i < l; ++i) {
                    if (i > 0 && !p->shown[i]) continue;
                    const bool deform = i == 0 ? p->bkgd_use_deform_time != 0 : p->use_deform_time != 0;
                    if ((pass == 0) != deform) continue;
                    const bool timed = (i > 0 ? 1 : p->bkgd_use_space_time) && p->use_space_time;
                    stnerf_stage_layer& e = sl[ns_l++];
                    e.space = i == 0 ? (fine ? nets->bkgd_fine : nets->bkgd) : (fine ? nets->space_fine[i] : nets->space[i]);
                    e.motion = deform ? nets->motion[i] : nullptr;
                    e.ray_list = i == 0 ? nullptr : ray_list + (int64_t)i * n;
                    e.ray_count = i == 0 ? nullptr : ray_count + i;
                    e.xyz = xyz + (int64_t)i * ns * 3;
                    e.raw = raw + (int64_t)i * ns * 4;
                    e.times = (timed || deform) ? rays + (p->retiming ? 6 + i : 6) : nullptr;
                    e.use_time = timed ? 1 : 0;
                    e.motion_flags = i == 0 ? STNERF_MOTION_PLAIN_TIME : 0;
                }
            }
            set_launch_tag(fine ? 1 : 0);
            const int r2 = stnerf_mlp_stage(sl, ns_l, n, ns, rays + 3, rs, rs, xs, ws_,
                                            (p->deep_rgb ? STNERF_STAGE_DEEP_RGB : 0) | STNERF_STAGE_SIGMOID_RGB | (p->precision == 3 ? STNERF_STAGE_BF16X3 : 0),
                                            reinterpret_cast<uint32_t*>(ray_count + STNERF_MAX_LAYERS + (fine ? 1 : 0)), ray_bias, stream);
            set_launch_tag(-1);
            return r2;
        }
        for (int i = 0; i < l; ++i) {
            if (i == 0 ? !p->bkgd_use_deform_time : !p->use_deform_time) continue;
            if (i > 0 && !p->shown[i]) continue;  // a hidden layer's points are never consumed
            set_launch_tag(i);
            // performers: time_deform_nets[i-1] on the hit rays, fractional frame ids lerp the encodings (:340-356);
            // background: bkgd_time_deform_net on every ray, MotionNet(input_time=False) (:358-367)
            const float* times = rays + (p->retiming ? 6 + i : 6);
            const int32_t* lst = i == 0 ? nullptr : ray_list + (int64_t)i * n;
            const int32_t* cnt = i == 0 ? nullptr : ray_count + i;
            const int flags = STNERF_MOTION_ADD_TO_XYZ | (i == 0 ? STNERF_MOTION_PLAIN_TIME : 0);
            const int r2 = stnerf_motionnet_fwd(nets->motion[i], n, ns, lst, cnt, xyz + (int64_t)i * ns * 3, xs, times,
                                                      rs, nullptr, 0, flags, stream);
            if (r2) return r2;
        }
        for (int i = 0; i < l; ++i) {
            if (i > 0 && !p->shown[i]) continue;
            set_launch_tag(i);
            const void* net = i == 0 ? (fine ? nets->bkgd_fine : nets->bkgd) : (fine ? nets->space_fine[i] : nets->space[i]);
            const bool timed = (i > 0 ? 1 : p->bkgd_use_space_time) && p->use_space_time;
            const int kind = timed ? (p->deep_rgb ? STNERF_NET_SPACE_TIME_DEEP : STNERF_NET_SPACE_TIME)
                                   : (p->deep_rgb ? STNERF_NET_SPACE_DEEP : STNERF_NET_SPACE);
            const float* times = timed ? rays + (p->retiming ? 6 + i : 6) : nullptr;
            const int32_t* lst = i == 0 ? nullptr : ray_list + (int64_t)i * n;
            const int32_t* cnt = i == 0 ? nullptr : ray_count + i;
            const int r2 = stnerf_spacenet_fwd(kind, net, n, ns, lst, cnt, xyz + (int64_t)i * ns * 3, xs, rays + 3, rs,
                                                     times, rs, raw + (int64_t)i * ns * 4, ws_, ray_bias + (int64_t)i * n * 128, stream);
            if (r2) return r2;
        }
        set_launch_tag(-1);
        return STNERF_OK;
    };
    rc = stage(xyz_c, raw_c, n1, false);
    if (rc) return rc;

    // ---- coarse: density edits, per-layer + merged composite (:414-448)
    stnerf_composite_params cp;
    memset(&cp, 0, sizeof(cp));
    cp.border = p->border;
    cp.near = p->near;
    cp.fine = 0;
    cp.cut_negative_t = 1;
    cp.rgb_activated = p->precision == 0 || p->precision == 3;  // the persistent stage kernels store sigmoid(rgb)
    for (int i = 0; i < STNERF_MAX_LAYERS; ++i) {
        cp.sigma_scale[i] = 1.f;
        cp.evaluated[i] = i < l ? (i == 0 ? 2 : p->shown[i]) : 1;  // background: every ray, mask or not (:382-392)
        cp.use_threshold[i] = (p->retiming && i >= 1) ? 1 : 0;  // :416-418 (performers, retiming only)
        cp.threshold[i] = p->density_threshold;
    }
    rc = stnerf_composite(t_c, raw_c, mask, n, l, n1, &cp, layer_coarse, mixed_coarse, p->only_coarse ? nullptr : w_c,
                          nullptr, ray_flags, stream);
    if (rc) return rc;
    if (p->only_coarse) return clear_mask_hints(mask, n * l, as_stream(stream));

    // ---- resample + fine points (:459-475), fine networks, fine composite (:538-606)
    rc = stnerf_resample(t_c, w_c, n, l, n1, n2, u, p->seed, p->ray_index_base, p->ray_index_stripe, p->ray_index_period,
                         rays, rs,
                         p->has_edits ? p->edits_fine : nullptr, p->pivot, mask, t_f, xyz_f, nullptr, nullptr, nullptr, stream);
    if (rc) return rc;
    rc = stage(xyz_f, raw_f, S, true);
    if (rc) return rc;
    cp.fine = 1;
    cp.cut_negative_t = 0;
    for (int i = 0; i < STNERF_MAX_LAYERS; ++i) {
        cp.use_threshold[i] = p->retiming ? 1 : 0;                       // :538-547 (bkgd), :564-566 (performers)
        cp.threshold[i] = i == 0 ? p->bkgd_density_threshold : p->density_threshold;
    }
    if (l > 2) cp.sigma_scale[2] = p->alpha;                             // :575-576
    rc = stnerf_composite(t_f, raw_f, mask, n, l, S, &cp, layer_fine, mixed_fine, nullptr, nullptr, ray_flags, stream);
    if (rc) return rc;
    return clear_mask_hints(mask, n * l, as_stream(stream));
}
