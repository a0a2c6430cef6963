// Fused positional-encoding + MLP kernels for SpaceNet and MotionNet on gfx950 (fp32 MFMA).
//
// Design (DESIGN.md section "MLP kernel"):
//   * One workgroup = 4 waves (one per SIMD) = one tile of TM = 128 samples.  The tile's activations
//     live in LDS for the whole network: `act` [64 quads][TM] float4 (feature quad-major, so a
//     sample's 4 consecutive features are one 16-byte word) + `enc` [16 quads][TM] float4 for the
//     positional encodings (skip connection / direction+time inputs).  128 KiB + 32 KiB = all 160 KiB
//     of the CU's LDS -> 1 workgroup per CU, nothing ever spills to HBM between layers (the
//     reference writes every 256-wide activation to memory: modeling/spacenet.py:136-151).
//   * Each layer is computed transposed, out^T[feature][sample] = W[feature][k] * act^T[k][sample],
//     with v_mfma_f32_32x32x2_f32: A operand = weights, B operand = activations.  With that
//     orientation a lane's 16 accumulator registers are 4 runs of 4 consecutive FEATURES of one
//     sample = exactly one float4 of the quad-major LDS layout: the epilogue is ds_write_b128 and
//     the next layer's B operand is ds_read_b128, both conflict-free, no transposes anywhere.
//   * Weights are pre-packed [K/4][N][4] (stnerf_pack_net) so a lane's A operand for 4 consecutive
//     MFMAs is one coalesced 16-byte global load; every workgroup streams the same ~1.9 MB per
//     network from L2.  The k index is permuted consistently for A and B (lane half h takes
//     k = 8s + 4h + {0..3}), which only reorders the fp32 accumulation.
//   * exact fp32 arithmetic (f32 MFMA == an fmaf chain); sin/cos are < 1 ulp Cody-Waite + minimax kernels
//     (arguments reach 2^9 * |x|, utils/dimension_kernel.py:20-27): sincos_pe below.
//
// Reference: modeling/spacenet.py:16-160, modeling/motion_net.py:7-71, utils/dimension_kernel.py:3-73.
#include <stdlib.h>
#include <string.h>


#include "mlp_blocks.h"

namespace stnerf {



template <int TM, int NW, bool USE_TIME, bool DEEP = false>
__global__ __launch_bounds__(NW * 64, (TM == 128 ? NW / 4 : 2)) void spacenet_kernel(SpaceArgs a) {
    constexpr int NTHREADS = NW * 64;
    constexpr int NPARTS = NTHREADS / TM;  // threads cooperating on one sample in the VALU phases
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    float4* act = smem;             // [64][TM]
    float4* enc = smem + 64 * TM;   // [16][TM]
    float* encf = reinterpret_cast<float*>(enc);
    float* scratch = reinterpret_cast<float*>(enc + 12 * TM);  // quads 12..15: 16*TM floats
    const SpaceLayout L = space_layout(USE_TIME, DEEP);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int part = __builtin_amdgcn_readfirstlane(tid / TM);
    const int s = tid & (TM - 1);
    const int64_t rows = worklist_rows(a.wl);
    const int ns = a.wl.ns;
    PH_DECL
    // step-0 weight fragments, prefetched one layer ahead (layer 0 of the first tile here, of every later
    // tile by the previous tile's last layer)
    WFrag<WaveSplit<TM, NW, 256>::NFB> wA, wB;
    WFrag<WaveSplit<TM, NW, 128>::NFB> wR, wR2;
    load_wfrag(wA, weight_lane_ptr(a.net, L.w[0], 256, WaveSplit<TM, NW, 256>::n0(wave), lane),
               a.net + L.b[0] + WaveSplit<TM, NW, 256>::n0(wave) + 4 * (lane >> 5));

    for (int64_t tile = blockIdx.x; tile * TM < rows; tile += gridDim.x) {
        PH(PH_MISC);
        // The weights/biases are loop-invariant; without this the compiler hoists ~10 layers of bias and
        // head-weight loads out of the tile loop and then spills them.  An opaque zero keeps every load
        // inside the iteration that uses it.
        int64_t opaque_zero = 0;
        asm volatile("" : "+s"(opaque_zero));
        const float* net = a.net + opaque_zero;
        // ---- row -> (ray, sample)
        const int64_t row = tile * TM + s;
        const bool valid = row < rows;
        int64_t ray = 0;
        int k = 0;
        if (valid) {
            const int64_t slot = row / ns;
            k = (int)(row - slot * ns);
            ray = a.wl.ray_list ? (int64_t)a.wl.ray_list[slot] : slot;
        }
        float* col = encf + s * 4;
        // ---- PE_10(pos) -> enc quads 0..15 (63 features + 1 zero pad); utils/dimension_kernel.py:8-33
        {
            float p[3] = {0.f, 0.f, 0.f};
            if (valid) {
                const float* src = a.xyz + ray * a.xyz_ray_stride + 3 * k;
                p[0] = src[0];
                p[1] = src[1];
                p[2] = src[2];
            }
            if (part == 0) {
#pragma unroll
                for (int dmn = 0; dmn < 3; ++dmn) ENC_AT(col, dmn) = p[dmn];
            }
            if (part == NPARTS - 1) ENC_AT(col, 63) = 0.f;  // zero pad
            for (int fq = part; fq < 10; fq += NPARTS) {
                const float freq = (float)(1 << fq);
#pragma unroll
                for (int dmn = 0; dmn < 3; ++dmn) {
                    float sn, cs;
                    sincos_pe(p[dmn] * freq, sn, cs);
                    const int fs = 3 + fq * 6 + dmn, fc = fs + 3;
                    ENC_AT(col, fs) = sn;
                    ENC_AT(col, fc) = cs;
                }
            }
        }
        PH(PH_PE);
        __syncthreads();
        PH(PH_BAR2);
        // ---- stage1 (modeling/spacenet.py:45-54)
        DENSE(TM, NW, 256, 256, net, L.w[0], L.b[0], enc, 16, nullptr, 0, act, wA, L.w[1], L.b[1], wB);
        __syncthreads();
        PH(PH_BAR2);
        DENSE(TM, NW, 256, 256, net, L.w[1], L.b[1], act, 64, nullptr, 0, act, wB, L.w[2], L.b[2], wA);
        __syncthreads();
        PH(PH_BAR2);
        DENSE(TM, NW, 256, 256, net, L.w[2], L.b[2], act, 64, nullptr, 0, act, wA, L.w[3], L.b[3], wB);
        __syncthreads();
        PH(PH_BAR2);
        DENSE(TM, NW, 256, 256, net, L.w[3], L.b[3], act, 64, nullptr, 0, act, wB, L.w[4], L.b[4], wA);
        __syncthreads();
        // ---- stage2.0 on [h, PE(pos)] (:56-57, :137)
        DENSE(TM, NW, 256, 256, net, L.w[4], L.b[4], act, 64, enc, 16, act, wA, L.w[5], L.b[5], wB);
        // enc is free now (all waves passed the barrier inside dense_layer).  rgb_net.1's direction / time columns are
        // not encoded here any more: they are part of the ray's C operand (mlp_raybias.hip); the tile keeps the ray of
        // every sample for that fetch
        int32_t* ray_of = reinterpret_cast<int32_t*>(enc);
        if (part == 0) ray_of[s] = valid ? (int32_t)ray : 0;
        PH(PH_ENC2);
        __syncthreads();
        PH(PH_BAR2);
        DENSE(TM, NW, 256, 256, net, L.w[5], L.b[5], act, 64, nullptr, 0, act, wB, L.w[6], L.b[6], wA);
        __syncthreads();
        PH(PH_BAR2);
        DENSE(TM, NW, 256, 128, net, L.w[6], L.b[6], act, 64, nullptr, 0, act, wA, L.w_rgb1, L.b_rgb1, wR);
        __syncthreads();
        PH(PH_BAR2);
        // rgb_net.1's C operands (bias + the ray's direction / time columns) on their way behind the sigma head
        RayC<WaveSplit<TM, NW, 128>::NFB, WaveSplit<TM, NW, 128>::NSB> rayc;
        load_rayc(rayc, a.raybias, ray_of, WaveSplit<TM, NW, 128>::n0(wave), WaveSplit<TM, NW, 128>::sb0(wave), lane);
        // ---- sigma = density_net(h) (:139), raw
        float sigma;
        {
            float ps[1];
            head_partial<TM, 1>(act, s, part * (64 / NPARTS), (part + 1) * (64 / NPARTS), net + L.w_sigma, 256, ps);
            scratch[part * TM + s] = ps[0];
            __syncthreads();
            sigma = net[L.b_sigma];
#pragma unroll
            for (int pp = 0; pp < NPARTS; ++pp) sigma += scratch[pp * TM + s];
        }
        PH(PH_HEAD);
        // ---- rgb_net: relu -> Linear(283|304,128) -> relu -> Linear(128,3)   (:80-86)
        // (h is already >= 0; the encodings were clamped when written)
        if constexpr (!DEEP) {
            DENSE_RAYC(TM, NW, 128, 256, net, L.w_rgb1, act, 64, act, wR, L.w[0], L.b[0], wA, rayc);  // + next tile's layer 0
        } else {  // deep_rgb (:68-79): two more 128-wide hidden layers before the 3-wide output
            DENSE_RAYC(TM, NW, 128, 128, net, L.w_rgb1, act, 64, act, wR, L.w_deep[0], L.b_deep[0], wR2, rayc);
            __syncthreads();
            DENSE(TM, NW, 128, 128, net, L.w_deep[0], L.b_deep[0], act, 32, nullptr, 0, act, wR2, L.w_deep[1], L.b_deep[1], wR);
            __syncthreads();
            DENSE(TM, NW, 128, 256, net, L.w_deep[1], L.b_deep[1], act, 32, nullptr, 0, act, wR, L.w[0], L.b[0], wA);
        }
        __syncthreads();
        {
            float ps[3];
            head_partial<TM, 3>(act, s, part * (32 / NPARTS), (part + 1) * (32 / NPARTS), net + L.w_rgb2, 128, ps);
            float* sc = scratch + NTHREADS;
            sc[(part * 3 + 0) * TM + s] = ps[0];
            sc[(part * 3 + 1) * TM + s] = ps[1];
            sc[(part * 3 + 2) * TM + s] = ps[2];
            __syncthreads();
            if (part == 0 && valid) {
                float4 o;
                o.x = net[L.b_rgb2 + 0];
                o.y = net[L.b_rgb2 + 1];
                o.z = net[L.b_rgb2 + 2];
#pragma unroll
                for (int pp = 0; pp < NPARTS; ++pp) {
                    o.x += sc[(pp * 3 + 0) * TM + s];
                    o.y += sc[(pp * 3 + 1) * TM + s];
                    o.z += sc[(pp * 3 + 2) * TM + s];
                }
                o.w = sigma;
                *reinterpret_cast<float4*>(a.raw + ray * a.raw_ray_stride + 4 * k) = o;
            }
        }
        __syncthreads();  // scratch/enc/act are rewritten by the next tile's prologue
        PH(PH_HEAD);
    }
    PH_FLUSH;
}

// ---------------------------------------------------------------------------------------------
// MotionNet
// ---------------------------------------------------------------------------------------------
template <int TM, int NW>
constexpr int motion_lds_bytes() { return 32 * TM * 16 + 3 * NW * 64 * 4; }

template <int TM, int NW>
__global__ __launch_bounds__(NW * 64, (TM == 128 ? NW / 4 : 2)) void motionnet_kernel(MotionArgs a) {
    constexpr int NTHREADS = NW * 64;
    constexpr int NPARTS = NTHREADS / TM;
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    float4* act = smem;            // [32][TM]
    float4* enc = smem;            // [22][TM]: 84 features + 4 zero pads.  ALIASES act: layer 0 is the only reader of enc and
                                   // every wave passes dense_layer's barrier before the first output quad is written, so
                                   // the tile needs 32 KiB (TM = 64) and four workgroups share a CU instead of two
    float* encf = reinterpret_cast<float*>(enc);
    float* scratch = reinterpret_cast<float*>(smem + 32 * TM);  // 3*NPARTS*TM floats
    const MotionLayout L = motion_layout();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int part = __builtin_amdgcn_readfirstlane(tid / TM);
    const int s = tid & (TM - 1);
    const int64_t rows = worklist_rows(a.wl);
    const int ns = a.wl.ns;
    PH_DECL
    WFrag<WaveSplit<TM, NW, 128>::NFB> wA, wB;  // step-0 weights, prefetched one layer ahead
    load_wfrag(wA, weight_lane_ptr(a.net, L.w[0], 128, WaveSplit<TM, NW, 128>::n0(wave), lane),
               a.net + L.b[0] + WaveSplit<TM, NW, 128>::n0(wave) + 4 * (lane >> 5));

    for (int64_t tile = blockIdx.x; tile * TM < rows; tile += gridDim.x) {
        int64_t opaque_zero = 0;  // see spacenet_kernel
        asm volatile("" : "+s"(opaque_zero));
        const float* net = a.net + opaque_zero;
        const int64_t row = tile * TM + s;
        const bool valid = row < rows;
        int64_t ray = 0;
        int k = 0;
        if (valid) {
            const int64_t slot = row / ns;
            k = (int)(row - slot * ns);
            ray = a.wl.ray_list ? (int64_t)a.wl.ray_list[slot] : slot;
        }
        float p[3] = {0.f, 0.f, 0.f};
        float tv = 0.f;
        if (valid) {
            const float* src = a.xyz + ray * a.xyz_ray_stride + 3 * k;
            p[0] = src[0];
            p[1] = src[1];
            p[2] = src[2];
            tv = a.times[ray * a.times_ray_stride];
        }
        // ---- PE_10([x,y,z,t]) with the fractional-time lerp of modeling/motion_net.py:49-60:
        // enc = (1-w) PE([xyz, floor t]) + w PE([xyz, floor t + 1]); w == 0 rows reduce to PE(input).
        {
            const float lo = (a.add_to_xyz & STNERF_MOTION_PLAIN_TIME) ? tv : floorf(tv);  // input_time=False: PE(input) as is
            const float wgt = tv - lo;
            const bool frac = wgt != 0.f;
            const float om = 1.f - wgt;
            float* col = encf + s * 4;
            auto mix = [&](float va, float vb) { return frac ? om * va + wgt * vb : va; };
            if (part == 0) {
#pragma unroll
                for (int dmn = 0; dmn < 3; ++dmn) ENC_AT(col, dmn) = mix(p[dmn], p[dmn]);
                ENC_AT(col, 3) = mix(lo, lo + 1.f);
            }
            if (part == NPARTS - 1) {
#pragma unroll
                for (int f = 84; f < 88; ++f) ENC_AT(col, f) = 0.f;
            }
            for (int fq = part; fq < 10; fq += NPARTS) {
                const float freq = (float)(1 << fq);
#pragma unroll
                for (int dmn = 0; dmn < 4; ++dmn) {
                    float sn, cs, sn2, cs2;
                    if (dmn < 3) {
                        sincos_pe(p[dmn] * freq, sn, cs);
                        sn2 = sn;
                        cs2 = cs;
                    } else {
                        sincos_pe(lo * freq, sn, cs);
                        sn2 = sn;
                        cs2 = cs;
                        if (frac) sincos_pe((lo + 1.f) * freq, sn2, cs2);
                    }
                    const int fs = 4 + fq * 8 + dmn, fc = fs + 4;
                    ENC_AT(col, fs) = mix(sn, sn2);
                    ENC_AT(col, fc) = mix(cs, cs2);
                }
            }
        }
        __syncthreads();
        DENSE(TM, NW, 128, 128, net, L.w[0], L.b[0], enc, 22, nullptr, 0, act, wA, L.w[1], L.b[1], wB);
        __syncthreads();
        DENSE(TM, NW, 128, 128, net, L.w[1], L.b[1], act, 32, nullptr, 0, act, wB, L.w[2], L.b[2], wA);
        __syncthreads();
        DENSE(TM, NW, 128, 128, net, L.w[2], L.b[2], act, 32, nullptr, 0, act, wA, L.w[3], L.b[3], wB);
        __syncthreads();
        DENSE(TM, NW, 128, 128, net, L.w[3], L.b[3], act, 32, nullptr, 0, act, wB, L.w[4], L.b[4], wA);
        __syncthreads();
        DENSE(TM, NW, 128, 128, net, L.w[4], L.b[4], act, 32, nullptr, 0, act, wA, L.w[0], L.b[0], wB);  // + next tile's layer 0
        wA = wB;
        __syncthreads();
        {
            float ps[3];
            head_partial<TM, 3>(act, s, part * (32 / NPARTS), (part + 1) * (32 / NPARTS), net + L.w_out, 128, ps);
            scratch[(part * 3 + 0) * TM + s] = ps[0];
            scratch[(part * 3 + 1) * TM + s] = ps[1];
            scratch[(part * 3 + 2) * TM + s] = ps[2];
            __syncthreads();
            if (part == 0 && valid) {
                float fl[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    fl[c] = net[L.b_out + c];
#pragma unroll
                    for (int pp = 0; pp < NPARTS; ++pp) fl[c] += scratch[(pp * 3 + c) * TM + s];
                }
                if (a.flow) {
                    float* dst = a.flow + ray * a.flow_ray_stride + 3 * k;
                    dst[0] = fl[0];
                    dst[1] = fl[1];
                    dst[2] = fl[2];
                }
                if (a.add_to_xyz & STNERF_MOTION_ADD_TO_XYZ) {
                    float* dst = a.xyz + ray * a.xyz_ray_stride + 3 * k;
                    dst[0] = p[0] + fl[0];  // modeling/layered_rfrender.py:356,510
                    dst[1] = p[1] + fl[1];
                    dst[2] = p[2] + fl[2];
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Stand-alone positional encoding (op-level API; the render path uses the fused prologues above)
// ---------------------------------------------------------------------------------------------
__global__ void encode_kernel(const float* __restrict__ x, int64_t n, int dim, int n_freq, int include_input,
                              float* __restrict__ y) {
    const int out_dim = dim * (include_input + 2 * n_freq);
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per output element
    if (e >= n * out_dim) return;
    const int64_t row = e / out_dim;
    int f = (int)(e - row * out_dim);
    const float* xr = x + row * dim;
    if (include_input) {
        if (f < dim) {
            y[e] = xr[f];
            return;
        }
        f -= dim;
    }
    const int fq = f / (2 * dim), w = f - fq * 2 * dim;
    const int d = w % dim;
    float sn, cs;
    sincos_pe(xr[d] * (float)(1 << fq), sn, cs);
    y[e] = w < dim ? sn : cs;
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
// (out,in) row-major -> [ceil(in/4) (padded to kq)][out][4], zero padded.
static void pack_linear(const float* w, int out_f, int in_f, int kq, float* dst) {
    memset(dst, 0, sizeof(float) * (size_t)kq * out_f * 4);
    for (int n = 0; n < out_f; ++n)
        for (int k = 0; k < in_f; ++k) dst[((size_t)(k >> 2) * out_f + n) * 4 + (k & 3)] = w[(size_t)n * in_f + k];
}

static int grid_for(int64_t n_rays, int ns, int tm) {
    const int64_t tiles = (n_rays * ns + tm - 1) / tm;
    return (int)(tiles < 8192 ? tiles : 8192);
}

// One tile configuration per network is compiled -- the fastest of the three measured in round 1 (128 samples x 4 waves,
// 128 x 8, 64 x 4: SpaceNet 138 / 141 / 137 TF/s, MotionNet 120 / 121 / 127 TF/s): SpaceNet 128 samples x 8 waves (two waves
// per SIMD: the VALU phases of one hide in the issue gaps of the other, 160 KiB LDS), MotionNet 64 samples x 4 waves (80 KiB
// LDS, two workgroups per CU).  The pipeline's stages do not come here (stage_entry.hip); these are the op-level entry points.

// LDS opt-in (per device, see reserve_dynamic_lds) + launch.
template <class Args>
static int launch_mlp(void (*kernel)(Args), int lds, int grid, int nthreads, stnerf_stream_t stream,
                      const Args& a, const char* what, int prof_kernel, int prof_kind) {
    if (const int rc = reserve_dynamic_lds(reinterpret_cast<const void*>(kernel), lds, what)) return rc;
    LaunchTimer timer(prof_kernel, prof_kind, a.wl.n_rays, a.wl.ns, 0, as_stream(stream));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(nthreads), lds, as_stream(stream), a);
    STNERF_CHECK_LAUNCH(what);
    return STNERF_OK;
}

}  // namespace stnerf

using namespace stnerf;

#ifdef STNERF_PHASE_PROF
extern "C" int stnerf_debug_read_phases(unsigned long long* host16, int reset) {
    if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 16) != hipSuccess) return STNERF_ELAUNCH;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) != hipSuccess) return STNERF_ELAUNCH;
    }
    return STNERF_OK;
}
#endif

extern "C" int64_t stnerf_packed_bytes(int kind) {
    switch (kind) {
        case STNERF_NET_SPACE: return space_layout(false).total * 4;
        case STNERF_NET_SPACE_TIME: return space_layout(true).total * 4;
        case STNERF_NET_SPACE_DEEP: return space_layout(false, true).total * 4;
        case STNERF_NET_SPACE_TIME_DEEP: return space_layout(true, true).total * 4;
        case STNERF_NET_MOTION: return motion_layout().total * 4;
        default: set_error("packed_bytes: unknown net kind %d", kind); return STNERF_EINVAL;
    }
}

extern "C" int stnerf_pack_net(int kind, const float* const* W, const float* const* B, int n_tensors, void* dst_host,
                               int64_t dst_bytes) {
    STNERF_REQUIRE(W && B && dst_host, "pack_net: null pointer");
    float* dst = static_cast<float*>(dst_host);
    if (STNERF_NET_IS_SPACE(kind)) {
        const bool ut = STNERF_NET_USES_TIME(kind), deep = STNERF_NET_IS_DEEP(kind);
        const SpaceLayout L = space_layout(ut, deep);
        const int nt = deep ? 12 : 10;
        STNERF_REQUIRE(n_tensors == nt, "pack_net: this SpaceNet kind takes %d tensors, got %d", nt, n_tensors);
        STNERF_REQUIRE(dst_bytes >= L.total * 4, "pack_net: dst too small");
        for (int i = 0; i < nt; ++i) STNERF_REQUIRE(W[i] && B[i], "pack_net: tensor %d is null", i);
        memset(dst, 0, (size_t)L.total * 4);
        const int in_f[7] = {63, 256, 256, 256, 319, 256, 256};
        for (int i = 0; i < 7; ++i) {
            pack_linear(W[i], 256, in_f[i], L.kq[i], dst + L.w[i]);
            memcpy(dst + L.b[i], B[i], 256 * sizeof(float));
        }
        memcpy(dst + L.w_sigma, W[7], 256 * sizeof(float));
        dst[L.b_sigma] = B[7][0];
        pack_linear(W[8], 128, 256 + 27 + (ut ? 21 : 0), L.kq_rgb1, dst + L.w_rgb1);
        memcpy(dst + L.b_rgb1, B[8], 128 * sizeof(float));
        for (int i = 0; i < 2 && deep; ++i) {
            pack_linear(W[9 + i], 128, 128, 32, dst + L.w_deep[i]);
            memcpy(dst + L.b_deep[i], B[9 + i], 128 * sizeof(float));
        }
        memcpy(dst + L.w_rgb2, W[nt - 1], 3 * 128 * sizeof(float));
        memcpy(dst + L.b_rgb2, B[nt - 1], 3 * sizeof(float));
        return STNERF_OK;
    }
    if (kind == STNERF_NET_MOTION) {
        const MotionLayout L = motion_layout();
        STNERF_REQUIRE(n_tensors == 6, "pack_net: MotionNet takes 6 tensors, got %d", n_tensors);
        STNERF_REQUIRE(dst_bytes >= L.total * 4, "pack_net: dst too small");
        for (int i = 0; i < 6; ++i) STNERF_REQUIRE(W[i] && B[i], "pack_net: tensor %d is null", i);
        memset(dst, 0, (size_t)L.total * 4);
        const int in_f[5] = {84, 128, 128, 128, 128};
        for (int i = 0; i < 5; ++i) {
            pack_linear(W[i], 128, in_f[i], L.kq[i], dst + L.w[i]);
            memcpy(dst + L.b[i], B[i], 128 * sizeof(float));
        }
        memcpy(dst + L.w_out, W[5], 3 * 128 * sizeof(float));
        memcpy(dst + L.b_out, B[5], 3 * sizeof(float));
        return STNERF_OK;
    }
    set_error("pack_net: unknown net kind %d", kind);
    return STNERF_EINVAL;
}

// ---------------------------------------------------------------------------------------------
// The exact-f32 packing ON THE DEVICE: the same blob as stnerf_pack_net from tensors that live in HBM -- what a training loop needs
// after every optimizer.step() (the host packer costs a D2H of the weights, a CPU loop and an H2D per network: 165 ms per
// iteration for the eight networks of a C3 model against 40 ms of kernels).  One launch per network: a table of segments, each
// either a linear layer's [Kq][N][4] re-blocking (zero padded) or a plain copy; blockIdx.y = the segment.
// ---------------------------------------------------------------------------------------------
namespace stnerf {
struct PackSeg {
    const float* src;
    int64_t dst_off;   // floats
    int32_t n, in_f, kq;   // kq > 0: W[n][in_f] -> [kq][n][4];  kq == 0: copy `n` floats
};
struct PackTable {
    PackSeg seg[26];
    int32_t count;
};
__global__ void pack_net_device_kernel(PackTable t, float* dst) {
    const PackSeg sg = t.seg[blockIdx.y];
    const int64_t total = sg.kq > 0 ? (int64_t)sg.kq * sg.n * 4 : sg.n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float v;
        if (sg.kq > 0) {
            const int r = (int)(i & 3);
            const int64_t q = i >> 2;
            const int n = (int)(q % sg.n), k = 4 * (int)(q / sg.n) + r;
            v = k < sg.in_f ? sg.src[(int64_t)n * sg.in_f + k] : 0.f;
        } else {
            v = sg.src[i];
        }
        dst[sg.dst_off + i] = v;
    }
}
}  // namespace stnerf

extern "C" int stnerf_pack_net_device(int kind, const float* const* W, const float* const* B, int n_tensors, void* dst_dev, int64_t dst_bytes,
                                      stnerf_stream_t stream) {
    STNERF_REQUIRE(W && B && dst_dev, "pack_net_device: null pointer");
    PackTable t;
    memset(&t, 0, sizeof(t));
    auto lin = [&](const float* w, int n, int in_f, int kq, int64_t off) { t.seg[t.count++] = PackSeg{w, off, n, in_f, kq}; };
    auto cpy = [&](const float* src, int count, int64_t off) { t.seg[t.count++] = PackSeg{src, off, count, 0, 0}; };
    int64_t total = 0;
    if (STNERF_NET_IS_SPACE(kind)) {
        const bool ut = STNERF_NET_USES_TIME(kind), deep = STNERF_NET_IS_DEEP(kind);
        const SpaceLayout L = space_layout(ut, deep);
        const int nt = deep ? 12 : 10;
        STNERF_REQUIRE(n_tensors == nt, "pack_net_device: this SpaceNet kind takes %d tensors, got %d", nt, n_tensors);
        for (int i = 0; i < nt; ++i) STNERF_REQUIRE(W[i] && B[i], "pack_net_device: tensor %d is null", i);
        total = L.total;
        const int in_f[7] = {63, 256, 256, 256, 319, 256, 256};
        for (int i = 0; i < 7; ++i) {
            lin(W[i], 256, in_f[i], L.kq[i], L.w[i]);
            cpy(B[i], 256, L.b[i]);
        }
        cpy(W[7], 256, L.w_sigma);
        cpy(B[7], 1, L.b_sigma);
        lin(W[8], 128, 256 + 27 + (ut ? 21 : 0), L.kq_rgb1, L.w_rgb1);
        cpy(B[8], 128, L.b_rgb1);
        for (int i = 0; i < 2 && deep; ++i) {
            lin(W[9 + i], 128, 128, 32, L.w_deep[i]);
            cpy(B[9 + i], 128, L.b_deep[i]);
        }
        cpy(W[nt - 1], 3 * 128, L.w_rgb2);
        cpy(B[nt - 1], 3, L.b_rgb2);
    } else if (kind == STNERF_NET_MOTION) {
        const MotionLayout L = motion_layout();
        STNERF_REQUIRE(n_tensors == 6, "pack_net_device: MotionNet takes 6 tensors, got %d", n_tensors);
        for (int i = 0; i < 6; ++i) STNERF_REQUIRE(W[i] && B[i], "pack_net_device: tensor %d is null", i);
        total = L.total;
        const int in_f[5] = {84, 128, 128, 128, 128};
        for (int i = 0; i < 5; ++i) {
            lin(W[i], 128, in_f[i], L.kq[i], L.w[i]);
            cpy(B[i], 128, L.b[i]);
        }
        cpy(W[5], 3 * 128, L.w_out);
        cpy(B[5], 3, L.b_out);
    } else {
        set_error("pack_net_device: unknown net kind %d", kind);
        return STNERF_EINVAL;
    }
    STNERF_REQUIRE(dst_bytes >= total * 4, "pack_net_device: dst too small");
    // (the pads between the sections: the host packer zeroes the whole blob first)
    if (hipMemsetAsync(dst_dev, 0, (size_t)total * 4, as_stream(stream)) != hipSuccess) return STNERF_ELAUNCH;
    hipLaunchKernelGGL(pack_net_device_kernel, dim3(64, t.count), dim3(256), 0, as_stream(stream), t, static_cast<float*>(dst_dev));
    STNERF_CHECK_LAUNCH("pack_net_device");
    return STNERF_OK;
}

// The transposed sections the fused backward chains read (csrc/train_wave.hip): out x in (row stride ldw) -> [out / 4][n_pad][4], zero
// for the padded inputs; n_pad == 0: a plain copy of out * in floats (the heads).  One launch for a network: blockIdx.y = section.
namespace stnerf {
struct TransposeTable {
    stnerf_transpose_section seg[12];
};
__global__ void pack_transposed_kernel(TransposeTable t, float* dst) {
    const stnerf_transpose_section sg = t.seg[blockIdx.y];
    const int64_t total = sg.n_pad > 0 ? (int64_t)sg.n_out * sg.n_pad : (int64_t)sg.n_out * sg.n_in;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float v;
        if (sg.n_pad > 0) {
            const int r = (int)(i & 3);
            const int64_t q = i >> 2;
            const int n = (int)(q % sg.n_pad), o = 4 * (int)(q / sg.n_pad) + r;
            v = n < sg.n_in ? sg.w[(int64_t)o * sg.ldw + n] : 0.f;
        } else {
            v = sg.w[(i / sg.n_in) * sg.ldw + (i % sg.n_in)];
        }
        dst[sg.dst_off + i] = v;
    }
}
}  // namespace stnerf

extern "C" int stnerf_pack_transposed(const stnerf_transpose_section* sections, int count, float* dst_dev, int64_t dst_floats, stnerf_stream_t stream) {
    STNERF_REQUIRE(sections && dst_dev && count >= 1 && count <= 12, "pack_transposed: 1 .. 12 sections");
    TransposeTable t;
    memset(&t, 0, sizeof(t));
    for (int i = 0; i < count; ++i) {
        const stnerf_transpose_section& q = sections[i];
        STNERF_REQUIRE(q.w && q.n_out >= 1 && q.n_in >= 1 && q.ldw >= q.n_in && q.dst_off >= 0, "pack_transposed: bad section %d", i);
        STNERF_REQUIRE(q.n_pad == 0 || ((q.n_out & 3) == 0 && q.n_pad >= q.n_in), "pack_transposed: section %d: out %% 4 == 0 and n_pad >= in", i);
        const int64_t floats = q.n_pad > 0 ? (int64_t)q.n_out * q.n_pad : (int64_t)q.n_out * q.n_in;
        STNERF_REQUIRE(q.dst_off + floats <= dst_floats, "pack_transposed: section %d ends beyond the destination", i);
        t.seg[i] = q;
    }
    hipLaunchKernelGGL(pack_transposed_kernel, dim3(64, count), dim3(256), 0, as_stream(stream), t, dst_dev);
    STNERF_CHECK_LAUNCH("pack_transposed");
    return STNERF_OK;
}

extern "C" int stnerf_encode(const float* x, int64_t n, int dim, int n_freq, int include_input, float* y,
                             stnerf_stream_t stream) {
    STNERF_REQUIRE(x && y, "encode: null pointer");
    STNERF_REQUIRE(n >= 0 && dim >= 1 && n_freq >= 0 && n_freq <= 30 && (include_input == 0 || include_input == 1),
                   "encode: bad shape n=%lld dim=%d n_freq=%d", (long long)n, dim, n_freq);
    const int64_t tot = n * dim * (include_input + 2 * n_freq);
    if (tot == 0) return STNERF_OK;
    hipLaunchKernelGGL(encode_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, as_stream(stream), x, n, dim,
                       n_freq, include_input, y);
    STNERF_CHECK_LAUNCH("encode");
    return STNERF_OK;
}

extern "C" int stnerf_spacenet_fwd(int kind, const void* packed, int64_t n_rays, int ns, const int32_t* ray_list,
                                   const int32_t* ray_count, const float* xyz, int64_t xyz_ray_stride,
                                   const float* dirs, int64_t dirs_ray_stride, const float* times,
                                   int64_t times_ray_stride, float* raw, int64_t raw_ray_stride, float* ray_bias,
                                   stnerf_stream_t stream) {
    STNERF_REQUIRE(STNERF_NET_IS_SPACE(kind), "spacenet_fwd: bad kind %d", kind);
    STNERF_REQUIRE(packed && xyz && dirs && raw && ray_bias, "spacenet_fwd: null pointer");
    STNERF_REQUIRE(((uintptr_t)ray_bias & 15) == 0, "spacenet_fwd: ray_bias must be 16-byte aligned");
    STNERF_REQUIRE(!STNERF_NET_USES_TIME(kind) || times, "spacenet_fwd: net takes time but times is null");
    STNERF_REQUIRE(n_rays >= 0 && ns >= 1, "spacenet_fwd: bad shape n_rays=%lld ns=%d", (long long)n_rays, ns);
    STNERF_REQUIRE((raw_ray_stride & 3) == 0 && ((uintptr_t)raw & 15) == 0, "spacenet_fwd: raw must be 16-byte aligned");
    STNERF_REQUIRE(((uintptr_t)packed & 15) == 0, "spacenet_fwd: packed weights must be 16-byte aligned");
    if (n_rays == 0) return STNERF_OK;
    // rgb_net.1's direction / time columns once per ray (mlp_raybias.hip) -> the C operands of that layer
    if (const int rc = launch_ray_bias(kind, static_cast<const float*>(packed), n_rays, ray_list, ray_count, dirs, dirs_ray_stride,
                                       times, times_ray_stride, ray_bias, as_stream(stream)))
        return rc;
    SpaceArgs a{static_cast<const float*>(packed), {n_rays, ns, ray_list, ray_count}, xyz, xyz_ray_stride, dirs,
                dirs_ray_stride, times, times_ray_stride, raw, raw_ray_stride, ray_bias};
    constexpr int TM = 128, NW = 8, LDS = (64 + 16) * TM * 16;
    const int grid = grid_for(n_rays, ns, TM);
    const bool ut = STNERF_NET_USES_TIME(kind);
    const char* what = "spacenet_fwd";
    if (STNERF_NET_IS_DEEP(kind))
        return ut ? launch_mlp(spacenet_kernel<TM, NW, true, true>, LDS, grid, NW * 64, stream, a, what, PROF_SPACENET, kind)
                  : launch_mlp(spacenet_kernel<TM, NW, false, true>, LDS, grid, NW * 64, stream, a, what, PROF_SPACENET, kind);
    return ut ? launch_mlp(spacenet_kernel<TM, NW, true>, LDS, grid, NW * 64, stream, a, what, PROF_SPACENET, kind)
              : launch_mlp(spacenet_kernel<TM, NW, false>, LDS, grid, NW * 64, stream, a, what, PROF_SPACENET, kind);
}

extern "C" int stnerf_motionnet_fwd(const void* packed, int64_t n_rays, int ns, const int32_t* ray_list,
                                    const int32_t* ray_count, float* xyz, int64_t xyz_ray_stride, const float* times,
                                    int64_t times_ray_stride, float* flow, int64_t flow_ray_stride, int add_to_xyz,
                                    stnerf_stream_t stream) {
    STNERF_REQUIRE(packed && xyz && times, "motionnet_fwd: null pointer");
    STNERF_REQUIRE(flow || (add_to_xyz & STNERF_MOTION_ADD_TO_XYZ), "motionnet_fwd: nothing to write");
    STNERF_REQUIRE(n_rays >= 0 && ns >= 1, "motionnet_fwd: bad shape");
    STNERF_REQUIRE(((uintptr_t)packed & 15) == 0, "motionnet_fwd: packed weights must be 16-byte aligned");
    if (n_rays == 0) return STNERF_OK;
    MotionArgs a{static_cast<const float*>(packed), {n_rays, ns, ray_list, ray_count}, xyz, xyz_ray_stride, times,
                 times_ray_stride, flow, flow_ray_stride, add_to_xyz};
    constexpr int TM = 64, NW = 4;
    return launch_mlp(motionnet_kernel<TM, NW>, motion_lds_bytes<TM, NW>(), grid_for(n_rays, ns, TM), NW * 64, stream, a,
                      "motionnet_fwd", PROF_MOTIONNET, STNERF_NET_MOTION);
}
