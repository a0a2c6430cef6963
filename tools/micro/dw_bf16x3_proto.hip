// Prototype (round 6, NOT product code): what would a split-bf16 ("bf16x3") weight-gradient tile run at?  DESIGN.md section 8.1.
//
// dW = dY^T X with BOTH operands activations: every fp32 value is split on the fly into three bf16 pieces (split8 of
// csrc/mlp_bf16x3.hip: 5.5 vector instructions per value) and a product takes six v_mfma_f32_32x32x16_bf16.  One wave owns a
// 128 x 64 tile of a dW (128 accumulator registers) and a range of samples; a K step is 16 samples: lane (h, c) loads, for its 8
// samples 8 h .. 8 h + 7, 16 bytes of dY (rows 4 c .. 4 c + 3 of the tile) and 8 bytes of X (columns 2 c, 2 c + 1) -- the loads of
// csrc/train_dw.hip's direct path --, splits the 48 values (264 vector instructions) and issues 8 block pairs x 6 = 48 MFMAs.
// The real tile would be 128 x 128 through the LDS ring (352 vector instructions beside 96 MFMAs: a better ratio than here).
//
//   variant 0: the MFMAs alone, on fixed planes                        -> the matrix pipe's own rate at this register pressure
//   variant 1: + the split of 48 fresh values per step (registers)     -> does the vector work hide behind the MFMAs?
//   variant 2: + the operand stream from HBM (each wave its own rows)  -> the whole step
// Each variant runs SEC seconds on 256 x 4 waves with the socket's energy counter, power and clock read beside it.
//   hipcc --offload-arch=gfx950 -O3 -o dw_bf16x3_proto dw_bf16x3_proto.hip -lrocm_smi64 ; ./dw_bf16x3_proto [SEC]
#include <hip/hip_runtime.h>
#include <rocm_smi/rocm_smi.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x2 = __attribute__((ext_vector_type(2))) int;

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    f32x2_ v = {a, b};
    bf16x2 h = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<unsigned*>(&h);
}
struct Planes { bf16x8 p[3]; };
__device__ __forceinline__ void split8(const float (&v)[8], Planes& o) {
    u32x4 w0, w1, w2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = v[2 * i], x1 = v[2 * i + 1];
        const unsigned u = pk_bf16(x0, x1);
        const float r0 = x0 - __uint_as_float(u << 16), r1 = x1 - __uint_as_float(u & 0xffff0000u);
        const unsigned m = pk_bf16(r0, r1);
        const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
        w0[i] = u, w1[i] = m, w2[i] = pk_bf16(s0, s1);
    }
    o.p[0] = *reinterpret_cast<bf16x8*>(&w0);
    o.p[1] = *reinterpret_cast<bf16x8*>(&w1);
    o.p[2] = *reinterpret_cast<bf16x8*>(&w2);
}
__device__ __forceinline__ f32x16 mma(const bf16x8& a, const bf16x8& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
// a b ~ a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0 (one accumulator: a dW sums thousands of samples, the f32 kernel rounds as often)
__device__ __forceinline__ void product(const Planes& a, const Planes& b, f32x16& c) {
    c = mma(a.p[0], b.p[0], c);
    c = mma(a.p[0], b.p[1], c);
    c = mma(a.p[1], b.p[0], c);
    c = mma(a.p[1], b.p[1], c);
    c = mma(a.p[0], b.p[2], c);
    c = mma(a.p[2], b.p[0], c);
}

struct Args {
    const float* dy;   // [m][256]
    const float* x;    // [m][256]
    float* out;        // [waves][128 * 64]: the partial tiles
    int m, rows_per_wave;
    unsigned long long* cycles;
};

template <int VARIANT>
__global__ __launch_bounds__(256, 1) void proto_kernel(Args a) {
    const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    // this wave's tile: row block (wave & 1), column quarter (wave >> 1) & 3 of a 256 x 256 layer; its sample range
    const int n0 = 128 * (wave & 1), k0 = 64 * ((wave >> 1) & 3);
    const int s0 = (int)(((long long)(wave >> 3) * a.rows_per_wave) % (a.m - a.rows_per_wave + 1)) & ~15;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, a.m * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.m * 1024, 0x00020000);
    const uint32_t va = (uint32_t)((8 * h) * 256 + n0 + 4 * c) * 4u, vb = (uint32_t)((8 * h) * 256 + k0 + 2 * c) * 4u;
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    i32x4 ar[2][8];
    i32x2 br[2][8];
    auto load = [&](int buf, int step) {
        const uint32_t so = (uint32_t)(s0 + 16 * step) * 1024u;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            ar[buf][s] = __builtin_amdgcn_raw_buffer_load_b128(ra, va, so + s * 1024u, 0);
            br[buf][s] = __builtin_amdgcn_raw_buffer_load_b64(rb, vb, so + s * 1024u, 0);
        }
    };
    // fixed operands for the variants that do not stream: a few rows of the matrices
    load(0, 0);
    load(1, 1);
    Planes pa[4], pb[2];
    auto split_all = [&](int buf, float bump) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) v[s] = VARIANT == 1 ? __int_as_float(ar[buf][s][i]) + bump : __int_as_float(ar[buf][s][i]);
            split8(v, pa[i]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) v[s] = VARIANT == 1 ? __int_as_float(br[buf][s][j]) + bump : __int_as_float(br[buf][s][j]);
            split8(v, pb[j]);
        }
    };
    split_all(0, 0.f);
    const int steps = a.rows_per_wave / 16;
    const unsigned long long t0 = clock64();
    for (int t = 0; t < steps; t += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // the products of step t + half on the planes split during the previous step ...
            Planes qa[4], qb[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) qa[i] = pa[i];
#pragma unroll
            for (int j = 0; j < 2; ++j) qb[j] = pb[j];
            if (VARIANT >= 1) {
                // ... while the next step's 48 values are split (VARIANT 1: the same registers, perturbed so that nothing is hoisted)
                if (VARIANT == 2) {
                    split_all(half ^ 1, 0.f);
                    load(half, t + half + 2);                         // two steps ahead, into the buffer the planes in use came from
                } else {
                    split_all(half ^ 1, (float)(t + half) * 1e-9f);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) product(qa[i], qb[j], acc[i][j]);
            // interleave: one MFMA, then a handful of the split's vector instructions (48 MFMAs, ~264 VALU)
#pragma unroll
            for (int g = 0; g < 48; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // 1 MFMA
                if (VARIANT >= 1) __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);   // 6 VALU
            }
        }
    }
    const unsigned long long t1 = clock64();
    float* o = a.out + (size_t)wave * 128 * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                *reinterpret_cast<float2*>(o + (4 * (8 * q + 4 * h + rr) + i) * 64 + 2 * c) = make_float2(acc[i][0][4 * q + rr], acc[i][1][4 * q + rr]);
    if (threadIdx.x == 0 && blockIdx.x == 0) *a.cycles = t1 - t0;
}

struct Meter {
    bool ok;
    Meter() { ok = rsmi_init(0) == RSMI_STATUS_SUCCESS; }
    double joules() {
        uint64_t e, ts; float res;
        if (!ok || rsmi_dev_energy_count_get(0, &e, &res, &ts) != RSMI_STATUS_SUCCESS) return -1.0;
        return (double)e * res * 1e-6;
    }
    double watts() {
        uint64_t p;
        if (!ok || rsmi_dev_current_socket_power_get(0, &p) != RSMI_STATUS_SUCCESS) return -1.0;
        return p * 1e-6;
    }
    double mhz() {
        rsmi_frequencies_t f;
        if (!ok || rsmi_dev_gpu_clk_freq_get(0, RSMI_CLK_TYPE_SYS, &f) != RSMI_STATUS_SUCCESS) return -1.0;
        return f.frequency[f.current] * 1e-6;
    }
};

template <int VARIANT>
static void run(Meter& mt, const char* name, double seconds, Args a, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    proto_kernel<VARIANT><<<blocks, 256>>>(a);
    hipDeviceSynchronize();
    auto wall0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count() < 1.0) {   // settle at the load's power
        proto_kernel<VARIANT><<<blocks, 256>>>(a);
        hipDeviceSynchronize();
    }
    double ms_sum = 0, w_sum = 0, f_sum = 0;
    int launches = 0, samples = 0;
    const double j0 = mt.joules();
    wall0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count() < seconds) {
        hipEventRecord(e0);
        for (int i = 0; i < 4; ++i) proto_kernel<VARIANT><<<blocks, 256>>>(a);
        hipEventRecord(e1);
        const double w = mt.watts(), f = mt.mhz();
        if (w > 0) w_sum += w, f_sum += f, ++samples;
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        ms_sum += ms;
        launches += 4;
    }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
    const double j1 = mt.joules();
    unsigned long long cyc; hipMemcpy(&cyc, a.cycles, 8, hipMemcpyDeviceToHost);
    const double steps = a.rows_per_wave / 16.0, waves = blocks * 4.0;
    const double flop = 2.0 * 128 * 64 * a.rows_per_wave * waves;               // algorithmic, per launch
    const double sec = ms_sum * 1e-3 / launches;
    printf("%-46s %7.3f ms per launch  %7.1f algorithmic TF/s (x6 executed: %6.0f)  %7.1f cycles per 16-sample step (48 MFMAs: %5.1f each)  %5.0f MHz  %5.0f W (smi, %d samples)  %5.0f W (energy ctr)\n",
           name, sec * 1e3, flop / sec / 1e12, 6 * flop / sec / 1e12, cyc / steps, cyc / steps / 48.0, samples ? f_sum / samples : -1.0, samples ? w_sum / samples : -1.0, samples,
           j0 >= 0 ? (j1 - j0) / wall : -1.0);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    const int m = 1 << 18, blocks = 256, rows_per_wave = 8192;     // 262,144 samples x 256 floats x 2 matrices = 512 MB
    std::vector<float> h((size_t)m * 256);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    Args a{};
    float *dy, *x;
    hipMalloc(&dy, h.size() * 4); hipMalloc(&x, h.size() * 4);
    hipMemcpy(dy, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (auto& v : h) v = (float)rand() / RAND_MAX;
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&a.out, (size_t)blocks * 4 * 128 * 64 * 4);
    hipMalloc(&a.cycles, 8);
    a.dy = dy, a.x = x, a.m = m, a.rows_per_wave = rows_per_wave;
    Meter mt;
    printf("split-bf16 dW tile prototype: 128 x 64 per wave, %d samples per wave, %d waves; %.1f s per variant after 1 s of the same load; energy counter %s\n",
           rows_per_wave, blocks * 4, seconds, mt.ok ? "rocm_smi" : "UNAVAILABLE");
    run<0>(mt, "0: 48 bf16 MFMAs per step, fixed planes", seconds, a, blocks);
    run<1>(mt, "1: + split of 48 values per step (264 VALU)", seconds, a, blocks);
    run<2>(mt, "2: + operands streamed from HBM (768 B/sample)", seconds, a, blocks);
    printf("for comparison, the f32 tile of csrc/train_dw.hip: 16 x v_mfma_f32_32x32x2_f32 per 2 samples of a 128 x 128 tile = 8192 cycles per 16 samples (4096 for this tile's 128 x 64)\n");
    return 0;
}
