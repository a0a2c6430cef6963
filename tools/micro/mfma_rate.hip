// Microbenchmark: sustained issue rate of v_mfma_f32_32x32x16_f16 (and the f32 32x32x2 form) from 1 or 2 waves
// per SIMD with 4 independent accumulators, no memory traffic.  hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
// `mfma_rate sustained` additionally runs each form for ~3 s with constant and with random operand data (the
// MFMA datapath toggles far more with real data, which matters once the chip is power-limited); sample
// rocm-smi next to it (tools/power_trace.sh).
#include <hip/hip_runtime.h>
#include <stdio.h>
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ inline float hash_unit(unsigned x) {  // (-1, 1)
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return (float)(int)x * (1.0f / 2147483648.0f);
}

template <bool F16, bool RANDOM = false>
__global__ __launch_bounds__(512, 2) void rate_kernel(float* out, int iters, unsigned long long* cycles) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + i + r);
    // RANDOM: four operand sets of random values per lane, cycled so consecutive MFMAs see different bits
    half8 a[4], b[4];
    float fa[4], fb[4];
    for (int q = 0; q < 4; ++q) {
        for (int j = 0; j < 8; ++j) {
            const unsigned id = (blockIdx.x * 512u + threadIdx.x) * 64u + q * 16u + j;
            a[q][j] = RANDOM ? (_Float16)(0.25f * hash_unit(id)) : (_Float16)(0.001f * (threadIdx.x + j));
            b[q][j] = RANDOM ? (_Float16)(0.25f * hash_unit(id + 8u)) : (_Float16)(0.002f * (j + 1));
        }
        fa[q] = RANDOM ? 0.25f * hash_unit(threadIdx.x * 8u + q) : 0.001f * threadIdx.x;
        fb[q] = RANDOM ? 0.25f * hash_unit(threadIdx.x * 8u + q + 4u) : 0.5f;
    }
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 3; ++rep)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = RANDOM ? (i + rep) & 3 : 0;
                if (F16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q], b[q], acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q], fb[q], acc[i], 0, 0, 0);
            }
    }
    const unsigned long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <bool F16>
void run(const char* name, int threads, int iters, int blocks = 256) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<F16><<<blocks, threads>>>(out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<F16><<<blocks, threads>>>(out, iters, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
    const double mfma_per_simd = (double)iters * 12 * (threads / 64) / 4;
    const double flop = F16 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2;
    printf("%s blocks %3d waves/SIMD %d: %.1f cyc/MFMA/SIMD (s_memtime), wall %.3f ms, s_memtime rate %.2f GHz, %.0f TFLOP/s\n", name,
           blocks, threads / 256, avg / mfma_per_simd, ms, avg / (ms * 1e6),
           (double)blocks * (threads / 64) * iters * 12 * flop / (ms * 1e-3) / 1e12);
}

template <bool F16, bool RANDOM>
void sustained(const char* name, int threads, int iters, double seconds) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<F16, RANDOM><<<256, threads>>>(out, iters, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<F16, RANDOM><<<256, threads>>>(out, iters, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms1; hipEventElapsedTime(&ms1, e0, e1);
    const int reps = (int)(seconds * 1e3 / ms1) + 1;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) rate_kernel<F16, RANDOM><<<256, threads>>>(out, iters, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = F16 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2;
    const double tf = 256.0 * (threads / 64) * iters * 12 * flop * reps / (ms * 1e-3) / 1e12;
    const double cyc_per = F16 ? 32.0 : 64.0;
    printf("sustained %s %s data, %d waves/SIMD: %.2f s, %.0f TFLOP/s, implied clock %.2f GHz\n", name,
           RANDOM ? "random" : "constant", threads / 256, ms * 1e-3, tf, tf * 1e12 / (256.0 * 4 * flop / cyc_per) / 1e9);
    fflush(stdout);
}

int main(int argc, char** argv) {
    if (argc > 1) {   // sustained mode
        sustained<true, false>("f16 32x32x16", 512, 20000, 3.0);
        sustained<true, true>("f16 32x32x16", 512, 20000, 3.0);
        sustained<false, false>("f32 32x32x2 ", 512, 5000, 3.0);
        sustained<false, true>("f32 32x32x2 ", 512, 5000, 3.0);
        return 0;
    }
    run<true>("f16 32x32x16", 256, 20000, 1);
    run<true>("f16 32x32x16", 512, 20000, 1);
    run<true>("f16 32x32x16", 256, 20000, 32);
    run<true>("f16 32x32x16", 512, 20000, 32);
    run<true>("f16 32x32x16", 256, 20000);
    run<true>("f16 32x32x16", 512, 20000);
    run<false>("f32 32x32x2 ", 256, 5000);
    run<false>("f32 32x32x2 ", 512, 5000);
    return 0;
}
