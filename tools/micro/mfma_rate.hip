// Microbenchmark: sustained issue rate of v_mfma_f32_32x32x16_f16 (and the f32 32x32x2 form) from 1 or 2 waves
// per SIMD with 4 independent accumulators, no memory traffic.  hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <bool F16>
__global__ __launch_bounds__(512, 2) void rate_kernel(float* out, int iters, unsigned long long* cycles) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + i + r);
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * (j + 1)); }
    float fa = 0.001f * threadIdx.x, fb = 0.5f;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 3; ++rep)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (F16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
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

int main() {
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
