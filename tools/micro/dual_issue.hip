// Microbenchmark: can the f32 MFMA pipe and the f32 vector ALU (v_pk_fma_f32) be kept busy at the same time?
// Each wave issues 4 independent v_mfma_f32_32x32x2_f32 per loop step plus V packed FMAs per MFMA on separate
// registers (random data).  2 waves per SIMD, all 256 CUs, ~2 s per setting.  Prints the combined f32 rate.
// hipcc --offload-arch=gfx950 -O3 -o dual_issue dual_issue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

__device__ inline float hash_unit(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return (float)(int)x * (1.0f / 2147483648.0f);
}

template <int V, bool MFMA>
__global__ __launch_bounds__(512, 2) void dual_kernel(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = hash_unit(threadIdx.x * 64 + i * 16 + r);
    float fa[4], fb[4];
    for (int q = 0; q < 4; ++q) { fa[q] = 0.25f * hash_unit(threadIdx.x * 8u + q); fb[q] = 0.25f * hash_unit(threadIdx.x * 8u + q + 4u); }
    constexpr int NV = V > 0 ? V : 1;
    f32x2 vacc[NV], vx[4], vw[4];
    for (int j = 0; j < NV; ++j) vacc[j] = f32x2{hash_unit(threadIdx.x + 1000 * j), hash_unit(threadIdx.x + 1000 * j + 500)};
    for (int q = 0; q < 4; ++q) {
        vx[q] = f32x2{0.5f * hash_unit(threadIdx.x * 4 + q + 77), 0.5f * hash_unit(threadIdx.x * 4 + q + 99)};
        vw[q] = f32x2{0.9f + 0.1f * hash_unit(q * 13 + blockIdx.x), 0.9f + 0.1f * hash_unit(q * 17 + blockIdx.x)};
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MFMA) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[i], acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < V; ++j) vacc[j] = __builtin_elementwise_fma(vacc[j], vw[(i + j) & 3], vx[(i + j) & 3]);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int j = 0; j < NV; ++j) s += vacc[j].x + vacc[j].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V, bool MFMA>
void run(double seconds) {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    dual_kernel<V, MFMA><<<256, 512>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    dual_kernel<V, MFMA><<<256, 512>>>(out, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms1; hipEventElapsedTime(&ms1, e0, e1);
    const int reps = (int)(seconds * 1e3 / ms1) + 1;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) dual_kernel<V, MFMA><<<256, 512>>>(out, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = 256.0 * 8, steps = (double)iters * 4 * reps;
    const double mfma_tf = MFMA ? waves * steps * 4096.0 / (ms * 1e-3) / 1e12 : 0.0;
    const double valu_tf = waves * steps * V * 64 * 4.0 / (ms * 1e-3) / 1e12;   // pk_fma: 2 FMA = 4 FLOP per lane
    printf("MFMA %s + %2d v_pk_fma_f32 per MFMA: %.2f s  MFMA %.1f TF/s  VALU %.1f TF/s  total %.1f TF/s\n",
           MFMA ? "on " : "off", V, ms * 1e-3, mfma_tf, valu_tf, mfma_tf + valu_tf);
    fflush(stdout);
    hipFree(out);
}

int main() {
    run<0, true>(2.0);
    run<4, true>(2.0);
    run<8, true>(2.0);
    run<12, true>(2.0);
    run<16, true>(2.0);
    run<16, false>(2.0);
    return 0;
}
