// Microbenchmark: which vector instructions cost the f32 MFMA pipe time on gfx950 when they sit between
// v_mfma_f32_32x32x2_f32 of the SAME wave (one wave per SIMD, the regime of csrc/mlp_wave.hip)?
// Every loop step issues 8 independent MFMAs (accumulators in AGPRs) with V instructions of one kind behind each, on
// registers the MFMAs do not touch.  Prints cycles per MFMA (64 = the pipe's own time).
// hipcc --offload-arch=gfx950 -O3 -o mfma_valu_mix mfma_valu_mix.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;

enum Kind { K_NONE = 0, K_MAX_I32 = 1, K_ACC_READ = 2, K_FMA = 3, K_MOV = 4, K_ACC_WRITE = 5, K_PK_FMA = 6, K_DS_READ = 7 };

template <int KIND>
__device__ __forceinline__ void filler(float& x, float& y, float& spare, const float* lds) {
    if (KIND == K_MAX_I32) asm volatile("v_max_i32 %0, 0, %0" : "+v"(x));
    if (KIND == K_ACC_READ) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(spare));
    if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
    if (KIND == K_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y));
    if (KIND == K_ACC_WRITE) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(spare) : "v"(y));
    if (KIND == K_PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double*>(&x)) : "v"(*reinterpret_cast<double*>(&y)));
    if (KIND == K_DS_READ) asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"((unsigned)(uintptr_t)lds) : "memory");
}

template <int KIND, int V>
__global__ __launch_bounds__(256, 1) void mix_kernel(float* out, int iters, unsigned long long* cycles) {
    __shared__ float lds[256];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    f32x16 acc[8];
    float spare = threadIdx.x;
    asm volatile("" : "+a"(spare));
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.001f * (threadIdx.x + i + r);
    float a = 0.5f + 0.001f * threadIdx.x, b = 0.25f;
    float x[2] = {1.f, 2.f}, y[2] = {0.5f, 0.25f};
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < V; ++j) filler<KIND>(x[0], y[0], spare, lds + (threadIdx.x & 63));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = clock64();
    float s = x[0] + x[1] + y[0] + spare;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int KIND, int V>
void run(const char* name) {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    unsigned long long* cyc; hipMalloc(&cyc, 8);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mix_kernel<KIND, V><<<256, 256>>>(out, iters, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) mix_kernel<KIND, V><<<256, 256>>>(out, iters, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double mf = (double)iters * 8;
    printf("%-18s V=%2d : %7.2f cycles per MFMA (s_memtime), %6.1f TF/s MFMA\n", name, V, (double)h / mf,
           256.0 * 4 * mf * 10 * 4096.0 / (ms * 1e-3) / 1e12);
    fflush(stdout);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<K_NONE, 0>("none");
    run<K_MAX_I32, 1>("v_max_i32"); run<K_MAX_I32, 2>("v_max_i32"); run<K_MAX_I32, 4>("v_max_i32"); run<K_MAX_I32, 8>("v_max_i32"); run<K_MAX_I32, 12>("v_max_i32");
    run<K_ACC_READ, 2>("v_accvgpr_read"); run<K_ACC_READ, 4>("v_accvgpr_read"); run<K_ACC_READ, 8>("v_accvgpr_read");
    run<K_ACC_WRITE, 2>("v_accvgpr_write"); run<K_ACC_WRITE, 8>("v_accvgpr_write");
    run<K_MOV, 2>("v_mov_b32"); run<K_MOV, 8>("v_mov_b32");
    run<K_FMA, 2>("v_fma_f32"); run<K_FMA, 4>("v_fma_f32"); run<K_FMA, 8>("v_fma_f32");
    run<K_PK_FMA, 2>("v_pk_fma_f32"); run<K_PK_FMA, 4>("v_pk_fma_f32");
    run<K_DS_READ, 2>("ds_read_b32"); run<K_DS_READ, 8>("ds_read_b32");
    return 0;
}
