// Microbenchmark: do the vector instructions of one wave hide behind the MFMAs of ANOTHER wave on the same SIMD (gfx950)?
// csrc/mlp_wave.hip runs one wave per SIMD (512 registers per lane); tools/micro/mfma_valu_mix.hip shows that there every
// vector instruction between MFMAs costs the matrix pipe its own issue time.  Here: bursts of NM independent MFMAs followed
// by NV vector instructions (v_max_i32 on four rotating registers), with one or two waves per SIMD, for
// v_mfma_f32_32x32x2_f32 (64 cycles, 16 accumulator registers) and v_mfma_f32_16x16x4_f32 (32 cycles, 4 registers; the
// same 64 flop per cycle and lane group).  Prints the achieved MFMA rate of the whole chip: with perfect overlap two waves
// per SIMD stay at the pipe's peak as long as the vector work of both fits beside it.
// hipcc --offload-arch=gfx950 -O3 -o mfma_two_waves mfma_two_waves.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <bool SMALL, int NM, int NV, int THREADS, bool OFFSET>
__global__ __launch_bounds__(THREADS, 1) void burst_kernel(float* out, int iters, unsigned long long* cycles) {
    f32x16 acc[4];
    f32x4 acs[8];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.001f * (threadIdx.x + i + r);
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 4; ++r) acs[i][r] = 0.002f * (threadIdx.x + i + r);
    float a = 0.5f + 0.001f * threadIdx.x, b = 0.25f;
    float x[4] = {1.f, 2.f, -3.f, 4.f};
    const unsigned long long t0 = clock64();
    // OFFSET: the second wave of a SIMD (waves w and w + 4 of the workgroup share one) starts with its vector burst, so the
    // bursts of the two interleave; without it the two waves stay in lock step and do their vector work at the same time
    if (OFFSET && threadIdx.x >= 256) {
#pragma unroll
        for (int j = 0; j < NV; ++j) asm volatile("v_max_i32 %0, 0, %0" : "+v"(x[j & 3]));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            if (SMALL) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acs[i & 7]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i & 3]) : "v"(a), "v"(b));
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) asm volatile("v_max_i32 %0, 0, %0" : "+v"(x[j & 3]));
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = clock64();
    float s = x[0] + x[1] + x[2] + x[3];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 4; ++r) s += acs[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <bool SMALL, int NM, int NV, int THREADS, bool OFFSET = false>
void run() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    unsigned long long* cyc; hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    burst_kernel<SMALL, NM, NV, THREADS, OFFSET><<<256, THREADS>>>(out, iters, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) burst_kernel<SMALL, NM, NV, THREADS, OFFSET><<<256, THREADS>>>(out, iters, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double waves = 256.0 * (THREADS / 64), flop = SMALL ? 2048.0 : 4096.0, pipe = SMALL ? 32.0 : 64.0;
    const double mf = (double)iters * NM;
    printf("%s  %d wave(s) per SIMD%s  bursts of %3d MFMA + %3d v_max_i32 : %7.2f wave-cycles per MFMA (pipe %2.0f), %6.1f TF/s\n",
           SMALL ? "16x16x4" : "32x32x2", THREADS / 256, OFFSET ? ", out of phase" : "", NM, NV, (double)h / mf, pipe, waves * mf * 10 * flop / (ms * 1e-3) / 1e12);
    fflush(stdout);
    hipFree(out); hipFree(cyc);
}

int main() {
    // 32x32x2: 64 MFMAs = 4096 pipe cycles per burst
    run<false, 64, 0, 256>(); run<false, 64, 32, 256>(); run<false, 64, 128, 256>();
    run<false, 64, 0, 512>(); run<false, 64, 32, 512>(); run<false, 64, 128, 512>(); run<false, 64, 256, 512>();
    // 16x16x4: 128 MFMAs = 4096 pipe cycles per burst
    run<true, 128, 0, 256>(); run<true, 128, 32, 256>(); run<true, 128, 128, 256>();
    run<true, 128, 0, 512>(); run<true, 128, 32, 512>(); run<true, 128, 128, 512>(); run<true, 128, 256, 512>();
    // two waves per SIMD, out of phase
    run<false, 64, 32, 512, true>(); run<false, 64, 128, 512, true>(); run<false, 64, 256, 512, true>(); run<false, 64, 512, 512, true>();
    run<true, 128, 32, 512, true>(); run<true, 128, 128, 512, true>(); run<true, 128, 256, 512, true>(); run<true, 128, 512, 512, true>();
    // fine-grained mix (a vector instruction every few MFMAs), two waves
    run<true, 8, 2, 512>(); run<true, 8, 4, 512>(); run<false, 4, 2, 512>(); run<false, 4, 4, 512>();
    return 0;
}
