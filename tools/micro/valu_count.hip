// Calibration of the SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS counters as rocprofv3 reports them on gfx950:
// a kernel that executes a KNOWN number of instructions per wave.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/valu_count tools/micro/valu_count.hip
//   rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -f csv -d out -o p -- tools/micro/valu_count
// Per wave: ITER iterations x (100 v_add_f32 + 10 v_exp_f32 + 4 ds_read_b32) + the loop's own s_add / s_cmp / s_cbranch.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ITER = 1000;

__global__ void __launch_bounds__(256) known_mix(float* out) {
    __shared__ float lds[256];
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    float a = (float)threadIdx.x, b = 1.0f, c = 0.f;
    const unsigned addr = (threadIdx.x & 255u) * 4u;
    for (int i = 0; i < ITER; ++i) {
        asm volatile(
            ".rept 100\n\tv_add_f32 %0, %0, %1\n\t.endr\n\t"
            ".rept 10\n\tv_exp_f32 %2, %1\n\t.endr\n\t"
            ".rept 4\n\tds_read_b32 %2, %3\n\t.endr\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "+v"(a), "+v"(b), "+v"(c) : "v"(addr) : "memory");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c;
}

int main() {
    const int blocks = 1024, threads = 256;
    float* d;
    hipMalloc(&d, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    known_mix<<<blocks, threads>>>(d);
    hipEventRecord(e0);
    known_mix<<<blocks, threads>>>(d);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = blocks * (threads / 64.0);
    printf("known_mix: %d workgroups x %d threads = %.0f waves; per wave %d x (100 v_add_f32 + 10 v_exp_f32 + 4 ds_read_b32)\n", blocks, threads, waves, ITER);
    printf("expected per launch: VALU %.4g (of which v_exp %.4g), LDS %.4g, loop SALU ~%.3g;  %.3f ms -> %.1f cycles per wave-iteration at 2.4 GHz / (waves per SIMD = %.0f)\n",
           waves * ITER * 110.0, waves * ITER * 10.0, waves * ITER * 4.0, waves * ITER * 3.0, ms, ms * 1e-3 * 2.4e9 / ITER, waves / 1024.0);
    return 0;
}
