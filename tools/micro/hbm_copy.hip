// HBM bandwidth microbenchmark for gfx950 ("confirm the ~8 TB/s vendor figure by a copy microbench").
// Three streaming kernels over a buffer far larger than L2 + MALL (default 2 GiB): float4 copy (read + write),
// read-only (float4 loads folded into one value per thread) and write-only (float4 stores) -- the compositor is
// read-mostly, the samplers are write-only, so each is quoted against the matching figure.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/hbm_copy tools/micro/hbm_copy.hip && tools/micro/hbm_copy [MiB] [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void read_kernel(const float4* __restrict__ src, float* __restrict__ sink, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        acc += (v.x + v.y) + (v.z + v.w);
    }
    if (acc == 123.456f) sink[0] = acc;  // never true for the zero-filled buffer: keeps the loads alive
}
__global__ void write_kernel(float4* __restrict__ dst, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = make_float4(v, v, v, v);
}

int main(int argc, char** argv) {
    const size_t mib = argc > 1 ? (size_t)atoll(argv[1]) : 2048;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const size_t bytes = mib << 20, n = bytes / sizeof(float4);
    float4 *a, *b;
    float* sink;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(a, 0, bytes));
    CHECK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grids[3] = {256 * 8, 256 * 16, 256 * 32};
    printf("{\"buffer_MiB\": %zu, \"reps\": %d", mib, reps);
    const char* names[3] = {"copy", "read", "write"};
    for (int k = 0; k < 3; ++k) {
        double best = 0.0;
        int best_grid = 0, best_bs = 0;
        for (int gi = 0; gi < 3; ++gi)
            for (int bs = 256; bs <= 1024; bs *= 2) {
                const dim3 g(grids[gi]), t(bs);
                auto launch = [&]() {
                    if (k == 0) hipLaunchKernelGGL(copy_kernel, g, t, 0, 0, a, b, n);
                    if (k == 1) hipLaunchKernelGGL(read_kernel, g, t, 0, 0, a, sink, n);
                    if (k == 2) hipLaunchKernelGGL(write_kernel, g, t, 0, 0, b, n, 1.0f);
                };
                launch();
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                for (int r = 0; r < reps; ++r) launch();
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms = 0.f;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double gbps = (k == 0 ? 2.0 : 1.0) * (double)bytes * reps / (ms * 1e-3) / 1e9;
                if (gbps > best) { best = gbps; best_grid = grids[gi]; best_bs = bs; }
            }
        printf(", \"%s_GBps\": %.1f, \"%s_grid\": [%d, %d]", names[k], best, names[k], best_grid, best_bs);
    }
    printf("}\n");
    return 0;
}
