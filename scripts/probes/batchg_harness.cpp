// batchg_harness.cpp — research tool: times cx::launch_batchg_scores alone (no Python, no other kernels between launches).
// hipcc --offload-arch=gfx950 -O2 -o _batchg_harness batchg_harness.cpp -L../../cortex_amd/lib -lcortex_hip -Wl,-rpath,$ORIGIN/../../cortex_amd/lib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
namespace cx {
int launch_batchg_scores(const float *rows, const float *norms, uint32_t n_rows, uint32_t dim, const float *d_queries, uint32_t nq,
                         char *d_qimg, float *d_qq, float *d_dense, uint32_t stride, hipStream_t stream);
size_t batchg_qimg_bytes(uint32_t dim);
}
__global__ void fill_kernel(uint32_t *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = 0x3c000000u | (x & 0x03ffffffu);
    }
}
int main(int argc, char **argv) {
    const uint32_t n = argc > 1 ? atoi(argv[1]) : 1000000, dim = argc > 2 ? atoi(argv[2]) : 1024, reps = argc > 3 ? atoi(argv[3]) : 40;
    float *rows, *norms, *q, *qq, *dense; char *qimg;
    const uint32_t stride = (n + 3u) & ~3u; const uint32_t pass_stride = getenv("NO_EPI") ? 0xffffffffu : stride;
    CK(hipMalloc((void **)&rows, ((size_t)n + 256) * dim * 4 + 64));
    CK(hipMalloc((void **)&norms, ((size_t)n + 64) * 4));
    CK(hipMalloc((void **)&q, 64 * (size_t)dim * 4));
    CK(hipMalloc((void **)&qq, 256));
    CK(hipMalloc((void **)&dense, (size_t)64 * stride * 4));
    CK(hipMalloc((void **)&qimg, cx::batchg_qimg_bytes(dim)));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t *)rows, ((size_t)n + 256) * dim);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, (uint32_t *)norms, (size_t)n + 64);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, (uint32_t *)q, (size_t)64 * dim);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e9, sum = 0;
    for (uint32_t r = 0; r < reps; r++) {
        CK(hipEventRecord(e0));
        if (cx::launch_batchg_scores(rows, norms, n, dim, q, 64, qimg, qq, dense, pass_stride, nullptr)) { printf("launch failed\n"); return 1; }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 5) { sum += ms; if (ms < best) best = ms; }
    }
    const double bytes = (double)n * dim * 4, avg = sum / (reps - 5);
    printf("n %u dim %u: best %.4f ms (%.3f of 8 TB/s)  avg %.4f ms (%.3f)\n", n, dim, best, bytes / (best * 1e-3) / 8e12, avg, bytes / (avg * 1e-3) / 8e12);
    return 0;
}
