// mfma_shape_probe.hip — research tool: sustained bf16 rate of v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16,
// registers only, random operands, 1 or 2 waves per SIMD, ~50 ms per arm (the clock the chip settles at is part of the answer).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ inline uint32_t rnd(uint32_t x) { x *= 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; return x; }
__device__ inline bf16x8 rand_frag(uint32_t seed) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (short)(0x3C00 | (rnd(seed * 8u + e) & 0x83FF));   // +-[0.008, 0.03), random mantissas
    return v;
}
template <int SHAPE>   // 32: 32x32x16, 16: 16x16x32; 16 accumulator tiles' worth of independent chains (128 regs)
__global__ __launch_bounds__(256) void mfma_kernel(uint32_t iters, float *sink) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { a[i] = rand_frag(t * 8u + i); b[i] = rand_frag(t * 8u + 4 + i); }
    float v = 0.0f;
    if constexpr (SHAPE == 32) {
        f32x16 c[8];
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int e = 0; e < 16; e++) c[i][e] = 0.0f;
        for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 1], c[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int e = 0; e < 16; e++) v += c[i][e];
    } else {
        f32x4 c[32];
#pragma unroll
        for (int i = 0; i < 32; i++) c[i] = f32x4{0, 0, 0, 0};
        for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], c[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 32; i++) v += c[i].x + c[i].y + c[i].z + c[i].w;
    }
    if (v == 123456.789f) *sink = v;
}
int main() {
    float *sink; CK(hipMalloc((void **)&sink, 4));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++)
    for (int wps : {1, 2})
    for (int shape : {32, 16}) {
        const uint32_t grid = cus * wps;   // 4 waves per block: wps blocks per CU = wps waves per SIMD
        // per iteration and wave: 8 x 32768 flop (32x32x16) or 16 x 16384 flop (16x16x32) = 262144 flop either way
        uint32_t iters = 200000;
        float ms = 0;
        for (int round = 0; round < 2; round++) {
            CK(hipEventRecord(e0));
            if (shape == 32) hipLaunchKernelGGL(mfma_kernel<32>, dim3(grid), dim3(256), 0, 0, iters, sink);
            else hipLaunchKernelGGL(mfma_kernel<16>, dim3(grid), dim3(256), 0, 0, iters, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (round == 0) iters = (uint32_t)(iters * 100.0 / ms);
        }
        const double flops = (double)grid * 4.0 * iters * 262144.0;
        printf("shape %s, %d wave(s) per SIMD: %.1f ms, %.0f TFLOP/s (%.3f of 2.5 PF)\n", shape == 32 ? "32x32x16" : "16x16x32", wps, ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15);
    }
    return 0;
}
