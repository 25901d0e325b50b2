// probe.hip — measured peaks of THIS box for the two rooflines ("re-measure on the box ... and use the
// measured peaks as denominators, reporting both").  Benchmark support, not part of the reference's interface.
//  - read stream: every CU streams a private slice of a buffer with 16-byte non-temporal loads (the scan
//    kernel's access pattern without its arithmetic) and folds it into one value per wave;
//  - MFMA: every wave issues a long run of independent v_mfma_f32_32x32x16_bf16 on registers (no memory at
//    all): the dense bf16 rate the chip SUSTAINS at its power limit, which is what a GEMM can hope for.
#include "common.hpp"
#include "../../include/cortex_hip_synth.h"

namespace cx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// each wave streams contiguous 16 KiB chunks (16 non-temporal 1-KiB wave loads in flight), chunks dealt round-robin
// to the waves of the grid: the DRAM-page-friendly pattern a row scan has
__global__ __launch_bounds__(256) void probe_read_kernel(const f32x4 *src, size_t n_vec, float *sink) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const size_t n_chunks = n_vec / 1024;   // 1024 vectors = 16 KiB
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    for (size_t ch = wave; ch < n_chunks; ch += n_waves) {
        const f32x4 *p = src + ch * 1024 + lane;
        f32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = __builtin_nontemporal_load(p + u * 64);
#pragma unroll
        for (int u = 0; u < 16; u += 4) { acc0 += v[u]; acc1 += v[u + 1]; acc2 += v[u + 2]; acc3 += v[u + 3]; }
    }
    const f32x4 t = (acc0 + acc1) + (acc2 + acc3);
    const float v = (t.x + t.y) + (t.z + t.w);
    if (v == 123456.789f) *sink = v;   // keeps the loads alive; never true for the probe's data
}

__global__ __launch_bounds__(256) void probe_mfma_kernel(uint32_t iters, float *sink) {
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; e++) { a[e] = (short)(0x3C00 + (threadIdx.x & 63) + e); b[e] = (short)(0x3B80 + (threadIdx.x & 31) * 3 + e); }
    f32x16 c0, c1, c2, c3;
#pragma unroll
    for (int e = 0; e < 16; e++) { c0[e] = 0.0f; c1[e] = 0.0f; c2[e] = 0.0f; c3[e] = 0.0f; }
    for (uint32_t i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    const f32x16 t = (c0 + c1) + (c2 + c3);
    float v = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; e++) v += t[e];
    if (v == 123456.789f) *sink = v;
}

}  // namespace cx

extern "C" {

/* Sustained read bandwidth (GB/s) of `bytes` of HBM streamed `reps` times by 8 blocks per CU; best of reps. */
int cx_probe_read_bw(int device, uint64_t bytes, uint32_t reps, double *out_gbs) {
    using namespace cx;
    if (!out_gbs || bytes < (1u << 20)) return set_err(CX_ERR_VALIDATION, "probe: need >= 1 MiB and an output");
    CX_HIP(hipSetDevice(device));
    float *buf = nullptr, *sink = nullptr;
    CX_HIP(hipMalloc((void **)&buf, bytes));
    CX_HIP(hipMalloc((void **)&sink, 4));
    CX_HIP(hipMemset(buf, 0x11, bytes));
    hipEvent_t e0, e1;
    CX_HIP(hipEventCreate(&e0));
    CX_HIP(hipEventCreate(&e1));
    double best = 0.0;
    for (uint32_t bpc = 1; bpc <= 4; bpc++) {   // blocks per CU: the best setting counts
    const uint32_t grid = device_cus() * bpc;
    for (uint32_t r = 0; r < reps + 1; r++) {
        CX_HIP(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(probe_read_kernel, dim3(grid), dim3(256), 0, nullptr, reinterpret_cast<const f32x4 *>(buf), (size_t)(bytes / 16), sink);
        CX_HIP(hipEventRecord(e1, nullptr));
        CX_HIP(hipEventSynchronize(e1));
        float ms = 0.0f;
        CX_HIP(hipEventElapsedTime(&ms, e0, e1));
        const double g = (double)bytes / (ms * 1e-3) / 1e9;
        if (r && g > best) best = g;
    }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(buf); (void)hipFree(sink);
    *out_gbs = best;
    return CX_OK;
}

/* Sustained dense bf16 MFMA rate (TFLOP/s) over ~`ms_target` ms of register-only v_mfma_f32_32x32x16_bf16 on every
 * SIMD (2 waves each); the average of a run that long includes the clock the chip settles at under that load. */
int cx_probe_mfma_tflops(int device, double ms_target, double *out_tflops) {
    using namespace cx;
    if (!out_tflops) return set_err(CX_ERR_VALIDATION, "probe: null output");
    CX_HIP(hipSetDevice(device));
    float *sink = nullptr;
    CX_HIP(hipMalloc((void **)&sink, 4));
    hipEvent_t e0, e1;
    CX_HIP(hipEventCreate(&e0));
    CX_HIP(hipEventCreate(&e1));
    const uint32_t grid = device_cus() * 2u;   // 2 blocks x 4 waves per CU = 2 waves per SIMD
    uint32_t iters = 20000;
    double tf = 0.0;
    for (int round = 0; round < 3; round++) {
        CX_HIP(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(probe_mfma_kernel, dim3(grid), dim3(256), 0, nullptr, iters, sink);
        CX_HIP(hipEventRecord(e1, nullptr));
        CX_HIP(hipEventSynchronize(e1));
        float ms = 0.0f;
        CX_HIP(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)grid * 4.0 * iters * 4.0 * (2.0 * 32 * 32 * 16);
        tf = flops / (ms * 1e-3) / 1e12;
        if (round == 0 && ms > 0.0f) {   // size the measured rounds to ms_target
            const double want = ms_target / ms * iters;
            iters = (uint32_t)(want < 1000.0 ? 1000.0 : (want > 4.0e8 ? 4.0e8 : want));
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    *out_tflops = tf;
    return CX_OK;
}

}  // extern "C"
