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

// Operands are RANDOM (hashed mantissas and signs): the clock the chip holds under an MFMA-dense loop depends on the data —
// the same loop on near-constant operands ran at 2.24 PFLOP/s, on random ones at 1.78-1.81 (scripts/probes/
// mfma_shape_probe.hip; MI355X_MICROARCH.md, DVFS give-back) — and the kernels this is a ceiling for see random data.
__device__ inline uint32_t probe_hash(uint32_t x) { x *= 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; return x; }
__device__ inline bf16x8 probe_rand_frag(uint32_t seed) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (short)(0x3C00 | (probe_hash(seed * 8u + (uint32_t)e) & 0x83FF));   // +-[0.0078, 0.0156)
    return v;
}
__global__ __launch_bounds__(256) void probe_mfma_kernel(uint32_t iters, float *sink) {
    const bf16x8 a = probe_rand_frag(2u * (blockIdx.x * 256u + threadIdx.x)), b = probe_rand_frag(2u * (blockIdx.x * 256u + threadIdx.x) + 1u);
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

// The filter GEMM's inner step without its global traffic: per step a wave reads 12 fragments from LDS (ds_read_b128,
// conflict-free: the filter GEMM's operand pattern) and issues 16 v_mfma_f32_32x32x16_bf16 on them, 2 waves per SIMD.
// What the matrix pipe sustains when it is fed from LDS at the GEMM's ratio — the ceiling of that kernel's main loop.
template <int DMA>   // 0 none, 1 LDS-DMA, 2 register-staged (global_load_dwordx4 -> ds_write_b128, three steps in flight)
__global__ __launch_bounds__(512) void probe_mfma_lds_kernel(uint32_t iters, float *sink, const char *src) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 32 KiB read slot (A 16 KiB + B 16 KiB) [+ 96 KiB of DMA targets]
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t i = tid; i < 32768u / 16u; i += 512u)   // random bf16 operands (see probe_rand_frag)
        *reinterpret_cast<bf16x8 *>(smem + i * 16u) = probe_rand_frag(blockIdx.x * 2048u + i);
    __syncthreads();
    const uint32_t wm = wave >> 2, wn = wave & 3u, fr = lane & 31u, fq = lane >> 5;
    uint32_t offA[4][2], offB[2][2];
    for (uint32_t m = 0; m < 4; m++)
        for (uint32_t h = 0; h < 2; h++) {
            const uint32_t row = wm * 128u + m * 32u + fr, piece = 2u * h + fq;
            offA[m][h] = row * 64u + ((piece ^ ((row >> 3) & 3u)) << 4);
        }
    for (uint32_t n = 0; n < 2; n++)
        for (uint32_t h = 0; h < 2; h++) {
            const uint32_t row = wn * 64u + n * 32u + fr, piece = 2u * h + fq;
            offB[n][h] = 16384u + row * 64u + ((piece ^ ((row >> 3) & 3u)) << 4);
        }
    f32x16 acc[4][2];
    for (int m = 0; m < 4; m++)
        for (int n = 0; n < 2; n++)
            for (int e = 0; e < 16; e++) acc[m][n][e] = 0.0f;
    f32x4 g0[4], g1[4], g2[4];
    if constexpr (DMA == 2) {
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) { g0[q] = f32x4{0, 0, 0, 0}; g1[q] = g0[q]; g2[q] = g0[q]; }
    }
    auto staged = [&](f32x4 (&g)[4], uint32_t it) {
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            *reinterpret_cast<f32x4 *>(smem + 32768u + ((it % 3u) * 32u + wave * 4u + q) * 1024u + lane * 16u) = g[q];   // store, then reload
            g[q] = *reinterpret_cast<const f32x4 *>(src + (((it * 8u + wave) * 4u + q) & 1023u) * 1024u + lane * 16u);
        }
    };
    for (uint32_t it = 0; it < iters; it += (DMA == 2 ? 3u : 1u)) {
        if constexpr (DMA == 2) {
            // three steps per loop iteration, one register set each
            for (uint32_t sub = 0; sub < 3u; sub++) {
                if (sub == 0) staged(g0, it); else if (sub == 1) staged(g1, it + 1); else staged(g2, it + 2);
                bf16x8 fa[8], fb[4];
#pragma unroll
                for (int n = 0; n < 2; n++)
#pragma unroll
                    for (int h = 0; h < 2; h++) fb[n * 2 + h] = *reinterpret_cast<const bf16x8 *>(smem + offB[n][h]);
#pragma unroll
                for (int m = 0; m < 4; m++)
#pragma unroll
                    for (int h = 0; h < 2; h++) fa[m * 2 + h] = *reinterpret_cast<const bf16x8 *>(smem + offA[m][h]);
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int m = 0; m < 4; m++)
#pragma unroll
                        for (int n = 0; n < 2; n++)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m * 2 + h], fb[n * 2 + h], acc[m][n], 0, 0, 0);
                asm volatile("" ::: "memory");
            }
            continue;
        }
        if constexpr (DMA == 1) {   // the GEMM's global traffic too: 4 LDS-DMAs of 1 KiB per wave and step, from an L2-resident buffer
#pragma unroll
            for (uint32_t q = 0; q < 4; q++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (((it * 8u + wave) * 4u + q) & 1023u) * 1024u + lane * 16u),
                                                 (__attribute__((address_space(3))) void *)(smem + 32768u + ((it % 3u) * 32u + wave * 4u + q) * 1024u), 16, 0, 0);
            if ((it & 1u) == 1u) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        bf16x8 fa[8], fb[4];
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int h = 0; h < 2; h++) fb[n * 2 + h] = *reinterpret_cast<const bf16x8 *>(smem + offB[n][h]);
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int h = 0; h < 2; h++) fa[m * 2 + h] = *reinterpret_cast<const bf16x8 *>(smem + offA[m][h]);
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < 2; n++)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m * 2 + h], fb[n * 2 + h], acc[m][n], 0, 0, 0);
        asm volatile("" ::: "memory");   // the reads are re-issued every step
    }
    float v = 0.0f;
    for (int m = 0; m < 4; m++)
        for (int n = 0; n < 2; n++)
            for (int e = 0; e < 16; e++) v += acc[m][n][e];
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

/* As cx_probe_mfma_tflops, but every step's operands are read from LDS at the filter GEMM's ratio (12 ds_read_b128 per
 * 16 MFMAs per wave, 8 waves per CU); with_dma adds its global traffic (4 LDS-DMAs of 1 KiB per wave and step from an
 * L2-resident buffer, counted waits, no barrier). */
int cx_probe_mfma_lds_tflops(int device, double ms_target, int with_dma, double *out_tflops) {
    using namespace cx;
    if (!out_tflops) return set_err(CX_ERR_VALIDATION, "probe: null output");
    CX_HIP(hipSetDevice(device));
    float *sink = nullptr;
    char *src = nullptr;
    CX_HIP(hipMalloc((void **)&sink, 4));
    CX_HIP(hipMalloc((void **)&src, 1 << 20));
    CX_HIP(hipMemset(src, 0x3c, 1 << 20));
    static bool attr = false;
    if (!attr) {
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(probe_mfma_lds_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(probe_mfma_lds_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        attr = true;
    }
    hipEvent_t e0, e1;
    CX_HIP(hipEventCreate(&e0));
    CX_HIP(hipEventCreate(&e1));
    const uint32_t grid = device_cus();
    uint32_t iters = 5000;
    double tf = 0.0;
    for (int round = 0; round < 3; round++) {
        CX_HIP(hipEventRecord(e0, nullptr));
        if (with_dma == 2) hipLaunchKernelGGL(probe_mfma_lds_kernel<2>, dim3(grid), dim3(512), 131072, nullptr, iters / 3 * 3, sink, src);
        else if (with_dma) hipLaunchKernelGGL(probe_mfma_lds_kernel<1>, dim3(grid), dim3(512), 131072, nullptr, iters, sink, src);
        else hipLaunchKernelGGL(probe_mfma_lds_kernel<0>, dim3(grid), dim3(512), 32768, nullptr, iters, sink, src);
        CX_HIP(hipEventRecord(e1, nullptr));
        CX_HIP(hipEventSynchronize(e1));
        float ms = 0.0f;
        CX_HIP(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)grid * 8.0 * (with_dma == 2 ? iters / 3 * 3 : iters) * 16.0 * (2.0 * 32 * 32 * 16);
        tf = flops / (ms * 1e-3) / 1e12;
        if (round == 0 && ms > 0.0f) {
            const double want = ms_target / ms * iters;
            iters = (uint32_t)(want < 500.0 ? 500.0 : (want > 1.0e8 ? 1.0e8 : want));
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    (void)hipFree(src);
    *out_tflops = tf;
    return CX_OK;
}

}  // extern "C"
