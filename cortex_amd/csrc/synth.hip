// synth.hip — on-device synthetic embedding generator (bench/test support).
// Clustered mixture of unit rows from a counter-based Philox4x32-10 stream;
// every float operation is a single IEEE op in a fixed order so that the CPU
// twin produces the same bits.  One wave per row, lane l owns columns
// l, l+64, ...; the row norm is 64 lane-strided sequential partial sums
// folded 32/16/8/4/2/1.
#include "common.hpp"
#include "../../include/cortex_hip_synth.h"

// every float op below must be a single IEEE operation: no FMA contraction, correctly rounded
// sqrtf and divide (the default for HIP device code)
#pragma clang fp contract(off)

namespace cx {

__device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                     uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ inline float gauss(uint64_t seed, uint64_t idx, uint32_t stream) {
    uint32_t x[4];
    philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), x);
    const int32_t s = (int32_t)((x[0] >> 8) + (x[1] >> 8) + (x[2] >> 8) + (x[3] >> 8)) - 33554430;
    return (float)s * 1.0323829e-07f;
}

// fold the 64 lane partials in the twin's order; the total ends up in every lane
__device__ inline float fold64(float part) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) part = part + __shfl_down(part, s, 64);
    return __shfl(part, 0, 64);
}

__device__ inline void normalise_row(float *row, uint32_t d, uint32_t lane) {
    float a = 0.0f;
    for (uint32_t j = lane; j < d; j += 64u) { const float p = row[j] * row[j]; a = a + p; }
    const float n = sqrtf(fold64(a));
    for (uint32_t j = lane; j < d; j += 64u) row[j] = row[j] / n;
}

__global__ __launch_bounds__(256) void synth_centres_kernel(float *centres, uint64_t seed, uint64_t n_centres, uint32_t d) {
    const uint64_t c = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (c >= n_centres) return;
    float *row = centres + c * d;
    for (uint32_t j = lane; j < d; j += 64u) row[j] = gauss(seed, c * (uint64_t)d + j, 0u);
    normalise_row(row, d, lane);
}

__device__ inline float sigma_of(uint32_t x1) {
    return x1 < 858993459u ? 0.25f : (x1 < 2576980378u ? 0.42f : 0.60f);
}

__global__ __launch_bounds__(256) void synth_rows_kernel(float *out, const float *centres, uint64_t seed_rows,
                                                         uint64_t seed_dup, uint64_t n_centres, uint64_t row_lo,
                                                         uint64_t n_rows, uint32_t d, uint32_t flags) {
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (i >= n_rows) return;
    const uint64_t r = row_lo + i;
    float *row = out + i * d;
    uint32_t x[4];
    philox4x32_10((uint32_t)r, (uint32_t)(r >> 32), 1u, 0u, (uint32_t)seed_rows, (uint32_t)(seed_rows >> 32), x);
    const uint32_t m = (uint32_t)(r % 1000u);
    const bool dup = (flags & 1u) && m >= 998u && r >= 998u;
    uint64_t src = r;
    if (dup) {
        src = x[2] % r;
        if (src % 1000u >= 998u) src -= 2;
    }
    uint32_t xb[4];
    philox4x32_10((uint32_t)src, (uint32_t)(src >> 32), 1u, 0u, (uint32_t)seed_rows, (uint32_t)(seed_rows >> 32), xb);
    const float *centre = centres + (uint64_t)(xb[0] % n_centres) * d;
    const float inv_sqrt_d = 1.0f / sqrtf((float)d);
    const float amp = sigma_of(xb[1]) * inv_sqrt_d;
    for (uint32_t j = lane; j < d; j += 64u)
        { const float t = amp * gauss(seed_rows, src * (uint64_t)d + j, 2u); row[j] = centre[j] + t; }
    normalise_row(row, d, lane);
    if (dup && m == 998u) {
        const float amp2 = 0.045f * inv_sqrt_d;
        for (uint32_t j = lane; j < d; j += 64u)
            { const float t = amp2 * gauss(seed_dup, r * (uint64_t)d + j, 3u); row[j] = row[j] + t; }
        normalise_row(row, d, lane);
    }
    if (flags & 2u) {
        const float u = (float)(x[3] >> 8) * 5.9604645e-08f;
        const float s15 = 1.5f * u;
        const float s = 0.5f + s15;
        for (uint32_t j = lane; j < d; j += 64u) row[j] = row[j] * s;
    }
}

}  // namespace cx

extern "C" int cx_synth_fill_dev(int device, float *d_out, uint64_t seed_centres, uint64_t seed_rows,
                                 uint64_t seed_dup, uint64_t n_centres, uint64_t row_lo, uint64_t n_rows,
                                 uint32_t dim, uint32_t flags) {
    using namespace cx;
    if (!d_out || !dim || !n_centres) return set_err(CX_ERR_VALIDATION, "synth: null output or zero dim/centres");
    if (!n_rows) return CX_OK;
    CX_HIP(hipSetDevice(device));
    float *centres = nullptr;
    CX_HIP(hipMalloc((void **)&centres, n_centres * dim * sizeof(float)));
    const uint64_t cb = (n_centres + 3) / 4, rb = (n_rows + 3) / 4;
    if (cb > 0x7FFFFFFFull || rb > 0x7FFFFFFFull) {
        (void)hipFree(centres);
        return set_err(CX_ERR_VALIDATION, "synth: too many rows for one launch");
    }
    hipLaunchKernelGGL(synth_centres_kernel, dim3((uint32_t)cb), dim3(256), 0, 0, centres, seed_centres, n_centres, dim);
    hipLaunchKernelGGL(synth_rows_kernel, dim3((uint32_t)rb), dim3(256), 0, 0, d_out, centres, seed_rows, seed_dup,
                       n_centres, row_lo, n_rows, dim, flags);
    hipError_t e = hipDeviceSynchronize();
    (void)hipFree(centres);
    if (e != hipSuccess) return set_err(CX_ERR_DEVICE, "synth kernels failed: %s", hipGetErrorString(e));
    return CX_OK;
}
