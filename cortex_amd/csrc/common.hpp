// common.hpp — error plumbing and shared device-side types for libcortex_hip.
#pragma once

#include <hip/hip_runtime.h>
#include <atomic>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/cortex_hip.h"

namespace cx {

// Thread-local message behind cx_last_error(); mapped to
// CortexError::Validation(String) by the caller's shim.
char *err_buf();
int set_err(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
// inside a catch (...) of an extern "C" entry point: the exception becomes a status + message (nothing unwinds
// across the C ABI)
int on_exception() noexcept;

#define CX_HIP(expr)                                                                       \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess)                                                             \
            return ::cx::set_err(CX_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr,            \
                                 hipGetErrorString(e__), __FILE__, __LINE__);              \
    } while (0)

// true exactly once per (call site, device): kernels that need more than 64 KiB of LDS set their
// hipFuncAttributeMaxDynamicSharedMemorySize on every device they are launched on (the attribute is per device;
// a process may hold indexes on several).  `mask` is the call site's static state.
inline bool first_use_on_device(std::atomic<uint64_t> &mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;
    const uint64_t bit = 1ull << dev;
    return (mask.fetch_or(bit) & bit) == 0;
}

// compute units of the current device (cached per device: hipGetDeviceProperties is slow)
inline uint32_t device_cus() {
    static std::atomic<uint32_t> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return 256;
    uint32_t v = cache[dev].load(std::memory_order_relaxed);
    if (!v) {
        hipDeviceProp_t p;
        v = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? (uint32_t)p.multiProcessorCount : 256u;
        cache[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

// A candidate's order key: (ord(score) << 32) | ~row.  Larger = better, so
// "score descending, then row ascending"; 0 = empty slot.  ord() maps a
// clamped score to an integer that grows with it: NaN -> 1 (after every
// number), x >= +0.0 -> bits(x) + 2.
__host__ __device__ inline uint32_t score_ord(float score) {
    if (score != score) return 1u;
    uint32_t b;
    memcpy(&b, &score, 4);
    return b + 2u;
}
__host__ __device__ inline uint64_t make_key(float score, uint32_t row) {
    return ((uint64_t)score_ord(score) << 32) | (uint32_t)(~row);
}
__host__ __device__ inline uint32_t key_row(uint64_t key) { return ~(uint32_t)key; }

// Device-side view of a VectorFilter plus per-row metadata
// (vector/index.rs:225-251).  meta[row]: bit0 = removed, bit1 = has metadata,
// bits 8.. = kind code; agent[row] = agent code.
struct DevFilter {
    const uint32_t *meta;          // [rows] never null once rows exist
    const uint32_t *agent;         // [rows]
    const uint32_t *exclude_rows;  // sorted ascending, n_exclude entries (device)
    const uint32_t *kind_codes;    // n_kinds entries (device)
    uint32_t n_exclude;
    uint32_t n_kinds;
    uint32_t has_kinds;
    uint32_t has_agent;
    uint32_t agent_code;
    uint32_t trivial;              // 1 = no filter and no removed rows: every row passes, skip the metadata read
};

constexpr uint32_t META_REMOVED = 1u;
constexpr uint32_t META_HAS = 2u;

// (m = f.meta[row], already loaded: callers that look at several rows issue their metadata loads together)
__device__ inline bool row_passes_meta(const DevFilter &f, uint32_t row, uint32_t m);
__device__ inline bool row_passes(const DevFilter &f, uint32_t row) {
    if (f.trivial) return true;
    return row_passes_meta(f, row, f.meta[row]);
}
__device__ inline bool row_passes_meta(const DevFilter &f, uint32_t row, uint32_t m) {
    if (m & META_REMOVED) return false;
    for (uint32_t i = 0; i < f.n_exclude; i++)
        if (f.exclude_rows[i] == row) return false;
    if (m & META_HAS) {  // filters bind only where metadata exists (:234)
        if (f.has_kinds) {
            const uint32_t kind = m >> 8;
            bool found = false;
            for (uint32_t i = 0; i < f.n_kinds; i++) found |= (f.kind_codes[i] == kind);
            if (!found) return false;
        }
        if (f.has_agent && f.agent[row] != f.agent_code) return false;
    }
    return true;
}

// The reference's epilogue, one IEEE operation per step
// (vector/index.rs:173-177 and :254-256).  sqrtf and '/' are correctly rounded
// in HIP device code (-fhip-fp32-correctly-rounded-divide-sqrt, the default);
// the __fsqrt_rn/__fdiv_rn "intrinsics" of this toolchain are NOT (native
// approximations), so they are not used anywhere.
__device__ inline float cosine_from_sums(float dot, float qq, float rr) {
#pragma clang fp contract(off)
    const float norm_a = sqrtf(qq);
    const float norm_b = sqrtf(rr);
    const float den = norm_a * norm_b;
    return dot / den;
}
// the same value from the two square roots (callers that reuse a norm over many pairs)
__device__ inline float cosine_from_norms(float dot, float norm_a, float norm_b) {
#pragma clang fp contract(off)
    const float den = norm_a * norm_b;
    return dot / den;
}
__host__ __device__ inline float distance_of(float sim) { return 1.0f - sim; }
// bf16 row stores (cx_create_ex, CX_DTYPE_BF16): an element is the upper half of an f32
__host__ __device__ inline float bf16_bits_to_f32(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// round to nearest even (what v_cvt_pk_bf16_f32 does); NaN stays NaN, overflow goes to infinity
__host__ __device__ inline uint16_t f32_to_bf16_bits(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x0040u);   // quiet NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
#ifdef __HIPCC__
// sum over the 64 lanes, result in every lane: four DPP steps inside each 16-lane row (VALU rate), then two
// cross-row exchanges
__device__ inline float wave_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));  // row_mirror
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
// row elements of either store type
__device__ inline float ldf(const float *p) { return *p; }
__device__ inline float ldf(const uint16_t *p) { return bf16_bits_to_f32(*p); }
__device__ inline void stf(float *p, float v) { *p = v; }
__device__ inline void stf(uint16_t *p, float v) { *p = f32_to_bf16_bits(v); }
typedef float cx_f32x4 __attribute__((ext_vector_type(4)));
// elements 4 j .. 4 j + 3 of row `row` (dim % 4 == 0)
__device__ inline cx_f32x4 row4(const float *rows, size_t row, uint32_t dim, uint32_t j) {
    return reinterpret_cast<const cx_f32x4 *>(rows + row * dim)[j];
}
__device__ inline cx_f32x4 row4(const uint16_t *rows, size_t row, uint32_t dim, uint32_t j) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 w = reinterpret_cast<const u32x2 *>(rows + row * dim)[j];
    return cx_f32x4{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xFFFF0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xFFFF0000u)};
}
#endif
__host__ __device__ inline float score_of(float distance) {
    float s = 1.0f - distance;
    if (s < 0.0f) s = 0.0f;
    if (s > 1.0f) s = 1.0f;
    return s;  // NaN falls through both compares
}

}  // namespace cx
