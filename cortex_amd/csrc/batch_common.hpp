// batch_common.hpp — the bf16 hi/lo split of an f32 value and the layout of the index's split store (cx_index::d_split,
// batch.hip).
//
// Split store, "fragment-major": a 16-row tile t (rows 16 t .. 16 t + 15) is D / 32 K-steps of 2 KiB:
//   [K-step ks: hi fragment 1 KiB | lo fragment 1 KiB], fragment = 64 lanes x 16 bytes, lane = 16 kq + i holding
//   elements 32 ks + 8 kq .. + 7 of row i as bf16 —
// byte for byte the A operand of v_mfma_f32_16x16x32_bf16 as its lanes hold it: a block that copies the tile into LDS
// linearly reads it back conflict-free with ds_read_b128 at `fragment + 16 lane` (batch.hip); a wave that loads 16 bytes
// per lane at the same offsets gets the operand straight into registers with one fully coalesced 1 KiB access (the layout
// batchs.hip's screening store uses for its single bf16 fragment per K-step).  (Rounds 1-2 kept row-major bf16 images
// with an XOR swizzle: the same bytes, permuted inside a tile.)
#pragma once

#include "common.hpp"

namespace cx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int SPLIT_TILE_ROWS = 16;
constexpr uint32_t SPLIT_STEP_BYTES = 2048;    // one K-step of a tile: hi fragment | lo fragment
constexpr uint32_t SPLIT_FRAG_BYTES = 1024;
__host__ __device__ constexpr size_t split_tile_bytes(uint32_t dim) { return (size_t)SPLIT_TILE_ROWS * dim * 4u; }

// byte offset, inside its tile, of the 8-byte piece that holds elements col .. col + 3 (col % 4 == 0) of tile row i
// (hi fragment; the lo fragment's piece sits SPLIT_FRAG_BYTES further)
__host__ __device__ inline uint32_t split_piece_off(uint32_t i, uint32_t col) {
    return (col >> 5) * SPLIT_STEP_BYTES + ((((col >> 3) & 3u) << 4) + i) * 16u + ((col >> 2) & 1u) * 8u;
}

#ifdef __HIPCC__
// hi = bf16(v) (round to nearest even, NaN stays NaN), lo = bf16(v - hi).  The packed conversion's result is unpacked
// with one shift / one mask per pair instead of being converted a second time element by element.
__device__ inline void split4(const f32x4 v, bf16x4_t &hi, bf16x4_t &lo) {
    const uint32_t p01 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){v.x, v.y}, bf16x2_t));
    const uint32_t p23 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){v.z, v.w}, bf16x2_t));
    const f32x2 b01 = {__uint_as_float(p01 << 16), __uint_as_float(p01 & 0xFFFF0000u)};
    const f32x2 b23 = {__uint_as_float(p23 << 16), __uint_as_float(p23 & 0xFFFF0000u)};
    const f32x2 d01 = (f32x2){v.x, v.y} - b01, d23 = (f32x2){v.z, v.w} - b23;   // v_pk_add_f32
    const uint32_t q01 = __builtin_bit_cast(uint32_t, __builtin_convertvector(d01, bf16x2_t));
    const uint32_t q23 = __builtin_bit_cast(uint32_t, __builtin_convertvector(d23, bf16x2_t));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(bf16x4_t, (u32x2){p01, p23});
    lo = __builtin_bit_cast(bf16x4_t, (u32x2){q01, q23});
}

// eight consecutive f32 values -> the hi and lo halves of an MFMA operand piece
__device__ inline void split8(const f32x4 v0, const f32x4 v1, s16x8 &hi, s16x8 &lo) {
    bf16x4_t h0, l0, h1, l1;
    split4(v0, h0, l0);
    split4(v1, h1, l1);
    hi = __builtin_bit_cast(s16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
    lo = __builtin_bit_cast(s16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
}

// minimum over the 64 lanes, result in every lane
__device__ inline uint32_t wave_min_u32(uint32_t v) {
    uint32_t o;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);  v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);  v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true); v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true); v = o < v ? o : v;
    o = (uint32_t)__shfl_xor((int)v, 16, 64); v = o < v ? o : v;
    o = (uint32_t)__shfl_xor((int)v, 32, 64); v = o < v ? o : v;
    return v;
}
#endif

}  // namespace cx
