// topk.hpp — wavefront-level top-k list kept in registers (gfx950, wave64).
//
// A wave owns up to 64*KS slots, slot (s, lane) enabled when s*64+lane < k.
// Keys are the 64-bit order keys of common.hpp (larger = better, 0 = empty).
// `tau` is the smallest enabled key and is wave-uniform: a candidate is
// worth inserting only if its key exceeds tau, so after warm-up almost every
// row is rejected by one scalar compare.  Insertion replaces the slot holding
// tau and recomputes tau with a 6-step butterfly — rare, so its cost does not
// matter next to the HBM stream.
#pragma once

#include "common.hpp"

namespace cx {

__device__ inline int lane_id() { return (int)(threadIdx.x & 63u); }

__device__ inline uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint64_t o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

__device__ inline uint64_t readlane_u64(uint64_t v, int src) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
__device__ inline float readlane_f32(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

template <int KS>
struct WaveTopK {
    uint64_t key[KS];
    float sim[KS];
    uint64_t tau;

    __device__ void init(uint32_t k) {
        const uint32_t lane = (uint32_t)lane_id();
#pragma unroll
        for (int s = 0; s < KS; s++) {
            key[s] = ((uint32_t)s * 64u + lane < k) ? 0ull : ~0ull;  // ~0 = disabled slot
            sim[s] = 0.0f;
        }
        tau = k ? 0ull : ~0ull;
    }

    // kn, sn wave-uniform; requires kn > tau
    __device__ void insert(uint64_t kn, float sn) {
        const int lane = lane_id();
        bool done = false;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const uint64_t m = __ballot(key[s] == tau);
            if (!done && m) {
                const int src = __ffsll((unsigned long long)m) - 1;
                if (lane == src) { key[s] = kn; sim[s] = sn; }
                done = true;
            }
        }
        uint64_t loc = key[0];
#pragma unroll
        for (int s = 1; s < KS; s++) loc = key[s] < loc ? key[s] : loc;
        tau = wave_min_u64(loc);
    }

    // offer the (per-lane) candidates of a whole wave: lanes with key > tau
    // are inserted one by one (uniform loop over the ballot mask)
    template <typename Pred>
    __device__ void offer_lanes(uint64_t kg, float sg, Pred pass) {
        uint64_t m = __ballot(kg > tau);
        while (m) {
            const int l = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            const uint64_t kk = readlane_u64(kg, l);
            if (kk > tau && pass(key_row(kk))) insert(kk, readlane_f32(sg, l));
        }
    }

    // slot i = s*64 + lane of the list, written to dst[i] for i < k (0 for empty)
    __device__ void store(uint64_t *dk, float *ds, uint32_t k) const {
        const uint32_t lane = (uint32_t)lane_id();
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const uint32_t i = (uint32_t)s * 64u + lane;
            if (i < k) { dk[i] = key[s]; ds[i] = sim[s]; }
        }
    }
};

}  // namespace cx
