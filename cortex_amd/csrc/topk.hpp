// topk.hpp — wavefront-level top-k list kept in registers (gfx950, wave64).
//
// A wave holds 64*KS entries sorted best-first across its lanes: rank
// r = s*64 + lane lives in (key[s], sim[s]) of that lane.  Keys are the 64-bit
// order keys of common.hpp (larger = better, 0 = empty, so empties sit at the
// tail).  `tau` is the key at rank k-1 and is wave-uniform: a candidate is
// worth inserting only if its key exceeds tau, so after warm-up almost every
// row is rejected by one scalar compare.
//
// Insertion is a sorted insert: one ballot per slot gives the position
// (entries better than the newcomer form a prefix), every worse entry moves
// one rank down with a single DPP wave shift (wave_shr:1 — VALU rate, no LDS
// crossbar round trip), the newcomer drops into the hole and tau is re-read
// with v_readlane.  ~20 instructions, no dependent shuffle chain.
#pragma once

#include "common.hpp"

namespace cx {

__device__ inline int lane_id() { return (int)(threadIdx.x & 63u); }

__device__ inline uint64_t readlane_u64(uint64_t v, int src) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
__device__ inline float readlane_f32(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// lane i <- v of lane i-1; lane 0 <- carry (DPP wave_shr:1, bound_ctrl off keeps `old`)
__device__ inline uint32_t wave_shr1(uint32_t carry, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, 0x138, 0xf, 0xf, false);
}

template <int KS>
struct WaveTopK {
    uint64_t key[KS];
    float sim[KS];
    uint64_t tau;
    uint32_t k;

    __device__ void init(uint32_t k_) {
#pragma unroll
        for (int s = 0; s < KS; s++) {
            key[s] = 0ull;
            sim[s] = 0.0f;
        }
        k = k_;
        tau = k_ ? 0ull : ~0ull;
    }

    // kn, sn wave-uniform; requires kn > tau (hence its rank is < k)
    __device__ void insert(uint64_t kn, float sn) {
        const uint32_t lane = (uint32_t)lane_id();
        uint32_t pos = 0;
#pragma unroll
        for (int s = 0; s < KS; s++) pos += (uint32_t)__popcll(__ballot(key[s] > kn));
        const uint32_t kn_lo = (uint32_t)kn, kn_hi = (uint32_t)(kn >> 32), sn_b = __float_as_uint(sn);
        uint32_t c_lo = 0, c_hi = 0, c_s = 0;  // what enters lane 0 of the current slot
#pragma unroll
        for (int s = 0; s < KS; s++) {
            uint32_t lo = (uint32_t)key[s], hi = (uint32_t)(key[s] >> 32), sb = __float_as_uint(sim[s]);
            const uint32_t n_lo = (uint32_t)__builtin_amdgcn_readlane((int)lo, 63);
            const uint32_t n_hi = (uint32_t)__builtin_amdgcn_readlane((int)hi, 63);
            const uint32_t n_s = (uint32_t)__builtin_amdgcn_readlane((int)sb, 63);
            const uint32_t sh_lo = wave_shr1(c_lo, lo), sh_hi = wave_shr1(c_hi, hi), sh_s = wave_shr1(c_s, sb);
            const uint32_t r = (uint32_t)s * 64u + lane;
            if (r > pos) { lo = sh_lo; hi = sh_hi; sb = sh_s; }
            else if (r == pos) { lo = kn_lo; hi = kn_hi; sb = sn_b; }
            key[s] = ((uint64_t)hi << 32) | lo;
            sim[s] = __uint_as_float(sb);
            c_lo = n_lo; c_hi = n_hi; c_s = n_s;
        }
        const uint32_t last = k - 1u;
#pragma unroll
        for (int s = 0; s < KS; s++)
            if ((last >> 6) == (uint32_t)s) tau = readlane_u64(key[s], (int)(last & 63u));
    }

    // offer the (per-lane) candidates of a whole wave: lanes whose key beats
    // tau are inserted one by one (uniform loop over the ballot mask)
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

    // rank i = s*64 + lane of the list, written to dst[i] for i < k (0 for empty)
    __device__ void store(uint64_t *dk, float *ds) const {
        const uint32_t lane = (uint32_t)lane_id();
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const uint32_t i = (uint32_t)s * 64u + lane;
            if (i < k) { dk[i] = key[s]; ds[i] = sim[s]; }
        }
    }
};

}  // namespace cx
