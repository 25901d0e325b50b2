// select.hpp — block-wide exact selection of the k-th largest key over keys held in registers (scan.hip's candidate and
// bound selections, batchs.hip's list selection).
#pragma once

#include "common.hpp"

namespace cx {

#ifdef __HIPCC__
// For the 256 threads that own a histogram bin: the number of entries in the bins ABOVE the thread's own.  Suffix sums
// inside each of the four waves by shuffles, then the waves' totals through four LDS words — every thread summing up to
// 255 bins itself took ~12 us per pass.  Called by every thread of the block (it contains a barrier); wtot: 4 uint32 of LDS.
__device__ inline uint32_t bins_above(const uint32_t *hist, uint32_t tid, uint32_t *wtot, uint32_t &mine) {
    uint32_t suf = 0;
    mine = 0;
    if (tid < 256u) {
        mine = hist[tid];
        suf = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_down(suf, off, 64);
            if ((tid & 63u) + (uint32_t)off < 64u) suf += t;
        }
        if ((tid & 63u) == 0u) wtot[tid >> 6] = suf;
    }
    __syncthreads();
    uint32_t above = suf - mine;
    if (tid < 256u)
        for (uint32_t w = (tid >> 6) + 1u; w < 4u; w++) above += wtot[w];
    return above;
}

// Block-wide exact selection over keys the threads hold in REGISTERS (0 = empty): the value of the k-th largest key,
// counting duplicates, or 0 when fewer than k keys are set.  A radix walk from bit `hi` down to bit `lo` (multiples of 8),
// one 256-bin LDS histogram and four barriers per byte — no global memory, so a pass costs a microsecond instead of a
// round trip per element.  Every thread of the block must call it; sh: 264 uint32 of LDS.
template <int NV, typename K = uint64_t>
__device__ inline K block_select_kth(const K (&key)[NV], uint32_t k, int hi, int lo, uint32_t *sh) {
    uint32_t *hist = sh;                      // [256]
    uint32_t *s_need = sh + 256, *s_found = sh + 257, *s_bin = sh + 258, *s_ok = sh + 259, *wtot = sh + 260;
    // once the bin that holds the answer has no more than 64 keys they are ranked directly by one wave: a walk over 64-bit
    // keys (score ordinal | ~row) needs two or three histogram passes instead of eight, one over 32-bit ordinals two
    __shared__ K s_list[64];
    __shared__ uint32_t s_ln, s_inbin;
    __shared__ K s_res;
    const uint32_t tid = threadIdx.x;
    K prefix = 0, mask = 0;
    if (tid == 0) { *s_need = k; *s_ok = 1u; }
    for (int shift = hi; shift >= lo; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        if (tid == 0) { *s_found = 0u; s_ln = 0u; }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NV; u++) {
            const bool act = key[u] != 0 && (key[u] & mask) == prefix;
            const uint32_t bin = (uint32_t)(key[u] >> shift) & 255u;
            const uint64_t am = __ballot(act);
            if (am == 0ull) continue;
            // leading bytes of score keys are nearly constant: a wave that lands in one bin adds its count once
            const int first = __ffsll((unsigned long long)am) - 1;
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)bin, first);
            if (__ballot(act && bin == b0) == am) {
                if ((int)(tid & 63u) == first) atomicAdd(&hist[b0], (uint32_t)__popcll(am));
            } else if (act) {
                atomicAdd(&hist[bin], 1u);
            }
        }
        __syncthreads();
        {
            uint32_t mine;
            const uint32_t above = bins_above(hist, tid, wtot, mine);
            const uint32_t need = *s_need;
            if (tid < 256 && above < need && need <= above + mine) { *s_bin = tid; *s_found = need - above; s_inbin = mine; }   // found: the new need (>= 1)
        }
        __syncthreads();
        const uint32_t found = *s_found;
        if (!found) { if (tid == 0) *s_ok = 0u; break; }   // uniform: every thread reads the same LDS word
        const uint32_t in_bin = s_inbin;
        prefix |= (K)(*s_bin) << shift;
        mask |= (K)0xFF << shift;
        if (in_bin <= 64u && shift > lo) {   // uniform.  The found-th largest of the bin's keys, ranked by one wave
#pragma unroll
            for (int u = 0; u < NV; u++)
                if (key[u] != 0 && (key[u] & mask) == prefix) s_list[atomicAdd(&s_ln, 1u)] = key[u];
            __syncthreads();
            if (tid < 64u) {
                const K mine_k = tid < in_bin ? s_list[tid] : (K)0;
                uint32_t rank = 0;   // keys that come before this one: larger, or equal and earlier in the list
                for (uint32_t j = 0; j < in_bin; j++) {
                    const K o = s_list[j];
                    rank += (o > mine_k || (o == mine_k && j < tid)) ? 1u : 0u;
                }
                if (tid < in_bin && rank == found - 1u) s_res = mine_k;
            }
            __syncthreads();
            return s_res;
        }
        __syncthreads();
        if (tid == 0) *s_need = found;
    }
    __syncthreads();
    return *s_ok ? prefix : (K)0;
}
#endif

}  // namespace cx
