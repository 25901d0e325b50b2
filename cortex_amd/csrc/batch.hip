// batch.hip — batched exact search: up to 64 queries per pass over the corpus
// (VectorIndex::search_batch, vector/index.rs:390-410; BASELINE config 4's
// inner loop).  The reference runs B independent scans (rayon, one per query);
// here the row store is read ONCE per batch.
//
// At B = 64 the contraction needs 32 flop per corpus byte — past what f32 VALU
// FMAs sustain next to an 8 TB/s stream — so the dot products run on the
// matrix cores with f32 operands (v_mfma_f32_16x16x4_f32: exact f32 FMA chain,
// no split-precision tricks, scores stay within the 5e-5 parity tolerance by
// construction).
//
// Shape (gfx950, wave64, one 256-thread block per CU, one wave per SIMD):
//  - a wave owns 16 queries for the whole launch and keeps them in registers
//    (dim/4 VGPRs, the MFMA B operand: lane (j, kq) holds q_j[16g + 4kq + t]);
//  - the block streams 16-row tiles of the f32 row store into LDS with
//    LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, double
//    buffered: 48 KiB in flight per CU at dim 768); all four waves read every
//    tile as the MFMA A operand: lane (i, kq) takes the 16-byte piece
//    [16g + 4kq, +4) of row i with one ds_read_b128 and feeds four MFMAs.
//    16-byte pieces are XOR-swizzled inside each 256-byte segment (applied to
//    the DMA's per-lane source address and again on the read) so the lane
//    groups of ds_read_b128 never share a bank group.
//  - row and query norms are accumulated from the same registers; the
//    reference's epilogue runs per (row, query) on the 16x16 accumulator tile:
//    lane (j, g) holds rows 4g..4g+3 of query j.
//  - top-k: each query has a small candidate buffer in LDS and a threshold tau
//    (the k-th best key at its last compaction).  Candidates beating tau are
//    appended (one LDS atomic); a buffer that could overflow is compacted by
//    its wave (keys ranked with v_readlane broadcasts, best k kept).  After
//    warm-up almost no candidate passes.
//  - per-block lists go to HBM, one merge block per query finishes
//    (merge_small_kernel, grid = nq).
#include <vector>

#include "kernels.hpp"
#include "topk.hpp"

namespace cx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BT_ROWS = 16;   // rows per tile
constexpr int BT_Q = 64;      // queries per pass (4 waves x 16)

template <int D>
struct BatchCfg {
    static constexpr int PPR = D / 4;                     // 16-byte pieces per row
    static constexpr int TILE_BYTES = BT_ROWS * D * 4;
    static constexpr int DMA_PER_TILE = BT_ROWS * PPR / 64;  // 1 KiB wave instructions per tile
    static_assert(D % 64 == 0, "dim must be a multiple of 64 floats (256-byte swizzle segments)");
    static_assert(DMA_PER_TILE % 4 == 0, "tile DMAs must split evenly over 4 waves");
};

// position (in 16-byte pieces, inside the tile) of logical piece P of row i
template <int D>
__device__ inline uint32_t piece_pos(uint32_t i, uint32_t P) {
    return i * BatchCfg<D>::PPR + ((P & ~15u) | ((P ^ i) & 15u));
}

// key of a stored candidate (row, sim): the same order key the scan path uses
__device__ inline uint64_t cand_key(uint32_t row, float sim) { return make_key(score_of(distance_of(sim)), row); }

template <int D, bool DIAG>
__global__ __launch_bounds__(256, 1) void batch_scan_kernel(const BatchArgs a) {
    using C = BatchCfg<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: [2 tiles][cand rows: 64 x capq u32][cand dots: f32][cand |r|^2: f32][tau: 64 u64][cnt: 64 u32][tsq: 64 f32]
    char *tiles = smem;
    const uint32_t capq = a.capq;
    uint32_t *c_rows = reinterpret_cast<uint32_t *>(smem + 2 * C::TILE_BYTES);
    float *c_dots = reinterpret_cast<float *>(c_rows + BT_Q * capq);
    float *c_rrs = c_dots + BT_Q * capq;
    uint64_t *c_tau = reinterpret_cast<uint64_t *>(c_rrs + BT_Q * capq);
    uint32_t *c_cnt = reinterpret_cast<uint32_t *>(c_tau + BT_Q);
    float *c_tsq = reinterpret_cast<float *>(c_cnt + BT_Q);

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t j = lane & 15u, kq = lane >> 4;   // B/C column (query), k quarter
    const uint32_t qslot = wave * 16u + j;           // query slot of this lane inside the pass
    const uint32_t k = a.k, n_rows = a.n_rows;

    if (tid < BT_Q) { c_cnt[tid] = 0; c_tau[tid] = 0ull; c_tsq[tid] = -1.0f; }

    // queries -> registers; |q|^2
    float qreg[D / 4];
    {
        const bool live = qslot < a.nq;
        const f32x4 *q4 = reinterpret_cast<const f32x4 *>(a.queries + (size_t)(live ? qslot : 0) * D);
#pragma unroll
        for (int g = 0; g < D / 16; g++) {
            f32x4 v = q4[4 * g + kq];
            if (!live) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            qreg[4 * g + 0] = v.x; qreg[4 * g + 1] = v.y; qreg[4 * g + 2] = v.z; qreg[4 * g + 3] = v.w;
        }
    }
    float qq = 0.0f;
#pragma unroll
    for (int t = 0; t < D / 4; t++) qq += qreg[t] * qreg[t];
    qq += __shfl_xor(qq, 16, 64);
    qq += __shfl_xor(qq, 32, 64);

    // A-operand read offsets: lane (i = j, kq) reads piece 4g + kq of row i
    const uint32_t ai = j;
    uint32_t a_off[4];
#pragma unroll
    for (uint32_t gl = 0; gl < 4; gl++) a_off[gl] = (ai * C::PPR + (((4u * gl + kq) ^ ai) & 15u)) * 16u;

    const uint32_t n_tiles = (n_rows + BT_ROWS - 1) / BT_ROWS;
    // LDS-DMA source offsets (floats, relative to the tile's first row) never change: compute them once.
    // Tail tiles read up to 15 rows past n_rows; the row store is allocated with one tile of padding and
    // those rows are masked in the epilogue.
    uint32_t src_off[C::DMA_PER_TILE / 4];
#pragma unroll
    for (int e = 0; e < C::DMA_PER_TILE / 4; e++) {
        const uint32_t inst = wave * (C::DMA_PER_TILE / 4) + (uint32_t)e;
        const uint32_t gp = inst * 64u + lane;            // linear piece inside the tile = LDS position
        const uint32_t i = gp / C::PPR, P = gp % C::PPR;   // row and stored position inside the row
        const uint32_t srcP = (P & ~15u) | ((P ^ i) & 15u);  // XOR is an involution: position P holds logical piece srcP
        src_off[e] = i * (uint32_t)D + srcP * 4u;
    }
    auto stage = [&](uint32_t buf, uint32_t tile) {
        char *dst = tiles + buf * C::TILE_BYTES + wave * (C::DMA_PER_TILE / 4) * 1024u;
        const float *base = a.rows + (size_t)tile * BT_ROWS * D;
#pragma unroll
        for (int e = 0; e < C::DMA_PER_TILE / 4; e++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + src_off[e]),
                                             (__attribute__((address_space(3))) void *)(dst + e * 1024), 16, 0, 0);
    };

    // Compaction of one query's candidate buffer by this wave: the exact epilogue (sqrt, divide, clamp)
    // runs HERE, two entries per lane, not in the per-tile path — there the whole wave would execute it
    // whenever any of its 64 lanes had a candidate.  Keeps the best k in rank order, refreshes tau.
    // qq_of = |q|^2 of this query (wave-uniform).
    auto compact = [&](uint32_t qs, float qq_of) {
        const uint32_t n = c_cnt[qs] < capq ? c_cnt[qs] : capq;
        uint32_t *rws = c_rows + qs * capq;
        float *dts = c_dots + qs * capq, *rrs = c_rrs + qs * capq;
        uint32_t r0 = 0, r1 = 0; float d0 = 0.0f, d1 = 0.0f, n0 = 1.0f, n1 = 1.0f; uint64_t k0 = 0ull, k1 = 0ull;
        if (lane < n) { r0 = rws[lane]; d0 = dts[lane]; n0 = rrs[lane]; k0 = cand_key(r0, cosine_from_sums(d0, qq_of, n0)); }
        if (lane + 64u < n) { r1 = rws[lane + 64u]; d1 = dts[lane + 64u]; n1 = rrs[lane + 64u]; k1 = cand_key(r1, cosine_from_sums(d1, qq_of, n1)); }
        uint32_t rank0 = 0, rank1 = 0;
        const uint32_t n_lo = n < 64u ? n : 64u;
        for (uint32_t f = 0; f < n_lo; f++) {
            const uint64_t kf = readlane_u64(k0, (int)f);
            rank0 += kf > k0 ? 1u : 0u;
            rank1 += kf > k1 ? 1u : 0u;
        }
        for (uint32_t f = 64u; f < n; f++) {
            const uint64_t kf = readlane_u64(k1, (int)(f - 64u));
            rank0 += kf > k0 ? 1u : 0u;
            rank1 += kf > k1 ? 1u : 0u;
        }
        // all reads are done (they live in registers): rewrite the buffer in rank order
        auto set_tau = [&](uint64_t kk, float dt, float nr) {
            const float sm = cosine_from_sums(dt, qq_of, nr);
            c_tau[qs] = kk;
            c_tsq[qs] = sm > 0.0f ? sm * sm * (1.0f - 1.0e-5f) : -1.0f;  // NaN compares false -> -1
        };
        if (lane < n && rank0 < k) { rws[rank0] = r0; dts[rank0] = d0; rrs[rank0] = n0; if (rank0 == k - 1u) set_tau(k0, d0, n0); }
        if (lane + 64u < n && rank1 < k) { rws[rank1] = r1; dts[rank1] = d1; rrs[rank1] = n1; if (rank1 == k - 1u) set_tau(k1, d1, n1); }
        if (lane == 0) c_cnt[qs] = n < k ? n : k;
    };

    // diagnostic stamps (DIAG build only; never in the shipped kernel): cycles per phase, per wave
    unsigned long long t_stage = 0, t_mfma = 0, t_epi = 0, t_bar = 0, t_prev = 0;
    auto stamp = [&](unsigned long long &acc) {
        if constexpr (DIAG) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            acc += t - t_prev;
            t_prev = t;
        }
    };
    if (blockIdx.x < n_tiles) stage(0, blockIdx.x);
    __syncthreads();
    if constexpr (DIAG) { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); t_prev = t; }
    uint32_t buf = 0;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t next = tile + gridDim.x;
        if (next < n_tiles) stage(buf ^ 1u, next);

        // make room: any of this wave's queries that could overflow in this tile (16 appends max)
        {
            uint64_t need = __ballot(kq == 0u && c_cnt[qslot] + 16u > capq);
            while (need) {
                const int l = __ffsll((unsigned long long)need) - 1;
                need &= need - 1;
                compact(wave * 16u + (uint32_t)l, readlane_f32(qq, l));
            }
        }

        stamp(t_stage);
        const char *T = tiles + buf * C::TILE_BYTES;
        f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
        float ss = 0.0f;
        // A fragments are read a chunk (4 groups = 16 MFMAs) ahead of their use, with scheduling fences
        // around each chunk: an in-order wave cannot issue a ds_read behind MFMAs that are still waiting
        // for the matrix pipe, so a read placed next to its use exposes the whole LDS latency every four
        // MFMAs (measured: 2x the MFMA time per tile); hipcc sinks unfenced reads to exactly that spot.
        constexpr int G = D / 16, CH = 4, NCH = G / CH;
        static_assert(G % CH == 0, "dim/16 must be a multiple of the chunk");
        auto rd = [&](int g) { return *reinterpret_cast<const f32x4 *>(T + a_off[g & 3] + (uint32_t)(g >> 2) * 256u); };
        f32x4 fa[CH], fb[CH];
#pragma unroll
        for (int u = 0; u < CH; u++) fa[u] = rd(u);
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            f32x4 *cur = (c & 1) ? fb : fa, *nxt = (c & 1) ? fa : fb;
            if (c + 1 < NCH) {
#pragma unroll
                for (int u = 0; u < CH; u++) nxt[u] = rd((c + 1) * CH + u);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < CH; u++) {
                const int g = c * CH + u;
                const f32x4 av = cur[u];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, qreg[4 * g + 0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, qreg[4 * g + 1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, qreg[4 * g + 2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, qreg[4 * g + 3], acc1, 0, 0, 0);
                ss += av.x * av.x + av.y * av.y + av.z * av.z + av.w * av.w;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp(t_mfma);
        // |row i|^2 for i = lane & 15: fold the four k quarters
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);

        // C tile: lane (j, g = kq) holds rows 4*kq + r of query j.  The per-tile test has no sqrt/divide:
        // with tau's cosine t > 0, cos >= t  <=>  dot > 0 and dot^2 >= t^2 |q|^2 |r|^2 (a 1e-5 relative margin
        // keeps rounding from rejecting a row the exact epilogue would accept; the few extra rows it lets
        // through are ranked exactly at compaction).  Survivors are appended raw: (row, dot, |r|^2).
        const uint32_t row0 = tile * BT_ROWS;
        const float tsq = c_tsq[qslot];  // t^2 * (1 - margin), or -1 while the list is not full / t <= 0
        const float tq = tsq * qq;
        const bool live = qslot < a.nq;
#pragma unroll
        for (uint32_t r = 0; r < 4; r++) {
            const uint32_t li = 4u * kq + r;
            const float rr = __shfl(ss, (int)li, 64);
            const float dot = (r == 0 ? acc0.x + acc1.x : r == 1 ? acc0.y + acc1.y : r == 2 ? acc0.z + acc1.z : acc0.w + acc1.w);
            const uint32_t row = row0 + li;
            const bool maybe = tsq < 0.0f || (dot > 0.0f && dot * dot >= tq * rr) || !(dot == dot) || !(rr == rr);
            if (maybe && live && row < n_rows && row_passes(a.flt, row)) {
                const uint32_t slot = atomicAdd(&c_cnt[qslot], 1u);
                if (slot < capq) { c_rows[qslot * capq + slot] = row; c_dots[qslot * capq + slot] = dot; c_rrs[qslot * capq + slot] = rr; }
            }
        }
        stamp(t_epi);
        __syncthreads();  // next tile landed (LDS-DMA drained), this one is free
        stamp(t_bar);
        buf ^= 1u;
    }

    if constexpr (DIAG) {
        if (lane == 0) {
            unsigned long long *o = a.diag + ((size_t)blockIdx.x * 4 + wave) * 5;
            o[0] = t_stage; o[1] = t_mfma; o[2] = t_epi; o[3] = t_bar; o[4] = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
        }
    }
    // final compaction and per-block lists: part[(q * grid + block) * k + r]
    for (uint32_t l = 0; l < 16u; l++) compact(wave * 16u + l, readlane_f32(qq, (int)l));
    for (uint32_t l = 0; l < 16u; l++) {
        const uint32_t qs = wave * 16u + l;
        if (qs >= a.nq) break;
        const uint32_t n = c_cnt[qs];
        const size_t base = ((size_t)qs * gridDim.x + blockIdx.x) * k;
        const float qq_l = readlane_f32(qq, (int)l);
        if (lane < k) {
            const bool valid = lane < n;
            const uint32_t row = valid ? c_rows[qs * capq + lane] : 0u;
            const float sim = valid ? cosine_from_sums(c_dots[qs * capq + lane], qq_l, c_rrs[qs * capq + lane]) : 0.0f;
            a.part_keys[base + lane] = valid ? cand_key(row, sim) : 0ull;
            a.part_sims[base + lane] = sim;
        }
    }
}

uint32_t batch_grid_blocks(uint32_t n_rows) {
    int dev = 0, cus = 256;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0)
        cus = p.multiProcessorCount;
    const uint32_t tiles = (n_rows + BT_ROWS - 1) / BT_ROWS;
    return tiles < (uint32_t)cus ? (tiles ? tiles : 1u) : (uint32_t)cus;
}

bool batch_supported(uint32_t dim, uint32_t k) { return (dim == 384 || dim == 768) && k >= 1 && k <= 32; }

template <int D>
static int launch_batch_d(BatchArgs a, uint32_t grid, hipStream_t stream) {
    const size_t lds = 2 * (size_t)BatchCfg<D>::TILE_BYTES + (size_t)BT_Q * a.capq * 12 + BT_Q * 8 + BT_Q * 4 + BT_Q * 4;
    static bool attr_set = false;
    if (!attr_set) {
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(batch_scan_kernel<D, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(batch_scan_kernel<D, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    if (getenv("CX_BATCH_DIAG")) {  // diagnostic build: per-phase cycle shares on stderr, results still valid
        const size_t n = (size_t)grid * 4 * 5;
        CX_HIP(hipMalloc((void **)&a.diag, n * 8));
        hipLaunchKernelGGL((batch_scan_kernel<D, true>), dim3(grid), dim3(256), lds, stream, a);
        CX_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long long> h(n);
        CX_HIP(hipMemcpy(h.data(), a.diag, n * 8, hipMemcpyDeviceToHost));
        CX_HIP(hipFree(a.diag));
        double s[4] = {0, 0, 0, 0}, tiles = 0;
        for (size_t w = 0; w < (size_t)grid * 4; w++) {
            for (int p = 0; p < 4; p++) s[p] += (double)h[w * 5 + p];
            tiles += (double)h[w * 5 + 4];
        }
        fprintf(stderr, "[batch diag] cycles per tile per wave: stage+check %.0f  mfma-loop %.0f  epilogue %.0f  barrier %.0f\n",
                s[0] / tiles, s[1] / tiles, s[2] / tiles, s[3] / tiles);
        return CX_OK;
    }
    hipLaunchKernelGGL((batch_scan_kernel<D, false>), dim3(grid), dim3(256), lds, stream, a);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

int launch_batch_scan(BatchArgs a, uint32_t grid, hipStream_t stream) {
    if (!batch_supported(a.dim, a.k)) return set_err(CX_ERR_VALIDATION, "batch scan: unsupported dim %u / k %u", a.dim, a.k);
    if (a.nq == 0 || a.nq > BT_Q) return set_err(CX_ERR_VALIDATION, "batch scan: 1..64 queries per pass");
    a.capq = a.k <= 16 ? 64u : 80u;  // >= k + 16 appends per tile, <= 128 (two entries per lane in compact); 12 B per entry
    if (a.dim == 384) return launch_batch_d<384>(a, grid, stream);
    return launch_batch_d<768>(a, grid, stream);
}

}  // namespace cx
