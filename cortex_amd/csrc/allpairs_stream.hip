// allpairs_stream.hip — the filter pass for a SMALL scan set (<= 64 rows: streaming ingest, BASELINE config 5), built
// like the batched search instead of like a tiled GEMM.  With 64 scanned rows the pass is HBM-bound — the bf16 shadow
// of the shard is streamed once — and what matters is how the stream is read.  The tiled kernel fetches 128-byte
// pieces of 128 rows per LDS-DMA step (5.8 TB/s); here a producer wave reads 1 KiB contiguous per instruction, whole
// 16-row tiles back to back, exactly like a row scan:
//  - 512-thread block per CU: waves 0-3 are consumers — 16 scanned rows each, as MFMA B-operand fragments in registers
//    for the whole launch (dim/8 VGPRs) — waves 4-7 are producers: 16-row tiles of the shadow HBM -> registers -> LDS
//    (two register sets; store, then reload into the same register), one XOR swizzle on the LDS address;
//  - per tile a consumer runs dim/32 v_mfma_f32_16x16x32_bf16 (tile rows x its 16 scanned rows) and screens the 16x16
//    scores against thr - eps; hits go to the scanned row's candidate list (rare);
//  - one raw barrier per tile (lgkmcnt(0) + s_barrier: the prefetched loads stay in flight); a tile is one 16-row image,
//    or two for dims <= 512, where one image alone is too little work per barrier.
//  - Round 3: the shard's rows come from the TILED shadow (kernels.hpp: tiled_shadow_off) — the only copy kept when
//    dim % 32 == 0.  A 16-row block of it is dim / 32 contiguous KiB, so the producers' loads are as before; the LDS
//    image keeps this kernel's own swizzle, the producers translate (a lane's 16 bytes are piece x ^ s of row r of
//    K-step ks, s from bits 3-4 of the global row: bit 4 is the block's parity, one XOR of the LDS address by 32).
// Same contract as pair_filter_kernel (allpairs.hip): candidate columns out, no score matrix.
#include "kernels.hpp"

namespace cx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

namespace pstream {
constexpr uint32_t IMG_ROWS = 16, MAX_SCAN = 64;   // a tile is one or two 16-row images (rows_per_tile)
// tile rows per barrier: narrow rows make a 16-row tile so small (12 KiB at 384-d) that the barrier and the loop
// overhead weigh as much as its MFMAs; two images per step halve that
constexpr uint32_t rows_per_tile(uint32_t dim) { return dim <= 512 ? 32u : 16u; }
// byte offset of 16-byte piece p of tile row i: pieces XOR-swizzled with the row inside each 256-byte segment, so the 16
// lanes of a ds_read_b128 group (rows 0..15, same piece) hit 16 different bank groups
template <int D>
__device__ inline uint32_t t_off(uint32_t i, uint32_t p) { return i * (uint32_t)(D * 2) + (((p & ~15u) | ((p ^ i) & 15u)) << 4); }
}  // namespace pstream

template <int D>
__global__ __launch_bounds__(512) void pair_filter_stream_kernel(const PairFilterArgs a) {
    using namespace pstream;
    constexpr uint32_t ROWS = rows_per_tile(D), NIMG = ROWS / IMG_ROWS;
    constexpr uint32_t ROW_BYTES = D * 2, IMG_BYTES = IMG_ROWS * ROW_BYTES, TILE_BYTES = ROWS * ROW_BYTES, LOADS = TILE_BYTES / 4 / 1024, KS = D / 32;
    static_assert(D % 128 == 0, "dim must be a multiple of 128 (whole KiB per producer wave)");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // two tile buffers
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, pw = wave & 3u;
    const bool consumer = __builtin_amdgcn_readfirstlane(wave) < 4u;
    const uint32_t n_tiles = (a.n_rows + ROWS - 1) / ROWS;
    const uint32_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - 1u - blockIdx.x) / gridDim.x + 1u : 0u;
    const uint32_t my_steps = (my_tiles + 1u) & ~1u;   // both roles step in pairs (one back edge, two register sets)
    auto tile_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    if (!consumer) {
        // ------------------------------------------------------------------ producer
        f32x4 ldA[LOADS], ldB[LOADS];
        const bool tiled = a.shadow_t != nullptr;
        uint32_t dst_off[LOADS];   // swizzled LDS offset of this lane's 16 bytes of each 1-KiB piece
#pragma unroll
        for (uint32_t e = 0; e < LOADS; e++) {
            const uint32_t o = (pw * LOADS + e) * 1024u + lane * 16u;   // byte offset inside the tile as it lies in memory
            if (tiled) {   // [image][K-step][16 rows x 64 B]: slot x of row r holds piece x ^ ((global row >> 3) & 3)
                const uint32_t img = o / IMG_BYTES, oi = o % IMG_BYTES, ks = oi >> 10, r = (oi >> 6) & 15u, x = (oi >> 4) & 3u;
                const uint32_t par = NIMG == 2u ? img : 0u;   // two images per tile: tiles start at even blocks; one: the tile's parity, applied per tile
                dst_off[e] = img * IMG_BYTES + t_off<D>(r, 4u * ks + (x ^ ((par << 1) | (r >> 3))));
            } else {
                const uint32_t row = o / ROW_BYTES;                      // 0 .. ROWS-1: image row / 16, swizzled inside its image
                dst_off[e] = (row / IMG_ROWS) * IMG_BYTES + t_off<D>(row % IMG_ROWS, (o % ROW_BYTES) >> 4);
            }
        }
        const uint32_t my_src = pw * LOADS * 1024u + lane * 16u;
        const char *shadow = reinterpret_cast<const char *>(tiled ? a.shadow_t : a.shadow);
        // the last tile may be ragged: rows past n_rows are read from the last full tile position instead (clamped
        // tile index) — their scores are masked at emit
        auto src_of = [&](uint32_t t) {
            const uint32_t tile = t < n_tiles ? t : n_tiles - 1u;
            // row-major: a ragged last tile would read past the shadow: shift it up to end exactly at the last row.
            // The tiled shadow is allocated in whole 256-row tiles: the last tile is read where it lies
            const size_t row0 = (tiled || (size_t)tile * ROWS + ROWS <= a.n_rows) ? (size_t)tile * ROWS : (size_t)a.n_rows - ROWS;
            return shadow + row0 * ROW_BYTES + my_src;
        };
        auto par_of = [&](uint32_t t) {   // LDS address bit 5 of a one-image tile from an odd 16-row block
            const uint32_t tile = t < n_tiles ? t : n_tiles - 1u;
            return (tiled && NIMG == 1u) ? (tile & 1u) << 5 : 0u;
        };
        auto issue_loads = [&](f32x4 (&ld)[LOADS], uint32_t t) {
            const char *base = src_of(t);
#pragma unroll
            for (uint32_t e = 0; e < LOADS; e++) ld[e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(base + e * 1024u));
        };
        auto write_tile = [&](f32x4 (&ld)[LOADS], uint32_t buf, uint32_t held, uint32_t reload) {   // held: the tile the registers hold
            const char *rbase = src_of(reload);
            char *dst = smem + buf * TILE_BYTES;
            const uint32_t px = par_of(held);
#pragma unroll
            for (uint32_t e = 0; e < LOADS; e++) {
                *reinterpret_cast<f32x4 *>(dst + (dst_off[e] ^ px)) = ld[e];   // store, then reload into the same register
                ld[e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(rbase + e * 1024u));
            }
        };
        uint32_t tile = blockIdx.x, buf = 0;
        if (my_tiles) {
            issue_loads(ldA, tile);
            issue_loads(ldB, tile + gridDim.x);
            write_tile(ldA, 0, tile, tile + 2u * gridDim.x);
        }
        tile_barrier();
        auto step = [&](f32x4 (&ld)[LOADS]) {
            const uint32_t next = tile + gridDim.x;
            if (next < n_tiles) write_tile(ld, buf ^ 1u, next, next + 2u * gridDim.x);
            tile_barrier();
            buf ^= 1u;
            tile += gridDim.x;
        };
        for (uint32_t i = 0; i < my_steps; i += 2u) {
            step(ldB);
            step(ldA);
        }
        return;
    }

    // ---------------------------------------------------------------------- consumer
    const uint32_t j = lane & 15u, kq = lane >> 4;
    const uint32_t si = pw * 16u + j;                       // this lane's scanned row (position in the scan set)
    bf16x8 qf[KS];
    {
        const bool live = si < a.n_scan;
        const uint32_t g = live ? (a.scan_rows ? a.scan_rows[si] : si) : 0u;
        const bool q_tiled = !a.shadow_q && a.shadow_t;   // the scanned rows are rows of this shard: read them where the shard keeps them
        const uint16_t *q_row = a.shadow_q ? a.shadow_q + (size_t)g * D : (a.shadow ? a.shadow + (size_t)g * D : nullptr);
#pragma unroll
        for (uint32_t ks = 0; ks < KS; ks++) {
            qf[ks] = q_tiled ? *reinterpret_cast<const bf16x8 *>(a.shadow_t + tiled_shadow_off(g, 4u * ks + kq, KS))
                             : reinterpret_cast<const bf16x8 *>(q_row)[4u * ks + kq];
            if (!live) qf[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    uint32_t a_off[4];   // tile-row fragment offsets for ks & 3 = 0..3 (the swizzle has period 16 pieces = 4 k-steps)
#pragma unroll
    for (uint32_t ksl = 0; ksl < 4; ksl++) a_off[ksl] = t_off<D>(j, 4u * ksl + kq);
    const bool wave_live = pw * 16u < a.n_scan;

    tile_barrier();   // tile 0 is in buffer 0
    uint32_t buf = 0;
    for (uint32_t st = 0, tile = blockIdx.x; st < my_steps; st++, tile += gridDim.x) {
        if (!wave_live || tile >= n_tiles) { tile_barrier(); buf ^= 1u; continue; }
        // row-major shadow: a ragged last tile was read shifted up (src_of): its row 0 is row n_rows - ROWS
        const size_t row0 = (a.shadow_t || (size_t)tile * ROWS + ROWS <= a.n_rows) ? (size_t)tile * ROWS : (size_t)a.n_rows - ROWS;
        const size_t first_new = (size_t)tile * ROWS;   // rows below this were already covered by the previous tile
#pragma unroll
        for (uint32_t img = 0; img < NIMG; img++) {
            const char *T = smem + buf * TILE_BYTES + img * IMG_BYTES;
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
            constexpr int CH = 4;
            static_assert(KS % CH == 0, "dim/32 must be a multiple of 4");
            auto rd = [&](uint32_t ks) { return *reinterpret_cast<const bf16x8 *>(T + a_off[ks & 3u] + (ks >> 2) * 256u); };
            bf16x8 fa[CH], fb[CH];
#pragma unroll
            for (int u = 0; u < CH; u++) fa[u] = rd(u);
#pragma unroll
            for (uint32_t c = 0; c < KS / CH; c++) {
                bf16x8 *cur = (c & 1u) ? fb : fa, *nxt = (c & 1u) ? fa : fb;
                if (c + 1 < KS / CH) {
#pragma unroll
                    for (int u = 0; u < CH; u++) nxt[u] = rd((c + 1) * CH + u);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < CH; u++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[u], qf[c * CH + u], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // C[image row 4 kq + e][scanned row j]: screen, then emit the rare hits
            const float mx = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
            if (__ballot(mx >= a.thr_lo) != 0ull) {
#pragma unroll
                for (uint32_t e = 0; e < 4; e++) {
                    const size_t row = row0 + img * IMG_ROWS + 4u * kq + e;
                    if (acc[e] >= a.thr_lo && si < a.n_scan && row >= first_new && row < a.n_rows) {
                        const uint32_t slot = atomicAdd(a.cand_cnt + si, 1u);
                        if (slot < a.cap) a.cand[(size_t)si * a.cap + slot] = (uint32_t)row;
                    }
                }
            }
        }
        tile_barrier();
        buf ^= 1u;
    }
}

bool pair_filter_stream_supported(const PairFilterArgs &a) {
    return a.n_scan >= 1 && a.n_scan <= pstream::MAX_SCAN && !a.symmetric && (a.shadow_t || a.n_rows >= pstream::rows_per_tile(a.dim)) &&
           (a.dim == 384 || a.dim == 512 || a.dim == 768 || a.dim == 1024);
}

template <int D>
static int launch_stream_d(const PairFilterArgs &a, hipStream_t stream) {
    constexpr size_t lds = 2 * (size_t)pstream::rows_per_tile(D) * D * 2;
    static std::atomic<uint64_t> attr_devices{0};
    if (first_use_on_device(attr_devices))
        CX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_filter_stream_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t n_tiles = (a.n_rows + pstream::rows_per_tile(D) - 1) / pstream::rows_per_tile(D);
    const uint32_t cus = device_cus();
    const uint32_t grid = n_tiles < cus ? n_tiles : cus;
    hipLaunchKernelGGL(pair_filter_stream_kernel<D>, dim3(grid), dim3(512), lds, stream, a);
    CX_HIP(hipGetLastError());
    return CX_OK;
}

int launch_pair_filter_stream(const PairFilterArgs &a, hipStream_t stream) {
    if (!pair_filter_stream_supported(a)) return set_err(CX_ERR_VALIDATION, "stream pair filter: unsupported shape (n_scan %u, dim %u)", a.n_scan, a.dim);
    // large shards: the same pass through batchs.hip's worker / service structure (no barrier per tile, tiles claimed
    // dynamically, hits through LDS rings): the caller lends a word of pair_ctl as the tile counter
    static const int thr_ok = getenv("CX_PAIR_STREAM_BATCHS") ? atoi(getenv("CX_PAIR_STREAM_BATCHS")) : 1;
    if (thr_ok && a.shadow_t && a.pair_ctl && batchs_thr_supported(a.n_rows, a.dim, a.n_scan)) {
        BatchSArgs b;
        memset(&b, 0, sizeof b);
        b.shadow_t = a.shadow_t;
        b.n_rows = a.n_rows;
        b.nq = a.n_scan;
        b.dim = a.dim;
        b.flt.trivial = 1;
        b.thr_lo = a.thr_lo;
        b.scan_rows = a.scan_rows;
        b.shadow_q = a.shadow_q;
        b.thr_cand_cnt = a.cand_cnt;
        b.thr_cand = a.cand;
        b.thr_cap = a.cap;
        b.thr_next = a.pair_ctl + 24;
        return launch_batchs_thr(b, stream);
    }
    if (a.dim == 384) return launch_stream_d<384>(a, stream);
    if (a.dim == 512) return launch_stream_d<512>(a, stream);
    if (a.dim == 768) return launch_stream_d<768>(a, stream);
    return launch_stream_d<1024>(a, stream);
}

}  // namespace cx
