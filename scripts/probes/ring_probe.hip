// ring_probe.hip — research tool (not part of the library): what read bandwidth does batchq.hip's load structure reach
// on its own?  A block of 8 waves (1 per CU: 117 KiB of LDS requested), every wave owns tiles of TILE bytes and keeps a ring
// of P steps in flight; a step is NSTR streams x STEP bytes (the tile is NSTR equal contiguous streams).  Loads are
// consumed by one VALU add each.  order 0: tile t of wave w = w + t * n_waves (interleaved); 1: every wave walks its own
// contiguous range.   hipcc --offload-arch=gfx950 -O3 -o _ring_probe ring_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int P, int NSTR, int STEP, int TILE, int BUF = 0, int WORK = 0>
__global__ __launch_bounds__(512, 2) void ring_kernel(const char *src, uint32_t n_tiles, float *sink, int order, int rot) {
    extern __shared__ char pad[];
    constexpr int LPS = STEP / 1024;               // load instructions per stream and step
    constexpr int KSN = TILE / (NSTR * STEP);      // steps per tile
    static_assert(KSN % P == 0, "steps per tile must be a multiple of the ring depth");
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t gw = blockIdx.x * 8u + wave, nw = gridDim.x * 8u;
    const uint32_t per = (n_tiles + nw - 1) / nw;
    f32x4 ring[P][NSTR][LPS];
    f32x4 acc = {0, 0, 0, 0};
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    f32x4 macc[4][2];
    s16x8 Bq[4][2];
#pragma unroll
    for (int g = 0; g < 4; g++) { macc[g][0] = acc; macc[g][1] = acc; Bq[g][0] = *reinterpret_cast<const s16x8 *>(pad + lane * 16u + g * 2048); Bq[g][1] = *reinterpret_cast<const s16x8 *>(pad + lane * 16u + g * 2048 + 1024); }
    auto tile_of = [&](uint32_t i) -> uint32_t { return order ? gw * per + i : gw + i * nw; };
    auto base_of = [&](uint32_t t) -> const char * {
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t < n_tiles ? t : n_tiles - 1u));
        return src + (size_t)t * TILE + lane * 16u;
    };
    const uint32_t r0 = rot == 1 ? (gw * 5u) % (uint32_t)KSN : 0u;   // rot: wave-specific starting step (breaks lockstep address patterns)
    auto issue = [&](f32x4 (&slot)[NSTR][LPS], const char *b, int ks) {
        uint32_t kk = (uint32_t)ks + r0;
        kk = kk >= (uint32_t)KSN ? kk - KSN : kk;
        if constexpr (BUF) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(b) - lane * 16u, 0, TILE, 0x00020000);
            // rot == 2: lane (i = lane & 15, kq = lane >> 4) reads piece kq ^ swizzle of row i: 16 rows x 64 B per KiB (the tiled shadow)
            const uint32_t i = lane & 15u, kq = lane >> 4;
            const uint32_t vo[2] = {rot == 2 ? i * 64u + (((kq ^ ((i >> 3) & 1u)) & 3u) << 4) : lane * 16u,
                                    rot == 2 ? i * 64u + (((kq ^ (((i >> 3) & 1u) | 2u)) & 3u) << 4) : lane * 16u};
#pragma unroll
            for (int s = 0; s < NSTR; s++)
#pragma unroll
                for (int l = 0; l < LPS; l++)
                    slot[s][l] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo[s & 1] + l * 1024, (int)(s * (TILE / NSTR) + kk * STEP), 2));
        } else {
#pragma unroll
        for (int s = 0; s < NSTR; s++)
#pragma unroll
            for (int l = 0; l < LPS; l++)
                slot[s][l] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(b + (size_t)s * (TILE / NSTR) + (size_t)kk * STEP + l * 1024));
        }
    };
    uint32_t i = 0;
    const uint32_t my = order ? (gw * per < n_tiles ? (n_tiles - gw * per < per ? n_tiles - gw * per : per) : 0u) : (gw < n_tiles ? (n_tiles - 1u - gw) / nw + 1u : 0u);
    if (!my) return;
    const char *cb = base_of(tile_of(0));
#pragma unroll
    for (int p = 0; p < P; p++) issue(ring[p], cb, p);
    for (; i < my; i++) {
        const char *nb = base_of(i + 1 < my ? tile_of(i + 1) : tile_of(i));
#pragma unroll
        for (int ks = 0; ks < KSN; ks++) {
            const int p = ks % P;
            if constexpr (WORK == 0) {
#pragma unroll
            for (int s = 0; s < NSTR; s++)
#pragma unroll
                for (int l = 0; l < LPS; l++) acc += ring[p][s][l];
            } else {
                // batchq's K-step: 24 MFMAs on the four loaded fragments (NSTR = 2, LPS = 2), 8 query fragments
                const s16x8 h0 = __builtin_bit_cast(s16x8, ring[p][0][0]), l0 = __builtin_bit_cast(s16x8, ring[p][0][1]);
                const s16x8 h1 = __builtin_bit_cast(s16x8, ring[p][NSTR - 1][0]), l1 = __builtin_bit_cast(s16x8, ring[p][NSTR - 1][LPS - 1]);
#pragma unroll
                for (int gp = 0; gp < 4; gp += 2) {
#pragma unroll
                    for (int g = gp; g < gp + 2; g++) {
                        macc[g][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0, Bq[g][0], macc[g][0], 0, 0, 0);
                        macc[g][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l1, Bq[g][0], macc[g][1], 0, 0, 0);
                        macc[g][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, Bq[g][1], macc[g][0], 0, 0, 0);
                        macc[g][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, Bq[g][1], macc[g][1], 0, 0, 0);
                        macc[g][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, Bq[g][0], macc[g][0], 0, 0, 0);
                        macc[g][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, Bq[g][0], macc[g][1], 0, 0, 0);
                    }
                    if constexpr (WORK == 2) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int g = gp; g < gp + 2; g++) {
                            Bq[g][0] = *reinterpret_cast<const s16x8 *>(pad + lane * 16u + (((ks + 1) % 12) * 4 + g) * 2048);
                            Bq[g][1] = *reinterpret_cast<const s16x8 *>(pad + lane * 16u + (((ks + 1) % 12) * 4 + g) * 2048 + 1024);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (ks + P < KSN) issue(ring[p], cb, ks + P);
            else issue(ring[p], nb, ks + P - KSN);
        }
        cb = nb;
    }
#pragma unroll
    for (int g = 0; g < 4; g++) acc += macc[g][0] + macc[g][1];
    if (acc.x + acc.y + acc.z + acc.w == 123456.789f) *sink = acc.x;
}

// reference: the linear kernel of shape_probe.hip (U KiB per wave per step, nothing in flight across steps)
template <int U>
__global__ __launch_bounds__(256) void linear_kernel(const f32x4 *src, size_t n_vec, float *sink) {
    extern __shared__ char pad[];
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const size_t n_chunks = n_vec / (64 * U);
    f32x4 acc = {0, 0, 0, 0};
    for (size_t ch = wave; ch < n_chunks; ch += n_waves) {
        const f32x4 *p = src + ch * (64 * U) + lane;
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(p + u * 64);
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123456.789f) *sink = acc.x;
}

__global__ void fill_kernel(uint32_t *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = 0x3c000000u | (x & 0x03ffffffu);
    }
}

static double g_avg;
template <typename F>
static double timed(F launch, double bytes, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 0, sum = 0;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double g = bytes / (ms * 1e-3) / 1e12;
        if (r && g > best) best = g;
        if (r) sum += g;
    }
    g_avg = sum / (reps - 1);
    return best;
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 12;
    const double gb = argc > 2 ? atof(argv[2]) : 1.92;   // bytes per launch, GB (1.92 = 1.25M x 384 x 4)
    const size_t bytes = (size_t)(gb * 1e9) / (48 * 1024) * (48 * 1024);
    char *buf; float *sink;
    CK(hipMalloc((void **)&buf, bytes + (1 << 20))); CK(hipMalloc((void **)&sink, 4));
    CK(hipMemset(buf, 0x11, bytes));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    printf("CUs %d, %.2f GB per launch, %d reps\n", cus, bytes / 1e9, reps);
#define RINGB(P, NSTR, STEP, TILE, BUF) \
    for (int order : {0}) for (int rot : {0, 2}) { \
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(ring_kernel<P, NSTR, STEP, TILE, BUF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        const uint32_t n_tiles = (uint32_t)(bytes / (TILE)); \
        double g = timed([&] { hipLaunchKernelGGL((ring_kernel<P, NSTR, STEP, TILE, BUF>), dim3(cus), dim3(512), 117 * 1024, 0, buf, n_tiles, sink, order, rot); }, (double)n_tiles * (TILE), reps); \
        printf("ring P=%d streams=%d step=%d B tile=%d KiB buffer-loads=%d rot=%d in flight %3d KiB/CU: best %.3f avg %.3f\n", P, NSTR, STEP, (TILE) / 1024, BUF, rot, 8 * P * NSTR * STEP / 1024, g / 8.0, g_avg / 8.0); \
    }
#define RINGW(P, NSTR, STEP, TILE, WORK) \
    for (int rot : {0}) { \
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(ring_kernel<P, NSTR, STEP, TILE, 1, WORK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        const uint32_t n_tiles = (uint32_t)(bytes / (TILE)); \
        double g = timed([&] { hipLaunchKernelGGL((ring_kernel<P, NSTR, STEP, TILE, 1, WORK>), dim3(cus), dim3(512), 117 * 1024, 0, buf, n_tiles, sink, 0, rot); }, (double)n_tiles * (TILE), reps); \
        printf("ring P=%d streams=%d step=%d B tile=%d KiB buffer loads, work=%d (1: 24 MFMAs per step, 2: + 8 ds_read_b128): best %.3f avg %.3f\n", P, NSTR, STEP, (TILE) / 1024, WORK, g / 8.0, g_avg / 8.0); \
    }
#define RING(P, NSTR, STEP, TILE) RINGB(P, NSTR, STEP, TILE, 1) RINGW(P, NSTR, STEP, TILE, 1) RINGW(P, NSTR, STEP, TILE, 2)
    if (argc > 3) {   // random data (floats of magnitude ~0.01 .. 0.03)
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t *)buf, bytes / 4); CK(hipDeviceSynchronize());
        printf("data: random\n");
    }
    RINGB(8, 2, 1024, 49152, 1)
    RINGB(4, 2, 2048, 49152, 1)
#define LIN(U) \
    for (int bpc : {2, 4}) { \
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(linear_kernel<U>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        double g = timed([&] { hipLaunchKernelGGL(linear_kernel<U>, dim3(cus * bpc), dim3(256), (160 * 1024 / bpc) & ~255, 0, (const f32x4 *)buf, bytes / 16, sink); }, (double)bytes, reps); \
        printf("linear U=%2d KiB/wave  %d waves/CU  in flight %4d KiB/CU : best %.3f avg %.3f\n", U, 4 * bpc, 4 * bpc * U, g / 8.0, g_avg / 8.0); \
    }
    LIN(8) LIN(16)
    return 0;
}
