// shape_probe.hip — research tool (not part of the library): what read bandwidth does MI355X give a kernel as a function of
// the bytes a CU keeps in flight and of the shape of a wave's load instructions?   hipcc --offload-arch=gfx950 -O3 -o _shape_probe shape_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// linear: a wave reads contiguous chunks of U KiB (U load instructions of 1 KiB in flight), chunks round-robin over waves
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

// sliced: a wave owns 16 rows of `pitch` bytes and walks them SEG bytes at a time (64 * 16 / SEG rows per instruction);
// one K-block = 16 rows x SEG... generalised: per K-block the wave reads 16 rows x KBYTES, as 16*KBYTES/1024 instructions,
// each covering (1024 / SEG) rows x SEG bytes; two K-blocks in flight (2 register sets)
template <int KBYTES, int SEG>
__global__ __launch_bounds__(512) void sliced_kernel(const char *src, uint32_t n_rows, uint32_t pitch, float *sink) {
    extern __shared__ char pad[];
    constexpr int NI = 16 * KBYTES / 1024;       // instructions per K-block
    constexpr int RPI = 1024 / SEG;              // rows per instruction
    constexpr int IPG = KBYTES / SEG;            // instructions to cover KBYTES of one row group
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n_tiles = n_rows / 128u, n_kb = pitch / KBYTES;
    const uint32_t lrow = lane / (SEG / 16), lpiece = lane % (SEG / 16);
    f32x4 acc = {0, 0, 0, 0};
    f32x4 xa[NI], xb[NI];
    auto fetch = [&](f32x4 (&dst)[NI], uint32_t tile, uint32_t kb) {
        const char *base = src + ((size_t)tile * 128u + wave * 16u) * pitch + (size_t)kb * KBYTES;
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const uint32_t rg = i / IPG, sg = i % IPG;   // row group, segment inside the K-block
            dst[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(base + (size_t)(rg * RPI + lrow) * pitch + sg * SEG + lpiece * 16u));
        }
    };
    auto eat = [&](const f32x4 (&s)[NI]) {
#pragma unroll
        for (int i = 0; i < NI; i++) acc += s[i];
    };
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        fetch(xa, tile, 0);
        for (uint32_t kb = 0; kb < n_kb; kb += 2) {
            fetch(xb, tile, kb + 1);
            eat(xa);
            if (kb + 2 < n_kb) fetch(xa, tile, kb + 2);
            eat(xb);
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123456.789f) *sink = acc.x;
}

__global__ void fill_kernel(uint32_t *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = 0x3c000000u | (x & 0x03ffffffu);   // floats of magnitude ~0.01 .. 0.03 with random mantissas
    }
}

// the sliced shape through LDS-DMA: wave w of 8 owns the w-th 512-byte slice of 16 whole rows (batchk's tile), 2 slots
__global__ __launch_bounds__(512) void dma_kernel(const char *src, uint32_t n_rows, uint32_t pitch, float *sink, int aux_nt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t n_tiles = n_rows / 16u;
    char *Rw = smem + wave * 16384u;
    const uint32_t lane_off = (lane >> 5) * pitch + (lane & 31u) * 16u;
    f32x4 acc = {0, 0, 0, 0};
    auto issue = [&](uint32_t slot, uint32_t tile) {
        const char *base = src + (size_t)tile * 16u * pitch + wave * 512u;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (aux_nt) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (size_t)i * 2u * pitch + lane_off), (__attribute__((address_space(3))) void *)(Rw + slot * 8192u + i * 1024), 16, 0, 2);
            else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (size_t)i * 2u * pitch + lane_off), (__attribute__((address_space(3))) void *)(Rw + slot * 8192u + i * 1024), 16, 0, 0);
        }
    };
    uint32_t tile = blockIdx.x;
    if (tile < n_tiles) issue(0, tile);
    if (tile + gridDim.x < n_tiles) issue(1, tile + gridDim.x);
    for (uint32_t slot = 0; tile < n_tiles; tile += gridDim.x, slot ^= 1u) {
        if (tile + gridDim.x < n_tiles) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += *reinterpret_cast<const f32x4 *>(Rw + slot * 8192u + lane * 16u);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (tile + 2u * gridDim.x < n_tiles) issue(slot, tile + 2u * gridDim.x);
    }
    if (acc.x + acc.y + acc.z + acc.w == 123456.789f) *sink = acc.x;
}

static double g_avg;
template <typename F>
static double timed(F launch, double bytes, int reps = 6) {
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
    const int random_data = argc > 1 ? atoi(argv[1]) : 0;
    const int reps = argc > 2 ? atoi(argv[2]) : 6;
    const size_t bytes = (size_t)1 << 32;   // 1M rows x 4 KiB
    char *buf; float *sink;
    CK(hipMalloc((void **)&buf, bytes + (1 << 20))); CK(hipMalloc((void **)&sink, 4));
    CK(hipMemset(buf, 0x11, bytes));
    if (random_data) { hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t *)buf, bytes / 4); CK(hipDeviceSynchronize()); }
    printf("data: %s, reps %d\n", random_data ? "random" : "0x11", reps);
    int cus = 256;
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); cus = pr.multiProcessorCount;
    printf("CUs %d\n", cus);
#define LIN(U) \
    for (int bpc : {2, 4, 8}) { \
        const int lds = bpc <= 8 ? (160 * 1024 / bpc) & ~255 : 0; \
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(linear_kernel<U>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        double g = timed([&] { hipLaunchKernelGGL(linear_kernel<U>, dim3(cus * bpc), dim3(256), lds, 0, (const f32x4 *)buf, bytes / 16, sink); }, (double)bytes, reps); \
        printf("linear U=%2d KiB/wave  %d waves/CU  in flight %4d KiB/CU : best %.3f avg %.3f\n", U, 4 * bpc, 4 * bpc * U, g / 8.0, g_avg / 8.0); \
    }
    LIN(8) LIN(16)
#define SL(KB, SEG) { \
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(sliced_kernel<KB, SEG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        for (int bpc : {1, 2}) { \
        double g = timed([&] { hipLaunchKernelGGL((sliced_kernel<KB, SEG>), dim3(cus * bpc), dim3(512), (160 * 1024 / bpc) & ~255, 0, buf, 1u << 20, 4096u, sink); }, (double)bytes, reps); \
        printf("sliced K-block %4d B, segment %4d B, %d waves/CU, in flight %d KiB/CU: best %.3f avg %.3f\n", KB, SEG, 8 * bpc, 8 * bpc * 2 * 16 * KB / 1024, g / 8.0, g_avg / 8.0); } }
    SL(256, 256) SL(512, 256) SL(512, 512) SL(1024, 512) SL(1024, 1024)
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int nt : {0, 1}) {
        double g = timed([&] { hipLaunchKernelGGL(dma_kernel, dim3(cus), dim3(512), 128 * 1024, 0, buf, 1u << 20, 4096u, sink, nt); }, (double)bytes, reps);
        printf("LDS-DMA split-K tile (16 rows x 4 KiB per block, 2 slots), aux %d: best %.3f avg %.3f\n", nt * 2, g / 8.0, g_avg / 8.0);
    }
    return 0;
}
