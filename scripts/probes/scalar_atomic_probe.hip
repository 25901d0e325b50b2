// scalar_atomic_probe.hip — research tool: does gfx950 execute s_atomic_add (a returning atomic that is counted in lgkmcnt,
// not in the in-order vmcnt queue)?  Every block draws tickets from one counter; the tickets must be a permutation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k(unsigned *ctr, unsigned *out, int per_block) {
    for (int i = 0; i < per_block; i++) {
        unsigned v = 1;
        asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
        if (threadIdx.x == 0) out[blockIdx.x * per_block + i] = v;
    }
}
int main() {
    const int blocks = 2048, per = 8;
    unsigned *ctr, *out;
    CK(hipMalloc((void **)&ctr, 4)); CK(hipMalloc((void **)&out, blocks * per * 4));
    CK(hipMemset(ctr, 0, 4));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, ctr, out, per);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(e)); return 1; }
    std::vector<unsigned> h(blocks * per);
    CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    unsigned c; CK(hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    bool ok = c == (unsigned)(blocks * per);
    for (size_t i = 0; i < h.size(); i++) ok = ok && h[i] == i;
    printf("s_atomic_add on this device: counter %u (expected %d), tickets %s\n", c, blocks * per, ok ? "are a permutation: works" : "WRONG");
    return ok ? 0 : 2;
}
