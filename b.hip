#include <hip/hip_runtime.h>
__global__ void k(unsigned *p, unsigned v) {
    asm volatile("s_atomic_add %0, %1, 0x0 glc" : "+s"(v) : "s"(p) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v) :: "memory");
    if (threadIdx.x == 0) p[1] = v;
}
