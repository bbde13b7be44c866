// Probe: lane mapping of v_permlane16_swap_b32 / v_permlane32_swap_b32 (gfx950) with both operands = the lane id.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
    unsigned v = threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
    auto r2 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    o[128 + threadIdx.x] = r2[0]; o[192 + threadIdx.x] = r2[1];
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[4] = {"p16 r0", "p16 r1", "p32 r0", "p32 r1"};
    for (int a = 0; a < 4; ++a) { printf("%s:", names[a]); for (int i = 0; i < 64; i += 4) printf(" %u", h[a * 64 + i]); printf("\n"); }
    return 0;
}
