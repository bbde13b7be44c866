// Probe: which SIMD does wave w of a workgroup run on?  (HW_ID register: wave [3:0], simd [5:4], cu [11:8], se [15:13])
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned *out) {
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = hw;
    __builtin_amdgcn_s_sleep(100);
}
int main() {
    for (int waves : {8, 12, 16}) {
        unsigned *d, h[64];
        hipMalloc(&d, sizeof(h));
        hipLaunchKernelGGL(probe, dim3(3), dim3(64 * waves), 0, 0, d);
        hipMemcpy(h, d, sizeof(unsigned) * 3 * waves, hipMemcpyDeviceToHost);
        for (int b = 0; b < 3; ++b) {
            printf("waves/wg %d block %d simd of wave:", waves, b);
            for (int w = 0; w < waves; ++w) printf(" %u", (h[b * waves + w] >> 4) & 3);
            printf("   cu %u\n", (h[b * waves] >> 8) & 15);
        }
        hipFree(d);
    }
    return 0;
}
