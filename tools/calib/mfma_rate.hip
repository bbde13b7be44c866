// Calibration: raw issue rate of v_mfma_f32_32x32x2_f32 (4 independent accumulators per wave).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float x = a + threadIdx.x * 1e-9f, y = b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[c], 0, 0, 0);
    }
    float s = 0; for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float *out; hipMalloc(&out, 1 << 26);
    for (int wpb : {1, 2}) {
        int blocks = 256 * wpb, iters = 2000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 10, 1.f, 2.f); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 * iters * 16 * (2.0 * 32 * 32 * 2);
        printf("blocks=%d (%d waves/SIMD): %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4GHz)\n", blocks, wpb, ms, flops / ms / 1e9,
               ms * 1e-3 * 2.4e9 / (iters * 16.0 * wpb));
    }
    return 0;
}
