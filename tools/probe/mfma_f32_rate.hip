// Probe: sustained rate of v_mfma_f32_32x32x2_f32 issued by W waves per SIMD, A accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(512) void k(float *out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int c = 0; c < NACC; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = (float)threadIdx.x;
    float x = a + threadIdx.x, y = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < NACC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < NACC; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 1.2345f) out[0] = s;
}
template <int NACC>
void run(int waves_per_cu, int n_cu) {
    float *d;
    hipMalloc(&d, 4);
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NACC>, dim3(n_cu), dim3(64 * waves_per_cu), 0, 0, d, iters, 1.0f, 2.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)n_cu * waves_per_cu * iters * 4 * NACC * 4096.0;
    printf("acc %d  waves/CU %d  %.3f ms  %.1f TF/s  cycles/MFMA/SIMD at 2.4GHz: %.1f\n", NACC, waves_per_cu, ms,
           flops / ms * 1e-9, ms * 1e-3 * 2.4e9 / ((double)iters * 4 * NACC * waves_per_cu / 4.0));
    hipFree(d);
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("CUs %d clock %d kHz\n", p.multiProcessorCount, p.clockRate);
    run<4>(4, p.multiProcessorCount);
    run<4>(8, p.multiProcessorCount);
    run<2>(4, p.multiProcessorCount);
    run<1>(4, p.multiProcessorCount);
    run<1>(8, p.multiProcessorCount);
    run<4>(4, 64);
    return 0;
}
