// Probe: do v_mfma_f32_32x32x2_f32 (waves 0-3 of a workgroup, one per SIMD) and plain fp32 VALU work (waves 4-7, the
// partner on each SIMD) overlap, or do they add up?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(float *out, int mfma_iters, int valu_iters, float a, float b) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        f32x16 acc[4];
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 16; ++r) acc[c][r] = (float)threadIdx.x;
        float x = a + threadIdx.x, y = b;
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[c], 0, 0, 0);
        }
        float s = 0;
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 16; ++r) s += acc[c][r];
        if (s == 1.2345f) out[0] = s;
    } else {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = a * i + threadIdx.x;
        for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], b, a);   // 128 independent-ish v_fma per iteration
        }
        float s = 0;
        for (int i = 0; i < 16; ++i) s += v[i];
        if (s == 1.2345f) out[1] = s;
    }
}
float run(int mi, int vi) {
    float *d;
    hipMalloc(&d, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, mi, vi, 1.0f, 0.999f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipFree(d);
    return ms;
}
int main() {
    const int mi = 2048;           // 2048 * 16 MFMAs * 64 cycles = 2.1M cycles = 0.87 ms
    const float m = run(mi, 0);
    printf("MFMA alone            %.3f ms\n", m);
    for (int vi : {1024, 2048, 4096, 8192}) {
        const float v = run(0, vi), both = run(mi, vi);
        printf("VALU alone (%5d x 128 fma) %.3f ms   together %.3f ms   sum %.3f  max %.3f\n", vi, v, both, m + v,
               m > v ? m : v);
    }
    return 0;
}
