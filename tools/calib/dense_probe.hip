// Ablation probe for the dense-chain inner loop (single layer, N = 256, K = 256, wave pair per 16 samples, G = 2):
// which of {global weight staging, LDS write + barrier, LDS operand reads, input loads} costs the MFMA rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TPW = 8, NTP = 16, G = 2, THREADS = 512, P = NTP * G * 64 / THREADS, SLAB = P * THREADS;

template <int ABL>
__global__ __launch_bounds__(THREADS, 4) void probe(const float *X, const float *Wp, float *out, int M, int K) {
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = wave >> 1, half = wave & 1;
    const int q = lane >> 4, j = lane & 15;
    const int64_t m = (int64_t)blockIdx.x * 64 + grp * 16 + j;
    const float *xa = X + m * K;
    const int ng = K / 16;
    f32x4 acc[TPW];
    for (int c = 0; c < TPW; ++c) acc[c] = (f32x4){0, 0, 0, 0};
    f32x4 wr[P], xr[G];
    const f32x4 *src = reinterpret_cast<const f32x4 *>(Wp);
    for (int e = 0; e < P; ++e) wr[e] = src[e * THREADS + tid];
    for (int s = 0; s < G; ++s) xr[s] = *reinterpret_cast<const f32x4 *>(xa + 16 * s + 4 * q);
    int buf = 0;
#pragma unroll 1
    for (int g0 = 0; g0 < ng; g0 += G) {
        f32x4 bv[G];
        for (int s = 0; s < G; ++s) bv[s] = xr[s];
        f32x4 *lw = lds + buf * SLAB;
        if (!(ABL & 2) || g0 == 0) {
#pragma unroll
            for (int e = 0; e < P; ++e) lw[e * THREADS + tid] = wr[e];
            __syncthreads();
        }
        if (g0 + G < ng) {
            if (!(ABL & 1)) {
#pragma unroll
                for (int e = 0; e < P; ++e) wr[e] = src[(int64_t)(g0 / G + 1) * SLAB + e * THREADS + tid];
            }
            if (!(ABL & 8)) {
#pragma unroll
                for (int s = 0; s < G; ++s) xr[s] = *reinterpret_cast<const f32x4 *>(xa + 16 * (g0 + G + s) + 4 * q);
            }
        }
        const f32x4 *p = ((ABL & 2) ? lds : lw) + (half * TPW) * 64 + lane;
        constexpr int T = G * TPW;
        auto at = [&](int t) -> f32x4 {
            if (ABL & 4) return bv[0] + (float)t;
            return p[((t / TPW) * NTP + (t % TPW)) * 64];
        };
        f32x4 n0 = at(0), n1 = at(1);
#pragma unroll
        for (int t = 0; t < T; t += 2) {
            const f32x4 a0 = n0, a1 = n1;
            if (t + 2 < T) n0 = at(t + 2);
            if (t + 3 < T) n1 = at(t + 3);
            const int s0 = t / TPW, c0 = t % TPW, s1 = (t + 1) / TPW, c1 = (t + 1) % TPW;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[c0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], bv[s0][u], acc[c0], 0, 0, 0);
                acc[c1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], bv[s1][u], acc[c1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(ABL & 2)) buf ^= 1;
    }
    for (int c = 0; c < TPW; ++c)
        *reinterpret_cast<f32x4 *>(out + m * 256 + 16 * (half * TPW + c) + 4 * q) = acc[c];
}

template <int ABL>
void run(const char *name, const float *X, const float *W, float *out, int M, int K) {
    size_t lds = 2 * SLAB * sizeof(f32x4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<ABL>, dim3(M / 64), dim3(THREADS), lds, 0, X, W, out, M, K);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe<ABL>, dim3(M / 64), dim3(THREADS), lds, 0, X, W, out, M, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%-44s M=%7d  %8.1f us  %6.1f TFLOP/s\n", name, M, ms * 1e3, 2.0 * M * K * 256 / ms / 1e9);
}

int main() {
    const int K = 256;
    for (int M : {32768, 262144}) {
        float *X, *W, *out;
        hipMalloc(&X, (size_t)M * K * 4); hipMalloc(&W, (size_t)(K / 16 / G) * SLAB * 16); hipMalloc(&out, (size_t)M * 256 * 4);
        hipMemset(X, 0, (size_t)M * K * 4); hipMemset(W, 0, (size_t)(K / 16 / G) * SLAB * 16);
        run<0>("full", X, W, out, M, K);
        run<1>("no global weight loads", X, W, out, M, K);
        run<8>("no input loads", X, W, out, M, K);
        run<9>("no global loads at all", X, W, out, M, K);
        run<3>("no staging (no weight loads, LDS write, barrier)", X, W, out, M, K);
        run<4>("no LDS operand reads", X, W, out, M, K);
        run<15>("MFMA skeleton", X, W, out, M, K);
        hipFree(X); hipFree(W); hipFree(out);
    }
    return 0;
}
