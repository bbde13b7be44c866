// Calibration micro-benchmarks (not part of the product): workgroup launch rate and dependent-load chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_empty(int *out) { if (threadIdx.x == 0 && blockIdx.x == 0xFFFFFFF) out[0] = 1; }

// each wave: `depth` dependent loads through a random permutation (pointer chase), then one store
__global__ void k_chase(const int *__restrict__ next, int *__restrict__ out, int depth, int n) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int idx = (int)(((long long)wave * 2654435761ll) % n);
    for (int d = 0; d < depth; ++d) idx = next[idx];
    if ((threadIdx.x & 63) == 0) out[wave] = idx;
}

template <typename F> float timeit(F f, int reps = 20) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms * 1000.f / reps;
}

int main() {
    int *out; CK(hipMalloc(&out, 1 << 24));
    const int n = 64 << 20;  // 256 MB table of ints
    std::vector<int> h(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int)(s % n); }
    int *next; CK(hipMalloc(&next, (size_t)n * 4)); CK(hipMemcpy(next, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    for (int threads : {64, 256}) {
        for (int blocks : {4096, 16384, 65536}) {
            float us = timeit([&] { hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(threads), 0, 0, out); });
            printf("empty  blocks=%6d threads=%3d : %8.1f us  (%.1f WG/us)\n", blocks, threads, us, blocks / us);
        }
    }
    for (int threads : {64, 256}) {
        for (int depth : {1, 4, 8}) {
            const int waves = 32768, blocks = waves * 64 / threads;
            float us = timeit([&] { hipLaunchKernelGGL(k_chase, dim3(blocks), dim3(threads), 0, 0, next, out, depth, n); });
            printf("chase  waves=%d threads=%3d depth=%d : %8.1f us\n", waves, threads, depth, us);
        }
    }
    // small table (L2 resident) for comparison
    for (int depth : {1, 4, 8}) {
        const int waves = 32768, blocks = waves / 4;
        float us = timeit([&] { hipLaunchKernelGGL(k_chase, dim3(blocks), dim3(256), 0, 0, next, out, depth, 1 << 18); });
        printf("chase-L2 waves=%d depth=%d : %8.1f us\n", waves, depth, us);
    }
    return 0;
}
