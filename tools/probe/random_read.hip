// Probe: latency and rate of dependent random reads as a function of the table size, the read width and the waves in
// flight -- is the selection's ~2 us per dependent round trip the memory system (TLB reach?) or the kernel?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/random_read.hip -o /tmp/random_read && /tmp/random_read
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__device__ __forceinline__ uint32_t mix(uint32_t h) {
    h *= 0x85EBCA6Bu; h ^= h >> 15; h *= 0xC2B2AE35u; h ^= h >> 13;
    return h;
}
// every lane chases `steps` dependent reads of W dwords at random 64-byte-aligned places of the table
template <int W>
__global__ __launch_bounds__(256) void chase(const uint32_t *table, uint64_t n_lines, int steps, uint32_t *out) {
    uint32_t h = mix(blockIdx.x * 256u + threadIdx.x + 1u);
    uint32_t acc = 0;
    for (int s = 0; s < steps; ++s) {
        const uint64_t line = ((uint64_t)h * n_lines) >> 32;
        const uint32_t *p = table + line * 16;
        if (W == 1) {
            acc += p[0];
        } else if (W == 4) {
            const uint4 v = *reinterpret_cast<const uint4 *>(p);
            acc += v.x + v.y + v.z + v.w;
        } else {
            const uint4 a = reinterpret_cast<const uint4 *>(p)[0], b = reinterpret_cast<const uint4 *>(p)[1];
            const uint4 c = reinterpret_cast<const uint4 *>(p)[2], d = reinterpret_cast<const uint4 *>(p)[3];
            acc += a.x + b.y + c.z + d.w;
        }
        h = mix(h + acc);
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int W>
void run(const uint32_t *table, uint64_t bytes, int waves_per_cu, int n_cu, uint32_t *out) {
    const int steps = 64;
    const uint64_t n_lines = bytes / 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(chase<W>, dim3(n_cu * waves_per_cu / 4), dim3(256), 0, 0, table, n_lines, steps, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double reads = (double)n_cu * waves_per_cu * 64 * steps;
    printf("table %7.0f MB  %2d B/read  waves/CU %2d : %.3f ms, %.2f us per dependent read, %.1f G reads/s, %.2f TB/s of 64-B lines\n",
           bytes / 1e6, 4 * W, waves_per_cu, ms, ms * 1e3 / steps, reads / ms * 1e-6, reads * 64 / ms * 1e-9);
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount;
    const uint64_t max_bytes = 8ull << 30;
    uint32_t *table, *out;
    hipMalloc(&table, max_bytes);
    hipMalloc(&out, 64);
    hipMemset(table, 1, max_bytes);
    hipDeviceSynchronize();
    for (uint64_t mb : {16ull, 128ull, 512ull, 1200ull, 4096ull, 8192ull})
        for (int w : {4, 8, 16, 32}) run<1>(table, mb << 20, w, n_cu, out);
    for (uint64_t mb : {128ull, 1200ull, 8192ull})
        for (int w : {8, 16}) { run<4>(table, mb << 20, w, n_cu, out); run<16>(table, mb << 20, w, n_cu, out); }
    return 0;
}
