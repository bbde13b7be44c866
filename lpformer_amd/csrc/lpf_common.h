// Shared helpers for the gfx950 kernels (wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lpformer_hip.h"

#define LPF_WAVE 64

void lpf_set_hip_error(hipError_t e);

// Launch facts are cached PER DEVICE: one process may drive several GPUs, from several host threads.  The guarded
// actions (hipFuncSetAttribute, an occupancy query) are idempotent, so relaxed atomics are enough -- two threads doing
// one of them twice is harmless, launching without it is not (function-local `static bool` caches).
#include <atomic>
constexpr int LPF_MAX_DEVICES = 64;
struct LpfPerDevice {
    std::atomic<int> v[LPF_MAX_DEVICES];
};
int lpf_current_device();                                             // -1: none (or an ordinal beyond the cache)
int lpf_cu_count();                                                   // CUs of the current device; 0: no device
int lpf_set_max_lds(LpfPerDevice &once, const void *kern, int bytes);  // hipFuncAttributeMaxDynamicSharedMemorySize
// resident workgroups per CU of `kern` (occupancy query, cached); `fallback` when the query fails
int lpf_blocks_per_cu(LpfPerDevice &cache, const void *kern, int threads, size_t lds, int fallback);

#define LPF_SET_MAX_LDS(kern, lds)                                                                       \
    do {                                                                                                 \
        static LpfPerDevice once__;                                                                      \
        if ((lds) > 64 * 1024) {                                                                         \
            const int rc__ = lpf_set_max_lds(once__, reinterpret_cast<const void *>(kern), (int)(lds));  \
            if (rc__ != LPF_OK) return rc__;                                                             \
        }                                                                                                \
    } while (0)

#define LPF_CHECK_LAUNCH()                      \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) {                \
            lpf_set_hip_error(e__);             \
            return LPF_ERR_LAUNCH;              \
        }                                       \
    } while (0)

#define LPF_REQUIRE(cond)                 \
    do {                                  \
        if (!(cond)) return LPF_ERR_INVALID; \
    } while (0)

static inline bool lpf_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ int lpf_lane() { return (int)(threadIdx.x & 63); }

// xor-butterfly reductions over the low `WIDTH` lanes of each aligned group (WIDTH power of two <= 64)
template <int WIDTH>
__device__ __forceinline__ float lpf_group_sum(float v) {
#pragma unroll
    for (int m = WIDTH >> 1; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
template <int WIDTH>
__device__ __forceinline__ float lpf_group_max(float v) {
#pragma unroll
    for (int m = WIDTH >> 1; m > 0; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}

// first index i in [lo, hi) with a[i] >= key (a sorted ascending); returns hi if none
__device__ __forceinline__ int64_t lpf_lower_bound(const int32_t *__restrict__ a, int64_t lo, int64_t hi, int32_t key) {
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// fp32 <-> bf16 bits (round to nearest even; NaN stays NaN)
__device__ __forceinline__ uint16_t lpf_f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float lpf_bf16_to_f32(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }

// Dropout inside a kernel (the training forward of the fused GCN layer and its LayerNorm/ReLU backward): element (row,
// col) of a launch with 64-bit seed s is KEPT iff lpf_drop_bits(row key, col, s) >= threshold, threshold =
// lpf_drop_threshold(p) -- a counter-based hash (two lowbias32 rounds), so the backward recomputes the forward's mask
// from (seed, row, col) and no mask tensor exists.  Reference: F.dropout(x, p, training=True) behind every GCN layer
// (src/models/other_models.py:69): Bernoulli(1 - p) keeps scaled by 1 / (1 - p); which elements are kept is random
// there as here (parity is distributional: tests/test_gpu_train.py), p is met to 2^-32.
__host__ __device__ __forceinline__ uint32_t lpf_mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
__host__ __device__ __forceinline__ uint32_t lpf_drop_row_key(int64_t row, uint64_t seed) {
    return lpf_mix32((uint32_t)row * 0x9E3779B1u + (uint32_t)seed) ^ (uint32_t)((uint64_t)row >> 32);
}
__host__ __device__ __forceinline__ uint32_t lpf_drop_bits(uint32_t row_key, int col, uint64_t seed) {
    return lpf_mix32(row_key + (uint32_t)col * 0x85EBCA77u + (uint32_t)(seed >> 32));
}
static inline uint32_t lpf_drop_threshold(float p) {   // keep iff bits >= threshold: P(drop) = threshold / 2^32
    const double t = (double)p * 4294967296.0;
    return t <= 0.0 ? 0u : (t >= 4294967295.0 ? 4294967295u : (uint32_t)t);
}
