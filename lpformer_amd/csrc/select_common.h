// Shared by the selection kernels (select2.hip: general path over raw PPR rows; select3.hip: walk-plan path over the
// per-model indexes): the chained scan (decoupled look-back) that turns per-item totals into output positions, and
// the reference's fp32 round trip.
#pragma once
#include "lpf_common.h"

// ------------------------------------------------------------------------------------------- chained scan helpers
// One 8-byte word per participant: [63:42] launch epoch, [41:40] state (1 = own total, 2 = inclusive prefix),
// [39:0] value.  Words are written and polled as single agent-scope relaxed 8-byte accesses (value and state travel
// together, nothing else is handed over), and a word of an older launch simply reads as "not ready": the arrays are
// never cleared.
constexpr uint64_t LB_VAL_MASK = (1ull << 40) - 1ull;

__device__ __forceinline__ void lb_store(uint64_t *p, uint32_t epoch, uint32_t state, uint64_t value) {
    __hip_atomic_store(p, ((uint64_t)epoch << 42) | ((uint64_t)state << 40) | (value & LB_VAL_MASK), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint64_t lb_wait(const uint64_t *p, uint32_t epoch) {
    while (true) {
        const uint64_t w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(w >> 42) == epoch && ((w >> 40) & 3ull) != 0ull) return w;
        __builtin_amdgcn_s_sleep(1);
    }
}

__device__ __forceinline__ uint64_t lb_wave_sum(uint64_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor((unsigned long long)v, d, 64);
    return v;
}

// Exclusive prefix of `own` over participants 0..k-1 (k = this participant), computed by ONE WHOLE WAVEFRONT (every
// lane calls it with the same k / own and gets the result): publishes own total, looks back 64 predecessors at a
// time -- a participant that already carries an inclusive prefix ends the walk --, then publishes its own prefix.
// (A 256-word window, four words per lane, measured slower: 326 vs 250 us -- the polling traffic grows with it.)
__device__ __forceinline__ void lb_publish(uint64_t *lb, int64_t k, uint32_t epoch, uint64_t own, int lane) {
    if (lane == 0) lb_store(lb + k, epoch, k == 0 ? 2 : 1, own);
}
// (second half: the walk.  A participant may do other work between the two halves -- nothing a predecessor needs is
// held back by that, its own total is already out.)
// The walk takes LB_BATCH windows of 64 predecessors per round trip (tuning knob).  Measured on select3_run, collab-like:
// 1 window 58.2 us, 8 windows 62.2 us -- the nearest inclusive prefix is usually inside the first window, and the extra
// uncached reads of the wider walk cost more than the rare second hop.
#ifndef LB_BATCH
#define LB_BATCH 1
#endif
__device__ __forceinline__ uint64_t lb_lookback(uint64_t *lb, int64_t k, uint32_t epoch, uint64_t own, int lane) {
    if (k == 0) return 0;
    uint64_t mine = 0;     // this lane's share of the exclusive prefix
    bool done = false;
    for (int64_t j = k - 1; !done; j -= 64 * LB_BATCH) {
        uint64_t w[LB_BATCH];
#pragma unroll
        for (int b = 0; b < LB_BATCH; ++b) {
            const int64_t idx = j - lane - 64 * b;
            w[b] = idx >= 0 ? __hip_atomic_load(lb + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                            : ((uint64_t)epoch << 42) | (2ull << 40);   // before participant 0: an inclusive prefix of 0
        }
#pragma unroll
        for (int b = 0; b < LB_BATCH; ++b) {
            if (done) break;
            const int64_t idx = j - lane - 64 * b;
            if (idx >= 0 && !((uint32_t)(w[b] >> 42) == epoch && ((w[b] >> 40) & 3ull) != 0ull)) w[b] = lb_wait(lb + idx, epoch);
            const uint64_t pm = __ballot(((w[b] >> 40) & 3ull) == 2ull);
            const uint64_t v = w[b] & LB_VAL_MASK;
            if (pm) {  // nearest predecessor with an inclusive prefix: take it and the totals of the nearer ones
                const int p = __ffsll((unsigned long long)pm) - 1;
                mine += lane <= p ? v : 0ull;
                done = true;
            } else {
                mine += v;
            }
        }
    }
    const uint64_t excl = lb_wave_sum(mine);
    if (lane == 0) lb_store(lb + k, epoch, 2, excl + own);
    return excl;
}
__device__ __forceinline__ uint64_t lb_exclusive(uint64_t *lb, int64_t k, uint32_t epoch, uint64_t own, int lane) {
    lb_publish(lb, k, epoch, own, lane);
    return lb_lookback(lb, k, epoch, own, lane);
}
