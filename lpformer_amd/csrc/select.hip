// PPR-thresholded node selection for a batch of candidate pairs (integer / bit-exact part of the pair stage).
//
// The reference builds BS x N sparse COO temporaries and coalesces (sorts) them seven times per batch
// (src/models/link_transformer.py:214-319,434-481).  Here the work is cut into ITEMS of at most SEL_CAP candidate
// nodes (a whole pair when deg(a)+deg(b) <= SEL_CAP, otherwise slices of N(a) and of N(b)); one wavefront handles
// one item.  Two kernels share the item table, the staging layout and the emit code:
//   * select_nodes_indexed_kernel (evaluation; needs the per-model indexes selfp / P1 of DESIGN.md section 3): the
//     candidates and their self-PPR values arrive in one coalesced burst, each lane owns one candidate, membership
//     in the other endpoint's row is a binary search in LDS, and at most one lookup in the prefiltered one-hop rows
//     follows -- no PPR row is streamed.  ~1,400 instructions per item; issue bound.
//   * select_nodes_kernel (general: a caller-supplied adjacency has no aligned self-PPR): candidates go into an
//     LDS open-addressing hash (ds_cmpswap inserts) and the raw PPR rows of a and b are STREAMED once through it
//     (coalesced (col, val) reads, several loads in flight, lock-step probes).  Bound: HBM bandwidth on
//     4(deg a + deg b) + 8(|P_a| + |P_b|) bytes per pair.
// Both type the candidates (2 = common neighbour, 1 = one-hop), apply the reference's fp32 round trip and thresholds
// op for op, and write kept nodes to the staging area: compacted runs for single-item pairs, a dense code per
// candidate for slices of hub pairs (lpformer_hip.h); kept counts are accumulated per pair with integer atomics
// (order independent).  The >1-hop candidates (per-threshold prefiltered PPR rows) are walked by the first item of
// each pair.  select_compact_kernel then produces the reference's layout (all CN entries sorted by (pair, node),
// then 1-hop, then >1-hop).  Order comes from the CSR order: no sort anywhere.
#include "lpf_common.h"

// the reference's fp32 round trip must be evaluated op by op: no fused multiply-add in this file
#pragma clang fp contract(off)

// Orders the calling wavefront's own LDS traffic (a wave's DS instructions execute in issue order; this only stops the
// compiler from moving LDS accesses across the point).  Not a barrier between wavefronts.
#define LPF_WAVE_SYNC()                                         \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

namespace {

constexpr int SEL_CAP = 512;           // candidates per item
constexpr int SEL_TBL = 1024;          // hash slots (load factor <= 0.5)
constexpr int32_t CN_BIT = 1 << 30;    // marks a kept common neighbour in the dense staging code
constexpr int DESC_I64 = 16;           // int64 words per pair descriptor (128 B)
constexpr uint8_t F_INB = 1, F_INA = 2;

// fl32((fl32(fl32(p*t)+t)-t)/t) for t in {1,2}, without letting the compiler contract or re-associate anything.
// p*1, p*2, x/1 and x/2 are exact in binary fp32, so only the add and the subtract round.
__device__ __forceinline__ float ppr_round_trip(float p, bool two) {
    if (two) return 0.5f * __fsub_rn(__fadd_rn(p * 2.0f, 2.0f), 2.0f);
    return __fsub_rn(__fadd_rn(p, 1.0f), 1.0f);
}

__device__ __forceinline__ bool csr_contains(const int32_t *__restrict__ col, int64_t lo, int64_t hi, int32_t key) {
    const int64_t i = lpf_lower_bound(col, lo, hi, key);
    return i < hi && col[i] == key;
}

// value stored at (row, key) or 0 when absent (a sparse entry that is not stored reads as 0)
__device__ __forceinline__ float csr_value(const int32_t *__restrict__ col, const float *__restrict__ val, int64_t lo,
                                           int64_t hi, int32_t key, bool *found) {
    const int64_t i = lpf_lower_bound(col, lo, hi, key);
    const bool f = i < hi && col[i] == key;
    *found = f;
    return f ? val[i] : 0.0f;
}

__device__ __forceinline__ int lanes_below(uint64_t mask, int lane) {
    return __popcll(mask & ((1ull << lane) - 1ull));
}

__device__ __forceinline__ uint32_t sel_hash(int32_t key) { return ((uint32_t)key * 2654435761u) >> 22; }  // 10 bits

// Block-wide sum of one int64 per thread (256 threads); result valid in every thread.
__device__ __forceinline__ int64_t block_sum_i64(int64_t v, int64_t *lds4) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

__device__ __forceinline__ int items_of(int64_t dA, int64_t dB) {
    if (dA + dB <= SEL_CAP) return 1;
    return (int)((dA + SEL_CAP - 1) / SEL_CAP + (dB + SEL_CAP - 1) / SEL_CAP);
}

// Per pair: descriptor (every row start/length the later kernels need, one 128-byte line), staging capacity and item
// count; also the 256-pair block sums for the two scans.
__global__ __launch_bounds__(256) void select_bound_kernel(
    int64_t bs, const int64_t *__restrict__ batch, int64_t batch_ld, const int64_t *__restrict__ adj_rowptr,
    const int64_t *__restrict__ ppr_rowptr, const int64_t *__restrict__ t0_rowptr, int64_t *__restrict__ offs,
    int64_t *__restrict__ desc, int64_t *__restrict__ blk, int64_t nb) {
    __shared__ int64_t red[4];
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) {
        offs[0] = 0;
        offs[bs + 1] = 0;
    }
    int64_t ub = 0, ni = 0;
    if (k < bs) {
        const int64_t a = batch[k], b = batch[batch_ld + k];
        const int64_t ra0 = adj_rowptr[a], ra1 = adj_rowptr[a + 1], rb0 = adj_rowptr[b], rb1 = adj_rowptr[b + 1];
        const int64_t pa0 = ppr_rowptr[a], pa1 = ppr_rowptr[a + 1], pb0 = ppr_rowptr[b], pb1 = ppr_rowptr[b + 1];
        int64_t ta0 = 0, ta1 = 0, tb0 = 0, tb1 = 0;
        if (t0_rowptr) {
            ta0 = t0_rowptr[a]; ta1 = t0_rowptr[a + 1]; tb0 = t0_rowptr[b]; tb1 = t0_rowptr[b + 1];
        }
        const int64_t dA = ra1 - ra0, dB = rb1 - rb0, ha = ta1 - ta0, hb = tb1 - tb0;
        ub = dA + dB + (ha < hb ? ha : hb);
        ni = items_of(dA, dB);
        offs[k + 1] = ub;
        offs[(bs + 1) + k + 1] = ni;
        int64_t *d = desc + k * DESC_I64;
        d[0] = ra0; d[1] = rb0; d[2] = pa0; d[3] = pb0; d[4] = ta0; d[5] = tb0;
        d[6] = dA; d[7] = dB; d[8] = pa1 - pa0; d[9] = pb1 - pb0; d[10] = ha; d[11] = hb;
        d[12] = a; d[13] = b; d[14] = 0; d[15] = 0;
    }
    const int64_t s0 = block_sum_i64(ub, red), s1 = block_sum_i64(ni, red);
    if (threadIdx.x == 0) {
        blk[blockIdx.x] = s0;
        blk[nb + blockIdx.x] = s1;
    }
}

// Second half of a two-kernel scan.  The producer kernel (256 threads, one element per thread) has written its
// values to data[q*stride + 1 + i] and the sum of every 256-element block to blk[q*nb + block]; here block B adds
// the sums of the blocks before it to an in-block inclusive scan.  Fully parallel: grid = nb blocks.
// The grand totals (= last element of each scanned sequence) are also written next to each other at
// blk[3 * nb + q], so the host can fetch them with one small contiguous copy.
template <int NSEQ>
__global__ __launch_bounds__(256) void scan_blocks_kernel(int64_t n, int64_t *__restrict__ data, int64_t stride,
                                                          int64_t *__restrict__ blk, int64_t nb) {
    __shared__ int64_t red[4];
    __shared__ int64_t wtot[NSEQ][4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t B = blockIdx.x, i = B * 256 + t;
#pragma unroll
    for (int q = 0; q < NSEQ; ++q) {
        int64_t pre = 0;
        for (int64_t j = t; j < B; j += 256) pre += blk[q * nb + j];
        pre = block_sum_i64(pre, red);
        int64_t x = (i < n) ? data[q * stride + 1 + i] : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wtot[q][wave] = x;
        __syncthreads();
        for (int w = 0; w < wave; ++w) pre += wtot[q][w];
        if (i < n) data[q * stride + 1 + i] = x + pre;
        if (i == n - 1) blk[3 * nb + q] = x + pre;
    }
}

// Work-item record, 64 bytes = one coalesced 16-lane read: everything the item kernel needs to start loading rows.
// kind 0 = whole pair, 1 = slice of N(a), 2 = slice of N(b)
struct ItemRec {
    int32_t p, kind, start, len, dA, dB, nPa, nPb;
    int64_t ra0, rb0, pa0, pb0;
};
static_assert(sizeof(ItemRec) == 64, "ItemRec must be 64 bytes");

__global__ __launch_bounds__(256) void select_items_kernel(int64_t bs, const int64_t *__restrict__ desc,
                                                           const int64_t *__restrict__ item_off,
                                                           ItemRec *__restrict__ items,
                                                           int32_t *__restrict__ stage_cnt) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= bs) return;
    *reinterpret_cast<int4 *>(stage_cnt + 4 * k) = make_int4(0, 0, 0, 0);  // the item kernel accumulates into these
    const int64_t *d = desc + k * DESC_I64;
    const int64_t dA = d[6], dB = d[7];
    ItemRec r;
    r.p = (int32_t)k; r.dA = (int32_t)dA; r.dB = (int32_t)dB; r.nPa = (int32_t)d[8]; r.nPb = (int32_t)d[9];
    r.ra0 = d[0]; r.rb0 = d[1]; r.pa0 = d[2]; r.pb0 = d[3];
    int64_t o = item_off[k];
    if (dA + dB <= SEL_CAP) {
        r.kind = 0; r.start = 0; r.len = (int32_t)(dA + dB);
        items[o] = r;
        return;
    }
    for (int64_t s = 0; s < dA; s += SEL_CAP) {
        r.kind = 1; r.start = (int32_t)s; r.len = (int32_t)((dA - s) < SEL_CAP ? (dA - s) : SEL_CAP);
        items[o++] = r;
    }
    for (int64_t s = 0; s < dB; s += SEL_CAP) {
        r.kind = 2; r.start = (int32_t)s; r.len = (int32_t)((dB - s) < SEL_CAP ? (dB - s) : SEL_CAP);
        items[o++] = r;
    }
}

// Stores one chunk of typed candidates and updates the per-item counters.
//   kind 0 (the item is the whole pair): COMPACTED runs, written as they are found -- common neighbours upwards from
//     s, kept one-hop nodes of N(a) DOWNWARDS from s+dA-1 (both fit the dA slots of the N(a) run), kept one-hop nodes
//     of N(b) upwards from s+dA.  Node order inside a run follows the CSR order (ascending; descending for the
//     reversed run).
//   slices of a hub pair: DENSE, one code per candidate (node | CN bit, or -1), compacted later per pair.
__device__ __forceinline__ void sel_emit(int kind, int lane, bool in_range, bool keep, bool cn, bool from_a,
                                         int32_t node, float va, float vb, int64_t s, int64_t dA, int64_t dense_slot,
                                         int &n_cn, int &n_l1, int &n_l2, int32_t *__restrict__ stage_node,
                                         float *__restrict__ stage_pa, float *__restrict__ stage_pb) {
    const uint64_t m_cn = __ballot(keep && cn), m_l1 = __ballot(keep && !cn && from_a),
                   m_l2 = __ballot(keep && !from_a);
    if (kind == 0) {
        if (keep) {
            const int64_t dst = cn ? s + n_cn + lanes_below(m_cn, lane)
                                   : (from_a ? s + dA - 1 - (n_l1 + lanes_below(m_l1, lane))
                                             : s + dA + n_l2 + lanes_below(m_l2, lane));
            stage_node[dst] = node;
            stage_pa[dst] = va;
            stage_pb[dst] = vb;
        }
    } else if (in_range) {
        stage_node[dense_slot] = keep ? (node | (cn ? CN_BIT : 0)) : -1;
        if (keep) {
            stage_pa[dense_slot] = va;
            stage_pb[dense_slot] = vb;
        }
    }
    n_cn += __popcll(m_cn);
    n_l1 += __popcll(m_l1);
    n_l2 += __popcll(m_l2);
}

struct alignas(16) SelLds {
    int32_t cand[SEL_CAP];
    float pa[SEL_CAP];
    float pb[SEL_CAP];
    int32_t table[SEL_TBL];
    uint8_t flag[SEL_CAP];
};

// slot of `key` in the item's hash, or -1
__device__ __forceinline__ int sel_probe(const SelLds &L, int32_t key) {
    uint32_t h = sel_hash(key);
    while (true) {
        const int s = L.table[h];
        if (s < 0) return -1;
        if (L.cand[s] == key) return s;
        h = (h + 1) & (SEL_TBL - 1);
    }
}

// Probe U keys per lane in lock step: the U table reads (and then the U key compares) of a round are independent, so
// their LDS latencies overlap; with a load factor <= 0.5 almost every key is settled in the first round.
template <int U>
__device__ __forceinline__ void sel_probe_values(SelLds &L, float *dst, const int32_t (&c)[U], const float (&v)[U]) {
    uint32_t h[U];
    bool act[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        h[u] = sel_hash(c[u]);
        act[u] = c[u] >= 0;
    }
    while (true) {
        int s[U];
        int32_t k[U];
#pragma unroll
        for (int u = 0; u < U; ++u) s[u] = act[u] ? L.table[h[u]] : -1;
#pragma unroll
        for (int u = 0; u < U; ++u) k[u] = (s[u] >= 0) ? L.cand[s[u]] : -1;
        bool again = false;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (act[u]) {
                if (s[u] < 0) {
                    act[u] = false;  // empty slot: the node is not a candidate
                } else if (k[u] == c[u]) {
                    dst[s[u]] = v[u];
                    act[u] = false;
                } else {
                    h[u] = (h[u] + 1) & (SEL_TBL - 1);
                    again = true;
                }
            }
        }
        if (__ballot(again) == 0) break;
    }
}

// stream the tail of a (col, val) CSR row (entries from `from` on) through the hash
__device__ __forceinline__ void sel_stream_values(SelLds &L, float *dst, const int32_t *__restrict__ col,
                                                  const float *__restrict__ val, int64_t lo, int from, int n,
                                                  int lane) {
    for (int i0 = from; i0 < n; i0 += 256) {  // four independent 64-entry loads in flight
        int32_t c[4];
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 64 * u + lane;
            c[u] = (i < n) ? col[lo + i] : -1;
            v[u] = (i < n) ? val[lo + i] : 0.f;
        }
        sel_probe_values<4>(L, dst, c, v);
    }
}

// stream a sorted adjacency row through the hash: hits set `bit` as the candidate's flag
__device__ __forceinline__ void sel_stream_flags(SelLds &L, uint8_t bit, const int32_t *__restrict__ col, int64_t lo,
                                                 int64_t n, int lane) {
    for (int64_t i0 = 0; i0 < n; i0 += 256) {
        int32_t c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + 64 * u + lane;
            c[u] = (i < n) ? col[lo + i] : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c[u] >= 0) {
                const int s = sel_probe(L, c[u]);
                if (s >= 0) L.flag[s] = bit;  // one writer per slot (row entries are distinct)
            }
        }
    }
}

constexpr int SEL_PRE = 6;  // 64-entry chunks of each PPR row fetched up front (384 entries)

__global__ __launch_bounds__(256) void select_nodes_kernel(
    const int64_t *__restrict__ item_total, const ItemRec *__restrict__ items, const int64_t *__restrict__ desc,
    const int32_t *__restrict__ adj_col, const int64_t *__restrict__ adjx_rowptr, const int32_t *__restrict__ adjx_col,
    int same_adj, const int32_t *__restrict__ ppr_col, const float *__restrict__ ppr_val,
    const int32_t *__restrict__ t0_col, const float *__restrict__ t0_val, int want_t0, float th_cn, float th_1,
    float th_n, const int64_t *__restrict__ stage_off, int32_t *__restrict__ stage_node, float *__restrict__ stage_pa,
    float *__restrict__ stage_pb, int32_t *__restrict__ stage_cnt) {
    // Four independent wavefronts per workgroup, one item each, private LDS images (workgroup dispatch rate, not
    // wave count, limits how fast tiny work items can be started).  The waves never synchronise with each other:
    // LPF_WAVE_SYNC orders a wave's own LDS traffic only (its DS instructions execute in issue order).
    __shared__ SelLds Ls[4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    SelLds &L = Ls[wave];
    const int64_t n_items = *item_total;

    // the record address is wave-uniform, so its fields arrive through scalar loads and every row base / bound
    // below is scalar; uneven items are balanced by the hardware dispatcher (about one item per wavefront)
    for (int64_t it = (int64_t)blockIdx.x * 4 + wave; it < n_items; it += (int64_t)gridDim.x * 4) {
        const ItemRec r = items[it];
        const int64_t p = r.p;
        const int kind = r.kind, start = r.start, len = r.len;
        const int64_t dA = r.dA, dB = r.dB;
        const int nPa = r.nPa, nPb = r.nPb;
        const int64_t ra0 = r.ra0, rb0 = r.rb0, pa0 = r.pa0, pb0 = r.pb0;
        const int64_t s = stage_off[p];

        // ---- 1. every global read of the item is issued up front: candidates and the head of both PPR rows
        //         (chunks past the end of a row are skipped by scalar branches)
        int32_t cnd[SEL_CAP / 64];
        const int32_t *rowA = adj_col + ra0, *rowB = adj_col + rb0;
        const int32_t *rowS = adj_col + (kind == 1 ? ra0 : rb0) + start;
#pragma unroll
        for (int u = 0; u < SEL_CAP / 64; ++u) {
            cnd[u] = -1;
            if (64 * u < len) {
                const int i = lane + 64 * u;
                if (i < len) {
                    if (kind == 0) cnd[u] = (i < dA) ? rowA[i] : rowB[i - dA];
                    else cnd[u] = rowS[i];
                }
            }
        }
        int32_t ca[SEL_PRE], cb[SEL_PRE];
        float wa[SEL_PRE], wb[SEL_PRE];
        const int32_t *pca = ppr_col + pa0, *pcb = ppr_col + pb0;
        const float *pva = ppr_val + pa0, *pvb = ppr_val + pb0;
#pragma unroll
        for (int u = 0; u < SEL_PRE; ++u) {
            const int i = lane + 64 * u;
            ca[u] = -1; wa[u] = 0.f; cb[u] = -1; wb[u] = 0.f;
            if (64 * u < nPa && i < nPa) { ca[u] = pca[i]; wa[u] = pva[i]; }
            if (64 * u < nPb && i < nPb) { cb[u] = pcb[i]; wb[u] = pvb[i]; }
        }
        LPF_WAVE_SYNC();  // the previous item's LDS image is no longer needed
        // empty table and value slots while the loads are in flight
        for (int i = lane; i < SEL_TBL / 4; i += 64)
            reinterpret_cast<int4 *>(L.table)[i] = make_int4(-1, -1, -1, -1);
        for (int i = lane; i < (len + 3) / 4; i += 64) {
            reinterpret_cast<float4 *>(L.pa)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            reinterpret_cast<float4 *>(L.pb)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int i = lane; i < (len + 3) / 4; i += 64) reinterpret_cast<uint32_t *>(L.flag)[i] = 0u;
#pragma unroll
        for (int u = 0; u < SEL_CAP / 64; ++u)
            if (lane + 64 * u < len) L.cand[lane + 64 * u] = cnd[u];
        LPF_WAVE_SYNC();
        // ---- 2. hash inserts (whole pair: N(a) first, then N(b); a node found again is a common neighbour)
        const int n_first = (kind == 0) ? (int)dA : len;
        for (int i = lane; i < n_first; i += 64) {
            uint32_t h = sel_hash(L.cand[i]);
            while (atomicCAS(&L.table[h], -1, i) != -1) h = (h + 1) & (SEL_TBL - 1);
        }
        if (kind == 0) {
            LPF_WAVE_SYNC();
            for (int i = (int)dA + lane; i < len; i += 64) {
                const int32_t key = L.cand[i];
                uint32_t h = sel_hash(key);
                while (true) {
                    const int prev = atomicCAS(&L.table[h], -1, i);
                    if (prev == -1) break;
                    if (L.cand[prev] == key) {  // prev is the copy from N(a)
                        L.flag[prev] = F_INB;
                        L.flag[i] = F_INA;
                        break;
                    }
                    h = (h + 1) & (SEL_TBL - 1);
                }
            }
        }
        LPF_WAVE_SYNC();
        // ---- 3. stream the rows that carry information about the candidates
        if (kind == 1) sel_stream_flags(L, F_INB, adj_col, rb0, dB, lane);
        if (kind == 2) sel_stream_flags(L, F_INA, adj_col, ra0, dA, lane);
        sel_probe_values<SEL_PRE>(L, L.pa, ca, wa);
        sel_probe_values<SEL_PRE>(L, L.pb, cb, wb);
        if (nPa > 64 * SEL_PRE) sel_stream_values(L, L.pa, ppr_col, ppr_val, pa0, 64 * SEL_PRE, nPa, lane);
        if (nPb > 64 * SEL_PRE) sel_stream_values(L, L.pb, ppr_col, ppr_val, pb0, 64 * SEL_PRE, nPb, lane);
        LPF_WAVE_SYNC();
        // ---- 4. type, round trip, threshold; dense code per candidate; counts
        int n_cn = 0, n_l1 = 0, n_l2 = 0;
        const int64_t slot0 = s + (kind == 2 ? dA : 0) + start;  // kind 0: candidate i <-> staging slot s + i
        for (int i0 = 0; i0 < len; i0 += 64) {
            const int i = i0 + lane;
            int32_t node = 0;
            bool from_a = false, cn = false, keep = false;
            float va = 0.f, vb = 0.f;
            if (i < len) {
                from_a = (kind == 1) || (kind == 0 && i < dA);
                const uint8_t f = L.flag[i];
                node = L.cand[i];
                if (from_a || !(f & F_INA)) {  // a node of N(b) that is also in N(a) is emitted through N(a)
                    cn = from_a && (f & F_INB);
                    va = ppr_round_trip(L.pa[i], cn);
                    vb = ppr_round_trip(L.pb[i], cn);
                    const float th = cn ? th_cn : th_1;
                    keep = (va >= th) && (vb >= th);
                }
            }
            sel_emit(kind, lane, i < len, keep, cn, from_a, node, va, vb, s, dA, slot0 + i, n_cn, n_l1, n_l2,
                     stage_node, stage_pa, stage_pb);
        }
        if (lane == 0) {
            if (n_cn) atomicAdd(&stage_cnt[4 * p + 0], n_cn);
            if (n_l1) atomicAdd(&stage_cnt[4 * p + 1], n_l1);
            if (n_l2) atomicAdd(&stage_cnt[4 * p + 2], n_l2);
        }
        // ---- 5. >1-hop nodes: stored in both T0 rows, adjacent to neither endpoint (UNMASKED adjacency); done once
        //         per pair, by its first item
        if (want_t0 && (kind == 0 || (kind == 1 && start == 0) || (kind == 2 && start == 0 && dA == 0))) {
            const int64_t *d = desc + p * DESC_I64;
            const int64_t ta0 = d[4], tb0 = d[5], nTa = d[10], nTb = d[11], a = d[12], b = d[13];
            const bool walk_a = nTa <= nTb;  // walk the shorter row, probe the longer
            const int64_t w0 = walk_a ? ta0 : tb0, wn = walk_a ? nTa : nTb;
            const int64_t o0 = walk_a ? tb0 : ta0, o1 = o0 + (walk_a ? nTb : nTa);
            const bool use_hash = (kind == 0) && same_adj;  // the hash holds all of N(a) u N(b)
            int64_t xa0 = ra0, xa1 = ra0 + dA, xb0 = rb0, xb1 = rb0 + dB;
            const int32_t *xcol = adj_col;
            if (!same_adj) {
                xa0 = adjx_rowptr[a]; xa1 = adjx_rowptr[a + 1]; xb0 = adjx_rowptr[b]; xb1 = adjx_rowptr[b + 1];
                xcol = adjx_col;
            }
            const int64_t t0_base = s + dA + dB;
            int n_t0 = 0;
            for (int64_t i0 = 0; i0 < wn; i0 += 64) {
                const int64_t i = i0 + lane;
                int32_t v = 0;
                bool keep = false;
                float sa = 0.f, sb = 0.f;
                if (i < wn) {
                    v = t0_col[w0 + i];
                    const float pw = t0_val[w0 + i];
                    const float sw = __fsub_rn(__fadd_rn(pw, 1.0f), 1.0f);
                    if (pw > 0.f && sw >= th_n) {
                        bool f;
                        const float po = csr_value(t0_col, t0_val, o0, o1, v, &f);
                        const float so = __fsub_rn(__fadd_rn(po, 1.0f), 1.0f);
                        if (f && po > 0.f && so >= th_n) {
                            const bool adjacent = use_hash ? (sel_probe(L, v) >= 0)
                                                           : (csr_contains(xcol, xa0, xa1, v) ||
                                                              csr_contains(xcol, xb0, xb1, v));
                            if (!adjacent) {
                                keep = true;
                                sa = walk_a ? sw : so;
                                sb = walk_a ? so : sw;
                            }
                        }
                    }
                }
                const uint64_t m = __ballot(keep);
                if (keep) {
                    const int64_t dst = t0_base + n_t0 + lanes_below(m, lane);
                    stage_node[dst] = v;
                    stage_pa[dst] = sa;
                    stage_pb[dst] = sb;
                }
                n_t0 += __popcll(m);
            }
            if (lane == 0) stage_cnt[4 * p + 3] = n_t0;
        }
    }
}

// counts -> type_ptr rows (offset by one, ready for scan_blocks_kernel) + block sums + float count features
__global__ __launch_bounds__(256) void select_counts_kernel(int64_t bs, const int32_t *__restrict__ stage_cnt,
                                                            int64_t *__restrict__ type_ptr,
                                                            float *__restrict__ counts_f, int64_t ldc, int want_t0,
                                                            int64_t *__restrict__ blk, int64_t nb) {
    __shared__ int64_t red[4];
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) {
        type_ptr[0] = 0;
        type_ptr[bs + 1] = 0;
        type_ptr[2 * (bs + 1)] = 0;
    }
    int n_cn = 0, n_1 = 0, n_0 = 0;
    if (k < bs) {
        const int4 c4 = *reinterpret_cast<const int4 *>(stage_cnt + 4 * k);
        n_cn = c4.x; n_1 = c4.y + c4.z; n_0 = c4.w;
        type_ptr[k + 1] = n_cn;
        type_ptr[(bs + 1) + k + 1] = n_1;
        type_ptr[2 * (bs + 1) + k + 1] = n_0;
        if (counts_f) {
            float *c = counts_f + k * ldc;
            c[0] = (float)n_cn;
            c[1] = (float)n_1;
            if (want_t0) {
                c[2] = (float)n_0;
                c[3] = (float)(n_cn + n_1);
            } else {
                c[2] = (float)(n_cn + n_1);
            }
        }
    }
    const int64_t s0 = block_sum_i64(n_cn, red), s1 = block_sum_i64(n_1, red), s2 = block_sum_i64(n_0, red);
    if (threadIdx.x == 0) {
        blk[blockIdx.x] = s0;
        blk[nb + blockIdx.x] = s1;
        blk[2 * nb + blockIdx.x] = s2;
    }
}

constexpr int CMP_CAP = 128;  // kept one-hop nodes per run that the compaction kernel merges through LDS
struct CmpLds {
    int32_t n1[CMP_CAP], s1[CMP_CAP], n2[CMP_CAP], s2[CMP_CAP];  // node ids and staging slots of the two runs
};

// One wavefront per pair: dense runs -> reference layout.  CN entries go straight to their final place; the kept
// one-hop nodes of the N(a) run and of the N(b) run are first compacted IN PLACE (writes never pass the read
// cursor), then merged: two sorted, disjoint runs, final rank = own index + lower_bound in the other run.
template <int WPB>  // wavefronts (= concurrent pairs) per workgroup; the waves never synchronise with each other
__global__ __launch_bounds__(64 * WPB) void select_compact_kernel(
    int64_t bs, const int64_t *__restrict__ desc, const int64_t *__restrict__ stage_off,
    int32_t *__restrict__ stage_node, float *__restrict__ stage_pa, float *__restrict__ stage_pb,
    const int32_t *__restrict__ stage_cnt, const int64_t *__restrict__ type_ptr, int32_t *__restrict__ sel_pair,
    int32_t *__restrict__ sel_node, float *__restrict__ sel_pa, float *__restrict__ sel_pb) {
    __shared__ CmpLds cmp_lds[WPB];
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * WPB;
    const int64_t tot_cn = type_ptr[bs], tot_1 = type_ptr[(bs + 1) + bs];
    for (int64_t p = wave_id; p < bs; p += n_waves) {
        const int4 kept = *reinterpret_cast<const int4 *>(stage_cnt + 4 * p);
        if ((kept.x | kept.y | kept.z | kept.w) == 0) continue;  // nothing selected for this pair (almost half of them)
        const int64_t dA = desc[p * DESC_I64 + 6], dB = desc[p * DESC_I64 + 7];
        const int64_t s = stage_off[p];
        const int n_l1 = kept.y, n_l2 = kept.z, n_t0 = kept.w;
        const int64_t d_cn = type_ptr[p], d_1 = tot_cn + type_ptr[(bs + 1) + p],
                      d_0 = tot_cn + tot_1 + type_ptr[2 * (bs + 1) + p];
        const int64_t l1 = s, l2 = s + dA, t0 = s + dA + dB;
        if (dA + dB <= SEL_CAP) {
            // the pair was a single work item: its runs are already compacted (the N(a) one-hop run stored downwards
            // from s+dA-1).  Copy CN, merge the two one-hop runs by rank = own index + lower_bound in the other run
            // (they were written by the previous kernel: plain global reads), copy the >1-hop run.
            const int n_cn = stage_cnt[4 * p];
            for (int i = lane; i < n_cn; i += 64) {
                sel_pair[d_cn + i] = (int32_t)p;
                sel_node[d_cn + i] = stage_node[s + i];
                sel_pa[d_cn + i] = stage_pa[s + i];
                sel_pb[d_cn + i] = stage_pb[s + i];
            }
            const int64_t top = s + dA - 1;  // element k of the N(a) one-hop run lives at top - k
            for (int i = lane; i < n_l1; i += 64) {
                const int32_t x = stage_node[top - i];
                const int64_t dst = d_1 + i + (lpf_lower_bound(stage_node, l2, l2 + n_l2, x) - l2);
                sel_pair[dst] = (int32_t)p;
                sel_node[dst] = x;
                sel_pa[dst] = stage_pa[top - i];
                sel_pb[dst] = stage_pb[top - i];
            }
            for (int j = lane; j < n_l2; j += 64) {
                const int32_t y = stage_node[l2 + j];
                int lo = 0, hi = n_l1;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (stage_node[top - mid] < y) lo = mid + 1; else hi = mid;
                }
                const int64_t dst = d_1 + j + lo;
                sel_pair[dst] = (int32_t)p;
                sel_node[dst] = y;
                sel_pa[dst] = stage_pa[l2 + j];
                sel_pb[dst] = stage_pb[l2 + j];
            }
            for (int i = lane; i < n_t0; i += 64) {
                sel_pair[d_0 + i] = (int32_t)p;
                sel_node[d_0 + i] = stage_node[t0 + i];
                sel_pa[d_0 + i] = stage_pa[t0 + i];
                sel_pb[d_0 + i] = stage_pb[t0 + i];
            }
            continue;
        }
        if (n_l1 <= CMP_CAP && n_l2 <= CMP_CAP) {
            // medium path: the kept one-hop nodes of both runs (with the staging slot they came from) go to this
            // wave's LDS lists; ranks come from binary searches in LDS, values are fetched from the original slots
            CmpLds &W = cmp_lds[threadIdx.x >> 6];
            int c_cn = 0, c_l1 = 0, c_l2 = 0;
            LPF_WAVE_SYNC();
            for (int64_t i0 = 0; i0 < dA; i0 += 64) {
                const int64_t i = i0 + lane;
                const int32_t code = (i < dA) ? stage_node[s + i] : -1;
                const bool is_cn = code >= 0 && (code & CN_BIT), is_l1 = code >= 0 && !(code & CN_BIT);
                const uint64_t m_cn = __ballot(is_cn), m_l1 = __ballot(is_l1);
                if (is_cn) {
                    const int64_t dst = d_cn + c_cn + lanes_below(m_cn, lane);
                    sel_pair[dst] = (int32_t)p;
                    sel_node[dst] = code & ~CN_BIT;
                    sel_pa[dst] = stage_pa[s + i];
                    sel_pb[dst] = stage_pb[s + i];
                }
                if (is_l1) {
                    const int k = c_l1 + lanes_below(m_l1, lane);
                    W.n1[k] = code;
                    W.s1[k] = (int32_t)i;
                }
                c_cn += __popcll(m_cn);
                c_l1 += __popcll(m_l1);
            }
            for (int64_t j0 = 0; j0 < dB; j0 += 64) {
                const int64_t j = j0 + lane;
                const int32_t code = (j < dB) ? stage_node[l2 + j] : -1;
                const uint64_t m = __ballot(code >= 0);
                if (code >= 0) {
                    const int k = c_l2 + lanes_below(m, lane);
                    W.n2[k] = code;
                    W.s2[k] = (int32_t)j;
                }
                c_l2 += __popcll(m);
            }
            LPF_WAVE_SYNC();
            for (int i = lane; i < n_l1; i += 64) {
                const int32_t x = W.n1[i];
                int lo = 0, hi = n_l2;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (W.n2[mid] < x) lo = mid + 1; else hi = mid;
                }
                const int64_t dst = d_1 + i + lo, src = l1 + W.s1[i];
                sel_pair[dst] = (int32_t)p;
                sel_node[dst] = x;
                sel_pa[dst] = stage_pa[src];
                sel_pb[dst] = stage_pb[src];
            }
            for (int j = lane; j < n_l2; j += 64) {
                const int32_t y = W.n2[j];
                int lo = 0, hi = n_l1;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (W.n1[mid] < y) lo = mid + 1; else hi = mid;
                }
                const int64_t dst = d_1 + j + lo, src = l2 + W.s2[j];
                sel_pair[dst] = (int32_t)p;
                sel_node[dst] = y;
                sel_pa[dst] = stage_pa[src];
                sel_pb[dst] = stage_pb[src];
            }
            for (int i = lane; i < n_t0; i += 64) {
                sel_pair[d_0 + i] = (int32_t)p;
                sel_node[d_0 + i] = stage_node[t0 + i];
                sel_pa[d_0 + i] = stage_pa[t0 + i];
                sel_pb[d_0 + i] = stage_pb[t0 + i];
            }
            continue;
        }
        int c_cn = 0, c_l1 = 0, c_l2 = 0;
        for (int64_t i0 = 0; i0 < dA; i0 += 64) {  // N(a) run: CN -> final, one-hop -> in place
            const int64_t i = i0 + lane;
            int32_t code = -1;
            float va = 0.f, vb = 0.f;
            if (i < dA) {
                code = stage_node[s + i];
                if (code >= 0) {
                    va = stage_pa[s + i];
                    vb = stage_pb[s + i];
                }
            }
            const bool is_cn = code >= 0 && (code & CN_BIT), is_l1 = code >= 0 && !(code & CN_BIT);
            const uint64_t m_cn = __ballot(is_cn), m_l1 = __ballot(is_l1);
            if (is_cn) {
                const int64_t dst = d_cn + c_cn + lanes_below(m_cn, lane);
                sel_pair[dst] = (int32_t)p;
                sel_node[dst] = code & ~CN_BIT;
                sel_pa[dst] = va;
                sel_pb[dst] = vb;
            }
            if (is_l1) {
                const int64_t dst = l1 + c_l1 + lanes_below(m_l1, lane);
                stage_node[dst] = code;
                stage_pa[dst] = va;
                stage_pb[dst] = vb;
            }
            c_cn += __popcll(m_cn);
            c_l1 += __popcll(m_l1);
        }
        for (int64_t j0 = 0; j0 < dB; j0 += 64) {  // N(b) run: one-hop -> in place
            const int64_t j = j0 + lane;
            int32_t code = -1;
            float va = 0.f, vb = 0.f;
            if (j < dB) {
                code = stage_node[l2 + j];
                if (code >= 0) {
                    va = stage_pa[l2 + j];
                    vb = stage_pb[l2 + j];
                }
            }
            const uint64_t m = __ballot(code >= 0);
            if (code >= 0) {
                const int64_t dst = l2 + c_l2 + lanes_below(m, lane);
                stage_node[dst] = code;
                stage_pa[dst] = va;
                stage_pb[dst] = vb;
            }
            c_l2 += __popcll(m);
        }
        // the compacted runs (global memory) are read back by other lanes of this wavefront below
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < n_l1; i += 64) {
            const int32_t x = stage_node[l1 + i];
            const int64_t dst = d_1 + i + (lpf_lower_bound(stage_node, l2, l2 + n_l2, x) - l2);
            sel_pair[dst] = (int32_t)p;
            sel_node[dst] = x;
            sel_pa[dst] = stage_pa[l1 + i];
            sel_pb[dst] = stage_pb[l1 + i];
        }
        for (int j = lane; j < n_l2; j += 64) {
            const int32_t y = stage_node[l2 + j];
            const int64_t dst = d_1 + j + (lpf_lower_bound(stage_node, l1, l1 + n_l1, y) - l1);
            sel_pair[dst] = (int32_t)p;
            sel_node[dst] = y;
            sel_pa[dst] = stage_pa[l2 + j];
            sel_pb[dst] = stage_pb[l2 + j];
        }
        for (int i = lane; i < n_t0; i += 64) {
            sel_pair[d_0 + i] = (int32_t)p;
            sel_node[d_0 + i] = stage_node[t0 + i];
            sel_pa[d_0 + i] = stage_pa[t0 + i];
            sel_pb[d_0 + i] = stage_pb[t0 + i];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Indexed variant (evaluation fast path).  Two per-model indexes over the PPR matrix remove the row streaming:
//   selfp[e] = P[i, j] for adjacency entry e = (i, j) (0 when not stored): the PPR of a node to its own neighbours,
//              aligned with the adjacency CSR, so "P[a, x] for x in N(a)" is a coalesced read;
//   P1       = the PPR rows restricted to entries that can pass the one-hop test (fl32(fl32(p+1)-1) >= theta_1).
// For a candidate x of N(a): if x is in N(b) (binary search in LDS) it is a common neighbour and P[b, x] is the self
// value of b at the position just found; otherwise P[b, x] is looked up in P1[b] -- and only when x's own value
// passes theta_1.  Each lane owns one candidate: no hash table, no cross-lane traffic, ~4x fewer instructions.
// Results are identical to select_nodes_kernel (tests compare both against the reference's golden vectors).
constexpr int IDX_P1_CAP = 256;  // P1 rows up to this length are searched in LDS (they hold at most ~1/theta_1 entries)
struct alignas(16) IdxLds {     // 4 KiB per wavefront: eight wavefronts per SIMD fit
    int32_t cand[SEL_CAP];      // kind 0: N(a) then N(b); otherwise the slice
    int32_t p1[2 * IDX_P1_CAP];  // columns of P1[a] | P1[b] when they fit (searched by the other endpoint's nodes);
                                 // reused as one buffer for the other endpoint's T0 row in the >1-hop pass
};

__device__ __forceinline__ int find_lds(const int32_t *a, int n, int32_t key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < n && a[lo] == key) ? lo : -1;
}

__device__ __forceinline__ int find_glb(const int32_t *__restrict__ a, int n, int32_t key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < n && a[lo] == key) ? lo : -1;
}

template <int WPB>  // wavefronts (= concurrent items) per workgroup
__global__ __launch_bounds__(64 * WPB) void select_nodes_indexed_kernel(
    const int64_t *__restrict__ item_total, const ItemRec *__restrict__ items, const int64_t *__restrict__ desc,
    const int32_t *__restrict__ adj_col, const float *__restrict__ selfp, const int64_t *__restrict__ adjx_rowptr,
    const int32_t *__restrict__ adjx_col, int same_adj, const int32_t *__restrict__ p1_col,
    const float *__restrict__ p1_val, const int32_t *__restrict__ t0_col, const float *__restrict__ t0_val,
    int want_t0, float th_cn, float th_1, float th_n, const int64_t *__restrict__ stage_off,
    int32_t *__restrict__ stage_node, float *__restrict__ stage_pa, float *__restrict__ stage_pb,
    int32_t *__restrict__ stage_cnt) {
    __shared__ IdxLds Ls[WPB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    IdxLds &L = Ls[wave];
    const int64_t n_items = *item_total;

    for (int64_t it = (int64_t)blockIdx.x * WPB + wave; it < n_items; it += (int64_t)gridDim.x * WPB) {
        const ItemRec r = items[it];
        const int64_t p = r.p;
        const int kind = r.kind, start = r.start, len = r.len;
        const int dA = r.dA, dB = r.dB, nPa = r.nPa, nPb = r.nPb;  // nPa/nPb, pa0/pb0: the P1 rows
        const int64_t ra0 = r.ra0, rb0 = r.rb0, pa0 = r.pa0, pb0 = r.pb0;
        const int64_t s = stage_off[p];
        const bool p1a_lds = nPa <= IDX_P1_CAP, p1b_lds = nPb <= IDX_P1_CAP;

        // ---- 1. all global reads up front: candidates with their self PPR, and the P1 columns
        int32_t cnd[SEL_CAP / 64], ta[IDX_P1_CAP / 64], tb[IDX_P1_CAP / 64];
        float sp[SEL_CAP / 64];
        const int64_t base_s = (kind == 1 ? ra0 : rb0) + start;
#pragma unroll
        for (int u = 0; u < SEL_CAP / 64; ++u) {
            cnd[u] = -1; sp[u] = 0.f;
            const int i = lane + 64 * u;
            if (64 * u < len && i < len) {
                const int64_t e = (kind == 0) ? ((i < dA) ? ra0 + i : rb0 + (i - dA)) : base_s + i;
                cnd[u] = adj_col[e];
                sp[u] = selfp[e];
            }
        }
#pragma unroll
        for (int u = 0; u < IDX_P1_CAP / 64; ++u) {
            ta[u] = 0; tb[u] = 0;
            const int i = lane + 64 * u;
            if (p1a_lds && 64 * u < nPa && i < nPa) ta[u] = p1_col[pa0 + i];
            if (p1b_lds && 64 * u < nPb && i < nPb) tb[u] = p1_col[pb0 + i];
        }
        LPF_WAVE_SYNC();  // the previous item's LDS image is no longer needed
#pragma unroll
        for (int u = 0; u < SEL_CAP / 64; ++u) {
            const int i = lane + 64 * u;
            if (64 * u < len && i < len) L.cand[i] = cnd[u];
        }
#pragma unroll
        for (int u = 0; u < IDX_P1_CAP / 64; ++u) {
            const int i = lane + 64 * u;
            if (p1a_lds && 64 * u < nPa && i < nPa) L.p1[i] = ta[u];
            if (p1b_lds && 64 * u < nPb && i < nPb) L.p1[IDX_P1_CAP + i] = tb[u];
        }
        LPF_WAVE_SYNC();

        // ---- 2. one lane per candidate: membership in the other adjacency row, then (maybe) one P1 lookup
        int n_cn = 0, n_l1 = 0, n_l2 = 0;
        const int64_t slot0 = s + (kind == 2 ? dA : 0) + start;
#pragma unroll
        for (int u = 0; u < SEL_CAP / 64; ++u) {
            if (64 * u >= len) break;
            const int i = lane + 64 * u;
            bool from_a = false, cn = false, keep = false;
            float va = 0.f, vb = 0.f;
            const int32_t x = cnd[u];
            if (i < len) {
                from_a = (kind == 1) || (kind == 0 && i < dA);
                int j;  // position of x in the other endpoint's adjacency row
                if (kind == 0) j = from_a ? find_lds(L.cand + dA, dB, x) : find_lds(L.cand, dA, x);
                else j = from_a ? find_glb(adj_col + rb0, dB, x) : find_glb(adj_col + ra0, dA, x);
                if (from_a || j < 0) {  // a node of N(b) that is also in N(a) is emitted through N(a)
                    cn = from_a && j >= 0;
                    float other = 0.f;
                    if (cn) {
                        other = selfp[rb0 + j];
                    } else if (ppr_round_trip(sp[u], false) >= th_1) {  // otherwise it is dropped anyway
                        int idx;
                        if (from_a) idx = p1b_lds ? find_lds(L.p1 + IDX_P1_CAP, nPb, x) : find_glb(p1_col + pb0, nPb, x);
                        else idx = p1a_lds ? find_lds(L.p1, nPa, x) : find_glb(p1_col + pa0, nPa, x);
                        if (idx >= 0) other = p1_val[(from_a ? pb0 : pa0) + idx];
                    }
                    va = ppr_round_trip(from_a ? sp[u] : other, cn);
                    vb = ppr_round_trip(from_a ? other : sp[u], cn);
                    const float th = cn ? th_cn : th_1;
                    keep = (va >= th) && (vb >= th);
                }
            }
            sel_emit(kind, lane, i < len, keep, cn, from_a, x, va, vb, s, dA, slot0 + i, n_cn, n_l1, n_l2, stage_node,
                     stage_pa, stage_pb);
        }
        if (lane == 0) {
            if (n_cn) atomicAdd(&stage_cnt[4 * p + 0], n_cn);
            if (n_l1) atomicAdd(&stage_cnt[4 * p + 1], n_l1);
            if (n_l2) atomicAdd(&stage_cnt[4 * p + 2], n_l2);
        }
        // ---- 3. >1-hop nodes, once per pair (first item): same as select_nodes_kernel, adjacency test in LDS
        if (want_t0 && (kind == 0 || (kind == 1 && start == 0) || (kind == 2 && start == 0 && dA == 0))) {
            const int64_t *d = desc + p * DESC_I64;
            const int64_t ta0 = d[4], tb0 = d[5], nTa = d[10], nTb = d[11], a = d[12], b = d[13];
            const bool walk_a = nTa <= nTb;
            const int64_t w0 = walk_a ? ta0 : tb0, wn = walk_a ? nTa : nTb;
            const int64_t o0 = walk_a ? tb0 : ta0, o1 = o0 + (walk_a ? nTb : nTa);
            const bool use_lds = (kind == 0) && same_adj;
            int64_t xa0 = ra0, xa1 = ra0 + dA, xb0 = rb0, xb1 = rb0 + dB;
            const int32_t *xcol = adj_col;
            if (!same_adj) {
                xa0 = adjx_rowptr[a]; xa1 = adjx_rowptr[a + 1]; xb0 = adjx_rowptr[b]; xb1 = adjx_rowptr[b + 1];
                xcol = adjx_col;
            }
            const int64_t t0_base = s + dA + dB;
            int n_t0 = 0;
            // the other endpoint's T0 columns go to LDS in one coalesced burst (rows hold at most ~1/theta_n entries):
            // each walked node then costs one LDS search + one value load instead of a global binary search
            const int n_o = (int)(o1 - o0);
            const bool o_lds = n_o <= 2 * IDX_P1_CAP;
            if (o_lds && wn > 0) {
                LPF_WAVE_SYNC();  // phase 2 is done with the P1 columns
                for (int i = lane; i < n_o; i += 64) L.p1[i] = t0_col[o0 + i];
                LPF_WAVE_SYNC();
            }
            for (int64_t i0 = 0; i0 < wn; i0 += 64) {
                const int64_t i = i0 + lane;
                int32_t v = 0;
                bool keep = false;
                float sa = 0.f, sb = 0.f;
                if (i < wn) {
                    v = t0_col[w0 + i];
                    const float pw = t0_val[w0 + i];
                    const float sw = __fsub_rn(__fadd_rn(pw, 1.0f), 1.0f);
                    if (pw > 0.f && sw >= th_n) {
                        bool f;
                        float po;
                        if (o_lds) {
                            const int idx = find_lds(L.p1, n_o, v);
                            f = idx >= 0;
                            po = f ? t0_val[o0 + idx] : 0.0f;
                        } else {
                            po = csr_value(t0_col, t0_val, o0, o1, v, &f);
                        }
                        const float so = __fsub_rn(__fadd_rn(po, 1.0f), 1.0f);
                        if (f && po > 0.f && so >= th_n) {
                            const bool adjacent =
                                use_lds ? (find_lds(L.cand, dA, v) >= 0 || find_lds(L.cand + dA, dB, v) >= 0)
                                        : (csr_contains(xcol, xa0, xa1, v) || csr_contains(xcol, xb0, xb1, v));
                            if (!adjacent) {
                                keep = true;
                                sa = walk_a ? sw : so;
                                sb = walk_a ? so : sw;
                            }
                        }
                    }
                }
                const uint64_t m = __ballot(keep);
                if (keep) {
                    const int64_t dst = t0_base + n_t0 + lanes_below(m, lane);
                    stage_node[dst] = v;
                    stage_pa[dst] = sa;
                    stage_pb[dst] = sb;
                }
                n_t0 += __popcll(m);
            }
            if (lane == 0) stage_cnt[4 * p + 3] = n_t0;
        }
    }
}

}  // namespace

extern "C" int lpf_select_bound(int64_t bs, const int64_t *batch, int64_t batch_ld, const int64_t *adj_rowptr,
                                const int64_t *ppr_rowptr, const int64_t *t0_rowptr, int64_t *offs, int64_t *desc,
                                int64_t *scratch, void *stream) {
    LPF_REQUIRE(bs >= 0 && offs && scratch);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (bs == 0) {
        (void)hipMemsetAsync(offs, 0, 2 * sizeof(int64_t), s);
        return LPF_OK;
    }
    LPF_REQUIRE(batch && adj_rowptr && ppr_rowptr && desc && batch_ld >= bs);
    const int64_t nb = (bs + 255) / 256;
    hipLaunchKernelGGL(select_bound_kernel, dim3((unsigned)nb), dim3(256), 0, s, bs, batch, batch_ld, adj_rowptr,
                       ppr_rowptr, t0_rowptr, offs, desc, scratch, nb);
    hipLaunchKernelGGL(scan_blocks_kernel<2>, dim3((unsigned)nb), dim3(256), 0, s, bs, offs, bs + 1, scratch, nb);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_select_nodes(int64_t bs, int64_t item_capacity, const int64_t *offs, const int64_t *desc,
                                int32_t *items, const int32_t *adj_col, const float *adj_selfp,
                                const int64_t *adjx_rowptr, const int32_t *adjx_col, int32_t same_adj,
                                const int32_t *ppr_col,
                                const float *ppr_val, const int32_t *t0_col, const float *t0_val, float th_cn,
                                float th_1hop, float th_non1hop, int32_t *stage_node, float *stage_pa, float *stage_pb,
                                int32_t *stage_cnt, void *stream) {
    if (bs == 0) return LPF_OK;
    LPF_REQUIRE(bs > 0 && bs < (1ll << 31) && item_capacity >= bs && offs && desc && items && adj_col && ppr_col &&
                ppr_val && stage_node && stage_pa && stage_pb && stage_cnt && lpf_aligned16(items) &&
                lpf_aligned16(stage_cnt));
    LPF_REQUIRE(same_adj || (adjx_rowptr && adjx_col));
    const int want_t0 = t0_col != nullptr;
    LPF_REQUIRE(!want_t0 || t0_val);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t *item_off = offs + (bs + 1);
    hipLaunchKernelGGL(select_items_kernel, dim3((unsigned)((bs + 255) / 256)), dim3(256), 0, s, bs, desc, item_off,
                       reinterpret_cast<ItemRec *>(items), stage_cnt);
    // four wavefronts (= four items) per block, about one item per wavefront: uneven items are balanced by the dispatcher
    int64_t blocks = (item_capacity + 3) / 4;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    if (adj_selfp)  // indexed fast path: ppr_col / ppr_val (and the descriptors) describe the P1 rows; one wavefront
                    // per workgroup, so a long item (hub slice) does not hold the LDS of three finished neighbours
        hipLaunchKernelGGL(select_nodes_indexed_kernel<1>, dim3((unsigned)(item_capacity < (1 << 22) ? item_capacity
                                                                                                    : (1 << 22))),
                           dim3(64), 0, s, item_off + bs,
                           reinterpret_cast<const ItemRec *>(items), desc, adj_col, adj_selfp, adjx_rowptr, adjx_col,
                           (int)same_adj, ppr_col, ppr_val, t0_col, t0_val, want_t0, th_cn, th_1hop, th_non1hop, offs,
                           stage_node, stage_pa, stage_pb, stage_cnt);
    else
        hipLaunchKernelGGL(select_nodes_kernel, dim3((unsigned)blocks), dim3(256), 0, s, item_off + bs,
                           reinterpret_cast<const ItemRec *>(items), desc, adj_col, adjx_rowptr, adjx_col,
                           (int)same_adj, ppr_col, ppr_val, t0_col, t0_val, want_t0, th_cn, th_1hop, th_non1hop, offs,
                           stage_node, stage_pa, stage_pb, stage_cnt);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_select_scan(int64_t bs, const int32_t *stage_cnt, int64_t *type_ptr, float *counts_f, int64_t ldc,
                               int32_t want_t0, int64_t *scratch, void *stream) {
    LPF_REQUIRE(bs >= 0 && type_ptr && scratch);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (bs == 0) {
        (void)hipMemsetAsync(type_ptr, 0, 3 * sizeof(int64_t), s);
        return LPF_OK;
    }
    LPF_REQUIRE(stage_cnt && lpf_aligned16(stage_cnt) && (!counts_f || ldc >= (want_t0 ? 4 : 3)));
    const int64_t nb = (bs + 255) / 256;
    hipLaunchKernelGGL(select_counts_kernel, dim3((unsigned)nb), dim3(256), 0, s, bs, stage_cnt, type_ptr, counts_f,
                       ldc, (int)want_t0, scratch, nb);
    hipLaunchKernelGGL(scan_blocks_kernel<3>, dim3((unsigned)nb), dim3(256), 0, s, bs, type_ptr, bs + 1, scratch, nb);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_select_compact(int64_t bs, const int64_t *desc, const int64_t *offs, int32_t *stage_node,
                                  float *stage_pa, float *stage_pb, const int32_t *stage_cnt, const int64_t *type_ptr,
                                  int32_t *sel_pair, int32_t *sel_node, float *sel_pa, float *sel_pb, void *stream) {
    if (bs == 0) return LPF_OK;
    LPF_REQUIRE(bs > 0 && bs < (1ll << 31) && desc && offs && stage_node && stage_pa && stage_pb && stage_cnt &&
                type_ptr && sel_pair && sel_node && sel_pa && sel_pb);
    // one wavefront per workgroup: a pair with long runs does not hold the LDS / wave slots of three finished ones
    const int64_t blocks = bs < (1 << 22) ? bs : (1 << 22);
    hipLaunchKernelGGL(select_compact_kernel<1>, dim3((unsigned)blocks), dim3(64), 0, static_cast<hipStream_t>(stream), bs,
                       desc, offs, stage_node, stage_pa, stage_pb, stage_cnt, type_ptr, sel_pair, sel_node, sel_pa,
                       sel_pb);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}
