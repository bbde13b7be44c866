// PPR-thresholded node selection, GENERAL path: two launches, nothing read back by the host.
//
// Reference: compute_node_mask + get_ppr_vals + get_non_1hop_ppr (src/models/link_transformer.py:214-319, 434-481),
// eval mode.  This file serves a caller-supplied typing adjacency (the training loop's masked adjacency,
// src/train/train_model.py:40-59), for which no per-model index exists: both PPR values of a candidate are looked up
// in the raw PPR rows (binary searches), the >1-hop exclusion uses the unmasked adjacency.  The model's own adjacency
// goes through select3.hip (walk indexes).  lpf_select_export (reference layout) lives here too.
//
// The candidates of a batch form ONE flat index space: pair k owns the slots
//     [offs[k], offs[k+1])  =  N(a_k)  |  N(b_k)  |  the shorter of T0[a_k], T0[b_k]     (at least one slot per pair)
// and the space is cut into work items of S2_ITEM consecutive slots, whatever pairs they belong to -- a hub pair
// simply spans many items, a run of small pairs shares one.  Each slot is owned by one thread:
//   * plan kernel   per pair: descriptor (row starts / lengths), slot count; one chained scan (decoupled look-back
//                   over the 256-pair blocks) gives offs[] and the first pair of every item.  Node ids are range
//                   checked here (a bad id raises a sticky error bit instead of reading out of bounds).
//   * run kernel    persistent workgroups draw items from a ticket counter.  A thread types its candidate (binary
//                   search in the other endpoint's adjacency run -- in LDS when that run lies inside the item --,
//                   then the two PPR values from the raw rows / the other >1-hop row), applies the
//                   reference's fp32 round trip and thresholds op for op, and keeps the result in registers.  Kept
//                   entries are ranked per type inside the item (ballots), the item's three totals go through a
//                   second chained scan ordered by ticket, and the entries land at their FINAL position: per type
//                   one dense region ordered by (pair, candidate slot).  The thread that owns a pair's first slot also
//                   writes the pair's three segment starts, so type_ptr[t][k+1] - type_ptr[t][k] is the count
//                   feature of get_structure_cnts (:340-356).
// Placement is deterministic (it follows the flat slot order, not the schedule).  Inside a pair's one-hop segment
// the kept nodes of N(a) come before those of N(b) (flag bit 31 of the pair word); lpf_select_export merges the two
// runs by node id when a caller wants the reference's exact layout (compute_node_mask, attention weights).
#include "select_common.h"

// the reference's fp32 round trip must be evaluated op by op: no fused multiply-add in this file
#pragma clang fp contract(off)

namespace {

constexpr int S2_ITEM = LPF_SELECT_ITEM;            // candidate slots per work item
constexpr int S2_THREADS = 256;
constexpr int S2_ROUNDS = S2_ITEM / S2_THREADS;     // slots per thread
constexpr int S2_WAVES = S2_THREADS / 64;
constexpr uint32_t S2_FROM_B = 0x80000000u;
constexpr int S2_WCACHE = 48;                       // window pairs whose descriptors are cached in LDS
constexpr int S2_MIN_SLOTS = (S2_ITEM + S2_WCACHE - 2) / (S2_WCACHE - 1);  // => at most S2_WCACHE pairs per item

struct alignas(16) PairDesc {
    int64_t ra0, rb0;              // adjacency rows (the typing adjacency: adj_mask or the caller's override)
    int64_t pa0, pb0;              // value rows: P1 index rows (indexed path) or raw PPR rows (general path)
    int64_t ta0, tb0;              // >1-hop candidate rows (T0 index)
    int32_t dA, dB, nPa, nPb;
    int32_t nTa, nTb, a, b;
    int64_t xa0, xb0;              // rows of the UNMASKED adjacency (>1-hop exclusion, link_transformer.py:443)
    int32_t dxA, dxB, pad0, pad1;
    int64_t pad2, pad3;
};
static_assert(sizeof(PairDesc) == 128, "PairDesc is one 128-byte line");

// fl32((fl32(fl32(p*t)+t)-t)/t) for t in {1,2}: p*1, p*2, x/1 and x/2 are exact, only the add and subtract round
__device__ __forceinline__ float s2_round_trip(float p, bool two) {
    if (two) return 0.5f * __fsub_rn(__fadd_rn(p * 2.0f, 2.0f), 2.0f);
    return __fsub_rn(__fadd_rn(p, 1.0f), 1.0f);
}

// position of key in the sorted array a[0..n), or -1 (a in global memory or LDS).  Lower bound that remembers the
// value under the final `hi`: when the loop ends lo == hi, and a[hi] was loaded the last time hi moved (or hi never
// moved and lo == n), so no load is needed after the loop.  (hipcc 7.2 miscompiled the usual "if (lo < n &&
// a[lo] == key)" tail of two back-to-back searches in the divergent >1-hop branch: a register pair copy on the
// conditional-load path overwrote the result of the lanes that took it.)
__device__ __forceinline__ int s2_find(const int32_t *a, int n, int32_t key) {
    int lo = 0, hi = n;
    int32_t at_hi = ~key;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int32_t v = a[mid];
        if (v < key) {
            lo = mid + 1;
        } else {
            hi = mid;
            at_hi = v;
        }
    }
    return at_hi == key ? lo : -1;
}

// Lookup in a row of a BLOCKED index (lpformer_amd/graph.py BlockedIndex): rows padded to multiples of 16 entries
// (padding column INT32_MAX), entries stored as interleaved {column, value} pairs so that a 16-entry block is one
// aligned 128-byte line carrying the values with the columns, skip[b] = last column of block b.  Two memory round
// trips instead of a binary search's eight plus one for the value: the row's skip entries (<= 16 of them in aligned
// groups of four; longer rows are narrowed with a few scalar probes first) give the block, the block gives the value.
__device__ __forceinline__ bool s2_lookup_blocked(const int2 *__restrict__ cv, const int32_t *__restrict__ skip,
                                                  int64_t row0, int n_real, int32_t key, float &val) {
    if (n_real <= 0) return false;
    const int nb = (n_real + 15) >> 4;
    int j = 0;
    if (nb > 1) {
        const int64_t sb = row0 >> 4;
        int lo = 0, hi = nb;  // the first block whose last column is >= key lies in [lo, hi] (hi: none below nb)
        while (hi - lo > 16) {
            const int mid = (lo + hi) >> 1;
            if (skip[sb + mid] < key) lo = mid + 1; else hi = mid;
        }
        const int64_t first = sb + lo, g0 = first & ~3ll;
        const int off = (int)(first - g0), n_in = hi - lo;
        int cnt = 0;
        // five aligned groups cover any 16 entries; all five are requested together (reads past the row's entries
        // stay inside the array -- it ends with spare entries -- and are masked out)
        int4 v[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) v[q] = *reinterpret_cast<const int4 *>(skip + g0 + 4 * q);
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int b = 4 * q - off;  // index of v[q].x relative to `first`
            cnt += (b + 0 >= 0 && b + 0 < n_in && v[q].x < key) + (b + 1 >= 0 && b + 1 < n_in && v[q].y < key) +
                   (b + 2 >= 0 && b + 2 < n_in && v[q].z < key) + (b + 3 >= 0 && b + 3 < n_in && v[q].w < key);
        }
        j = lo + cnt;
        if (j >= nb) return false;
    }
    const int4 *blk = reinterpret_cast<const int4 *>(cv + row0 + 16 * j);  // {col, val, col, val} x 8
    bool eq = false;
    int bits = 0;
    int4 bv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) bv[q] = blk[q];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (bv[q].x == key) { eq = true; bits = bv[q].y; }
        if (bv[q].z == key) { eq = true; bits = bv[q].w; }
    }
    val = __int_as_float(bits);
    return eq;
}

// Lookup in a row of a HASHED index (lpformer_amd/graph.py HashedIndex): the row owns n_buckets buckets of 8
// {column, value} entries (64 aligned bytes), the key's bucket follows from its hash, and the builder made sure no
// bucket overflows -- one memory round trip and no search (the blocked layout above needs the row's skip entries
// first: two dependent round trips and about twice the instructions).  Buckets of 8 rather than 16: every 16-byte load
// of a lane is a separate line look-up in the vector L1, and half of the slots of a batch come through here.
constexpr int S2_HASH_BUCKET = 8;
__device__ __forceinline__ bool s2_lookup_hashed(const int2 *__restrict__ cv, int64_t row0, int n_buckets, int32_t key,
                                                 float &val) {
    if (n_buckets <= 0) return false;
    const uint32_t b = __umulhi((uint32_t)key * 2654435761u, (uint32_t)n_buckets);
    const int4 *blk = reinterpret_cast<const int4 *>(cv + row0 + S2_HASH_BUCKET * (int64_t)b);  // {col, val, col, val} x 4
    bool eq = false;
    int bits = 0;
    int4 bv[S2_HASH_BUCKET / 2];
#pragma unroll
    for (int q = 0; q < S2_HASH_BUCKET / 2; ++q) bv[q] = blk[q];
#pragma unroll
    for (int q = 0; q < S2_HASH_BUCKET / 2; ++q) {
        if (bv[q].x == key) { eq = true; bits = bv[q].y; }
        if (bv[q].z == key) { eq = true; bits = bv[q].w; }
    }
    val = __int_as_float(bits);
    return eq;
}

// ------------------------------------------------------------------------------------------- plan
__global__ __launch_bounds__(S2_THREADS) void select_plan_kernel(
    int64_t bs, const int64_t *__restrict__ batch, int64_t batch_ld, int64_t n_nodes,
    const int64_t *__restrict__ adj_rowptr, const int64_t *__restrict__ val_rowptr,
    const int64_t *__restrict__ t0_rowptr, const int64_t *__restrict__ adjx_rowptr,
    const int32_t *__restrict__ val_len, const int32_t *__restrict__ t0_len, PairDesc *__restrict__ desc,
    int64_t *__restrict__ offs, int32_t *__restrict__ item_pair, int64_t item_cap, int64_t *__restrict__ ctl,
    uint64_t *__restrict__ plan_lb) {
    __shared__ int64_t wtot[S2_WAVES];
    __shared__ int64_t s_base, s_blk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Logical block index = order of arrival: a block then only ever waits for blocks that are already running,
    // whatever the dispatch order.  ctl[7] only ever grows and every launch on this control block draws exactly
    // gridDim.x tickets (one batch size per control block), so ticket / gridDim.x is the launch number: it tags the
    // chained-scan words of this launch (plan and run kernel), nothing is reset and nothing comes from the host --
    // the same two launches can be replayed from a captured graph.
    if (tid == 0) s_blk = (int64_t)atomicAdd(reinterpret_cast<unsigned long long *>(ctl + 7), 1ull);
    __syncthreads();
    const int64_t blk = s_blk % (int64_t)gridDim.x;
    const uint32_t epoch = (uint32_t)((s_blk / (int64_t)gridDim.x) & ((1 << 22) - 1));
    const int64_t k = blk * S2_THREADS + tid;
    int64_t ub = 0;
    if (k < bs) {
        const int64_t a = batch[k], b = batch[batch_ld + k];
        PairDesc d;
        d.pad0 = d.pad1 = 0; d.pad2 = d.pad3 = 0;
        if ((uint64_t)a >= (uint64_t)n_nodes || (uint64_t)b >= (uint64_t)n_nodes) {
            atomicOr(reinterpret_cast<unsigned long long *>(ctl + 3), (unsigned long long)LPF_SELECT_ERR_NODE_RANGE);
            d.ra0 = d.rb0 = d.pa0 = d.pb0 = d.ta0 = d.tb0 = d.xa0 = d.xb0 = 0;
            d.dA = d.dB = d.nPa = d.nPb = d.nTa = d.nTb = d.dxA = d.dxB = 0;
            d.a = d.b = 0;
        } else {
            d.ra0 = adj_rowptr[a]; d.dA = (int32_t)(adj_rowptr[a + 1] - d.ra0);
            d.rb0 = adj_rowptr[b]; d.dB = (int32_t)(adj_rowptr[b + 1] - d.rb0);
            // (blocked indexes carry their real row lengths separately: the row pointers count padded entries)
            d.pa0 = val_rowptr[a]; d.nPa = val_len ? val_len[a] : (int32_t)(val_rowptr[a + 1] - d.pa0);
            d.pb0 = val_rowptr[b]; d.nPb = val_len ? val_len[b] : (int32_t)(val_rowptr[b + 1] - d.pb0);
            d.ta0 = d.tb0 = 0; d.nTa = d.nTb = 0;
            if (t0_rowptr) {
                d.ta0 = t0_rowptr[a]; d.nTa = t0_len ? t0_len[a] : (int32_t)(t0_rowptr[a + 1] - d.ta0);
                d.tb0 = t0_rowptr[b]; d.nTb = t0_len ? t0_len[b] : (int32_t)(t0_rowptr[b + 1] - d.tb0);
            }
            d.xa0 = d.ra0; d.xb0 = d.rb0; d.dxA = d.dA; d.dxB = d.dB;
            if (adjx_rowptr) {
                d.xa0 = adjx_rowptr[a]; d.dxA = (int32_t)(adjx_rowptr[a + 1] - d.xa0);
                d.xb0 = adjx_rowptr[b]; d.dxB = (int32_t)(adjx_rowptr[b + 1] - d.xb0);
            }
            d.a = (int32_t)a; d.b = (int32_t)b;
        }
        ub = (int64_t)d.dA + d.dB + (d.nTa < d.nTb ? d.nTa : d.nTb);
        // Every pair owns at least S2_MIN_SLOTS slots (the surplus ones hold no candidate): the owner of the first
        // writes the pair's segment starts, and an item of S2_ITEM slots then never spans more than S2_WCACHE pairs,
        // so all its descriptors sit in LDS -- items made of hundreds of one-neighbour pairs were the slow ones, and
        // in a chained scan everything behind a slow item waits for it.
        if (ub < S2_MIN_SLOTS) ub = S2_MIN_SLOTS;
        __builtin_memcpy(desc + k, &d, sizeof(PairDesc));  // (no type punning of the local struct)
    }
    // inclusive scan of ub inside the block, then the block's base through the chained scan
    int64_t x = ub;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const int64_t y = __shfl_up(x, dlt, 64);
        if (lane >= dlt) x += y;
    }
    if (lane == 63) wtot[wave] = x;
    __syncthreads();
    int64_t pre = 0, btot = 0;
#pragma unroll
    for (int w = 0; w < S2_WAVES; ++w) {
        if (w < wave) pre += wtot[w];
        btot += wtot[w];
    }
    if (wave == 0) {
        const int64_t base = (int64_t)lb_exclusive(plan_lb, blk, epoch, (uint64_t)btot, lane);
        if (lane == 0) s_base = base;
    }
    __syncthreads();
    const int64_t o = s_base + pre + x - ub;  // exclusive
    if (k < bs) {
        offs[k] = o;
        // first pair of every item that starts inside this pair's slots
        for (int64_t it = (o + S2_ITEM - 1) / S2_ITEM; it * S2_ITEM < o + ub; ++it) {
            if (it < item_cap) item_pair[it] = (int32_t)k;
            else atomicOr(reinterpret_cast<unsigned long long *>(ctl + 3), (unsigned long long)LPF_SELECT_ERR_ITEM_CAP);
        }
        if (k == bs - 1) {
            const int64_t total = o + ub;
            offs[bs] = total;
            ctl[0] = total;
            ctl[1] = (total + S2_ITEM - 1) / S2_ITEM;
            ctl[2] = 0;  // ticket counter of the run kernel (stream order: the previous run kernel has finished)
            ctl[8] = epoch;
        }
    }
}

// ------------------------------------------------------------------------------------------- run
struct RunArgs {
    int64_t bs;
    const PairDesc *desc;
    const int64_t *offs;
    const int32_t *item_pair;
    int64_t item_cap;
    int64_t *ctl;
    uint64_t *run_lb;  // [3][item_cap]
    const int32_t *adj_col;    // typing adjacency
    const float *selfp;        // aligned with adj_col (indexed path) or NULL (general path)
    const int32_t *adjx_col;   // unmasked adjacency (== adj_col when the typing adjacency is the unmasked one)
    const int32_t *val_col;    // general path: raw PPR rows (plain sorted CSR)
    const float *val_val;
    const int2 *val_cv;        // indexed path: P1 index, hashed buckets of 16 {column, value} entries
    const int2 *t0_cv;         // T0 index, blocked layout (both paths)
    const int32_t *t0_skip;
    float th_cn, th_1, th_n;
    int mode_cn;   // mask mode "cn" (link_transformer.py:232-247): common neighbours only, round trip with t = 1
    int32_t *type_ptr;         // [3][bs+1]
    int4 *entries;             // [3][ent_cap]
    int64_t ent_cap;
};

constexpr int S2_PARK = 256;  // kept entries of one item that can wait in LDS for the item's place in the output
struct RunLds {
    int4 pk[S2_PARK];           // parked entries: type 0 | type 1 | type 2, each in slot order
    int32_t loc[S2_ITEM + 1];   // slot (relative to the item) at which pair pf + i starts
    int32_t cand[S2_ITEM];      // node id in every slot: the adjacency runs double as searchable rows
    float va[S2_ITEM], vb[S2_ITEM];  // emitted values of the kept slots
    int16_t win[S2_ITEM];       // window index of the slot's pair
    uint8_t code[S2_ITEM];      // 0 dropped, 1 cn, 2 one-hop, 3 >1-hop; bit 2: the one-hop node comes from N(b)
    PairDesc dsc[S2_WCACHE];    // descriptors of the first window pairs (one coalesced copy per item)
    int32_t cnt[S2_ROUNDS * S2_WAVES][4];
    int64_t base[3];
    int64_t ticket;
    int32_t n_pairs;
    int32_t run[3];             // kept entries per type of the item being built
    int32_t n_t0;               // >1-hop candidates of the item that passed their own threshold ...
    int16_t t0list[S2_ITEM];    // ... and their slots: typed densely by s2_type_t0 (3 % of the slots, but half of the
                                // wavefronts hold some and would otherwise run both typing paths)
    // the parked item: what its deferred write-out needs once cand / code / loc belong to the next item
    int64_t p_item, p_pf;
    int32_t p_run[3], p_np, p_last, p_live;
    int16_t ps[S2_WCACHE][4];   // {rank of the pair's first slot per type, pair starts in the item}
};
static_assert(sizeof(RunLds) <= 32 * 1024, "five workgroups per CU");

// Descriptor fields of window pair w in registers: from the LDS cache (first S2_WCACHE pairs of the window) or from
// global memory.  Only what a caller uses is actually loaded.
struct PairLite {
    int64_t ra0, rb0, pa0, pb0, ta0, tb0, xa0, xb0;
    int dA, dB, nPa, nPb, nTa, nTb, dxA, dxB;
};
__device__ __forceinline__ PairLite s2_pair(const RunArgs &A, const RunLds &L, int64_t pf, int w) {
    PairLite p;
#define S2_COPY(src)                                                                                       \
    p.ra0 = (src).ra0; p.rb0 = (src).rb0; p.pa0 = (src).pa0; p.pb0 = (src).pb0; p.ta0 = (src).ta0;         \
    p.tb0 = (src).tb0; p.xa0 = (src).xa0; p.xb0 = (src).xb0; p.dA = (src).dA; p.dB = (src).dB;             \
    p.nPa = (src).nPa; p.nPb = (src).nPb; p.nTa = (src).nTa; p.nTb = (src).nTb; p.dxA = (src).dxA;         \
    p.dxB = (src).dxB
    // (an item never spans more than S2_WCACHE pairs -- S2_MIN_SLOTS --, so the window's descriptors are all in LDS)
    (void)A; (void)pf;
    S2_COPY(L.dsc[w]);
#undef S2_COPY
    return p;
}

// slot index inside pair (window index w) of item slot l
__device__ __forceinline__ int64_t s2_slot_in_pair(const RunArgs &A, const RunLds &L, int64_t pf, int64_t c0, int w,
                                                   int l) {
    const int s0 = L.loc[w];
    return s0 > -(1 << 30) ? (int64_t)(l - s0) : (c0 + l) - A.offs[pf + w];  // (clamped only ~2^30 slots back)
}

// Types one ADJACENCY candidate slot (phase B of select_run_kernel).  own: P[own endpoint, node] from the aligned
// self-PPR array (indexed path).  Returns true -- and touches nothing -- for a >1-hop slot: those are typed by
// s2_type_t0 from a compacted list.
template <bool INDEXED>
__device__ __forceinline__ bool s2_type_slot(const RunArgs &A, const RunLds &L, int64_t pf, int64_t c0, int n_here,
                                             int l, float own_in, int &code, float &va, float &vb, bool &fromb) {
    const int32_t x = L.cand[l];
    const int w = L.win[l];
    const PairLite d = s2_pair(A, L, pf, w);
    const int dA = d.dA, dB = d.dB;
    const int i = (int)s2_slot_in_pair(A, L, pf, c0, w, l);
    const int s0 = L.loc[w];  // slot of the pair's first candidate (may lie before the item)
    if (i >= dA + dB) return true;
    const bool from_a = i < dA;
    // position of x in the OTHER endpoint's adjacency run: in LDS when that run lies inside the item
    const int o_lo = from_a ? s0 + dA : s0, o_n = from_a ? dB : dA;
    int j;
    if (s0 > -(1 << 30) && o_lo >= 0 && o_lo + o_n <= n_here) j = s2_find(L.cand + o_lo, o_n, x);
    else j = s2_find(A.adj_col + (from_a ? d.rb0 : d.ra0), o_n, x);
    if (!from_a && j >= 0) return false;  // a node of N(b) that is also in N(a) is emitted through N(a)
    const bool cn = from_a && j >= 0;
    float own, other = 0.f;
    if (INDEXED) {
        own = own_in;
        if (cn) {
            other = A.selfp[d.rb0 + j];
        } else if (s2_round_trip(own, false) >= A.th_1) {  // otherwise it is dropped anyway
            float v;
            if (s2_lookup_hashed(A.val_cv, from_a ? d.pb0 : d.pa0, from_a ? d.nPb : d.nPa, x, v)) other = v;
        }
    } else {  // general path: both values from the raw PPR rows (absent entries read as 0)
        const int64_t m0 = from_a ? d.pa0 : d.pb0, v0 = from_a ? d.pb0 : d.pa0;
        const int im = s2_find(A.val_col + m0, from_a ? d.nPa : d.nPb, x);
        own = im >= 0 ? A.val_val[m0 + im] : 0.f;
        if (cn || s2_round_trip(own, false) >= A.th_1) {
            const int idx = s2_find(A.val_col + v0, from_a ? d.nPb : d.nPa, x);
            if (idx >= 0) other = A.val_val[v0 + idx];
        }
    }
    const bool two = cn && !A.mode_cn;   // (t = 2 for a common neighbour; mode "cn": t = 1, as in select3.hip)
    const float pa = s2_round_trip(from_a ? own : other, two);
    const float pb = s2_round_trip(from_a ? other : own, two);
    const float th = cn ? A.th_cn : A.th_1;
    const bool keep = pa >= th && pb >= th && (cn || !A.mode_cn);
    code = keep ? (cn ? 1 : 2) : 0;
    va = pa; vb = pb; fromb = !from_a;
    return false;
}

// Types one >1-hop candidate slot whose own value pw already passed (pw > 0, round trip >= theta_n): stored in the
// other T0 row too, adjacent to neither endpoint in the UNMASKED adjacency; the endpoints themselves may be selected
// (link_transformer.py:438-443).
template <bool INDEXED>
__device__ __forceinline__ void s2_type_t0(const RunArgs &A, const RunLds &L, int64_t pf, int n_here, int l, float pw,
                                           int &code, float &va, float &vb) {
    const int32_t x = L.cand[l];
    const int w = L.win[l];
    const PairLite d = s2_pair(A, L, pf, w);
    const int dA = d.dA, dB = d.dB;
    const int s0 = L.loc[w];
    const bool walk_a = d.nTa <= d.nTb;
    const float sw = __fsub_rn(__fadd_rn(pw, 1.0f), 1.0f);
    float so = 0.f, po;
    bool ok = s2_lookup_blocked(A.t0_cv, A.t0_skip, walk_a ? d.tb0 : d.ta0, walk_a ? d.nTb : d.nTa, x, po);
    if (ok) {
        so = __fsub_rn(__fadd_rn(po, 1.0f), 1.0f);
        ok = po > 0.f && so >= A.th_n;
    }
    if (ok) {
        int ja, jb;
        if (INDEXED && s0 >= 0 && s0 + dA + dB <= n_here) {  // both adjacency runs are in LDS
            ja = s2_find(L.cand + s0, dA, x);
            jb = s2_find(L.cand + s0 + dA, dB, x);
        } else {
            ja = s2_find(A.adjx_col + d.xa0, d.dxA, x);
            jb = s2_find(A.adjx_col + d.xb0, d.dxB, x);
        }
        ok = (ja & jb) < 0;  // both searches came back -1
    }
    code = ok ? 3 : 0;
    va = walk_a ? sw : so;
    vb = walk_a ? so : sw;
}

// Entries [p_lo, p_lo + S2_PARK) of the item described by L.p_* (type 0 | type 1 | type 2 order) from L.pk to their
// final place; with `heads`, also the segment starts of the pairs that begin in the item and, from the last item, the
// totals.  L.base holds the item's place per type.
__device__ __forceinline__ void s2_write_out(const RunArgs &A, RunLds &L, int64_t bs, int tid, int p_lo, bool heads) {
    const int r0 = L.p_run[0], r1 = L.p_run[1], r2 = L.p_run[2];
    const int64_t b0 = L.base[0], b1 = L.base[1], b2 = L.base[2];
    const int n = r0 + r1 + r2 - p_lo < S2_PARK ? r0 + r1 + r2 - p_lo : S2_PARK;
    for (int i = tid; i < n; i += S2_THREADS) {
        const int g = p_lo + i;
        const int t = g < r0 ? 0 : (g < r0 + r1 ? 1 : 2);
        const int64_t dst = t == 0 ? b0 + g : (t == 1 ? b1 + (g - r0) : b2 + (g - r0 - r1));
        if (dst < A.ent_cap) {
            A.entries[(int64_t)t * A.ent_cap + dst] = L.pk[i];
        } else {
            atomicOr(reinterpret_cast<unsigned long long *>(A.ctl + 3), (unsigned long long)LPF_SELECT_ERR_ENTRY_CAP);
        }
    }
    if (!heads) return;
    if (tid < L.p_np && L.ps[tid][3]) {
        const int64_t p = L.p_pf + tid;
        A.type_ptr[p] = (int32_t)(b0 + L.ps[tid][0]);
        A.type_ptr[(bs + 1) + p] = (int32_t)(b1 + L.ps[tid][1]);
        A.type_ptr[2 * (bs + 1) + p] = (int32_t)(b2 + L.ps[tid][2]);
    }
    if (tid == 0 && L.p_last) {
        const int64_t t0 = b0 + r0, t1 = b1 + r1, t2 = b2 + r2;
        A.type_ptr[bs] = (int32_t)t0;
        A.type_ptr[(bs + 1) + bs] = (int32_t)t1;
        A.type_ptr[2 * (bs + 1) + bs] = (int32_t)t2;
        A.ctl[4] = t0; A.ctl[5] = t1; A.ctl[6] = t2;
    }
}

// Write-out of the parked item, if there is one (whole workgroup; barriers at both ends).
__device__ __forceinline__ void s2_finish_parked(const RunArgs &A, RunLds &L, uint32_t epoch, int64_t bs, int lane,
                                                 int wave, int tid) {
    __syncthreads();
    if (L.p_live) {
        if (wave < 3) {
            const int64_t base = (int64_t)lb_lookback(A.run_lb + (int64_t)wave * A.item_cap, L.p_item, epoch,
                                                      (uint64_t)L.p_run[wave], lane);
            if (lane == 0) L.base[wave] = base;
        }
        __syncthreads();
        s2_write_out(A, L, bs, tid, 0, true);
    }
    __syncthreads();
}

// (five workgroups per CU = 96 registers.  Forcing six -- 80 registers, 32 bytes of scratch -- measured slower, and
// so did an index lookup shared by eight lanes per slot, one 128-byte line per instruction and group instead of one per
// lane: 177 us against 133 us, the extra LDS traffic and lane shuffling cost more than the L1 line look-ups saved.)
template <bool INDEXED>
__global__ __launch_bounds__(S2_THREADS, 5) void select_run_kernel(const RunArgs A) {
    __shared__ RunLds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const int64_t total = A.ctl[0];
    int64_t n_items = A.ctl[1];
    if (n_items > A.item_cap) n_items = A.item_cap;
    const int64_t bs = A.bs;
    const uint32_t epoch = (uint32_t)A.ctl[8];  // launch number, written by the plan kernel
    if (tid == 0) L.p_live = 0;

    while (true) {
        __syncthreads();  // the previous item's LDS image is no longer needed
        if (tid == 0) {
            L.ticket = (int64_t)atomicAdd(reinterpret_cast<unsigned long long *>(A.ctl + 2), 1ull);
        }
        __syncthreads();
        const int64_t it = L.ticket;
        if (it >= n_items) break;
        const int64_t c0 = it * S2_ITEM;
        const int n_here = (int)((total - c0) < S2_ITEM ? (total - c0) : S2_ITEM);
        const int64_t pf = A.item_pair[it];

        // ---- pair window: loc[i] = offs[pf + i] - c0 clamped to [-2^30, n_here]; pairs past the item read n_here.
        //      (every pair owns >= S2_MIN_SLOTS slots, so the window holds at most S2_WCACHE pairs)
        if (tid < 64) {
            int64_t v = n_here;
            if (pf + tid <= bs) v = A.offs[pf + tid] - c0;
            if (v > n_here) v = n_here;
            if (v < -(1 << 30)) v = -(1 << 30);
            L.loc[tid] = (int32_t)v;
            const int cntp = __popcll(__ballot(v < n_here));
            if (tid == 0) { L.n_pairs = cntp; L.n_t0 = 0; }
        }
        __syncthreads();
        const int np = L.n_pairs;  // pairs with at least one slot in this item (>= 1)
        {   // descriptors of the first window pairs -> LDS (one coalesced copy; every slot reads them several times)
            const int n4 = (np < S2_WCACHE ? np : S2_WCACHE) * 8;
            const int4 *src = reinterpret_cast<const int4 *>(A.desc + pf);
            int4 *dst = reinterpret_cast<int4 *>(L.dsc);
            for (int i = tid; i < n4; i += S2_THREADS) dst[i] = src[i];
        }
        __syncthreads();

        // ---- phase A: every thread identifies its slots and loads the candidate node (+ its own PPR value)
        // Per-slot state lives in LDS (cand / meta) and the rounds are real loops: four copies of the typing code in
        // one kernel body are what hipcc 7.2 miscompiled (see s2_find), and the per-round register arrays cost
        // occupancy.
        float cown[S2_ROUNDS];     // P[own endpoint, node] (indexed path) / T0 value (>1-hop slots)
#pragma unroll
        for (int r = 0; r < S2_ROUNDS; ++r) {
            const int l = tid + S2_THREADS * r;
            cown[r] = 0.f;
            bool want_t0 = false;
            if (l < n_here) {
                int lo = 0, hi = np;  // last window pair with loc <= l
                while (lo + 1 < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (L.loc[mid] <= l) lo = mid; else hi = mid;
                }
                const int w = lo;
                const PairLite d = s2_pair(A, L, pf, w);
                const int64_t i64 = s2_slot_in_pair(A, L, pf, c0, w, l);
                const int dA = d.dA, dB = d.dB;
                int32_t x = -1;
                if (i64 < dA) {
                    const int64_t e = d.ra0 + i64;
                    x = A.adj_col[e];
                    if (INDEXED) cown[r] = A.selfp[e];
                } else if (i64 < (int64_t)dA + dB) {
                    const int64_t e = d.rb0 + (i64 - dA);
                    x = A.adj_col[e];
                    if (INDEXED) cown[r] = A.selfp[e];
                } else {
                    const int nW = d.nTa < d.nTb ? d.nTa : d.nTb;
                    const int64_t wi = i64 - dA - dB;
                    L.code[l] = 0;  // (phase B leaves >1-hop slots alone)
                    if (wi < nW) {
                        const int2 cv = A.t0_cv[(d.nTa <= d.nTb ? d.ta0 : d.tb0) + wi];
                        x = cv.x;
                        const float pw = __int_as_float(cv.y);
                        want_t0 = pw > 0.f && __fsub_rn(__fadd_rn(pw, 1.0f), 1.0f) >= A.th_n;
                        L.va[l] = pw;
                    }
                }
                L.cand[l] = x;
                L.win[l] = (int16_t)w;
            }
            const uint64_t bt = __ballot(want_t0);
            if (bt) {  // (uniform) the slots whose own T0 value passes go on the list of the >1-hop typing pass
                int base = 0;
                if (lane == 0) base = atomicAdd(&L.n_t0, __popcll(bt));
                base = __builtin_amdgcn_readfirstlane(base);
                if (want_t0) L.t0list[base + __popcll(bt & lt_mask)] = (int16_t)l;
            }
        }
        __syncthreads();

        // ---- >1-hop candidates, densely from the list
        for (int k = tid; k < L.n_t0; k += S2_THREADS) {
            const int l = L.t0list[k];
            int code;
            float va, vb;
            s2_type_t0<INDEXED>(A, L, pf, n_here, l, L.va[l], code, va, vb);
            L.code[l] = (uint8_t)code;
            L.va[l] = va;
            L.vb[l] = vb;
        }
        __syncthreads();

        // ---- phase B: type, look up the other endpoint's value, round trip, thresholds
#pragma unroll 1
        for (int r = 0; r < S2_ROUNDS; ++r) {
            const int l = tid + S2_THREADS * r;
            int code = 0;  // 0 dropped, 1 common neighbour, 2 one-hop, 3 >1-hop
            float va = 0.f, vb = 0.f;
            bool fromb = false;
            const float ownv = r == 0 ? cown[0] : (r == 1 ? cown[1] : (r == 2 ? cown[2] : cown[3]));
            bool is_t0 = false;
            if (l < n_here && L.cand[l] >= 0)
                is_t0 = s2_type_slot<INDEXED>(A, L, pf, c0, n_here, l, ownv, code, va, vb, fromb);
            if (is_t0) {
                code = L.code[l] & 3;  // typed above
            } else {
                L.code[l] = (uint8_t)(code | (fromb ? 4 : 0));
                L.va[l] = va;
                L.vb[l] = vb;
            }
            const uint64_t b0 = __ballot(code == 1), b1 = __ballot(code == 2), b2 = __ballot(code == 3);
            if (lane == 0) {
                int32_t *c = L.cnt[r * S2_WAVES + wave];
                c[0] = __popcll(b0); c[1] = __popcll(b1); c[2] = __popcll(b2);
            }
        }
        __syncthreads();

        // ---- phase C: ranks inside the item; the item's totals go out to the chained scan at once
        if (wave < 3) {  // wavefront t: exclusive scan of type t over the (round, wave) groups in slot order
            constexpr int NG = S2_ROUNDS * S2_WAVES;
            const int v = lane < NG ? L.cnt[lane][wave] : 0;
            int x = v;
#pragma unroll
            for (int dlt = 1; dlt < NG; dlt <<= 1) {
                const int y = __shfl_up(x, dlt, 64);
                if (lane >= dlt) x += y;
            }
            if (lane < NG) L.cnt[lane][wave] = x - v;
            const int run = __shfl(x, NG - 1, 64);
            if (lane == 0) L.run[wave] = run;
            lb_publish(A.run_lb + (int64_t)wave * A.item_cap, it, epoch, (uint64_t)run, lane);
        }
        // ---- phase D, deferred: the item's place in the output depends on every earlier item, and the slowest of the
        //      ~1300 in flight decides when that is known.  So the kept entries are parked in LDS, the workgroup goes
        //      on to type the next item, and only then asks for the parked item's place (by then an answer that
        //      needs no waiting) and writes it out.
        s2_finish_parked(A, L, epoch, bs, lane, wave, tid);  // (starts and ends with a barrier)
        const int run0 = L.run[0], run1 = L.run[1], run2 = L.run[2];
        const int n_kept = run0 + run1 + run2;
        const bool now = n_kept > S2_PARK;  // (does not fit: wait for the place here and write in several passes)
        if (tid == 0) {
            L.p_item = it; L.p_pf = pf; L.p_np = np; L.p_last = (c0 + n_here == total);
            L.p_run[0] = run0; L.p_run[1] = run1; L.p_run[2] = run2;
            L.p_live = now ? 0 : 1;
        }
        if (tid < S2_WCACHE) L.ps[tid][3] = 0;
        if (now && wave < 3) {
            const int64_t base = (int64_t)lb_lookback(A.run_lb + (int64_t)wave * A.item_cap, it, epoch, (uint64_t)L.run[wave], lane);
            if (lane == 0) L.base[wave] = base;
        }
        __syncthreads();
        for (int pass = 0; pass * S2_PARK < (n_kept > 0 ? n_kept : 1); ++pass) {
            const int p_lo = pass * S2_PARK;
#pragma unroll 1
            for (int r = 0; r < S2_ROUNDS; ++r) {
                const int l = tid + S2_THREADS * r;
                const bool live = l < n_here;
                const int cf = live ? L.code[l] : 0;
                const int code = cf & 3;
                const uint64_t b0 = __ballot(code == 1), b1 = __ballot(code == 2), b2 = __ballot(code == 3);
                if (!live) continue;
                const int32_t *gp = L.cnt[r * S2_WAVES + wave];
                const int k0 = gp[0] + __popcll(b0 & lt_mask);
                const int k1 = gp[1] + __popcll(b1 & lt_mask);
                const int k2 = gp[2] + __popcll(b2 & lt_mask);
                const int w = L.win[l];
                if (code) {
                    const int pos = (code == 1 ? k0 : (code == 2 ? run0 + k1 : run0 + run1 + k2)) - p_lo;
                    if (pos >= 0 && pos < S2_PARK)
                        L.pk[pos] = make_int4((int32_t)((uint32_t)(pf + w) | ((cf & 4) ? S2_FROM_B : 0u)), L.cand[l],
                                              __float_as_int(L.va[l]), __float_as_int(L.vb[l]));
                }
                if (pass == 0 && l == L.loc[w]) {  // first slot of the pair: its three segment starts
                    L.ps[w][0] = (int16_t)k0; L.ps[w][1] = (int16_t)k1; L.ps[w][2] = (int16_t)k2; L.ps[w][3] = 1;
                }
            }
            if (!now) break;  // parked: written out after the next item has been typed (or at the end)
            __syncthreads();
            s2_write_out(A, L, bs, tid, p_lo, pass == 0);
            __syncthreads();
        }
    }
    s2_finish_parked(A, L, epoch, bs, lane, wave, tid);
}

// ------------------------------------------------------------------------------------------- export
// Reference layout for callers that want it (compute_node_mask, attention weights, the module-by-module path):
// all CN entries sorted by (pair, node), then all 1-hop, then all >1-hop (link_transformer.py:161-162), separate
// arrays, int64 segment pointers, float count features.  One wavefront per pair.
__global__ __launch_bounds__(256) void select_export_kernel(
    int64_t bs, const int32_t *__restrict__ type_ptr, const int4 *__restrict__ entries, int64_t ent_cap,
    int64_t *__restrict__ type_ptr64, float *__restrict__ counts_f, int64_t ldc, int n_counts,
    int32_t *__restrict__ sel_pair, int32_t *__restrict__ sel_node, float *__restrict__ sel_pa,
    float *__restrict__ sel_pb) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    int64_t tot[3], obase[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        tot[t] = type_ptr[(int64_t)t * (bs + 1) + bs];
        if (tot[t] > ent_cap) tot[t] = ent_cap;  // (overflow: the sticky error bit is set; stay inside the arrays)
    }
    obase[0] = 0; obase[1] = tot[0]; obase[2] = tot[0] + tot[1];
    for (int64_t p = wave_id; p <= bs; p += n_waves) {
        if (p == bs) {
            if (lane < 3) type_ptr64[(int64_t)lane * (bs + 1) + bs] = tot[lane];
            continue;
        }
        int64_t lo[3], n[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            lo[t] = type_ptr[(int64_t)t * (bs + 1) + p];
            int64_t hi = type_ptr[(int64_t)t * (bs + 1) + p + 1];
            if (lo[t] > tot[t]) lo[t] = tot[t];
            if (hi > tot[t]) hi = tot[t];
            n[t] = hi - lo[t];
        }
        if (lane < 3) type_ptr64[(int64_t)lane * (bs + 1) + p] = lane == 0 ? lo[0] : (lane == 1 ? lo[1] : lo[2]);
        if (counts_f && lane == 0) {
            float *c = counts_f + p * ldc;
            c[0] = (float)n[0];
            if (n_counts == 4) {          // get_structure_cnts (:340-356): n_cn, n_1hop, n_non1hop, n_cn + n_1hop
                c[1] = (float)n[1];
                c[2] = (float)n[2];
                c[3] = (float)(n[0] + n[1]);
            } else if (n_counts == 3) {   // mask mode "1-hop"
                c[1] = (float)n[1];
                c[2] = (float)(n[0] + n[1]);
            }                             // (1: mask mode "cn", get_count alone, :154-155)
        }
        // common neighbours and >1-hop nodes are already in node order
#pragma unroll
        for (int t = 0; t < 3; t += 2) {
            const int4 *src = entries + (int64_t)t * ent_cap + lo[t];
            for (int64_t i = lane; i < n[t]; i += 64) {
                const int4 e = src[i];
                const int64_t dst = obase[t] + lo[t] + i;
                sel_pair[dst] = (int32_t)p;
                sel_node[dst] = e.y;
                sel_pa[dst] = __int_as_float(e.z);
                sel_pb[dst] = __int_as_float(e.w);
            }
        }
        // one-hop: kept nodes of N(a) (ascending), then kept nodes of N(b) (ascending, flagged); disjoint sets:
        // final rank = own index + number of nodes of the other run below it
        if (n[1] > 0) {
            const int4 *src = entries + ent_cap + lo[1];
            int64_t na = 0;
            for (int64_t i0 = 0; i0 < n[1]; i0 += 64) {
                const int64_t i = i0 + lane;
                const bool isa = i < n[1] && !((uint32_t)src[i].x & S2_FROM_B);
                na += __popcll(__ballot(isa));
            }
            const int64_t nb = n[1] - na;
            for (int64_t i = lane; i < n[1]; i += 64) {
                const int4 e = src[i];
                int64_t r;
                if (i < na) {
                    int64_t l2 = 0, h2 = nb;
                    while (l2 < h2) {
                        const int64_t mid = (l2 + h2) >> 1;
                        if (src[na + mid].y < e.y) l2 = mid + 1; else h2 = mid;
                    }
                    r = i + l2;
                } else {
                    int64_t l2 = 0, h2 = na;
                    while (l2 < h2) {
                        const int64_t mid = (l2 + h2) >> 1;
                        if (src[mid].y < e.y) l2 = mid + 1; else h2 = mid;
                    }
                    r = (i - na) + l2;
                }
                const int64_t dst = obase[1] + lo[1] + r;
                sel_pair[dst] = (int32_t)p;
                sel_node[dst] = e.y;
                sel_pa[dst] = __int_as_float(e.z);
                sel_pb[dst] = __int_as_float(e.w);
            }
        }
    }
}

}  // namespace

extern "C" int64_t lpf_select_plan_blocks(int64_t bs) { return (bs + S2_THREADS - 1) / S2_THREADS; }

extern "C" int lpf_select_plan(int64_t bs, const int64_t *batch, int64_t batch_ld, int64_t n_nodes,
                               const int64_t *adj_rowptr, const int64_t *val_rowptr, const int64_t *t0_rowptr,
                               const int64_t *adjx_rowptr, const int32_t *val_len, const int32_t *t0_len, void *desc,
                               int64_t *offs, int32_t *item_pair,
                               int64_t item_cap, int64_t *ctl, uint64_t *plan_lb, void *stream) {
    if (bs == 0) return LPF_OK;
    LPF_REQUIRE(bs > 0 && bs < (1ll << 31) && batch && batch_ld >= bs && n_nodes > 0 && adj_rowptr && val_rowptr &&
                desc && offs && item_pair && item_cap > 0 && ctl && plan_lb && lpf_aligned16(desc));
    const int64_t nb = (bs + S2_THREADS - 1) / S2_THREADS;
    if (nb > 2048) return LPF_ERR_UNSUPPORTED;  // the chained scan wants every block resident: split larger batches
    hipLaunchKernelGGL(select_plan_kernel, dim3((unsigned)nb), dim3(S2_THREADS), 0, static_cast<hipStream_t>(stream), bs,
                       batch, batch_ld, n_nodes, adj_rowptr, val_rowptr, t0_rowptr, adjx_rowptr, val_len, t0_len,
                       static_cast<PairDesc *>(desc), offs, item_pair, item_cap, ctl, plan_lb);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_select_run(int64_t bs, const void *desc, const int64_t *offs, const int32_t *item_pair,
                              int64_t item_cap, int64_t *ctl, uint64_t *run_lb, const int32_t *adj_col,
                              const float *adj_selfp, const int32_t *adjx_col, const int32_t *val_col,
                              const float *val_val, const void *val_cv, const void *t0_cv,
                              const int32_t *t0_skip, float th_cn, float th_1hop,
                              float th_non1hop, int32_t mode_cn, int32_t *type_ptr, void *entries, int64_t ent_cap,
                              int32_t grid_blocks, void *stream) {
    if (bs == 0) return LPF_OK;
    LPF_REQUIRE(bs > 0 && desc && offs && item_pair && item_cap > 0 && ctl && run_lb && adj_col && type_ptr &&
                entries && ent_cap > 0 && lpf_aligned16(entries) && lpf_aligned16(desc));
    // Round 2's indexed path (self-PPR aligned with the adjacency + hashed one-hop index) was replaced by the walk
    // indexes of select3.hip; what remains here is the general path over the raw PPR rows.
    if (adj_selfp || val_cv) return LPF_ERR_UNSUPPORTED;
    LPF_REQUIRE(val_col && val_val);
    LPF_REQUIRE((t0_cv == nullptr) == (t0_skip == nullptr) && lpf_aligned16(t0_cv) && lpf_aligned16(t0_skip));
    RunArgs a;
    a.bs = bs; a.desc = static_cast<const PairDesc *>(desc); a.offs = offs; a.item_pair = item_pair;
    a.item_cap = item_cap; a.ctl = ctl; a.run_lb = run_lb;
    a.adj_col = adj_col; a.selfp = adj_selfp; a.adjx_col = adjx_col ? adjx_col : adj_col;
    a.val_col = val_col; a.val_val = val_val;
    a.val_cv = static_cast<const int2 *>(val_cv);
    a.t0_cv = static_cast<const int2 *>(t0_cv); a.t0_skip = t0_skip;
    a.th_cn = th_cn; a.th_1 = th_1hop; a.th_n = th_non1hop; a.mode_cn = mode_cn;
    a.type_ptr = type_ptr; a.entries = static_cast<int4 *>(entries); a.ent_cap = ent_cap;
    // one resident round of workgroups (they are persistent; more than fit only queue up behind the others:
    // 1024 / 1280 / 2048 / 4096 workgroups measured 132 / 132 / 138 / 158 us on 256 CUs)
    static LpfPerDevice occ_cache;
    const int n_cu = lpf_cu_count();
    if (n_cu == 0) return LPF_ERR_NO_DEVICE;
    const int resident =
        n_cu * lpf_blocks_per_cu(occ_cache, reinterpret_cast<const void *>(select_run_kernel<false>), S2_THREADS, 0, 4);
    int64_t blocks = grid_blocks > 0 ? grid_blocks : resident;
    if (blocks > item_cap) blocks = item_cap;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(select_run_kernel<false>, dim3((unsigned)blocks), dim3(S2_THREADS), 0, s, a);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_select_export(int64_t bs, const int32_t *type_ptr, const void *entries, int64_t ent_cap,
                                 int64_t *type_ptr64, float *counts_f, int64_t ldc, int32_t n_counts,
                                 int32_t *sel_pair, int32_t *sel_node, float *sel_pa, float *sel_pb, void *stream) {
    if (bs == 0) return LPF_OK;
    LPF_REQUIRE(bs > 0 && type_ptr && entries && ent_cap > 0 && type_ptr64 && sel_pair && sel_node && sel_pa &&
                sel_pb && (n_counts == 1 || n_counts == 3 || n_counts == 4) && (!counts_f || ldc >= n_counts));
    int64_t blocks = (bs + 1 + 3) / 4;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    hipLaunchKernelGGL(select_export_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), bs,
                       type_ptr, static_cast<const int4 *>(entries), ent_cap, type_ptr64, counts_f, ldc, (int)n_counts,
                       sel_pair, sel_node, sel_pa, sel_pb);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}
