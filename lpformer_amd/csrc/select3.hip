// PPR-thresholded node selection over per-model WALK INDEXES: two launches, nothing read back by the host.
//
// Reference: compute_node_mask + get_ppr_vals + get_non_1hop_ppr (src/models/link_transformer.py:214-319, 434-481),
// eval mode, typing adjacency == the model's own adj_mask / full_adj_mask (a caller-supplied override goes through
// select2.hip, which looks values up in the raw PPR rows).  The reference materialises BS x N sparse temporaries and
// coalesces them seven times; select2.hip (round 2) walked BOTH adjacency rows of every pair and searched the other
// row for every candidate.  Here the selected sets are written as intersections and every intersection is evaluated
// from its SHORTER side, one hashed look-up per candidate and no search:
//
//   U(i)   = N(i)  u  {v : P[i,v] passes the weaker of the one-hop / >1-hop tests}, hashed by v, each entry carrying
//            P[i,v] and, in the value's sign bit, "v is adjacent to i"                       ("what is v to i?" in ONE read)
//   CN(a,b)           = N(a) n N(b)                       walk the SHORTER adjacency row s, look v up in U(other): adjacent
//   1-hop, s's side   = N(s) \ N(l), both values >= th_1  same walk (FULL): found non-adjacent / not found
//   1-hop, l's side   = N(l) \ N(s), both values >= th_1  = A1(l) n P1x(s):  A1(l) = entries of N(l) whose own value passes,
//                                                         P1x(s) = non-neighbours of s whose value passes; walk the
//                                                         shorter of the two and look up in U of the other endpoint
//   >1-hop            = T0x(a) n T0x(b), T0x(i) = non-neighbours of i passing the >1-hop test (link_transformer.py:
//                       443-478: the UNMASKED adjacency, which here is the typing adjacency); walk the shorter one
//
// On the collab-like bench batches that is 1.53 M candidate slots instead of 3.76 M (hub rows are only ever looked up
// in, never walked), and a slot costs one 8-byte read + one 64-byte bucket instead of two binary searches.
// Every value that is compared or emitted is the same fp32 number as in the reference: the typing comes from the flag
// (built from the same adjacency), the thresholds are applied in the kernel to the round-tripped values op for op.
//
// Skeleton (unchanged from select2.hip): the candidates of a batch form ONE flat slot space, pair k owning
// [offs[k], offs[k+1]) = a-side walk | b-side walk | >1-hop walk (at least S3_MIN_SLOTS slots), cut into work items of
// S3_ITEM slots whatever pairs they belong to; a plan kernel builds the walk descriptors and offs[] (chained scan over
// 256-pair blocks); persistent workgroups of the run kernel draw items from a ticket, type one slot per thread and
// round, rank the kept entries per type, publish the item's three totals to a second chained scan ordered by ticket,
// park the entries in LDS, type the NEXT item and only then fetch the parked item's place and write it out.
// Placement is deterministic (flat slot order): per type one dense region ordered by (pair, slot); inside a pair's
// one-hop segment the kept nodes of N(a) come before those of N(b) (flag bit 31 of the pair word) -- exactly the layout
// select2.hip produces, so lpf_select_export and the attention kernels do not care which path ran.

#include "walk_common.h"   // node records, walk descriptors, the per-candidate typing (shared with select4.hip)

namespace {

using namespace walk;

constexpr int S3_ITEM = LPF_SELECT_ITEM;             // candidate slots per work item
constexpr int S3_THREADS = 256;
constexpr int S3_ROUNDS = S3_ITEM / S3_THREADS;      // slots per thread
constexpr int S3_WAVES = S3_THREADS / 64;
constexpr int S3_GROUPS = S3_ITEM / 64;              // 64-slot groups of an item (one per wavefront and round)
constexpr int S3_MIN_SLOTS = 16;                     // slots a pair owns at least => at most S3_PAIRS pairs per item
constexpr int S3_PAIRS = S3_ITEM / S3_MIN_SLOTS + 2;
constexpr uint32_t S3_FROM_B = FROM_B;
constexpr int S3_BUCKET = BUCKET;
constexpr int S3_DESC_AHEAD = S3_THREADS / 8;        // descriptors copied before the window is known (one int4 per thread)
constexpr int S3_MINI_WORDS = MINI_WORDS;
constexpr int S3_FLT_LOADS = (2 * (LPF_SELECT_ITEM / 16 + 2) * (S3_MINI_WORDS / 4) + 255) / 256;   // int4 reads per thread

// ------------------------------------------------------------------------------------------- plan
struct PlanArgs {
    int64_t bs;
    const int64_t *batch;
    int64_t batch_ld, n_nodes;
    const NodeRec *rec;
    const int2 *adj_cv, *a1_cv, *px_cv, *t0_cv;   // t0_cv NULL: no >1-hop walk (modes "1-hop", "cn")
    int32_t mode_cn, use_px;
    PairDesc3 *desc;
    int64_t *offs;
    int32_t *item_pair;
    int64_t item_cap;
    int64_t *ctl;
    uint64_t *plan_lb;
};

__global__ __launch_bounds__(S3_THREADS) void select3_plan_kernel(const PlanArgs A) {
    __shared__ int64_t wtot[S3_WAVES];
    __shared__ int64_t s_base, s_blk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Logical block index = order of arrival (a block only ever waits for blocks that are already running); ctl[7]
    // only ever grows and every launch on this control block draws exactly gridDim.x tickets, so ticket / gridDim.x is
    // the launch number that tags the chained-scan words of this launch (see select2.hip, same control block layout).
    if (tid == 0) s_blk = (int64_t)atomicAdd(reinterpret_cast<unsigned long long *>(A.ctl + 7), 1ull);
    __syncthreads();
    const int64_t blk = s_blk % (int64_t)gridDim.x;
    const uint32_t epoch = (uint32_t)((s_blk / (int64_t)gridDim.x) & ((1 << 22) - 1));
    const int64_t k = blk * S3_THREADS + tid;
    int64_t ub = 0;
    if (k < A.bs) {
        const int64_t a = A.batch[k], b = A.batch[A.batch_ld + k];
        PairDesc3 d;
        __builtin_memset(&d, 0, sizeof(d));
        if ((uint64_t)a >= (uint64_t)A.n_nodes || (uint64_t)b >= (uint64_t)A.n_nodes) {
            atomicOr(reinterpret_cast<unsigned long long *>(A.ctl + 3), (unsigned long long)LPF_SELECT_ERR_NODE_RANGE);
        } else {
            NodeRec r[2];
            __builtin_memcpy(&r[0], A.rec + a, sizeof(NodeRec));
            __builtin_memcpy(&r[1], A.rec + b, sizeof(NodeRec));
            build_desc(d, a, b, r, A.adj_cv, A.a1_cv, A.px_cv, A.t0_cv, A.mode_cn, A.use_px);
        }
        ub = d.total < S3_MIN_SLOTS ? S3_MIN_SLOTS : d.total;
        __builtin_memcpy(A.desc + k, &d, sizeof(PairDesc3));
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
    for (int w = 0; w < S3_WAVES; ++w) {
        if (w < wave) pre += wtot[w];
        btot += wtot[w];
    }
    if (wave == 0) {
        const int64_t base = (int64_t)lb_exclusive(A.plan_lb, blk, epoch, (uint64_t)btot, lane);
        if (lane == 0) s_base = base;
    }
    __syncthreads();
    const int64_t o = s_base + pre + x - ub;  // exclusive
    if (k < A.bs) {
        A.offs[k] = o;
        // first pair of every item that starts inside this pair's slots
        for (int64_t it = (o + S3_ITEM - 1) / S3_ITEM; it * S3_ITEM < o + ub; ++it) {
            if (it < A.item_cap) A.item_pair[it] = (int32_t)k;
            else atomicOr(reinterpret_cast<unsigned long long *>(A.ctl + 3), (unsigned long long)LPF_SELECT_ERR_ITEM_CAP);
        }
        if (k == A.bs - 1) {
            const int64_t total = o + ub;
            A.offs[A.bs] = total;
            A.ctl[0] = total;
            A.ctl[1] = (total + S3_ITEM - 1) / S3_ITEM;
            A.ctl[2] = 0;  // ticket counter of the run kernel (stream order: the previous run kernel has finished)
            A.ctl[8] = epoch;
        }
    }
}

// ------------------------------------------------------------------------------------------- run
struct RunArgs3 {
    int64_t bs;
    const PairDesc3 *desc;
    const int64_t *offs;
    const int32_t *item_pair;
    int64_t item_cap;
    int64_t *ctl;
    uint64_t *run_lb;   // [3][item_cap]
    const int2 *u_cv;   // the hashed union index: buckets of 8 {node, value bits | adjacent << 31}
    const uint32_t *mini;   // [n_nodes][32]: the union rows' mini filters
    float th_cn, th_1, th_n;
    int32_t mode_cn;
    int32_t *type_ptr;  // [3][bs+1]
    int4 *entries;      // [3][ent_cap]
    int64_t ent_cap;
};

struct RunLds3 {
    int4 pk[S3_ITEM];             // parked entries: type 0 | type 1 | type 2, each in slot order
    PairDesc3 dsc[S3_PAIRS];      // descriptors of the window pairs (one coalesced copy per item)
    int32_t loc[S3_PAIRS + 1];    // slot (relative to the item) at which window pair i starts (pair 0: <= 0)
    uint32_t bits[2 * S3_GROUPS]; // bit l: a window pair (other than the first) starts in slot l
    int32_t pre[S3_GROUPS];       // window pairs that start before group g (beyond the first)
    int32_t cnt[S3_GROUPS][4];    // kept entries per type of group g, then their exclusive scan
    int64_t base[3];
    int64_t nx_ticket, nx_pf;     // the NEXT item and its first pair, drawn while the current one is being typed
    int32_t n_pairs;
    int32_t run[3];               // kept entries per type of the item being built
    // the parked item: what its deferred write-out needs once the window belongs to the next item
    int64_t p_item, p_pf;
    int32_t p_run[3], p_np, p_last, p_live;
    int16_t ps[S3_PAIRS][4];      // {rank of the pair's first slot per type, pair starts in the item}
    uint4 flt[S3_PAIRS][2][S3_MINI_WORDS / 4];   // mini filters of the window pairs' endpoints (a, b)
};
static_assert(sizeof(RunLds3) <= 48 * 1024, "three workgroups per CU");

// Workgroup barrier that orders LDS ONLY.  __syncthreads() is a workgroup-scope fence + barrier, and the fence makes the
// compiler drain the vector-memory counter -- loads included -- in front of every barrier: in the run kernel that turned
// each of the item loop's thirteen barriers into a wait for whatever global reads were in flight (in-kernel stamps: ~2 us
// per dependent round trip, seven of them per item).  Nothing global is handed from one wavefront of a workgroup to
// another here -- what leaves a workgroup for others goes through agent-scope atomics --, so the barriers only have to
// order LDS, and the loads requested ahead (next item's ticket, window, descriptors) stay in flight across them.
__device__ __forceinline__ void s3_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Entries of the parked item from L.pk to their final place, the segment starts of the pairs that begin in it and,
// from the last item, the totals.  L.base holds the item's place per type.
__device__ __forceinline__ void s3_write_out(const RunArgs3 &A, RunLds3 &L, int64_t bs, int tid) {
    const int r0 = L.p_run[0], r1 = L.p_run[1], r2 = L.p_run[2];
    const int64_t b0 = L.base[0], b1 = L.base[1], b2 = L.base[2];
    const int n = r0 + r1 + r2;
    for (int g = tid; g < n; g += S3_THREADS) {
        const int t = g < r0 ? 0 : (g < r0 + r1 ? 1 : 2);
        const int64_t dst = t == 0 ? b0 + g : (t == 1 ? b1 + (g - r0) : b2 + (g - r0 - r1));
#ifdef S3_ABL_NOWRITE
        if (dst < 0) {
#else
        if (dst < A.ent_cap) {
#endif
            A.entries[(int64_t)t * A.ent_cap + dst] = L.pk[g];
        } else {
            atomicOr(reinterpret_cast<unsigned long long *>(A.ctl + 3), (unsigned long long)LPF_SELECT_ERR_ENTRY_CAP);
        }
    }
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
__device__ __forceinline__ void s3_finish_parked(const RunArgs3 &A, RunLds3 &L, uint32_t epoch, int64_t bs, int lane,
                                                 int wave, int tid) {
    s3_lds_barrier();
    if (L.p_live) {
        if (wave < 3) {
#ifdef S3_ABL_NOLB
            const int64_t base = L.p_item * 300;
#else
            const int64_t base = (int64_t)lb_lookback(A.run_lb + (int64_t)wave * A.item_cap, L.p_item, epoch,
                                                      (uint64_t)L.p_run[wave], lane);
#endif
            if (lane == 0) L.base[wave] = base;
        }
        s3_lds_barrier();
        s3_write_out(A, L, bs, tid);
    }
    s3_lds_barrier();
}

// (tuning builds, -DS3_STAMPS: where does an item's time go?  Thread 0 of every workgroup accumulates the wall-clock
//  ticks -- 100 MHz -- between the marks below and leaves them, with its item count, in a debug buffer.)
#ifdef S3_STAMPS
__device__ uint64_t *s3_stamp_buf = nullptr;
#define S3_STAMP(k) do { if (tid == 0) { const uint64_t n__ = wall_clock64(); st_acc[k] += n__ - st_last; st_last = n__; } } while (0)
#define S3_STAMP_WAIT(k) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); S3_STAMP(k); } while (0)
#else
#define S3_STAMP(k) do { } while (0)
#define S3_STAMP_WAIT(k) do { } while (0)
#endif
#ifndef S3_MIN_WAVES      // (tuning) resident wavefronts per SIMD the register budget allows: workgroups per CU
#define S3_MIN_WAVES 2
#endif
#ifndef S3_PER_CU         // (tuning) persistent workgroups per CU the launch asks for
#define S3_PER_CU 2
#endif
__global__ __launch_bounds__(S3_THREADS, S3_MIN_WAVES) void select3_run_kernel(const RunArgs3 A) {
    __shared__ RunLds3 L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const int64_t total = A.ctl[0];
    int64_t n_items = A.ctl[1];
    if (n_items > A.item_cap) n_items = A.item_cap;
    const int64_t bs = A.bs;
    const uint32_t epoch = (uint32_t)A.ctl[8];  // launch number, written by the plan kernel
    // The ticket of an item and its first pair are drawn one item AHEAD (by the last wavefront, between the phases of
    // the item being typed): at the top of the loop both are known and the window and the descriptors are requested
    // at once -- ticket -> item_pair -> offsets -> descriptors used to be four dependent round trips per item.
    const bool drawer = tid == S3_THREADS - 1;
#ifdef S3_STAMPS
    uint64_t st_acc[16] = {0}, st_last = wall_clock64(), st_t0 = st_last;
#endif
    if (tid == 0) {
        L.p_live = 0;
        const int64_t t = (int64_t)atomicAdd(reinterpret_cast<unsigned long long *>(A.ctl + 2), 1ull);
        L.nx_ticket = t;
        L.nx_pf = t < n_items ? A.item_pair[t] : 0;
    }
    if (tid < 2 * S3_GROUPS) L.bits[tid] = 0u;
    s3_lds_barrier();
    // An item's window (slot offsets of its pairs) and its first S3_DESC_AHEAD descriptors (one int4 per thread; an item
    // of 1,024 slots rarely holds more pairs) are REQUESTED as soon as the item is known -- for the first item here, for
    // every other one behind the typing of the item before it -- and consumed at the top of the loop: they travel while
    // the previous item is ranked, looked back and parked.
    struct Req {
        int64_t it, pf, c0, v;
        int n_here, n4a;
        int4 dreg;
    };
    auto request = [&]() __attribute__((always_inline)) {
        Req q;
        const int64_t t = L.nx_ticket, p = L.nx_pf;
        q.it = ((int64_t)__builtin_amdgcn_readfirstlane((int)(t >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(t & 0xffffffff));
        q.pf = ((int64_t)__builtin_amdgcn_readfirstlane((int)(p >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(p & 0xffffffff));
        q.c0 = q.it * S3_ITEM;
        q.n_here = (int)((total - q.c0) < S3_ITEM ? (total - q.c0) : S3_ITEM);
        const int64_t avail = bs - q.pf;
        q.n4a = (int)(avail < S3_DESC_AHEAD ? avail : S3_DESC_AHEAD) * 8;
        q.dreg = make_int4(0, 0, 0, 0);
        q.v = q.n_here;
        if (q.it < n_items) {
            if (tid < q.n4a) q.dreg = reinterpret_cast<const int4 *>(A.desc + q.pf)[tid];
            if (tid <= S3_PAIRS && q.pf + tid <= bs) q.v = A.offs[q.pf + tid] - q.c0;
        }
        return q;
    };
    Req cur = request();

    while (true) {
        const int64_t it = cur.it, pf = cur.pf, c0 = cur.c0;
        const int n_here = cur.n_here, n4a = cur.n4a;
        const int4 dreg = cur.dreg;
        int64_t v = cur.v;
        s3_lds_barrier();  // the previous item's window is no longer needed (and its bit map is zero again)
        S3_STAMP(1);    // top barrier (waits for the slowest wavefront of the previous item)
        if (it >= n_items) break;
        int64_t nx_t = 0;
        if (drawer) nx_t = (int64_t)atomicAdd(reinterpret_cast<unsigned long long *>(A.ctl + 2), 1ull);

        // ---- pair window: loc[i] = offs[pf + i] - c0, pairs past the item read n_here; a pair other than the first
        //      raises the bit of the slot it starts in (every pair owns >= S3_MIN_SLOTS slots: at most S3_PAIRS pairs,
        //      no two in one slot)
        S3_STAMP(2);    // (the window / descriptor requests were made an item ago)
        if (tid < n4a) reinterpret_cast<int4 *>(L.dsc)[tid] = dreg;
        if (tid < 128) {
            if (v > n_here) v = n_here;
            if (tid <= S3_PAIRS) L.loc[tid] = (int32_t)v;
            const bool in = tid <= S3_PAIRS && v < n_here;
            if (in && tid > 0) atomicOr(&L.bits[v >> 5], 1u << (v & 31));
            const int cntp = __popcll(__ballot(in));
            if (lane == 0) L.pre[wave] = cntp;       // (scratch: two partial counts)
        }
        s3_lds_barrier();
        const int np = L.pre[0] + L.pre[1];  // pairs with at least one slot in this item (>= 1)
        s3_lds_barrier();
        S3_STAMP(3);    // window processed
        if (tid < S3_GROUPS) {               // exclusive popcount scan over the 64-slot groups
            int s = 0;
            for (int g = 0; g < tid; ++g) s += __popc(L.bits[2 * g]) + __popc(L.bits[2 * g + 1]);
            L.pre[tid] = s;
        }
        {   // the rest of the window's descriptors (every slot reads its pair's walk from LDS)
            const int n4 = np * 8;
            const int4 *src = reinterpret_cast<const int4 *>(A.desc + pf);
            int4 *dst = reinterpret_cast<int4 *>(L.dsc);
            for (int i = S3_DESC_AHEAD * 8 + tid; i < n4; i += S3_THREADS) dst[i] = src[i];
        }
        // the next item's first pair: requested now, in flight beside this item's walked entries
        int64_t nx_p = 0;
        if (drawer && nx_t < n_items) nx_p = A.item_pair[nx_t];
        s3_lds_barrier();
        S3_STAMP(4);    // rest of the descriptors + barrier

        // ---- typing: one slot per thread and round; everything a kept slot needs later stays in registers.  The rounds
        //      are taken together, phase by phase, so that their memory round trips overlap: first every round's walked
        //      entry is requested, then every round's bucket (four 16-byte reads each), then the arithmetic -- two
        //      dependent round trips per item instead of eight.
        int code[S3_ROUNDS], node[S3_ROUNDS], win[S3_ROUNDS];
        float va[S3_ROUNDS], vb[S3_ROUNDS];
        int kindr[S3_ROUNDS], unbr[S3_ROUNDS];
        int64_t u0r[S3_ROUNDS];
        int2 cvr[S3_ROUNDS];
        bool act[S3_ROUNDS];
        // The mini filters of the window pairs' endpoints (one 128-byte line per node, eight lanes each) are requested
        // first and land in LDS while the walked entries travel: "is x in the other endpoint's union row AT ALL?" is
        // then answered from LDS for most candidates, and a candidate that is not there needs no bucket -- "not found"
        // is what the bucket would have said.  (Random reads are what bounds this kernel: the chip retires ~54 G
        // lane-reads of distinct lines per second whatever is in flight, tools/probe/random_read.hip.)
        uint4 fr[S3_FLT_LOADS];
#pragma unroll
        for (int f = 0; f < S3_FLT_LOADS; ++f) {
            const int i = f * S3_THREADS + tid;                    // (pair w, endpoint e, 16-byte piece q)
            const int w = i / (2 * (S3_MINI_WORDS / 4));
            fr[f] = make_uint4(0u, 0u, 0u, 0u);
#ifdef S3_ABL_NOFLT
            if (false) {
#else
            if (w < np) {
#endif
                const int node = (i & (S3_MINI_WORDS / 4)) ? L.dsc[w].b : L.dsc[w].a;
                fr[f] = reinterpret_cast<const uint4 *>(A.mini + (int64_t)node * S3_MINI_WORDS)[i & (S3_MINI_WORDS / 4 - 1)];
            }
        }
#pragma unroll
        for (int r = 0; r < S3_ROUNDS; ++r) {
            const int g = S3_WAVES * r + wave;          // the wavefront's 64-slot group
            const int l = 64 * g + lane;
            code[r] = 0; node[r] = 0; win[r] = 0; va[r] = 0.f; vb[r] = 0.f;
            act[r] = false; kindr[r] = 0; unbr[r] = 0; u0r[r] = 0; cvr[r] = make_int2(0, 0);
            if (l < n_here) {
                // window pair of slot l = pairs that start at or before it
                const uint32_t blo = L.bits[2 * g], bhi = L.bits[2 * g + 1];
                const uint64_t bm = ((uint64_t)bhi << 32) | blo;
                const int w = L.pre[g] + __popcll(bm & lt_mask) + (int)((bm >> lane) & 1ull);
                const PairDesc3 &d = L.dsc[w];
                const int i = l - L.loc[w];
                win[r] = w;
                if (i < d.total) {
                    const int k = (i >= d.w[1].start) + (i >= d.w[2].start);
                    const Walk3 wk = d.w[k];
#ifdef S3_ABL_NOWALK
                    cvr[r] = make_int2(i, 0x3c000000);
#else
                    cvr[r] = wk.src[i - wk.start];
#endif
                    kindr[r] = wk.kind; unbr[r] = wk.unb; u0r[r] = wk.u0;
                    act[r] = true;
                }
            }
        }
#pragma unroll
        for (int f = 0; f < S3_FLT_LOADS; ++f) {
            const int i = f * S3_THREADS + tid;
            if (i < np * 2 * (S3_MINI_WORDS / 4)) (&L.flt[0][0][0])[i] = fr[f];
        }
        if (drawer) { L.nx_ticket = nx_t; L.nx_pf = nx_p; }   // (read behind the typing barrier below)
        s3_lds_barrier();
        S3_STAMP_WAIT(5);   // walked entries and mini filters arrived
        int4 bv[S3_ROUNDS][S3_BUCKET / 2];
#pragma unroll
        for (int r = 0; r < S3_ROUNDS; ++r) {
            // what is x to the other endpoint?  one bucket of its hashed union row -- if the endpoint's mini filter lets
            // x through (the looked-up endpoint is b when the walked row is a's, and the other way round)
#if defined(S3_ABL_NOBUCKET)   // (ablation builds: wrong results, what does the phase cost?)
            const bool look = false;
#elif defined(S3_NO_BLOOM)   // (tuning: every candidate fetches its bucket, as before the filter)
            const bool look = act[r] && unbr[r] > 0;
#else
            const bool look = act[r] && unbr[r] > 0 &&
                              mini_pass(reinterpret_cast<const uint32_t *>(&L.flt[win[r]][(kindr[r] & KF_SRC_A) ? 1 : 0][0]), cvr[r].x);
#endif
            const uint32_t b = look ? bucket_of(cvr[r].x, unbr[r]) : 0u;
            const int4 *blk = reinterpret_cast<const int4 *>(A.u_cv + (look ? u0r[r] + S3_BUCKET * (int64_t)b : 0));
#pragma unroll
            for (int q = 0; q < S3_BUCKET / 2; ++q) bv[r][q] = look ? blk[q] : make_int4(-1, 0, -1, 0);
        }
        S3_STAMP_WAIT(7);   // buckets arrived
#pragma unroll
        for (int r = 0; r < S3_ROUNDS; ++r) {
            const int g = S3_WAVES * r + wave;
            if (act[r]) {
                const Typed ty = type_slot(cvr[r].x, __int_as_float(cvr[r].y), kindr[r], bv[r], A.th_cn, A.th_1, A.th_n,
                                           A.mode_cn);
                code[r] = ty.code;
                node[r] = cvr[r].x;
                va[r] = ty.va;
                vb[r] = ty.vb;
            }
            const int c3 = code[r] & 3;
            const uint64_t b0 = __ballot(c3 == 1), b1 = __ballot(c3 == 2), b2 = __ballot(c3 == 3);
            if (lane == 0) {
                int32_t *c = L.cnt[g];
                c[0] = __popcll(b0); c[1] = __popcll(b1); c[2] = __popcll(b2);
            }
        }
        s3_lds_barrier();
        S3_STAMP(8);    // arithmetic + ballots + barrier
        // the bit map was read by the typing above only; the next item (drawn meanwhile) is requested now
        if (tid < 2 * S3_GROUPS) L.bits[tid] = 0u;
        const Req nxt = request();

        // ---- ranks inside the item; the item's totals go out to the chained scan at once
        if (wave < 3) {  // wavefront t: exclusive scan of type t over the groups in slot order
            const int v = lane < S3_GROUPS ? L.cnt[lane][wave] : 0;
            int x = v;
#pragma unroll
            for (int dlt = 1; dlt < S3_GROUPS; dlt <<= 1) {
                const int y = __shfl_up(x, dlt, 64);
                if (lane >= dlt) x += y;
            }
            if (lane < S3_GROUPS) L.cnt[lane][wave] = x - v;
            const int run = __shfl(x, S3_GROUPS - 1, 64);
            if (lane == 0) L.run[wave] = run;
            lb_publish(A.run_lb + (int64_t)wave * A.item_cap, it, epoch, (uint64_t)run, lane);
        }
        S3_STAMP(9);    // rank scan + publish
        // ---- deferred write-out: the item's place in the output depends on every earlier item, and the slowest of the
        //      ones in flight decides when that is known.  So the kept entries are parked in LDS, the workgroup went on
        //      to type this item first, and only now asks for the PREVIOUS item's place (by now an answer that needs
        //      no waiting) and writes it out.
        s3_finish_parked(A, L, epoch, bs, lane, wave, tid);  // (starts and ends with a barrier)
        S3_STAMP(10);   // look-back + write-out of the parked item
        const int run0 = L.run[0], run1 = L.run[1];
        if (tid == 0) {
            L.p_item = it; L.p_pf = pf; L.p_np = np; L.p_last = (c0 + n_here == total);
            L.p_run[0] = run0; L.p_run[1] = run1; L.p_run[2] = L.run[2];
            L.p_live = 1;
        }
        if (tid < S3_PAIRS) L.ps[tid][3] = 0;
        s3_lds_barrier();
#pragma unroll
        for (int r = 0; r < S3_ROUNDS; ++r) {
            const int g = S3_WAVES * r + wave;
            const int l = 64 * g + lane;
            const int c3 = code[r] & 3;
            const uint64_t b0 = __ballot(c3 == 1), b1 = __ballot(c3 == 2), b2 = __ballot(c3 == 3);
            if (l >= n_here) continue;
            const int32_t *gp = L.cnt[g];
            const int k0 = gp[0] + __popcll(b0 & lt_mask);
            const int k1 = gp[1] + __popcll(b1 & lt_mask);
            const int k2 = gp[2] + __popcll(b2 & lt_mask);
            const int w = win[r];
            if (c3) {
                const int pos = c3 == 1 ? k0 : (c3 == 2 ? run0 + k1 : run0 + run1 + k2);
                L.pk[pos] = make_int4((int32_t)((uint32_t)(pf + w) | ((code[r] & 4) ? S3_FROM_B : 0u)), node[r],
                                      __float_as_int(va[r]), __float_as_int(vb[r]));
            }
            if (l == L.loc[w]) {  // first slot of the pair: its three segment starts
                L.ps[w][0] = (int16_t)k0; L.ps[w][1] = (int16_t)k1; L.ps[w][2] = (int16_t)k2; L.ps[w][3] = 1;
            }
        }
        S3_STAMP(11);   // parking
        cur = nxt;
    }
    s3_finish_parked(A, L, epoch, bs, lane, wave, tid);
#ifdef S3_STAMPS
    if (tid == 0 && s3_stamp_buf) {
        S3_STAMP(12);   // last look-back + write-out
        uint64_t *o = s3_stamp_buf + (int64_t)blockIdx.x * 16;
        for (int k = 0; k < 13; ++k) o[k] = st_acc[k];
        o[13] = st_t0; o[14] = st_last;
    }
#endif
}

}  // namespace

#ifdef S3_STAMPS
extern "C" int lpf_select3_set_stamps(void *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(s3_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? LPF_OK : LPF_ERR_LAUNCH;
}
#endif

/* ---- C ABI ---------------------------------------------------------------------------------------------------- */
extern "C" int lpf_select3_plan(int64_t bs, const int64_t *batch, int64_t batch_ld, int64_t n_nodes, const void *node_rec,
                                const void *adj_cv, const void *a1_cv, const void *px_cv, const void *t0_cv,
                                int32_t mode_cn, int32_t use_px, void *desc, int64_t *offs, int32_t *item_pair,
                                int64_t item_cap, int64_t *ctl, uint64_t *plan_lb, void *stream) {
    if (bs == 0) return LPF_OK;
    LPF_REQUIRE(bs > 0 && bs < (1ll << 31) && batch && batch_ld >= bs && n_nodes > 0 && node_rec && adj_cv && a1_cv &&
                (px_cv || !use_px) && desc && offs && item_pair && item_cap > 0 && ctl && plan_lb &&
                lpf_aligned16(desc) && lpf_aligned16(node_rec));
    const int64_t nb = (bs + S3_THREADS - 1) / S3_THREADS;
    if (nb > 2048) return LPF_ERR_UNSUPPORTED;  // the chained scan wants every block resident: split larger batches
    PlanArgs a;
    a.bs = bs; a.batch = batch; a.batch_ld = batch_ld; a.n_nodes = n_nodes;
    a.rec = static_cast<const NodeRec *>(node_rec);
    a.adj_cv = static_cast<const int2 *>(adj_cv); a.a1_cv = static_cast<const int2 *>(a1_cv);
    a.px_cv = static_cast<const int2 *>(px_cv); a.t0_cv = static_cast<const int2 *>(t0_cv);
    a.mode_cn = mode_cn; a.use_px = use_px;
    a.desc = static_cast<PairDesc3 *>(desc); a.offs = offs; a.item_pair = item_pair; a.item_cap = item_cap;
    a.ctl = ctl; a.plan_lb = plan_lb;
    hipLaunchKernelGGL(select3_plan_kernel, dim3((unsigned)nb), dim3(S3_THREADS), 0, static_cast<hipStream_t>(stream), a);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_select3_run(int64_t bs, const void *desc, const int64_t *offs, const int32_t *item_pair,
                               int64_t item_cap, int64_t *ctl, uint64_t *run_lb, const void *u_cv, const void *mini,
                               float th_cn,
                               float th_1hop, float th_non1hop, int32_t mode_cn, int32_t *type_ptr, void *entries,
                               int64_t ent_cap, int32_t grid_blocks, void *stream) {
    if (bs == 0) return LPF_OK;
    LPF_REQUIRE(bs > 0 && desc && offs && item_pair && item_cap > 0 && ctl && run_lb && u_cv && mini && type_ptr &&
                entries && ent_cap > 0 && lpf_aligned16(entries) && lpf_aligned16(desc) && lpf_aligned16(u_cv) &&
                lpf_aligned16(mini));
    RunArgs3 a;
    a.bs = bs; a.desc = static_cast<const PairDesc3 *>(desc); a.offs = offs; a.item_pair = item_pair;
    a.item_cap = item_cap; a.ctl = ctl; a.run_lb = run_lb; a.u_cv = static_cast<const int2 *>(u_cv);
    a.mini = static_cast<const uint32_t *>(mini);
    a.th_cn = th_cn; a.th_1 = th_1hop; a.th_n = th_non1hop; a.mode_cn = mode_cn;
    a.type_ptr = type_ptr; a.entries = static_cast<int4 *>(entries); a.ent_cap = ent_cap;
    // Persistent workgroups: by default TWO per CU, half of what fits -- the kernel waits for memory, not for issue slots,
    // and at half the footprint it is faster by itself (collab-like 60.5 -> 55.9 us: fewer workgroups contend for the
    // same DRAM pages and the look-back chains are shorter) and leaves room for the kernels of other streams (pipelined
    // step 0.187 -> 0.181 ms; one per CU: 79 us by itself, the same pipelined step).
    static LpfPerDevice occ_cache;
    const int n_cu = lpf_cu_count();
    if (n_cu == 0) return LPF_ERR_NO_DEVICE;
    const int occ = lpf_blocks_per_cu(occ_cache, reinterpret_cast<const void *>(select3_run_kernel), S3_THREADS, 0, 4);
    const int resident = n_cu * (occ < S3_PER_CU ? occ : S3_PER_CU);
    int64_t blocks = grid_blocks > 0 ? grid_blocks : resident;
    if (blocks > item_cap) blocks = item_cap;
    hipLaunchKernelGGL(select3_run_kernel, dim3((unsigned)blocks), dim3(S3_THREADS), 0, static_cast<hipStream_t>(stream), a);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}
