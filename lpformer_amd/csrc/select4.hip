// PPR-thresholded node selection over the per-model WALK INDEXES, ONE launch, PAIR-MAJOR output, no chained scan.
//
// Reference: compute_node_mask + get_ppr_vals + get_non_1hop_ppr (src/models/link_transformer.py:214-319, 434-481), eval
// mode, typing adjacency == the model's own adj_mask / full_adj_mask.  The sets, the walks and the per-candidate
// arithmetic are select3.hip's (walk_common.h: every selected set an intersection evaluated from its shorter side, one
// hashed look-up per candidate, the reference's fp32 round trip op for op) -- what is different is the SKELETON.
//
// select3.hip lays all candidates of a batch out in one flat slot space, which takes a plan launch (descriptors + a
// chained scan over the pairs), a ticket, a window and eleven barriers per 1,024-slot item, and a second chained scan
// that turns per-item totals into positions in three type-major regions: 34 of its 57 us are that skeleton
// (DESIGN.md 5.2b), and the pair-major attention kernel then pays a search per entry to undo the type-major order.
// Here a workgroup owns a BLOCK of 64 consecutive pairs, everything about the block is decided inside the workgroup,
// and nothing is waited for that another workgroup produces:
//
//   plan      wavefront 0, one lane per pair: ids -> the two 64-byte node records -> the pair's three walks (LDS) and a
//             wave scan of the slot counts; meanwhile every thread fetches one 16-byte piece of the endpoints' mini
//             filters.  The block's place in the entry buffer is ONE atomic add of its slot count (an upper bound of
//             what it keeps) on a counter -- where a block lands is irrelevant, its pairs say where they start.
//   typing    the block's slots are dealt to the wavefronts 64 at a time, four rounds per thread in flight (walked
//             entries, then buckets, then the arithmetic: two dependent round trips per 4 x NTH slots); the pair of a
//             slot comes from a ballot over the pair starts.
//   placing   kept entries are compacted IN SLOT ORDER -- which is pair-major, a pair's slots being contiguous -- by
//             ballot ranks and one scan over the rounds of the block: a pair's entries end up contiguous, its start and
//             its three counts go to pair_tab[pair], the type of an entry travels in its record.
//
// Result: entries {pair | type << 29 | from_N(b) << 31, node, pa, pb}; pair_tab[pair] = {start, n_cn, n_1hop, n_far};
// blk_cnt[block] = {entries, pairs with entries} of the block (the attention kernel splits its work by it).  Deterministic up to the block
// bases: every consumer addresses entries through pair_tab, so scores do not depend on where a block landed.
#include "walk_common.h"

namespace {

using namespace walk;

constexpr int S4_PAIRS = LPF_SELECT4_BLOCK;   // pairs per workgroup: one lane of the planning wavefront each
#ifndef S4_ROUNDS_N       // (tuning: slots per thread in flight)
#define S4_ROUNDS_N 4
#endif
constexpr int S4_ROUNDS = S4_ROUNDS_N;        // slots per thread in flight
constexpr int CTL_SLOTS = 1, CTL_ERR = 3, CTL_DONE = 10, CTL_ALLOC = 16;   // CTL_ALLOC: first of S4_SHARDS allocation counters
// The entry buffer is cut into S4_SHARDS equal regions, a workgroup allocates in region blockIdx.x % S4_SHARDS: all
// workgroups of a launch reach their allocation within a microsecond of each other, and 256 returning atomics on ONE
// word take ~3 us to drain (~12 ns each) -- the wavefront that issued one waits that long at its next load.
constexpr int S4_SHARDS = 8;
static_assert(CTL_ALLOC + S4_SHARDS <= LPF_SELECT4_CTL_WORDS, "control block too small");

struct Args4 {
    int64_t bs;
    const int64_t *batch;
    int64_t batch_ld, n_nodes;
    const NodeRec *rec;
    const int2 *adj_cv, *a1_cv, *px_cv, *t0_cv;   // t0_cv NULL: no >1-hop walk (modes "1-hop", "cn")
    const int2 *u_cv;
    const uint32_t *mini;
    int32_t mode_cn, use_px;
    float th_cn, th_1, th_n;
    int64_t *ctl;
    int4 *pair_tab;      // [bs]
    int32_t *blk_cnt;    // [ceil(bs / 64)][2]: entries, pairs with entries
    int4 *blk_types;     // optional [ceil(bs / 64)]: {common neighbours, one-hop, >1-hop, 0} kept per block (lpf_select4_regions)
    int4 *entries;       // [ent_cap]
    int64_t ent_cap;
};

// NB: blocks of 64 pairs a workgroup takes together (one planning wavefront each; their slots form one space, their
// entries one contiguous run): the plan's two round trips are paid once per NB blocks and a workgroup's batches are fuller
template <int NTH, int NB>
struct Lds4 {
    static constexpr int WAVES = NTH / 64, NR = WAVES * S4_ROUNDS, PAIRS = S4_PAIRS * NB;
    PairDesc3 dsc[PAIRS];                        // the walk descriptors
    uint4 flt[PAIRS][2][MINI_WORDS / 4];         // mini filters of the endpoints (a, b)
    int32_t sflag[WAVES][S4_ROUNDS][64];         // per wavefront and round: "a pair starts in this slot"
    int32_t loc[PAIRS];                          // first slot of pair j, relative to its block of 64; pairs past the batch: the block's slots
    int32_t pcnt[PAIRS][4];                      // per pair: kept common neighbours, one-hop, >1-hop; start (workgroup-relative)
    int32_t rcnt[NR];                            // kept entries per round of the batch, then their exclusive scan
    int32_t sub[NB];                             // slots of every block
    int64_t base;                                // the workgroup's place in the entry buffer
    int32_t run, ovf;
};

// Workgroup barrier that orders LDS only (select3.hip: __syncthreads() would drain the vector-memory counter -- here
// the allocation atomic of wavefront 0 and the loads requested ahead stay in flight across it).  Nothing global is
// handed between the wavefronts of a workgroup.
__device__ __forceinline__ void s4_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// (tuning builds, -DS4_STAMPS: thread 0 of every workgroup leaves the wall clock -- 100 MHz -- at the marks below in a
//  debug buffer; tools/select4_stamps.py)
#ifdef S4_STAMPS
__device__ uint64_t *s4_stamp_buf = nullptr;
#define S4_STAMP(k) do { if (tid == 0) st_t[k] = wall_clock64(); } while (0)
#define S4_STAMP_WAIT(k) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); S4_STAMP(k); } while (0)
#else
#define S4_STAMP(k) do { } while (0)
#define S4_STAMP_WAIT(k) do { } while (0)
#endif

template <int NTH, int NB>
__global__ __launch_bounds__(NTH, 4) void select4_kernel(const Args4 A) {
    using LT = Lds4<NTH, NB>;
    constexpr int WAVES = LT::WAVES, NR = LT::NR, PAIRS = LT::PAIRS;
    static_assert(NB <= WAVES && NB <= 4, "one planning wavefront per block of 64 pairs");
    __shared__ LT L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull, le_mask = lt_mask | (1ull << lane);
    const int64_t p0 = (int64_t)blockIdx.x * PAIRS;
    const int np = (int)(A.bs - p0 < PAIRS ? A.bs - p0 : PAIRS);
#ifdef S4_STAMPS
    uint64_t st_t[16] = {0};
    S4_STAMP(0);
#endif
#ifdef S4_STAGGER   /* (tuning aid: every other workgroup of an XCD starts S4_STAGGER x 3.9 us late -- phases out of lockstep) */
    if (blockIdx.x & 8)
        for (int i = 0; i < S4_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif

    // ---- the endpoints' mini filters: one 16-byte piece per thread and trip (id -> piece: two dependent reads, beside
    //      the plan's id -> node record)
    constexpr int FP = PAIRS * 2 * (MINI_WORDS / 4), FL = (FP + NTH - 1) / NTH;
    uint4 fr[FL];
#pragma unroll
    for (int f = 0; f < FL; ++f) {
        const int i = f * NTH + tid, j = i / (2 * (MINI_WORDS / 4)), e = (i / (MINI_WORDS / 4)) & 1;
        fr[f] = make_uint4(0u, 0u, 0u, 0u);
        if (i < FP && j < np) {
            const int64_t id = A.batch[(int64_t)e * A.batch_ld + p0 + j];
            if ((uint64_t)id < (uint64_t)A.n_nodes)
                fr[f] = reinterpret_cast<const uint4 *>(A.mini + id * MINI_WORDS)[i & (MINI_WORDS / 4 - 1)];
        }
    }
    // ---- plan (wavefront q < NB: block q of 64 pairs, lane = pair)
    if (wave < NB) {
        const int j = S4_PAIRS * wave + lane;
        PairDesc3 d;
        __builtin_memset(&d, 0, sizeof(d));
        int ub = 0;
        if (j < np) {
            const int64_t a = A.batch[p0 + j], b = A.batch[A.batch_ld + p0 + j];
            if ((uint64_t)a >= (uint64_t)A.n_nodes || (uint64_t)b >= (uint64_t)A.n_nodes) {
                atomicOr(reinterpret_cast<unsigned long long *>(A.ctl + CTL_ERR), (unsigned long long)LPF_SELECT_ERR_NODE_RANGE);
            } else {
                NodeRec r[2];
                __builtin_memcpy(&r[0], A.rec + a, sizeof(NodeRec));
                __builtin_memcpy(&r[1], A.rec + b, sizeof(NodeRec));
                build_desc(d, a, b, r, A.adj_cv, A.a1_cv, A.px_cv, A.t0_cv, A.mode_cn, A.use_px);
            }
            ub = d.total < 1 ? 1 : d.total;     // (every pair owns a slot: no two pairs start in the same one)
        }
        int x = ub;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const int y = __shfl_up(x, dlt, 64);
            if (lane >= dlt) x += y;
        }
        L.dsc[j] = d;
        L.loc[j] = x - ub;
        *reinterpret_cast<int4 *>(L.pcnt[j]) = make_int4(0, 0, 0, 0);
        if (lane == 63) L.sub[wave] = x;
        if (tid == 0) L.run = 0;
    }
#pragma unroll
    for (int f = 0; f < FL; ++f) {
        const int i = f * NTH + tid;
        if (i < FP) (&L.flt[0][0][0])[i] = fr[f];
    }
    S4_STAMP(1);    // plan issued (wavefront 0: ids -> node records -> descriptors; the others: ids -> filter pieces)
    s4_lds_barrier();
    S4_STAMP(2);    // plan barrier
    // the blocks' slots one after the other: lane j holds the first slot of pair j of every block
    int off[NB], myloc[NB], S = 0;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        off[q] = S;
        myloc[q] = L.loc[S4_PAIRS * q + lane] + S;
        S += L.sub[q];
    }
    // the workgroup's place in the entry buffer: its slot count bounds what it keeps (rounded to whole 128-byte lines);
    // requested now, needed when the first entries are written
    const int shard = (int)(blockIdx.x % S4_SHARDS);
    const int64_t shard_cap = (A.ent_cap / S4_SHARDS) & ~7ll;
    int64_t base_reg = 0;
    if (tid == 0)
        base_reg = (int64_t)atomicAdd(reinterpret_cast<unsigned long long *>(A.ctl + CTL_ALLOC + shard),
                                      (unsigned long long)((S + 7) & ~7));
    volatile int32_t *const sfw = &L.sflag[wave][0][0];

    for (int s0 = 0; s0 < S; s0 += NTH * S4_ROUNDS) {
        if (s0 > 0) s4_lds_barrier();   // (the previous batch's round offsets are no longer needed)
        // ---- typing: one slot per thread and round; the rounds are taken together, phase by phase, so that their memory
        //      round trips overlap -- every round's walked entry, then every round's bucket, then the arithmetic
        int code[S4_ROUNDS], node[S4_ROUNDS], win[S4_ROUNDS];
        float va[S4_ROUNDS], vb[S4_ROUNDS];
        int kindr[S4_ROUNDS], unbr[S4_ROUNDS];
        int64_t u0r[S4_ROUNDS];
        int2 cvr[S4_ROUNDS];
        bool act[S4_ROUNDS], first[S4_ROUNDS];
#pragma unroll
        for (int r = 0; r < S4_ROUNDS; ++r) {
            const int g = WAVES * r + wave;          // the wavefront's 64-slot round inside the batch
            const int r0 = s0 + 64 * g, l = r0 + lane;
            code[r] = 0; node[r] = 0; win[r] = 0; va[r] = 0.f; vb[r] = 0.f;
            act[r] = false; first[r] = false; kindr[r] = 0; unbr[r] = 0; u0r[r] = 0; cvr[r] = make_int2(0, 0);
            if (r0 < S) {                            // (wave-uniform)
                // pair of slot l = pairs that start at or before it: those at or before the round's first slot by a
                // ballot over the pairs, those inside the round by flags scattered to the slots they start in
                volatile int32_t *sf = sfw + 64 * r;
                sf[lane] = 0;
                int before = 0;
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    before += __popcll(__ballot(myloc[q] <= r0));
                    if (myloc[q] > r0 && myloc[q] < r0 + 64) sf[myloc[q] - r0] = 1;
                }
                const uint64_t starts = __ballot(sf[lane] != 0);
                if (l < S) {
                    const int w = before - 1 + __popcll(starts & le_mask);
                    const PairDesc3 &d = L.dsc[w];
                    int i = l - L.loc[w];
#pragma unroll
                    for (int q = 1; q < NB; ++q) i -= (w >> 6) == q ? off[q] : 0;
                    win[r] = w;
                    first[r] = i == 0;
                    if (i < d.total) {
                        const int k = (i >= d.w[1].start) + (i >= d.w[2].start);
                        const Walk3 wk = d.w[k];
                        cvr[r] = wk.src[i - wk.start];
                        kindr[r] = wk.kind; unbr[r] = wk.unb; u0r[r] = wk.u0;
                        act[r] = true;
                    }
                }
            }
        }
        if (s0 == 0) S4_STAMP_WAIT(3);    // walked entries arrived
        int4 bv[S4_ROUNDS][BUCKET / 2];
#pragma unroll
        for (int r = 0; r < S4_ROUNDS; ++r) {
            // what is x to the other endpoint?  one bucket of its hashed union row -- if the endpoint's mini filter lets
            // x through (the looked-up endpoint is b when the walked row is a's, and the other way round)
            const bool look = act[r] && unbr[r] > 0 &&
                              mini_pass(reinterpret_cast<const uint32_t *>(&L.flt[win[r]][(kindr[r] & KF_SRC_A) ? 1 : 0][0]), cvr[r].x);
            const uint32_t b = look ? bucket_of(cvr[r].x, unbr[r]) : 0u;
            const int4 *blk = reinterpret_cast<const int4 *>(A.u_cv + (look ? u0r[r] + BUCKET * (int64_t)b : 0));
#if defined(S4_ABL_ONEPIECE)   // (timing only, wrong results: one 16-byte piece of the bucket instead of four)
#pragma unroll
            for (int q = 0; q < BUCKET / 2; ++q) bv[r][q] = (look && q == 0) ? blk[q] : make_int4(-1, 0, -1, 0);
#elif defined(S4_ABL_NOBUCKET)
#pragma unroll
            for (int q = 0; q < BUCKET / 2; ++q) bv[r][q] = make_int4(-1, (int)b, -1, 0);
#else
#pragma unroll
            for (int q = 0; q < BUCKET / 2; ++q) bv[r][q] = look ? blk[q] : make_int4(-1, 0, -1, 0);
#endif
        }
        if (s0 == 0) S4_STAMP_WAIT(4);    // buckets arrived
        uint64_t keptb[S4_ROUNDS];
#pragma unroll
        for (int r = 0; r < S4_ROUNDS; ++r) {
            const int g = WAVES * r + wave;
            if (act[r]) {
                const Typed ty = type_slot(cvr[r].x, __int_as_float(cvr[r].y), kindr[r], bv[r], A.th_cn, A.th_1, A.th_n,
                                           A.mode_cn);
                code[r] = ty.code;
                node[r] = cvr[r].x;
                va[r] = ty.va;
                vb[r] = ty.vb;
            }
            const int c3 = code[r] & 3;
            keptb[r] = __ballot(c3 != 0);
            if (c3) atomicAdd(&L.pcnt[win[r]][c3 - 1], 1);
            if (lane == 0) L.rcnt[g] = __popcll(keptb[r]);
        }
        if (s0 == 0) S4_STAMP(5);         // arithmetic + ballots
        s4_lds_barrier();
        if (s0 == 0) S4_STAMP(6);         // typing barrier
        // ---- the rounds' places inside the block: one scan over the rounds of the batch (wavefront 0)
        if (wave == 0) {
            const int v = lane < NR ? L.rcnt[lane] : 0;
            int x = v;
#pragma unroll
            for (int dlt = 1; dlt < NR; dlt <<= 1) {
                const int y = __shfl_up(x, dlt, 64);
                if (lane >= dlt) x += y;
            }
            const int run = L.run;
            if (lane < NR) L.rcnt[lane] = run + x - v;
            if (lane == NR - 1) L.run = run + x;
            if (lane == 0 && s0 == 0) {
                L.base = (int64_t)shard * shard_cap + base_reg;
                L.ovf = base_reg + ((S + 7) & ~7) > shard_cap ? 1 : 0;
            }
        }
        s4_lds_barrier();
        if (s0 == 0) S4_STAMP(7);         // scan + barrier
        const int64_t base = L.base;
        const bool ovf = L.ovf != 0;
#pragma unroll
        for (int r = 0; r < S4_ROUNDS; ++r) {
            const int g = WAVES * r + wave;
            const int c3 = code[r] & 3;
            const int pos = L.rcnt[g < NR ? g : 0] + __popcll(keptb[r] & lt_mask);
            if (c3 && !ovf)
                A.entries[base + pos] = make_int4((int32_t)((uint32_t)(p0 + win[r]) | ((uint32_t)c3 << 29) |
                                                            ((code[r] & 4) ? FROM_B : 0u)),
                                                  node[r], __float_as_int(va[r]), __float_as_int(vb[r]));
            if (first[r]) L.pcnt[win[r]][3] = pos;   // first slot of the pair: where its entries start
        }
    }
    S4_STAMP(8);    // entries written (issued), further batches
    s4_lds_barrier();
    S4_STAMP(9);
    const bool ovf = S > 0 && L.ovf != 0;
    if (wave < NB) {    // block q's {entries, pairs with entries}: lane = pair
        const int j = S4_PAIRS * wave + lane;
        const int4 cc = *reinterpret_cast<const int4 *>(L.pcnt[j]);
        int kept = (j < np && !ovf) ? cc.x + cc.y + cc.z : 0;
        const int ne = __popcll(__ballot(kept > 0));
#pragma unroll
        for (int dlt = 32; dlt > 0; dlt >>= 1) kept += __shfl_xor(kept, dlt, 64);
        const int64_t blk = (int64_t)blockIdx.x * NB + wave;
        if (lane == 0 && blk * S4_PAIRS < A.bs) reinterpret_cast<int2 *>(A.blk_cnt)[blk] = make_int2(kept, ne);
        if (A.blk_types) {   // (wave-uniform) the block's kept entries by type: the type-major form's block bases are their sums
            int c0 = (j < np && !ovf) ? cc.x : 0, c1 = (j < np && !ovf) ? cc.y : 0, c2 = (j < np && !ovf) ? cc.z : 0;
#pragma unroll
            for (int dlt = 32; dlt > 0; dlt >>= 1) {
                c0 += __shfl_xor(c0, dlt, 64);
                c1 += __shfl_xor(c1, dlt, 64);
                c2 += __shfl_xor(c2, dlt, 64);
            }
            if (lane == 0 && blk * S4_PAIRS < A.bs) A.blk_types[blk] = make_int4(c0, c1, c2, 0);
        }
    }
    if (tid < np) {
        const int4 c = *reinterpret_cast<const int4 *>(L.pcnt[tid]);
        // (a workgroup that does not fit leaves empty pairs -- nothing is read past the buffer -- and raises the sticky bit:
        //  the scores of the batch come out as NaN and the caller sizes the workspace again)
        A.pair_tab[p0 + tid] = ovf ? make_int4(0, 0, 0, 0) : make_int4((int32_t)(L.base + c.w), c.x, c.y, c.z);
    }
    if (tid == 0) {
        if (ovf) atomicOr(reinterpret_cast<unsigned long long *>(A.ctl + CTL_ERR), (unsigned long long)LPF_SELECT_ERR_ENTRY_CAP);
        // the last workgroup leaves the counters as the next launch on this control block wants them (stream order)
        const unsigned long long done = atomicAdd(reinterpret_cast<unsigned long long *>(A.ctl + CTL_DONE), 1ull);
        if (done == (unsigned long long)gridDim.x - 1ull) {
            unsigned long long most = 0ull, sum = 0ull;
            for (int sh = 0; sh < S4_SHARDS; ++sh) {
                const unsigned long long v = atomicExch(reinterpret_cast<unsigned long long *>(A.ctl + CTL_ALLOC + sh), 0ull);
                most = v > most ? v : most;
                sum += v;
            }
            atomicExch(reinterpret_cast<unsigned long long *>(A.ctl + CTL_DONE), 0ull);
            A.ctl[0] = (int64_t)(most * S4_SHARDS);   // entries the batch needs room for (what ent_cap is sized from)
            A.ctl[CTL_SLOTS] = (int64_t)sum;          // its candidate slots (block by block rounded to 8)
        }
    }
#ifdef S4_STAMPS
    if (tid == 0 && s4_stamp_buf) {
        S4_STAMP(10);
        uint64_t *o = s4_stamp_buf + (int64_t)blockIdx.x * 16;
        for (int k = 0; k < 11; ++k) o[k] = st_t[k];
        o[11] = (uint64_t)S;
    }
#endif
}

}  // namespace

#ifdef S4_STAMPS
extern "C" int lpf_select4_set_stamps(void *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(s4_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? LPF_OK : LPF_ERR_LAUNCH;
}
#endif

/* ---- C ABI ---------------------------------------------------------------------------------------------------- */
extern "C" int lpf_select4(int64_t bs, const int64_t *batch, int64_t batch_ld, int64_t n_nodes, const void *node_rec,
                           const void *adj_cv, const void *a1_cv, const void *px_cv, const void *t0_cv, const void *u_cv,
                           const void *mini, int32_t mode_cn, int32_t use_px, float th_cn, float th_1hop,
                           float th_non1hop, int64_t *ctl, void *pair_tab, int32_t *blk_cnt, void *blk_types,
                           void *entries, int64_t ent_cap, int32_t threads, void *stream) {
    if (bs == 0) return LPF_OK;
    LPF_REQUIRE(bs > 0 && bs < (1ll << 29) && batch && batch_ld >= bs && n_nodes > 0 && node_rec && adj_cv && a1_cv &&
                (px_cv || !use_px) && u_cv && mini && ctl && pair_tab && blk_cnt && entries && ent_cap > 0 &&
                ent_cap < (1ll << 31) && lpf_aligned16(node_rec) && lpf_aligned16(u_cv) && lpf_aligned16(mini) &&
                lpf_aligned16(pair_tab) && lpf_aligned16(entries) && (reinterpret_cast<uintptr_t>(blk_cnt) & 7) == 0 &&
                lpf_aligned16(blk_types));
    Args4 a;
    a.bs = bs; a.batch = batch; a.batch_ld = batch_ld; a.n_nodes = n_nodes;
    a.rec = static_cast<const NodeRec *>(node_rec);
    a.adj_cv = static_cast<const int2 *>(adj_cv); a.a1_cv = static_cast<const int2 *>(a1_cv);
    a.px_cv = static_cast<const int2 *>(px_cv); a.t0_cv = static_cast<const int2 *>(t0_cv);
    a.u_cv = static_cast<const int2 *>(u_cv); a.mini = static_cast<const uint32_t *>(mini);
    a.mode_cn = mode_cn; a.use_px = use_px; a.th_cn = th_cn; a.th_1 = th_1hop; a.th_n = th_non1hop;
    a.ctl = ctl; a.pair_tab = static_cast<int4 *>(pair_tab); a.blk_cnt = blk_cnt;
    a.blk_types = static_cast<int4 *>(blk_types);
    a.entries = static_cast<int4 *>(entries); a.ent_cap = ent_cap;
    const int64_t nb = (bs + S4_PAIRS - 1) / S4_PAIRS;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // threads = workgroup size + 4096 * (blocks of 64 pairs per workgroup - 1); 0: the default -- 1,024 threads and two
    // blocks per workgroup while that still gives every CU a workgroup, one block otherwise (a batch of 8,192 pairs of a
    // dense graph -- ddi-like, ~1,000 slots per pair -- is 64 workgroups of 32 batches each with two: 127 us against 69)
    if (threads == 0) {
        const int n_cu = lpf_cu_count();
        if (n_cu == 0) return LPF_ERR_NO_DEVICE;
        threads = 1024 + ((nb + 1) / 2 >= n_cu ? 4096 : 0);
    }
    const int nth = threads & 4095, per = threads / 4096 + 1;
    const dim3 grid((unsigned)((nb + per - 1) / per));
    if (nth == 1024 && per == 2) hipLaunchKernelGGL((select4_kernel<1024, 2>), grid, dim3(1024), 0, s, a);
    else if (nth == 1024 && per == 1) hipLaunchKernelGGL((select4_kernel<1024, 1>), grid, dim3(1024), 0, s, a);
    else if (nth == 1024 && per == 4) hipLaunchKernelGGL((select4_kernel<1024, 4>), grid, dim3(1024), 0, s, a);
    else if (nth == 512 && per == 1) hipLaunchKernelGGL((select4_kernel<512, 1>), grid, dim3(512), 0, s, a);
    else if (nth == 512 && per == 2) hipLaunchKernelGGL((select4_kernel<512, 2>), grid, dim3(512), 0, s, a);
    else if (nth == 256 && per == 1) hipLaunchKernelGGL((select4_kernel<256, 1>), grid, dim3(256), 0, s, a);
    else return LPF_ERR_INVALID;
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

/* ---- pair-major -> type-major ---------------------------------------------------------------------------------- */
namespace {

struct ArgsR {
    int64_t bs;
    const int4 *pair_tab, *blk_types, *entries4;
    int64_t ent_cap4;
    int32_t *type_ptr;      // [3][bs + 1]
    int4 *regions;          // [3][ent_cap]
    int64_t ent_cap;
    int64_t *ctl;
};

// One wavefront per block of 64 pairs (lane = pair).  A block's entries are one contiguous run of the pair-major buffer,
// pair after pair, and the type-major regions are ordered by pair too: the k-th entry of type T of the run -- in run
// order -- is entry  base_T(block) + k  of region T, base_T = the kept type-T entries of all blocks in front.  So the
// per-pair pointers are an in-wave scan on top of the block bases, and the entries move with ballot ranks and three
// running counters -- no search, no per-pair loop, a hub pair of hundreds of entries is walked by 64 lanes.
__global__ __launch_bounds__(64) void s4_regions_kernel(const ArgsR A) {
    const int lane = threadIdx.x & 63;
    const int64_t nblk = (A.bs + S4_PAIRS - 1) / S4_PAIRS;
    const int64_t B = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (B >= nblk) return;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    // kept entries, by type, of the blocks in front
    // (eight requests of a lane in flight at a time: 512 blocks -- a 32,768-pair batch -- are ONE round trip; a loop of
    //  single requests waited for each in turn, 10 us for the last blocks of a launch that moves a few thousand records)
    int64_t base[3] = {0, 0, 0};
    const int4 e_raw = B * S4_PAIRS + lane < A.bs ? A.pair_tab[B * S4_PAIRS + lane] : make_int4(0, 0, 0, 0);
    for (int64_t i0 = 0; i0 < B; i0 += 64 * 8) {
        int4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = i0 + 64 * u + lane;
            v[u] = A.blk_types[i < B ? i : 0];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + 64 * u + lane < B) {
                base[0] += v[u].x > 0 ? v[u].x : 0;
                base[1] += v[u].y > 0 ? v[u].y : 0;
                base[2] += v[u].z > 0 ? v[u].z : 0;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int dlt = 32; dlt > 0; dlt >>= 1) base[t] += __shfl_xor((long long)base[t], dlt, 64);
    const int64_t p = B * S4_PAIRS + lane;
    const bool in = p < A.bs;
    int4 e = e_raw;      // (requested beside the block sums)
    // (a table entry that does not lie inside the buffer counts as empty: nothing is read outside it)
    if (e.x < 0 || e.y < 0 || e.z < 0 || e.w < 0 || (int64_t)e.x + e.y + e.z + e.w > A.ent_cap4) e = make_int4(0, 0, 0, 0);
    int c[3] = {e.y, e.z, e.w}, x[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        int v = c[t];
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const int y = __shfl_up(v, dlt, 64);
            if (lane >= dlt) v += y;
        }
        x[t] = v;
        if (in) A.type_ptr[(int64_t)t * (A.bs + 1) + p] = (int32_t)(base[t] + v - c[t]);
    }
    int tot[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) tot[t] = __shfl(x[t], 63, 64);
    if (B == nblk - 1 && lane == 0) {
        bool over = false;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int64_t all = base[t] + tot[t];
            A.type_ptr[(int64_t)t * (A.bs + 1) + A.bs] = (int32_t)(all < 0x7fffffff ? all : 0x7fffffff);
            A.ctl[4 + t] = all;
            over = over || all > A.ent_cap;
        }
        if (over) atomicOr(reinterpret_cast<unsigned long long *>(A.ctl + CTL_ERR), (unsigned long long)LPF_SELECT_ERR_ENTRY_CAP);
    }
    // the block's run: from the first entry of its first pair with entries
    const uint64_t has = __ballot(c[0] + c[1] + c[2] > 0);
    if (!has) return;
    const int64_t start = __shfl(e.x, __builtin_ctzll(has), 64);
    const int n_run = tot[0] + tot[1] + tot[2];
    int run[3] = {0, 0, 0};
    for (int i0 = 0; i0 < n_run; i0 += 64) {
        const int i = i0 + lane;
        int4 rec = make_int4(0, 0, 0, 0);
        if (i < n_run) rec = A.entries4[start + i];
        const int ty = i < n_run ? (int)(((uint32_t)rec.x >> 29) & 3u) : 0;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const uint64_t m = __ballot(ty == t + 1);
            if (ty == t + 1) {
                const int64_t dst = base[t] + run[t] + __popcll(m & lt_mask);
                if (dst < A.ent_cap)      // (else: the sticky bit above -- the consumers write NaN rows)
                    A.regions[(int64_t)t * A.ent_cap + dst] =
                        make_int4((int32_t)((uint32_t)rec.x & ~(3u << 29)), rec.y, rec.z, rec.w);
            }
            run[t] += __popcll(m);
        }
    }
}

}  // namespace

extern "C" int lpf_select4_regions(int64_t bs, const void *pair_tab, const void *blk_types, const void *entries4,
                                   int64_t ent_cap4, int32_t *type_ptr, void *regions, int64_t ent_cap, int64_t *ctl,
                                   void *stream) {
    if (bs == 0) return LPF_OK;
    LPF_REQUIRE(bs > 0 && bs < (1ll << 29) && pair_tab && blk_types && entries4 && ent_cap4 > 0 && type_ptr && regions &&
                ent_cap > 0 && ent_cap < (1ll << 29) && ctl && lpf_aligned16(pair_tab) && lpf_aligned16(blk_types) &&
                lpf_aligned16(entries4) && lpf_aligned16(regions));
    const ArgsR a{bs, static_cast<const int4 *>(pair_tab), static_cast<const int4 *>(blk_types),
                  static_cast<const int4 *>(entries4), ent_cap4, type_ptr, static_cast<int4 *>(regions), ent_cap, ctl};
    const int64_t nblk = (bs + S4_PAIRS - 1) / S4_PAIRS;
    // (a wavefront per workgroup: nothing is shared, and 512 single wavefronts spread over every CU)
    hipLaunchKernelGGL(s4_regions_kernel, dim3((unsigned)nblk), dim3(64), 0, static_cast<hipStream_t>(stream), a);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}
