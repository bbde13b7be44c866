// Pairwise PPR-positional attention, PAIR-MAJOR: one finished feature row per candidate pair.
//
// Reference: LinkAttention.message + PyG softmax + scatter-sum + post_att_norm (src/modules/layers.py:66-78,193-224)
// with get_pos_encodings (src/models/link_transformer.py:182-211) folded in, evaluated through the activation pattern
// of the PE hidden layer exactly as in pair_flip.hip (same tables, same per-entry arithmetic).
//
// What is different from pair_flip.hip is the ORDER the entries are walked in and what leaves the kernel.  There a
// group of G = D/4 lanes walks a unit of 16 consecutive entries of ONE TYPE REGION and leaves online-softmax records (one
// per (pair, type) segment inside the unit, boundary records for segments that cross units) which the dense tail reads
// back and merges -- on the collab-like batch half of the pairs have ~22 selected entries, two thirds of the segments
// cross a unit, a pair costs 1.7 records of D + 4 floats written and re-read, and the tail spent a third of its time
// chasing and merging them before its first matrix instruction.  Here the entries are walked PAIR-MAJOR -- a pair's
// common-neighbour, one-hop and >1-hop segments one after the other: the three segment pointers are prefix sums, so
// C[p] = sum_t ptr_t[p] IS the pair-major position of pair p's first entry -- and what leaves the kernel is the pair's
// finished row
//     out[p] = [ post_att_norm( sum_e alpha_e k_e + bias ) | n_cn, n_1hop, (n_non1hop,) n_cn + n_1hop ],
// D + c floats written once, read once: no merge and no LayerNorm left for the consumer.
//
// Work: a workgroup owns a contiguous range of pairs holding 1/gridDim.x of (entries + pairs) -- found by a 64-ary
// search over the pointers -- and stages the range's pointers in LDS (PR_CHUNK pairs at a time).  The unit of work stays
// what it is in pair_flip.hip, 16 consecutive entries -- now of the pair-major order --, handed out inside the
// workgroup by a ticket in LDS: a pair of 500 entries is 32 units on 32 groups, a round is always full, no lane waits
// for a longer neighbour.  A pair that lies inside one unit is finished in registers (online softmax -> bias ->
// LayerNorm -> row).  A pair that crosses unit boundaries leaves one partial state per unit it touches in a scratch
// buffer (slot 1 of the unit it starts in, slot 0 of every later one); after the workgroup's units are done -- one
// barrier -- its groups merge those pieces in unit order and finish the rows.  The pieces are written and read by the
// same CU microseconds apart: they never leave L2.  A pair's row does not depend on the batch around it except through
// where the 16-entry grid cuts it (the same dependence pair_flip.hip has).
//
// Two forms share the kernel.  Type-major (behind select3): three entry regions + type_ptr, the base vectors of ONE
// activation pattern (the one of (0, 0)), every entry outside the no-flip square looks at its 2 D hidden units and
// corrects the flipped ones from an LDS copy of Wfold^T.  PT (behind select4, the hot path): entries pair-major already
// (pair_tab / blk_cnt), the activation patterns of an entry's two argument orders come from a TABLE over the plane of PPR
// value pairs (lpformer_amd/patterns.py: a log grid whose cells are proven to hold one pattern; the base vectors of
// the tabulated patterns in LDS; only entries of flagged cells look at their units, starting from the pattern the cell
// names), and at D >= 128 a lane holds two feature quads (NV = 2: 16 / 32 lanes per entry).
#include <type_traits>

#include "pe_common.h"
#include "select_common.h"   // the chained scan (lb_*): start of every workgroup's pairs in the order of the dense tail

namespace {

constexpr uint32_t PR_PAIR_MASK = 0x7fffffffu;
constexpr int PR_LB_WORDS = LPF_ROWS_PERM_LB_WORDS - 1;   // scan words of the perm order (>= the largest grid: 3 workgroups x 256+ CUs)
#ifndef PR_CHUNK_N    /* (tuning aids: -DPR_CHUNK_N / -DPR_FLAGS_N / -DPR_NP shrink the LDS image -- co-residency probes) */
#define PR_CHUNK_N 512
#endif
#ifndef PR_FLAGS_N
#define PR_FLAGS_N 2048
#endif
constexpr int PR_CHUNK = PR_CHUNK_N;   // pairs of the workgroup's range staged in LDS at a time
constexpr int PR_FLAGS = PR_FLAGS_N;   // units of a chunk whose completion is tracked in LDS (more: one barrier, then the merges)

struct RowsArgs {
    int64_t bs;
    const int32_t *type_ptr;   // [3][bs+1]
    const int4 *entries;       // [3][ent_cap]
    int64_t ent_cap;
    const float *Z; uint32_t ldz;   // (ZB: bf16 rows, ldz in bf16 elements)
    const float *q; uint32_t ldq;
    const float *pe_tab;       // [3][D][4]   (ta, tc, td, beta) per hidden unit, times +1 (unit in S0) or -1
    const float *pe_stat;      // [3][8]
    const float *base;         // [3][4][D]   P0, Q0, R0, C0 = 2 B0 + bfold
    const float *wfoldT;       // [3][D][D]   wfoldT[t][k][c] = Wfold_t[c][k]
    const float *att;          // [D]
    const float *att_bias, *ln_g, *ln_b;   // [D] each: LinkAttention.bias, post_att_norm
    float *out; int64_t ldo;   // [bs][ldo]: D features, then n_counts count features
    int32_t n_counts;
    const int64_t *sel_ctl;    // selection control block: word 3 != 0 => the batch did not fit its workspace, rows = NaN
    float *pieces;             // [units_cap][2][RSP]: partial states of the pairs that cross 16-entry units
    int64_t units_cap;
    // optional: the order in which the dense tail takes the pairs -- those WITH selected nodes first (ascending), those
    // without behind them (filled from the back) -- so that whole tiles of the tail hold pairs of one kind.  Deterministic
    // (a chained scan over the workgroups' counts, not atomics): a replayed step stays bitwise the eager one.
    int32_t *perm;             // [bs]
    uint64_t *perm_lb;         // [PR_LB_WORDS + 1] chained-scan words (tagged with the launch number, never cleared) + the
                               // launch counter
    int64_t *n_nonempty;       // receives the number of pairs with selected nodes
    // PT (pair-table) form -- the selection of select4.hip: entries pair-major already, the type in the record
    const int4 *pair_tab;      // [bs] {first entry, n_cn, n_1hop, n_non1hop}
    const int32_t *blk_cnt;    // [ceil(bs / LPF_SELECT4_BLOCK)] entries per block of pairs
    // PT form: activation patterns by table (lpformer_amd/patterns.py).  The plane of PPR value pairs is cut into
    // grid_n x grid_n cells, cell(v) = (bits(v + grid_ofs) >> grid_shift) - grid_base; a cell holds the id of the pattern
    // that is provably the pattern of every point of the cell or 0x80 (then the entry takes the exact detect-and-correct
    // path against pattern 0, as every entry outside the no-flip square does in the type-major form)
    const float *pat_base;     // [3][LPF_ROWS_PATTERNS][4][D]: (P, Q, R, B + bfold / 2) of pattern s of type t
    const uint8_t *pat_grid;   // [3][grid_n][grid_n]: pattern id (bits 0-4); bit 7: a boundary may cross the cell or its
                               // pattern is not tabulated -- the exact path, starting from the NAMED (nearest) pattern
    const uint32_t *pat_sign;  // [3][LPF_ROWS_PATTERNS][D / 32]: bit k = unit k's state in the pattern differs from pattern 0
    int32_t grid_n, grid_shift, grid_base;
    float grid_ofs;
};

template <int CTRL>
__device__ __forceinline__ float pr_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over the G lanes of a group, the same bits in every lane (pair_flip.hip::fl_group_sum)
template <int G>
__device__ __forceinline__ float pr_group_sum(float v) {
    v += pr_dpp<0xB1>(v);
    v += pr_dpp<0x4E>(v);
    if constexpr (G >= 8) v += pr_dpp<0x141>(v);
    if constexpr (G >= 16) v += pr_dpp<0x140>(v);
    if constexpr (G >= 32)
        v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401f));
    if constexpr (G >= 64) {
        const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
        const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
        v = a + b;
    }
    return v;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Workgroup barrier that orders LDS only.  __syncthreads() is a workgroup fence + barrier and the fence drains the
// vector-memory counter: behind a loop of global stores (count features, constant rows, finished rows) every barrier of
// the set-up would wait out a store round trip.  The wavefronts of a workgroup hand each other nothing through global
// memory here except the pieces, which have their own wait + flag.
__device__ __forceinline__ void pr_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// (tuning builds, -DPR_STAMPS: wall-clock marks -- 100 MHz -- of lane 0 of every wavefront; tools/rows_stamps.py)
#ifdef PR_STAMPS
__device__ uint64_t *pr_stamp_buf = nullptr;
#define PR_STAMP(k) do { if (lane == 0) st_t[k] = wall_clock64(); } while (0)
#else
#define PR_STAMP(k) do { } while (0)
#endif

// floats of one piece record: D accumulators, m, l, padded to whole 128-byte lines (no line is shared by two records:
// neighbouring units may belong to different workgroups)
constexpr int pr_piece_floats(int D) { return (D + 4 + 31) / 32 * 32; }

// LDS carve-up, shared by the kernel and the launcher (float4 units unless noted)
// (PT form: NP patterns per type instead of one, and no copy of Wfold^T -- its corrections are the exception there)
#ifdef PR_NP
constexpr int pr_patterns(int G, bool PT) { return !PT ? 1 : (PR_NP); }
#else
constexpr int pr_patterns(int G, bool PT) { return !PT ? 1 : (G >= 64 ? LPF_ROWS_PATTERNS / 2 : LPF_ROWS_PATTERNS); }
#endif

template <int G, int NTH, int WTL, bool PT = false, int NV = 1>
struct PrLds {
    static constexpr int GG = G * NV, D = 4 * GG, NG = (NTH / 64) * (64 / G), NP = pr_patterns(GG, PT);
    static constexpr int TAB = 0, BASE = TAB + 3 * D, VEC = BASE + 3 * NP * D, CONST_ROW = VEC + 3 * GG, STAT = CONST_ROW + GG;
    static constexpr int WT = STAT + 6;                      // (pe_stat: 3 x 8 floats)
    static constexpr int REC = WT + (PT ? 0 : WTL) * D * GG; // int4 [NG][16]
    static constexpr int SC = REC + NG * 16;                 // f32x2 [NG][16]  (NG * 8 float4)
    static constexpr int META = SC + NG * 8;                 // int [NG][16]    (NG * 4 float4)
    static constexpr int TP = META + NG * 4;                 // int [3][PR_CHUNK + 4]
    static constexpr int CUM = TP + 3 * (PR_CHUNK / 4 + 1);  // int [PR_CHUNK + 4]: pair-major start of every pair, chunk-relative
    static constexpr int LISTS = CUM + (PR_CHUNK / 4 + 1);   // int [2][PR_CHUNK]: empty pairs, pairs in several pieces
    static constexpr int CTL = LISTS + 2 * (PR_CHUNK / 4);   // int [16]: counters, ticket, range
    static constexpr int FLAGS = CTL + 4;                    // int [PR_FLAGS]: unit u of the chunk is done
    static constexpr int WCNT = FLAGS + PR_FLAGS / 4;        // int [32]: per-wavefront counts (ranks of the tail's order)
    static constexpr int SIGN = WCNT + 8;                    // PT: uint32 [3][NP][D / 32] (pat_sign)
    static constexpr int TOTAL = SIGN + (PT ? (3 * NP * ((D + 31) / 32) + 3) / 4 : 0);
    static constexpr size_t BYTES = (size_t)TOTAL * 16;
};

// ZB: the node table Z is stored in bf16 (the bf16 throughput mode)
// PT: the selection came from select4.hip (pair_tab / blk_cnt instead of the three type-major regions and type_ptr)
// NV: float4 vectors per lane -- a group of G lanes owns an entry, lane j its features (and hidden units) 4 (j + v G) ...
//     + 3, v < NV; D = 4 G NV.  NV = 2 at D = 128 behind select4: four entries per wavefront step instead of two, i.e.
//     half the per-entry work that does not depend on the feature (records, softmax scalars, reductions, branches)
template <int G, int NTH, int WTL_, bool ZB = false, bool PT = false, int NV = 1>
__global__ __launch_bounds__(NTH, NTH >= 512 ? 4 : 3) void pair_rows_kernel(const RowsArgs A) {
    using ZT = typename std::conditional<ZB, uint2, float4>::type;
    using L = PrLds<G, NTH, WTL_, PT, NV>;
    constexpr int WTL = PT ? 0 : WTL_, NP = L::NP, GG = L::GG;
    constexpr int D = 4 * GG, EPW = 64 / G, NG = L::NG, T_LO = WTL == 1 ? 1 : 0, TPS = PR_CHUNK + 4, RSP = pr_piece_floats(D);
    constexpr uint32_t PAIR_MASK = PT ? 0x1fffffffu : PR_PAIR_MASK;
    constexpr int SB = LPF_SELECT4_BLOCK;
    // PT: what a workgroup's share is balanced by is  EW * entries + pairs  -- the units of 16 entries are what takes time
    // (a workgroup walks them in rounds of NG: one unit more than a multiple of NG is a whole round more), a pair without
    // entries costs a row store
    constexpr int EW = 4;
    extern __shared__ __attribute__((aligned(16))) float4 pr_lds[];
    float4 *const ltab = pr_lds + L::TAB;        // [4][3][GG]: row j of hidden unit 4 q + j, type t -> ((j * 3 + t) * GG + q)
    float4 *const lbase = pr_lds + L::BASE;      // [3][NP][4][GG]
    float4 *const lvec = pr_lds + L::VEC;        // [3][GG]: att_bias, ln_g, ln_b by feature quad
    float4 *const lconst = pr_lds + L::CONST_ROW;   // [GG]: the row of a pair without entries
    float *const lstat = reinterpret_cast<float *>(pr_lds + L::STAT);   // [3][8]
    float4 *const lwt = pr_lds + L::WT;          // [WTL][D][GG]
    int4 *const lrec = reinterpret_cast<int4 *>(pr_lds + L::REC);
    f32x2 *const lsc = reinterpret_cast<f32x2 *>(pr_lds + L::SC);
    int *const lmeta = reinterpret_cast<int *>(pr_lds + L::META);
    int *const ltp = reinterpret_cast<int *>(pr_lds + L::TP);
    int *const lcum = reinterpret_cast<int *>(pr_lds + L::CUM);
    int *const llist = reinterpret_cast<int *>(pr_lds + L::LISTS);
    int *const lctl = reinterpret_cast<int *>(pr_lds + L::CTL);   // 0: n_empty 1: n_multi 2: unit ticket 4,5: P0 6,7: P1 8-11: the order (base, own count, launch number)
    int *const lflag = reinterpret_cast<int *>(pr_lds + L::FLAGS);
    int *const lwcnt = reinterpret_cast<int *>(pr_lds + L::WCNT);
    uint32_t *const lsign = reinterpret_cast<uint32_t *>(pr_lds + L::SIGN);
    constexpr int SW = (D + 31) / 32;            // words of a pattern's sign vector
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane / G, lj = lane % G, off = 4 * lj;
    const int gid = wave * EPW + grp;            // this group among the workgroup's NG
#ifdef PR_STAMPS
    uint64_t st_t[8] = {0};
    int st_rounds = 0;
    PR_STAMP(0);
#endif
    int64_t n[3] = {0, 0, 0};
    if constexpr (!PT) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            n[t] = A.type_ptr[(int64_t)t * (A.bs + 1) + A.bs];
            if (n[t] > A.ent_cap) n[t] = A.ent_cap;   // (overflow: flagged by the selection kernel, stay inside the region)
            if (n[t] < 0) n[t] = 0;
        }
    }
    // PT: entries of pair p (a pair whose table entry does not lie inside the buffer counts as empty: nothing is read
    // outside it whatever the table holds)
    auto pt_count = [&](int64_t p) __attribute__((always_inline)) {
        const int4 e = A.pair_tab[p];
        const int64_t c = (int64_t)e.y + e.z + e.w;
        return (e.x < 0 || e.y < 0 || e.z < 0 || e.w < 0 || e.x + c > A.ent_cap) ? 0 : (int)c;
    };
    bool bad = A.sel_ctl && A.sel_ctl[3] != 0;
    // PT, wavefronts 0 / 1 (lower / upper end of the workgroup's range): the blocks' {entries, pairs with entries} are
    // requested first -- lane l owns the blocks [l per, (l + 1) per) -- and travel beside the table fill below
    const int64_t nblk = (A.bs + SB - 1) / SB, per = (nblk + 63) / 64;
    auto blk_at = [&](int64_t B) __attribute__((always_inline)) {
        int2 c = reinterpret_cast<const int2 *>(A.blk_cnt)[B];
        const int np = (int)(A.bs - B * SB < SB ? A.bs - B * SB : SB);
        c.x = c.x < 0 ? 0 : c.x;
        c.y = c.y < 0 ? 0 : (c.y > np ? np : c.y);
        return make_int4(c.x, c.y, np, 0);
    };
    int64_t own_w = 0;
    int own_ne = 0;
    constexpr int BLK_KEEP = 8;          // the first blocks of a lane's share stay in registers for the walk below
    int2 kept_blk[BLK_KEEP];             // {weight, pairs with entries}
    if constexpr (PT) {
        if (wave < 2) {
            // (all of them requested before the first is looked at: inside their own branches the compiler waited for
            //  every one in turn -- eight round trips in front of the search)
            int2 raw_blk[BLK_KEEP];
#pragma unroll
            for (int i = 0; i < BLK_KEEP; ++i) {
                const int64_t B = (int64_t)lane * per + i;
                raw_blk[i] = reinterpret_cast<const int2 *>(A.blk_cnt)[B < nblk ? B : nblk - 1];
            }
#pragma unroll
            for (int i = 0; i < BLK_KEEP; ++i) {
                const int64_t B = (int64_t)lane * per + i;
                const int np = (int)(A.bs - B * SB < SB ? A.bs - B * SB : SB);
                const int cx = raw_blk[i].x < 0 ? 0 : raw_blk[i].x;
                const int cy = raw_blk[i].y < 0 ? 0 : (raw_blk[i].y > np ? np : raw_blk[i].y);
                kept_blk[i] = (i < per && B < nblk) ? make_int2(EW * cx + np, cy) : make_int2(0, 0);
            }
#pragma unroll
            for (int i = 0; i < BLK_KEEP; ++i) { own_w += kept_blk[i].x; own_ne += kept_blk[i].y; }
            for (int64_t i = BLK_KEEP; i < per; ++i) {
                const int64_t B = (int64_t)lane * per + i;
                if (B < nblk) {
                    const int4 c = blk_at(B);
                    own_w += (int64_t)EW * c.x + c.z;
                    own_ne += c.y;
                }
            }
        }
    }
    {   // the tables: every load of a thread in flight before its first store (filling them with the wavefronts that do
        // not look for the range only was measured: 9.4 us to the first barrier against 7.6)
        constexpr int FT = NTH;
        const int ft = tid;
        constexpr int N3 = (3 * D + FT - 1) / FT;
        float4 ta[N3], tb[N3];
#pragma unroll
        for (int u = 0; u < N3; ++u) {
            const int i = u * FT + ft;
            if (i < 3 * D) {
                ta[u] = reinterpret_cast<const float4 *>(A.pe_tab)[i];
                if constexpr (!PT) tb[u] = reinterpret_cast<const float4 *>(A.base)[i];
            }
        }
        constexpr int NW = WTL > 0 ? (WTL * D * GG + FT - 1) / FT : 1;
        float4 tw[NW];
        if constexpr (WTL > 0) {
            const float4 *src = reinterpret_cast<const float4 *>(A.wfoldT) + (int64_t)T_LO * D * GG;
#pragma unroll
            for (int u = 0; u < NW; ++u) {
                const int i = u * FT + ft;
                if (i < WTL * D * GG) tw[u] = src[i];
            }
        }
        constexpr int NVF = (3 * GG + FT - 1) / FT;
        float4 tv[NVF];
#pragma unroll
        for (int u = 0; u < NVF; ++u) {
            const int i = u * FT + ft;
            if (i < 3 * GG) {
                const float *src = i < GG ? A.att_bias : (i < 2 * GG ? A.ln_g : A.ln_b);
                tv[u] = *reinterpret_cast<const float4 *>(src + 4 * (i % GG));
            }
        }
        const float ts = ft < 24 ? A.pe_stat[ft] : 0.f;
#pragma unroll
        for (int u = 0; u < N3; ++u) {
            const int i = u * FT + ft;
            if (i < 3 * D) {
                const int t = i / D, k = i % D;
                ltab[((k & 3) * 3 + t) * GG + (k >> 2)] = ta[u];
                if constexpr (!PT) lbase[i] = tb[u];
            }
        }
        if constexpr (PT) {
            if (ft < 3 * NP * SW)
                lsign[ft] = A.pat_sign[((ft / (NP * SW)) * LPF_ROWS_PATTERNS + (ft / SW) % NP) * SW + ft % SW];
            // the pattern tables: [3][LPF_ROWS_PATTERNS][4][G] float4 in memory, the first NP patterns of every type kept
            constexpr int PB = 3 * NP * D, PBU = 6;   // (six loads of a thread in flight at a time)
            const float4 *src = reinterpret_cast<const float4 *>(A.pat_base);
            for (int i0 = 0; i0 < PB; i0 += PBU * FT) {
                float4 tp[PBU];
#pragma unroll
                for (int u = 0; u < PBU; ++u) {
                    const int i = i0 + u * FT + ft;
                    if (i < PB) tp[u] = src[(i / (NP * D)) * (LPF_ROWS_PATTERNS * D) + i % (NP * D)];
                }
#pragma unroll
                for (int u = 0; u < PBU; ++u) {
                    const int i = i0 + u * FT + ft;
                    if (i < PB) lbase[i] = tp[u];
                }
            }
        }
        if constexpr (WTL > 0) {
#pragma unroll
            for (int u = 0; u < NW; ++u) {
                const int i = u * FT + ft;
                if (i < WTL * D * GG) lwt[i] = tw[u];
            }
        }
#pragma unroll
        for (int u = 0; u < NVF; ++u) {
            const int i = u * FT + ft;
            if (i < 3 * GG) lvec[i] = tv[u];
        }
        if (ft < 24) lstat[ft] = ts;
    }

    // pair-major position of pair p's first entry (the pointers clamped into their regions)
    auto cum_at = [&](int64_t p) __attribute__((always_inline)) {
        int64_t c = 0;
        if constexpr (!PT) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int64_t v = A.type_ptr[(int64_t)t * (A.bs + 1) + p];
                c += v < n[t] ? (v < 0 ? 0 : v) : n[t];
            }
        }
        return c;
    };
    // does pair p select anything?
    auto nonempty = [&](int64_t p) __attribute__((always_inline)) {
        if constexpr (PT) return pt_count(p) > 0;
        else return cum_at(p + 1) > cum_at(p);
    };
    // ---- the workgroup's pair range: first pair p with  p + C[p]  >= b * (entries + pairs) / gridDim.x
    if constexpr (PT) {
        // The selection left {entries, pairs with entries} per BLOCK of 64 pairs.  One wavefront per end, no LDS and no
        // barrier: a scan over the lanes' shares of the blocks finds the share a target falls into, its lane walks it,
        // then the pair inside that block comes from the block's 64 table entries -- two dependent round trips.
        if (wave < 2) {
            int64_t x = own_w;
            int xn = own_ne;
#pragma unroll
            for (int dlt = 1; dlt < 64; dlt <<= 1) {
                const int64_t y = __shfl_up(x, dlt, 64);
                const int yn = __shfl_up(xn, dlt, 64);
                if (lane >= dlt) { x += y; xn += yn; }
            }
            const int64_t total = __shfl(x, 63, 64);
            const int total_ne = __shfl(xn, 63, 64);
            const int64_t b = (int64_t)blockIdx.x + wave;
            const int64_t target = (total * b) / (int64_t)gridDim.x;
            int64_t P = 0, C = 0;
            int NE = 0;
            if (b >= (int64_t)gridDim.x) {
                P = A.bs; C = (total - A.bs) / EW; NE = total_ne;
            } else if (b > 0 && target > 0) {   // (target 0: fewer entries + pairs than workgroups, nothing in front)
                // last block B with (weight in front of B) < target: it lies in the share with ex < target <= ex + own
                const int64_t ex = x - own_w;
                const uint64_t holder = __ballot(ex < target && target <= x);
                const int hl = holder ? __builtin_ctzll(holder) : 0;
                int64_t B = (int64_t)lane * per, f = ex;
                int nef = xn - own_ne;
                if (lane == hl) {
                    // (the share's first blocks from registers -- no round trip per step; the rest, if any, from memory)
                    const int64_t lim = (int64_t)(lane + 1) * per < nblk ? (int64_t)(lane + 1) * per : nblk;
                    bool stop = false;
#pragma unroll
                    for (int i = 0; i < BLK_KEEP; ++i) {
                        if (!stop && B + 1 < lim) {
                            const int64_t fn = f + kept_blk[i].x;
                            if (fn >= target) stop = true;
                            else { f = fn; nef += kept_blk[i].y; ++B; }
                        }
                    }
                    while (!stop && B + 1 < lim) {
                        const int4 c = blk_at(B);
                        const int64_t fn = f + (int64_t)EW * c.x + c.z;
                        if (fn >= target) break;
                        f = fn;
                        nef += c.y;
                        ++B;
                    }
                }
                B = __shfl(B, hl, 64);
                f = __shfl(f, hl, 64);
                nef = __shfl(nef, hl, 64);
                const int64_t p = B * SB + lane;
                const int cnt = p < A.bs ? pt_count(p) : 0;
                int xs = cnt;     // inclusive scan of the entries
#pragma unroll
                for (int dlt = 1; dlt < 64; dlt <<= 1) {
                    const int y = __shfl_up(xs, dlt, 64);
                    if (lane >= dlt) xs += y;
                }
                // f(p) = f + lane + EW * (entries of the block's pairs in front of p)
                const bool below = p < A.bs && f + lane + (int64_t)EW * (xs - cnt) < target;
                const uint64_t bm = __ballot(below);      // (monotone: the first c pairs are below the target)
                const int c = __popcll(bm);
                const int before = c == 0 ? 0 : __shfl(xs, c - 1, 64);
                P = B * SB + c;
                // entries in front of P: in front of block B (f = EW * entries + pairs = EW * entries + B * SB) + inside it
                C = (f - B * SB) / EW + before;
                NE = nef + __popcll(__ballot(below && cnt > 0));
            }
            if (lane == 0) {
                lctl[4 + 2 * wave] = (int)(P & 0xffffffff); lctl[5 + 2 * wave] = (int)(P >> 32);
                lctl[12 + 2 * wave] = (int)(C & 0xffffffff); lctl[13 + 2 * wave] = (int)(C >> 32);
                if (wave == 0) {
                    lctl[8] = NE; lctl[9] = 0; lctl[10] = total_ne;   // pairs with entries in front of the range, in all
                    // more entries than the scratch for the pieces has units for: nothing is walked, the rows are NaN and
                    // the sticky bit tells the caller to size the workspace again (as when the selection does not fit)
                    const bool over = (total - A.bs) / EW > 16 * (A.units_cap - 1);
                    lctl[11] = over ? 1 : 0;
                    if (over && blockIdx.x == 0 && A.sel_ctl)
                        atomicOr(reinterpret_cast<unsigned long long *>(const_cast<int64_t *>(A.sel_ctl) + 3),
                                 (unsigned long long)LPF_SELECT_ERR_ENTRY_CAP);
                }
            }
        }
    } else
    // (type-major form: 64-ary search over the pointers, three dependent rounds for 32 k pairs; wavefront 0 looks for
    //  the lower end, wavefront 1 for the upper one, the others fill the tables above meanwhile)
    if (wave < 2) {
        const int64_t total = n[0] + n[1] + n[2] + A.bs;
        const int64_t b = (int64_t)blockIdx.x + wave;
        int64_t lo = 0, hi = A.bs;   // answer in [lo, hi]
        if (b >= (int64_t)gridDim.x) {
            lo = A.bs;
        } else if (b > 0) {
            const int64_t target = (total * b) / (int64_t)gridDim.x;
            while (hi - lo > 64) {
                const int64_t step = (hi - lo + 63) / 64;
                const int64_t idx = lo + (int64_t)lane * step;
                const bool below = idx < hi && idx + cum_at(idx < hi ? idx : 0) < target;
                const int c = __popcll(__ballot(below));   // (monotone: the first c probes are below the target)
                if (c == 0) { hi = lo; break; }
                const int64_t nlo = lo + (int64_t)(c - 1) * step + 1;
                const int64_t nhi = lo + (int64_t)c * step;
                lo = nlo;
                hi = nhi < hi ? nhi : hi;
            }
            if (hi > lo) {
                const int64_t idx = lo + lane;
                const bool below = idx < hi && idx + cum_at(idx < hi ? idx : 0) < target;
                lo += __popcll(__ballot(below));
            }
        }
        if (lane == 0) { lctl[4 + 2 * wave] = (int)(lo & 0xffffffff); lctl[5 + 2 * wave] = (int)(lo >> 32); }
    }
#ifdef PR_STAMPS
    if (lane == 0) st_t[7] = wall_clock64();      // this wavefront's own work before the first barrier is done
#endif
    __syncthreads();
    PR_STAMP(1);
    const int64_t P0 = (int64_t)(uint32_t)lctl[4] | ((int64_t)lctl[5] << 32);
    const int64_t P1 = (int64_t)(uint32_t)lctl[6] | ((int64_t)lctl[7] << 32);
    int64_t vpos = PT ? ((int64_t)(uint32_t)lctl[12] | ((int64_t)lctl[13] << 32)) : 0;   // PT: pair-major position of the chunk
    const bool over = PT && lctl[11] != 0;
    bad = bad || over;

    // ---- the tail's order, first half: publish how many pairs of this workgroup's range select anything (word
    //      blockIdx.x of perm_lb, tagged with this launch's number: word PR_LB_WORDS counts the launches).  The second half
    //      -- the sum over the workgroups in front, then the order itself -- waits until the rows are done: by then every
    //      predecessor's word has long been out, nobody spins on a workgroup that is not resident yet.
    if (!PT && A.perm) {
        int cnt = 0;
        for (int64_t k = P0 + tid; k < P1; k += NTH) cnt += nonempty(k) ? 1 : 0;
#pragma unroll
        for (int dlt = 32; dlt > 0; dlt >>= 1) cnt += __shfl_xor(cnt, dlt, 64);
        if (lane == 0) lwcnt[wave] = cnt;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < NTH / 64; ++w) tot += lwcnt[w];
            const uint32_t epoch = (uint32_t)(A.perm_lb[PR_LB_WORDS] + 1ull) & ((1u << 22) - 1u);
            lb_store(A.perm_lb + blockIdx.x, epoch, 1, (uint64_t)tot);
            lctl[10] = tot;
            lctl[11] = (int)epoch;
        }
    }

    // a lane's NV feature quads as pairs of packed floats
    struct Q4 { f32x2 a, b; };
    auto q4_zero = [&](Q4 (&x)[NV]) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < NV; ++v) x[v] = Q4{f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
    };
    // finished row of a pair from its (merged) softmax state: post_att_norm(o / (l + 1e-16) + bias); quad v of the lane
    // goes to  out + row * ldo + 4 (lj + v G)
    auto finish_row = [&](const Q4 (&o)[NV], float l, float4 (&r)[NV]) __attribute__((always_inline)) {
        const float inv = __builtin_amdgcn_rcpf(l + 1e-16f);   // (1 ulp: v_rcp_f32, not the IEEE division sequence)
        float4 y[NV];
        float sum = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 vb = lvec[lj + v * G];                // (once per pair: not worth registers)
            y[v] = make_float4(o[v].a.x * inv + vb.x, o[v].a.y * inv + vb.y, o[v].b.x * inv + vb.z, o[v].b.y * inv + vb.w);
            sum += (y[v].x + y[v].y) + (y[v].z + y[v].w);
        }
        const float mean = pr_group_sum<G>(sum) * (1.0f / (float)D);
        float sq = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            y[v] = make_float4(y[v].x - mean, y[v].y - mean, y[v].z - mean, y[v].w - mean);
            sq += (y[v].x * y[v].x + y[v].y * y[v].y) + (y[v].z * y[v].z + y[v].w * y[v].w);
        }
        const float var = pr_group_sum<G>(sq) * (1.0f / (float)D);
        const float rstd = __builtin_amdgcn_rsqf(var + 1e-5f);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 vg = lvec[GG + lj + v * G], vbeta = lvec[2 * GG + lj + v * G];
            r[v] = make_float4(y[v].x * rstd * vg.x + vbeta.x, y[v].y * rstd * vg.y + vbeta.y,
                               y[v].z * rstd * vg.z + vbeta.z, y[v].w * rstd * vg.w + vbeta.w);
            if (bad) r[v] = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
        }
    };
    auto store_row = [&](int64_t pair, const float4 (&r)[NV]) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < NV; ++v) *reinterpret_cast<float4 *>(A.out + pair * A.ldo + off + 4 * G * v) = r[v];
    };
    // (the constant row of a pair without entries; PT with an order for the tail never writes it -- the tail has row_empty)
    if (!(PT && A.perm != nullptr) && wave == 0 && grp == 0) {
        Q4 zero[NV];
        q4_zero(zero);
        float4 r[NV];
        finish_row(zero, 0.f, r);
#pragma unroll
        for (int v = 0; v < NV; ++v) lconst[lj + v * G] = r[v];
    }

    f32x2 at01[NV], at23[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const float4 at = *reinterpret_cast<const float4 *>(A.att + off + 4 * G * v);
        at01[v] = f32x2{at.x, at.y};
        at23[v] = f32x2{at.z, at.w};
    }
    int4 *const lr = lrec + gid * 16;
    f32x2 *const ls = lsc + gid * 16;
    int *const lm = lmeta + gid * 16;

    auto z_row = [&](int node, ZT (&z)[NV]) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if constexpr (ZB)
                z[v] = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(A.Z) +
                                                        (uint64_t)(uint32_t)node * A.ldz + off + 4 * G * v);
            else
#ifdef PR_ABL_NOZ   /* (timing only, wrong results: every entry reads one of two rows of Z -- no random gather) */
                z[v] = *reinterpret_cast<const float4 *>(A.Z + (uint64_t)(uint32_t)(node & 1) * A.ldz + off + 4 * G * v);
#else
                z[v] = *reinterpret_cast<const float4 *>(A.Z + (uint64_t)(uint32_t)node * A.ldz + off + 4 * G * v);
#endif
        }
    };
    auto z_wide = [&](const ZT &z) __attribute__((always_inline)) {
        if constexpr (ZB)
            return make_float4(__uint_as_float(z.x << 16), __uint_as_float(z.x & 0xffff0000u),
                               __uint_as_float(z.y << 16), __uint_as_float(z.y & 0xffff0000u));
        else
            return z;
    };
    // PT: the cell of a PPR value on either axis of the pattern grid (NaN, negative, > 1: the last cell, never tabulated)
    auto grid_cell = [&](float v) __attribute__((always_inline)) {
        const int c = (int)(__float_as_uint(v + A.grid_ofs) >> A.grid_shift) - A.grid_base;
        return c < 0 ? A.grid_n - 1 : (c < A.grid_n ? c : A.grid_n - 1);
    };
    auto q_row = [&](int pair, float4 (&q)[NV]) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
            q[v] = *reinterpret_cast<const float4 *>(A.q + (uint64_t)(uint32_t)pair * A.ldq + off + 4 * G * v);
    };

    for (int64_t c0 = P0; c0 < P1; c0 += PR_CHUNK) {
        const int cn = (int)(P1 - c0 < PR_CHUNK ? P1 - c0 : PR_CHUNK);
        // PT: the chunk's table entries are requested in front of the barrier (they travel while the last wavefront
        // arrives; the barrier guards LDS, not them)
        constexpr int PER = (PR_CHUNK + NTH - 1) / NTH;
        int4 e_pre[PER];
        if constexpr (PT) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int k = tid * PER + i;
                e_pre[i] = A.pair_tab[c0 + (k < cn ? k : cn - 1)];
            }
        }
        pr_lds_barrier();   // the previous chunk's lists and pointers are no longer needed
        int64_t v0, v1;
        if constexpr (PT) {
            // ltp[k] = first entry, ltp[TPS + k] / ltp[2 TPS + k] = common neighbours / one-hop nodes of pair k, lcum = the
            // entries in front of pair k inside the chunk (a scan over the workgroup)
            int cnt[PER], own = 0;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int k = tid * PER + i;
                cnt[i] = 0;
                if (k < cn) {
                    const int4 e = e_pre[i];
                    const int64_t ce = (int64_t)e.y + e.z + e.w;   // (as pt_count: an entry outside the buffer counts as empty)
                    const bool ok = !over && !(e.x < 0 || e.y < 0 || e.z < 0 || e.w < 0 || e.x + ce > A.ent_cap) && ce > 0;
                    cnt[i] = ok ? e.y + e.z + e.w : 0;
                    ltp[k] = ok ? e.x : 0;
                    ltp[TPS + k] = ok ? e.y : 0;
                    ltp[2 * TPS + k] = ok ? e.z : 0;
                }
                own += cnt[i];
            }
            int x = own;
#pragma unroll
            for (int dlt = 1; dlt < 64; dlt <<= 1) {
                const int y = __shfl_up(x, dlt, 64);
                if (lane >= dlt) x += y;
            }
            if (lane == 63) lwcnt[wave] = x;
            if (tid < 4) lctl[tid] = 0;
            pr_lds_barrier();
            int pre = 0, tot = 0;
            for (int w = 0; w < NTH / 64; ++w) {
                if (w < wave) pre += lwcnt[w];
                tot += lwcnt[w];
            }
            int run = pre + x - own;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int k = tid * PER + i;
                if (k < cn) lcum[k] = run;
                run += cnt[i];
            }
            if (tid == 0) lcum[cn] = tot;
            v0 = vpos;
            v1 = vpos + tot;
            vpos = v1;
        } else {
        for (int i = tid; i < 3 * (cn + 1); i += NTH) {
            const int t = i / (cn + 1), k = i % (cn + 1);
            int64_t v = A.type_ptr[(int64_t)t * (A.bs + 1) + c0 + k];
            v = v < n[t] ? (v < 0 ? 0 : v) : n[t];
            ltp[t * TPS + k] = (int)v;
        }
        if (tid < 4) lctl[tid] = 0;
        pr_lds_barrier();
        // pair-major base of the chunk, then every pair's start relative to it (fits 31 bits: ent_cap does)
        v0 = (int64_t)ltp[0] + ltp[TPS] + ltp[2 * TPS];
        v1 = (int64_t)ltp[cn] + ltp[TPS + cn] + ltp[2 * TPS + cn];
        for (int k = tid; k <= cn; k += NTH)
            lcum[k] = (int)(((int64_t)ltp[k] + ltp[TPS + k] + ltp[2 * TPS + k]) - v0);
        }
        const int64_t u_first = v0 >> 4;
        const int n_units = v1 > v0 ? (int)(((v1 - 1) >> 4) - u_first + 1) : 0;
        // A pair in several pieces is merged by the group that walks its LAST piece, right behind that unit, once the
        // units in front of it are flagged done (they were drawn earlier: nobody waits for a later unit) -- no barrier,
        // no separate merge phase.  Only a chunk of more than PR_FLAGS units falls back to barrier + merge list.
        const bool inl = n_units <= PR_FLAGS;
        if (inl)
            for (int k = tid; k < n_units; k += NTH) lflag[k] = 0;
        pr_lds_barrier();
        PR_STAMP(2);
        PR_STAMP(3);
        // ---- the chunk's units: 16 consecutive entries of the pair-major order each, EPW of them per wavefront and
        //      ticket.  Lane i < 16 of a group finds entry i of its unit (binary search in the chunk's starts), fetches
        //      its record and computes its two 1 / std; then the walk of pair_flip.hip, four entries per batch with the Z
        //      rows requested a batch ahead.  meta: bits 0-1 type, 2 first entry of a piece, 3 last one, 4 the piece is
        //      not the whole pair, 5 head piece (the pair started in an earlier unit), 6 the slot holds an entry.
        while (true) {
            int tk = 0;
            if (lane == 0) tk = atomicAdd(&lctl[2], 1);
            tk = __builtin_amdgcn_readfirstlane(tk);
            if (tk * EPW >= n_units) break;
            const int ul = tk * EPW + grp;                     // unit of this group, chunk-relative
            const int64_t x0 = (u_first + ul) << 4;            // its first pair-major position (global)
            for (int i = lj; i < 16; i += G) {
                int4 rec = make_int4(0, 0, 0, 0);
                int meta = 0;
                f32x2 sc = {0.f, 0.f};
                const int64_t x = x0 + i;
                if (ul < n_units && x >= v0 && x < v1) {
                    const int xr = (int)(x - v0);
                    int lo = 0, hi = cn;                       // largest k with lcum[k] <= xr (skips the empty pairs)
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (lcum[mid] <= xr) lo = mid; else hi = mid;
                    }
                    const int k = lo, ps = lcum[k], pe = lcum[k + 1], j = xr - ps;
                    int t;
                    if constexpr (PT) {      // the pair's entries are contiguous, the type travels in the record
                        rec = A.entries[(int64_t)ltp[k] + j];
                        t = (int)(((uint32_t)rec.x >> 29) & 3u) - 1;
                        t = t < 0 ? 0 : t;
                    } else {
                        const int *tp = ltp + k;
                        const int lo0 = tp[0], n0 = tp[1] - lo0, lo1 = tp[TPS], n1 = tp[TPS + 1] - lo1, lo2 = tp[2 * TPS];
                        t = (j >= n0) + (j >= n0 + n1);
                        const int64_t idx = t == 0 ? lo0 + j : (t == 1 ? lo1 + (j - n0) : lo2 + (j - n0 - n1));
                        rec = A.entries[(int64_t)t * A.ent_cap + idx];
                    }
                    const int64_t ustart = x0 - v0, uend = ustart + 16;        // the unit in chunk-relative positions
                    const bool head = ps < ustart, more = pe > uend;
                    meta = t | ((j == 0 || i == 0) ? 4 : 0) | ((xr == pe - 1 || i == 15) ? 8 : 0) |
                           ((head || more) ? 16 : 0) | (head ? 32 : 0) | 64 | ((head && !more && inl) ? 128 : 0);
                    const float *st = lstat + 8 * t;
                    const float pa = __int_as_float(rec.z), pb = __int_as_float(rec.w);
                    // pe_stat[t][7]: inside [0, c]^2 no hidden unit leaves the pattern of (0, 0), in either argument order
                    // (fold.no_flip_radius): such an entry needs no look at its units at all
                    if constexpr (PT) {
                        // bits 9-14 / 15-20: row t * NP + s of the pattern table for (pa, pb) / (pb, pa).  Outside the
                        // square the two cells of the grid say which patterns these are -- or (bit 8: the exact path)
                        // which tabulated patterns come nearest where a boundary may cross a cell or its pattern is not
                        // tabulated
                        int i1 = t * NP, i2 = t * NP;
                        if (!(fmaxf(pa, pb) <= st[7])) {
                            const int ia = grid_cell(pa), ib = grid_cell(pb);
                            const uint8_t *gt = A.pat_grid + (int64_t)t * A.grid_n * A.grid_n;
                            const int g1 = gt[ia * A.grid_n + ib], g2 = gt[ib * A.grid_n + ia];
                            const int s1 = g1 & 31, s2 = g2 & 31;
                            if (((g1 | g2) & 0x80) || s1 >= NP || s2 >= NP) meta |= 256;   // the exact path, from (i1, i2)
                            i1 += s1 < NP ? s1 : 0;
                            i2 += s2 < NP ? s2 : 0;
                        }
                        meta |= (i1 << 9) | (i2 << 15);
                    } else {
                        if (!(fmaxf(pa, pb) <= st[7])) meta |= 256;
                    }
                    const float vab = st[0] * pa * pa + st[1] * pb * pb + st[2] + 2.0f * (st[3] * pa * pb + st[4] * pa + st[5] * pb);
                    const float vba = st[0] * pb * pb + st[1] * pa * pa + st[2] + 2.0f * (st[3] * pa * pb + st[4] * pb + st[5] * pa);
                    sc = f32x2{__builtin_amdgcn_rsqf(fmaxf(vab, 0.0f) + 1e-5f), __builtin_amdgcn_rsqf(fmaxf(vba, 0.0f) + 1e-5f)};
                }
                lr[i] = rec;
                lm[i] = meta;
                ls[i] = sc;
            }
            int lro = 0;                 // (opaque zero: the reads below must stay behind the stores above)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(lro) :: "memory");
            constexpr int ZBATCH = NV == 1 ? 4 : 1;     // entries whose Z rows are requested together, a batch ahead
            ZT za[ZBATCH][NV], zb[ZBATCH][NV];
#pragma unroll
            for (int u = 0; u < ZBATCH; ++u) z_row(lr[lro + u].y, za[u]);
            float4 qc[NV];
            q_row((int)((uint32_t)lr[lro].x & PAIR_MASK), qc);
            float *const piece_u = A.pieces + (u_first + ul) * 2 * RSP;
            float m = -INFINITY, l = 0.f;
            Q4 o[NV];
            q4_zero(o);
            // the pair that started in an earlier unit and ends in this one (bit 7 of meta): merged behind the walk
            int hpair = -1;

            auto entry = [&](const int i, const ZT (&zraw)[NV]) __attribute__((always_inline)) {
                const int4 rc = lr[lro + i];
                const f32x2 r12 = ls[lro + i];
                const int meta = lm[lro + i];
                // (the NEXT entry's pair: its query row is fetched at the end of this step, and only where the pair changes)
                const int pair_n = (int)((uint32_t)lr[lro + (i < 15 ? i + 1 : 15)].x & PAIR_MASK);
                const int pair_i = (int)((uint32_t)rc.x & PAIR_MASK);
                const bool on = meta & 64;
                const int t = meta & 3;
                const float pa = __int_as_float(rc.z), pb = __int_as_float(rc.w);
                const f32x2 pab = {pa, pb}, pba = {pb, pa};
                const float4 *tabl = ltab + t * GG + lj;
                Q4 k[NV];
                if constexpr (PT) {
                    // the base vectors of the pattern of (pa, pb) and of the pattern of (pb, pa), one order after the
                    // other (one association whatever the neighbours in the wavefront are: a pair's row depends on its
                    // own entries only)
                    const int i1 = (meta >> 9) & 63, i2 = (meta >> 15) & 63;
                    const float4 *b1 = lbase + i1 * D + lj, *b2 = lbase + i2 * D + lj;
                    const f32x2 c1 = r12.x * pab, c2 = r12.y * pba;
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const float4 zc = z_wide(zraw[v]);
                        const float4 P1 = b1[v * G], Q1 = b1[GG + v * G], R1 = b1[2 * GG + v * G], B1 = b1[3 * GG + v * G];
                        k[v].a = f32x2{zc.x, zc.y} + (f32x2{P1.x, P1.y} * c1.x + (f32x2{Q1.x, Q1.y} * c1.y +
                                                      (f32x2{R1.x, R1.y} * r12.x + f32x2{B1.x, B1.y})));
                        k[v].b = f32x2{zc.z, zc.w} + (f32x2{P1.z, P1.w} * c1.x + (f32x2{Q1.z, Q1.w} * c1.y +
                                                      (f32x2{R1.z, R1.w} * r12.x + f32x2{B1.z, B1.w})));
                    }
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const float4 P2 = b2[v * G], Q2 = b2[GG + v * G], R2 = b2[2 * GG + v * G], B2 = b2[3 * GG + v * G];
                        k[v].a += f32x2{P2.x, P2.y} * c2.x + (f32x2{Q2.x, Q2.y} * c2.y + (f32x2{R2.x, R2.y} * r12.y + f32x2{B2.x, B2.y}));
                        k[v].b += f32x2{P2.z, P2.w} * c2.x + (f32x2{Q2.z, Q2.w} * c2.y + (f32x2{R2.z, R2.w} * r12.y + f32x2{B2.z, B2.w}));
                    }
                } else {
                    const float4 *basel = lbase + t * D + lj;
                    const f32x2 cab = r12.x * pab + r12.y * pba;
                    const float cr = r12.x + r12.y;
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const float4 zc = z_wide(zraw[v]);
                        const float4 P0v = basel[v * G], Q0v = basel[GG + v * G], R0v = basel[2 * GG + v * G], C0v = basel[3 * GG + v * G];
                        k[v].a = f32x2{zc.x, zc.y} + (f32x2{P0v.x, P0v.y} * cab.x +
                                                      (f32x2{Q0v.x, Q0v.y} * cab.y + (f32x2{R0v.x, R0v.y} * cr + f32x2{C0v.x, C0v.y})));
                        k[v].b = f32x2{zc.z, zc.w} + (f32x2{P0v.z, P0v.w} * cab.x +
                                                      (f32x2{Q0v.z, Q0v.w} * cab.y + (f32x2{R0v.z, R0v.w} * cr + f32x2{C0v.z, C0v.w})));
                    }
                }
                // whose units are looked at: PT -- only the entries of cells without a tabulated pattern (the others
                // already carry the vectors of their own patterns: a correction against pattern 0 would count twice)
                const bool det = PT ? (on && (meta & 256) != 0) : on;
#ifdef PR_ABL_NOFLIP   /* (timing only, wrong results: what do detection and corrections of flipped units cost?) */
                if (false) {
#elif defined(PR_ABL_ALLDETECT)   /* (timing only: every entry looks at its units, as before the no-flip box) */
                if (true) {
#else
                if (__ballot((meta & 256) != 0)) {   // some entry of this wavefront lies outside its type's no-flip box
#endif
                // one feature quad's four units (both orders) at a time: a unit that left the pattern of (0, 0) owes every
                // lane of its group Wfold[:, k] |y_k| (pair_flip.hip: flipped lanes one at a time, the weights a step ahead)
                const bool wt_lds = t >= T_LO && t < T_LO + WTL;
                const float4 *lw = lwt + (wt_lds ? (t - T_LO) * D * GG : 0) + lj;
                const float *wT = A.wfoldT + (int64_t)t * D * D + off;
#pragma unroll
                for (int vu = 0; vu < NV; ++vu) {
                    f32x2 zz[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 tj = tabl[3 * GG * j + vu * G];
                        zz[j] = r12 * (tj.x * pab + (tj.y * pba + tj.z)) + tj.w;
                    }
                    if constexpr (PT) {
                        // the table is signed for pattern 0, the entry starts from the patterns its cells name: a unit
                        // whose state differs between the two changes sign (.x: the order (pa, pb), .y: (pb, pa))
                        const int qd = lj + vu * G;
                        const uint32_t w1 = lsign[((meta >> 9) & 63) * SW + (qd >> 3)] >> (4 * (qd & 7));
                        const uint32_t w2 = lsign[((meta >> 15) & 63) * SW + (qd >> 3)] >> (4 * (qd & 7));
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            zz[j].x = __uint_as_float(__float_as_uint(zz[j].x) ^ ((w1 >> j) << 31));
                            zz[j].y = __uint_as_float(__float_as_uint(zz[j].y) ^ ((w2 >> j) << 31));
                        }
                    }
                    const float zmin = fminf(fminf(fminf(zz[0].x, zz[0].y), fminf(zz[1].x, zz[1].y)),
                                             fminf(fminf(zz[2].x, zz[2].y), fminf(zz[3].x, zz[3].y)));
                    const bool fl = zmin < 0.f && det;
                    if (__ballot(fl)) {
                        const bool all_lds = WTL == 3 || __all(wt_lds || !fl);
                        Q4 wp[NV];
                        q4_zero(wp);
                        float vp = 0.f;
#pragma unroll
                        for (int oj = 0; oj < 8; ++oj) {
                            const float zv = (oj & 4) ? zz[oj & 3].y : zz[oj & 3].x;
                            uint64_t bm = __ballot(zv < 0.f && det);
                            while (bm) {
                                const int b = __builtin_ctzll(bm);
                                bm &= bm - 1;
                                const float val = -__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zv), b));
                                const bool mine = b / G == grp;
                                const int kk = 4 * (b % G + vu * G) + (oj & 3);
                                float4 w[NV];
#pragma unroll
                                for (int v = 0; v < NV; ++v)
                                    w[v] = WTL > 0 ? lw[kk * GG + v * G] : make_float4(0.f, 0.f, 0.f, 0.f);
                                if (!all_lds) {
#pragma unroll
                                    for (int v = 0; v < NV; ++v) {
                                        float4 wg = make_float4(0.f, 0.f, 0.f, 0.f);
                                        if (mine && !wt_lds) wg = *reinterpret_cast<const float4 *>(wT + (int64_t)kk * D + 4 * G * v);
                                        w[v].x = wt_lds ? w[v].x : wg.x; w[v].y = wt_lds ? w[v].y : wg.y;
                                        w[v].z = wt_lds ? w[v].z : wg.z; w[v].w = wt_lds ? w[v].w : wg.w;
                                    }
                                }
#pragma unroll
                                for (int v = 0; v < NV; ++v) {
                                    k[v].a += wp[v].a * vp;
                                    k[v].b += wp[v].b * vp;
                                    wp[v] = Q4{f32x2{w[v].x, w[v].y}, f32x2{w[v].z, w[v].w}};
                                }
                                vp = mine ? val : 0.f;
                            }
                        }
#pragma unroll
                        for (int v = 0; v < NV; ++v) {
                            k[v].a += wp[v].a * vp;
                            k[v].b += wp[v].b * vp;
                        }
                    }
                }
                }
                f32x2 sp = {0.f, 0.f};
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const f32x2 x01 = k[v].a * f32x2{qc[v].x, qc[v].y}, x23 = k[v].b * f32x2{qc[v].z, qc[v].w};
                    const f32x2 y01 = x01 * 0.2f, y23 = x23 * 0.2f;
                    const f32x2 l01 = {fmaxf(x01.x, y01.x), fmaxf(x01.y, y01.y)}, l23 = {fmaxf(x23.x, y23.x), fmaxf(x23.y, y23.y)};
                    if (v == 0) sp = l01 * at01[v] + l23 * at23[v];
                    else sp += l01 * at01[v] + l23 * at23[v];
                }
                const float s = pr_group_sum<G>(sp.x + sp.y);
                if (on) {
                    if (meta & 4) {          // first entry of a piece: fresh state
                        m = -INFINITY; l = 0.f;
                        q4_zero(o);
                    }
                    const float d = s - m;
                    const float e = __expf(-fabsf(d));
                    const bool up = d > 0.f;
                    const float sca = up ? e : 1.f, w = up ? 1.f : e;
                    l = fmaf(l, sca, w);
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        o[v].a = o[v].a * sca + k[v].a * w;
                        o[v].b = o[v].b * sca + k[v].b * w;
                    }
                    m = fmaxf(m, s);
                    if (meta & 8) {          // last entry of the piece
                        if (meta & 128) hpair = pair_i;   // ... the LAST piece of a pair in several pieces: merged behind the walk
                        if (meta & 16) {     // ... a piece of such a pair: its state waits for the merge
                            float *dst = piece_u + ((meta & 32) ? 0 : RSP);
#pragma unroll
                            for (int v = 0; v < NV; ++v)
                                *reinterpret_cast<float4 *>(dst + off + 4 * G * v) = make_float4(o[v].a.x, o[v].a.y, o[v].b.x, o[v].b.y);
                            if (lj == 0) *reinterpret_cast<float2 *>(dst + D) = make_float2(m, l);
                        } else {
                            float4 r[NV];
                            finish_row(o, l, r);
                            store_row(pair_i, r);
                        }
                    }
                }
                if (pair_n != pair_i) q_row(pair_n, qc);   // (behind the last use of this entry's row; most steps stay inside a pair)
                // (consecutive entries are kept apart.  Without the fence a ticket of the 8-feature form takes 25.5 us
                //  instead of 26.8 alone -- and the pipelined step 0.144 ms instead of 0.1415: six more registers spilled)
                __builtin_amdgcn_sched_barrier(0);
            };
            auto batch = [&](const int qt, const ZT (&zc4)[ZBATCH][NV], ZT (&zn4)[ZBATCH][NV]) __attribute__((always_inline)) {
                constexpr int NQ = 16 / ZBATCH;
                const int nb = qt < NQ - 1 ? ZBATCH * qt + ZBATCH : 16 - ZBATCH;
#pragma unroll
                for (int u = 0; u < ZBATCH; ++u) z_row(lr[lro + nb + u].y, zn4[u]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < ZBATCH; ++u) entry(ZBATCH * qt + u, zc4[u]);
            };
#pragma unroll 1
            for (int h = 0; h < 8 / ZBATCH; ++h) {
                batch(2 * h, za, zb);
                batch(2 * h + 1, zb, za);
            }
            if (inl) {
                // this unit's pieces are in L2: flag it; then the merge of the pair that ended in it, if there is one
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lj == 0 && ul < n_units) __hip_atomic_store(&lflag[ul], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (hpair >= 0) {
                    const int k = (int)(hpair - c0);
                    const int ua_l = (int)(((v0 + lcum[k]) >> 4) - u_first);      // first unit of the pair, chunk-relative
                    for (int u = ua_l + lj; u < ul; u += G)
                        while (__hip_atomic_load(&lflag[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0)
                            __builtin_amdgcn_s_sleep(2);
                    asm volatile("" ::: "memory");
                    float mx = -INFINITY, den = 0.f;
                    Q4 acc[NV];
                    q4_zero(acc);
                    for (int u = ua_l; u <= ul; u += 4) {   // (four pieces requested together: a hub pair has dozens)
                        float2 h[4];
                        float4 b[4][NV];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int uu = u + i <= ul ? u + i : ul;
                            const float *rp = A.pieces + ((u_first + uu) * 2 + (uu == ua_l ? 1 : 0)) * RSP;
                            h[i] = *reinterpret_cast<const float2 *>(rp + D);
#pragma unroll
                            for (int v = 0; v < NV; ++v) b[i][v] = *reinterpret_cast<const float4 *>(rp + off + 4 * G * v);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (u + i > ul) break;
                            const float mn = fmaxf(mx, h[i].x);
                            const float sa = __expf(mx - mn), sb = __expf(h[i].x - mn);
                            den = fmaf(den, sa, h[i].y * sb);
#pragma unroll
                            for (int v = 0; v < NV; ++v) {
                                acc[v].a = acc[v].a * sa + f32x2{b[i][v].x, b[i][v].y} * sb;
                                acc[v].b = acc[v].b * sa + f32x2{b[i][v].z, b[i][v].w} * sb;
                            }
                            mx = mn;
                        }
                    }
                    float4 r[NV];
                    finish_row(acc, den, r);
                    store_row(hpair, r);
                }
            }
#ifdef PR_STAMPS
            ++st_rounds;
#endif
        }
        PR_STAMP(4);
        // ---- the chunk's count features, and its pairs sorted (empty / in several pieces) -- BEHIND the units: nothing of
        //      this is needed to walk them, and the wavefronts that run out of units early do their share meanwhile.  PT
        //      with an order for the tail and a chunk whose pieces are merged inline: no list is read at all
        const bool lists = !(PT && A.perm != nullptr) || !inl;
        for (int k = tid; k < cn; k += NTH) {
            int n0, n1, n2;
            if constexpr (PT) {
                n0 = ltp[TPS + k]; n1 = ltp[2 * TPS + k]; n2 = lcum[k + 1] - lcum[k] - n0 - n1;
            } else {
                n0 = ltp[k + 1] - ltp[k]; n1 = ltp[TPS + k + 1] - ltp[TPS + k]; n2 = ltp[2 * TPS + k + 1] - ltp[2 * TPS + k];
            }
            const int np = n0 + n1 + n2;
            // PT with an order for the tail: a pair without entries gets neither row nor counts -- the tail knows the
            // constant row (lpf_tail_chain_rows_perm_*: row_empty) and never reads them
            const bool skip = PT && A.perm != nullptr && np == 0;
            if (skip) continue;
            if (lists) {
                if (np == 0) {
                    llist[atomicAdd(&lctl[0], 1)] = k;
                } else if (((v0 + lcum[k]) >> 4) != ((v0 + lcum[k + 1] - 1) >> 4)) {
                    llist[PR_CHUNK + atomicAdd(&lctl[1], 1)] = k;
                }
            }
            if (A.n_counts > 0) {
                float *o = A.out + (c0 + k) * A.ldo + D;
                const float f0 = (float)n0, f1 = (float)n1, f2 = (float)n2;
                if (A.n_counts == 4) { o[0] = f0; o[1] = f1; o[2] = f2; o[3] = f0 + f1; }
                else if (A.n_counts == 3) { o[0] = f0; o[1] = f1; o[2] = f0 + f1; }
                else { o[0] = f0; }
            }
        }
        int n_multi = 0;
        if (lists) {
            pr_lds_barrier();
            const int n_empty = lctl[0];
            n_multi = lctl[1];
            // ---- pairs without entries: the constant row
            float4 cr[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) cr[v] = lconst[lj + v * G];
            for (int k = gid; k < n_empty; k += NG) store_row(c0 + llist[k], cr);
        }
        // ---- (chunks of more than PR_FLAGS units) pairs in several pieces: one barrier, then merged in unit order
        if (!inl) __syncthreads();
        for (int mk = gid; mk < (inl ? 0 : n_multi); mk += NG) {
            const int k = llist[PR_CHUNK + mk];
            const int64_t ua = (v0 + lcum[k]) >> 4, ub = (v0 + lcum[k + 1] - 1) >> 4;
            float mx = -INFINITY, den = 0.f;
            Q4 acc[NV];
            q4_zero(acc);
            for (int64_t u = ua; u <= ub; u += 4) {   // (four pieces requested together: a hub pair has dozens)
                float2 h[4];
                float4 b[4][NV];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t uu = u + i <= ub ? u + i : ub;
                    const float *rp = A.pieces + (uu * 2 + (uu == ua ? 1 : 0)) * RSP;
                    h[i] = *reinterpret_cast<const float2 *>(rp + D);
#pragma unroll
                    for (int v = 0; v < NV; ++v) b[i][v] = *reinterpret_cast<const float4 *>(rp + off + 4 * G * v);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (u + i > ub) break;
                    const float mn = fmaxf(mx, h[i].x);
                    const float sa = __expf(mx - mn), sb = __expf(h[i].x - mn);
                    den = fmaf(den, sa, h[i].y * sb);
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        acc[v].a = acc[v].a * sa + f32x2{b[i][v].x, b[i][v].y} * sb;
                        acc[v].b = acc[v].b * sa + f32x2{b[i][v].z, b[i][v].w} * sb;
                    }
                    mx = mn;
                }
            }
            float4 r[NV];
            finish_row(acc, den, r);
            store_row(c0 + k, r);
        }
        PR_STAMP(5);
    }
    // ---- the tail's order, second half: pairs with selected nodes in ascending order from the front, the others from
    //      the back.  Deterministic (sums of published counts, ranks from ballots -- no atomics on a counter): a replayed
    //      step stays bitwise the eager one.
    if (PT && A.perm && tid == 0 && blockIdx.x == gridDim.x - 1) *A.n_nonempty = (int64_t)lctl[10];
    if (A.perm) {
        pr_lds_barrier();
        if (!PT && wave == 0) {   // (PT: the selection counted the pairs with entries per block -- known since the set-up)
            const uint32_t epoch = (uint32_t)lctl[11];
            const int64_t nb = blockIdx.x;
            uint64_t sum = 0;
            for (int64_t j0 = 0; j0 < nb; j0 += 256) {   // four words per lane and round trip
                uint64_t w[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int64_t idx = j0 + 64 * b + lane;
                    w[b] = idx < nb ? __hip_atomic_load(A.perm_lb + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int64_t idx = j0 + 64 * b + lane;
                    if (idx < nb) {
                        if (!((uint32_t)(w[b] >> 42) == epoch && ((w[b] >> 40) & 3ull) != 0ull))
                            w[b] = lb_wait(A.perm_lb + idx, epoch);
                        sum += w[b] & LB_VAL_MASK;
                    }
                }
            }
            sum = lb_wave_sum(sum);
            if (lane == 0) {
                lctl[8] = (int)(sum & 0xffffffff); lctl[9] = (int)(sum >> 32);
                if (blockIdx.x == gridDim.x - 1) {   // (it has seen every other workgroup's word: they all read the counter)
                    *A.n_nonempty = (int64_t)sum + lctl[10];
                    A.perm_lb[PR_LB_WORDS] = (uint64_t)epoch;
                }
            }
        }
        pr_lds_barrier();
        int64_t ne_base = (int64_t)(uint32_t)lctl[8] | ((int64_t)lctl[9] << 32);   // pairs with entries in front
        for (int64_t k0 = P0; k0 < P1; k0 += NTH) {
            const int64_t k = k0 + tid;
            const bool in = k < P1;
            const bool ne = in && nonempty(k);
            const uint64_t b_ne = __ballot(ne), b_all = __ballot(in);
            if (lane == 0) { lwcnt[wave] = __popcll(b_ne); lwcnt[16 + wave] = __popcll(b_all); }
            pr_lds_barrier();
            int pre_ne = 0, pre_all = 0, tot_ne = 0;
            for (int w = 0; w < NTH / 64; ++w) {
                if (w < wave) { pre_ne += lwcnt[w]; pre_all += lwcnt[16 + w]; }
                tot_ne += lwcnt[w];
            }
            const uint64_t lt = (1ull << lane) - 1ull;
            const int r_ne = pre_ne + __popcll(b_ne & lt), r_all = pre_all + __popcll(b_all & lt);
            if (ne) A.perm[ne_base + r_ne] = (int32_t)k;
            else if (in) A.perm[A.bs - 1 - ((k0 - ne_base) + (r_all - r_ne))] = (int32_t)k;
            pr_lds_barrier();
            ne_base += tot_ne;
        }
    }
#ifdef PR_STAMPS
    if (lane == 0 && pr_stamp_buf) {
        uint64_t *o = pr_stamp_buf + ((int64_t)blockIdx.x * (NTH / 64) + wave) * 8;
        for (int k = 0; k < 6; ++k) o[k] = st_t[k];
        o[3] = st_t[7];                           // (mark 3 -- the empties -- coincides with mark 2 since the lists moved)
        o[6] = (uint64_t)st_rounds;
        o[7] = (uint64_t)(P1 - P0) | ((uint64_t)lctl[1] << 32);
    }
#endif
}

template <bool ZB, bool PT = false>
int rows_launch(int32_t D, int64_t bs, const int32_t *type_ptr, const void *entries, int64_t ent_cap, const void *Z,
                int64_t ldz, const float *q, int64_t ldq, const float *pe_tab_signed, const float *pe_stat,
                const float *base, const float *wfold_t, const float *att, const float *att_bias, const float *ln_g,
                const float *ln_b, int32_t n_counts, const int64_t *sel_ctl, float *pieces, int64_t units_cap, float *out,
                int64_t ldo, void *stream, int32_t *perm = nullptr, uint64_t *perm_lb = nullptr,
                int64_t *n_nonempty = nullptr, const void *pair_tab = nullptr, const int32_t *blk_cnt = nullptr,
                const float *pat_base = nullptr, const uint8_t *pat_grid = nullptr, const uint32_t *pat_sign = nullptr,
                int32_t grid_n = 0,
                int32_t grid_shift = 0, int32_t grid_base = 0, float grid_ofs = 0.f) {
    if (bs == 0) return LPF_OK;
    LPF_REQUIRE(!PT || (pat_base && pat_grid && pat_sign && lpf_aligned16(pat_base) && grid_n >= 2 && grid_n <= 4096 &&
                        grid_shift >= 0 && grid_shift <= 23 && grid_base >= 0 && grid_ofs >= 0.f));
    LPF_REQUIRE(bs > 0 && bs < (PT ? (1ll << 29) : (1ll << 31)) &&
                (PT ? (pair_tab && blk_cnt && lpf_aligned16(pair_tab) && (reinterpret_cast<uintptr_t>(blk_cnt) & 7) == 0)
                    : type_ptr != nullptr) &&
                entries && ent_cap > 0 && ent_cap < (PT ? (1ll << 31) : (1ll << 29)) && Z && q &&
                pe_tab_signed && pe_stat && (PT || base) && wfold_t && att && att_bias && ln_g && ln_b && out && pieces &&
                units_cap >= (PT ? 2 : (3 * ent_cap + 15) / 16 + 1) && lpf_aligned16(pieces));
    LPF_REQUIRE(!perm || ((PT || perm_lb) && n_nonempty));
    LPF_REQUIRE((n_counts == 0 || n_counts == 1 || n_counts == 3 || n_counts == 4) && ldo >= D + n_counts && (ldo & 3) == 0);
    LPF_REQUIRE(ldz >= D && ldq >= D && ldz < (1ll << 31) && ldq < (1ll << 31) && (ldz & (ZB ? 7 : 3)) == 0 &&
                (ldq & 3) == 0 && lpf_aligned16(entries) && lpf_aligned16(Z) && lpf_aligned16(q) &&
                lpf_aligned16(pe_tab_signed) && (PT || lpf_aligned16(base)) && lpf_aligned16(wfold_t) && lpf_aligned16(att) &&
                lpf_aligned16(att_bias) && lpf_aligned16(ln_g) && lpf_aligned16(ln_b) && lpf_aligned16(out));
    const RowsArgs a{bs, type_ptr, static_cast<const int4 *>(entries), ent_cap, static_cast<const float *>(Z), (uint32_t)ldz,
                     q, (uint32_t)ldq, pe_tab_signed, pe_stat, base, wfold_t, att, att_bias, ln_g, ln_b, out, ldo,
                     n_counts, sel_ctl, pieces, units_cap, perm, perm_lb, n_nonempty,
                     static_cast<const int4 *>(pair_tab), blk_cnt, pat_base, pat_grid, pat_sign, grid_n, grid_shift, grid_base,
                     grid_ofs};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int n_cu = lpf_cu_count();
    if (n_cu == 0) return LPF_ERR_NO_DEVICE;
#ifdef PR_GRID2   /* tuning aid: twice the workgroups a CU holds at a time */
#define PR_GRID_MUL 2
#else
#define PR_GRID_MUL 1
#endif
#define LPF_ROWS_GO(GQ, NTH, WTL, PER_CU, NVV)                                                      \
    do {                                                                                            \
        auto kern = pair_rows_kernel<GQ, NTH, WTL, ZB, PT, NVV>;                                        \
        constexpr size_t lds = PrLds<GQ, NTH, WTL, PT, NVV>::BYTES;                                 \
        LPF_SET_MAX_LDS(kern, lds);                                                                 \
        int64_t groups = (int64_t)n_cu * PER_CU * PR_GRID_MUL;                                      \
        const int64_t most = (bs + 15) / 16;   /* (a workgroup per 16 pairs at the very least) */   \
        if (groups > most) groups = most;                                                           \
        if (groups > PR_LB_WORDS) groups = PR_LB_WORDS;                                             \
        hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(NTH), lds, s, a);                     \
    } while (0)
    switch (D) {
        case 32: LPF_ROWS_GO(8, 512, 3, 2, 1); break;
        case 64: LPF_ROWS_GO(16, 512, 3, 2, 1); break;
#ifdef PR_CFG128   /* tuning aid: G, NTH, WTL, PER_CU, NV of the D = 128 launch */
#define LPF_ROWS_GO_(...) LPF_ROWS_GO(__VA_ARGS__)
        case 128: LPF_ROWS_GO_(PR_CFG128); break;
#undef LPF_ROWS_GO_
#else
        case 128:
            // behind select4 (patterns by table): 16 lanes x 8 features per entry, four entries per wavefront step
            if constexpr (PT) LPF_ROWS_GO(16, 1024, 0, 1, 2);
            else LPF_ROWS_GO(32, 1024, 1, 1, 1);
            break;
#endif
        case 256:
            // (behind select4 the pattern tables fill the LDS: ONE workgroup per CU, so it has to be a large one -- 32
            //  lanes x 8 features per entry, sixteen wavefronts)
            if constexpr (PT) LPF_ROWS_GO(32, 1024, 0, 1, 2);
            else LPF_ROWS_GO(64, 256, 0, 3, 1);
            break;
        default: return LPF_ERR_UNSUPPORTED;
    }
#undef LPF_ROWS_GO
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

}  // namespace

extern "C" int lpf_pair_attention_rows_f32(int32_t D, int64_t bs, const int32_t *type_ptr, const void *entries,
                                           int64_t ent_cap, const float *Z, int64_t ldz, const float *q, int64_t ldq,
                                           const float *pe_tab_signed, const float *pe_stat, const float *base,
                                           const float *wfold_t, const float *att, const float *att_bias,
                                           const float *ln_g, const float *ln_b, int32_t n_counts,
                                           const int64_t *sel_ctl, float *pieces, int64_t units_cap, float *out,
                                           int64_t ldo, void *stream) {
    return rows_launch<false>(D, bs, type_ptr, entries, ent_cap, Z, ldz, q, ldq, pe_tab_signed, pe_stat, base, wfold_t, att,
                              att_bias, ln_g, ln_b, n_counts, sel_ctl, pieces, units_cap, out, ldo, stream);
}

extern "C" int lpf_pair_attention_rows_zbf16(int32_t D, int64_t bs, const int32_t *type_ptr, const void *entries,
                                             int64_t ent_cap, const void *Z_bf16, int64_t ldz, const float *q, int64_t ldq,
                                             const float *pe_tab_signed, const float *pe_stat, const float *base,
                                             const float *wfold_t, const float *att, const float *att_bias,
                                             const float *ln_g, const float *ln_b, int32_t n_counts,
                                             const int64_t *sel_ctl, float *pieces, int64_t units_cap, float *out,
                                             int64_t ldo, void *stream) {
    return rows_launch<true>(D, bs, type_ptr, entries, ent_cap, Z_bf16, ldz, q, ldq, pe_tab_signed, pe_stat, base, wfold_t,
                             att, att_bias, ln_g, ln_b, n_counts, sel_ctl, pieces, units_cap, out, ldo, stream);
}

#ifdef PR_STAMPS
extern "C" int lpf_pair_rows_set_stamps(void *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(pr_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? LPF_OK : LPF_ERR_LAUNCH;
}
#endif

/* lpf_pair_attention_rows_f32 / _zbf16 that also leave the ORDER for lpf_tail_chain_rows_perm_*: perm int32[bs] = the
 * pairs with selected nodes in ascending order, then (from the back) those without; perm_lb: uint64 scratch of
 * LPF_ROWS_PERM_LB_WORDS words, zero before the first launch and then left alone (chained-scan words tagged with a launch
 * number the kernel keeps in the last word); *n_nonempty: the count. */
extern "C" int lpf_pair_attention_rows_perm_f32(int32_t D, int64_t bs, const int32_t *type_ptr, const void *entries,
                                                int64_t ent_cap, const float *Z, int64_t ldz, const float *q, int64_t ldq,
                                                const float *pe_tab_signed, const float *pe_stat, const float *base,
                                                const float *wfold_t, const float *att, const float *att_bias,
                                                const float *ln_g, const float *ln_b, int32_t n_counts,
                                                const int64_t *sel_ctl, float *pieces, int64_t units_cap, float *out,
                                                int64_t ldo, int32_t *perm, uint64_t *perm_lb, int64_t *n_nonempty,
                                                void *stream) {
    LPF_REQUIRE(perm && perm_lb && n_nonempty);
    return rows_launch<false>(D, bs, type_ptr, entries, ent_cap, Z, ldz, q, ldq, pe_tab_signed, pe_stat, base, wfold_t, att,
                              att_bias, ln_g, ln_b, n_counts, sel_ctl, pieces, units_cap, out, ldo, stream, perm, perm_lb,
                              n_nonempty);
}

extern "C" int lpf_pair_attention_rows_perm_zbf16(int32_t D, int64_t bs, const int32_t *type_ptr, const void *entries,
                                                  int64_t ent_cap, const void *Z_bf16, int64_t ldz, const float *q,
                                                  int64_t ldq, const float *pe_tab_signed, const float *pe_stat,
                                                  const float *base, const float *wfold_t, const float *att,
                                                  const float *att_bias, const float *ln_g, const float *ln_b,
                                                  int32_t n_counts, const int64_t *sel_ctl, float *pieces,
                                                  int64_t units_cap, float *out, int64_t ldo, int32_t *perm,
                                                  uint64_t *perm_lb, int64_t *n_nonempty, void *stream) {
    LPF_REQUIRE(perm && perm_lb && n_nonempty);
    return rows_launch<true>(D, bs, type_ptr, entries, ent_cap, Z_bf16, ldz, q, ldq, pe_tab_signed, pe_stat, base, wfold_t,
                             att, att_bias, ln_g, ln_b, n_counts, sel_ctl, pieces, units_cap, out, ldo, stream, perm, perm_lb,
                             n_nonempty);
}

/* The same kernel behind lpf_select4 (pair-major entries, the type in the record; pair_tab / blk_cnt as that call leaves
 * them).  perm / n_nonempty: both or none (NULL); no scan words -- the selection counted the pairs with entries. */
extern "C" int lpf_pair_attention_rows4_f32(int32_t D, int64_t bs, const void *pair_tab, const int32_t *blk_cnt,
                                            const void *entries, int64_t ent_cap, const float *Z, int64_t ldz,
                                            const float *q, int64_t ldq, const float *pe_tab_signed, const float *pe_stat,
                                            const float *pat_base, const void *pat_grid, const void *pat_sign,
                                            int32_t grid_n, int32_t grid_shift, int32_t grid_base, float grid_ofs,
                                            const float *wfold_t,
                                            const float *att, const float *att_bias, const float *ln_g, const float *ln_b,
                                            int32_t n_counts, const int64_t *sel_ctl, float *pieces, int64_t units_cap,
                                            float *out, int64_t ldo, int32_t *perm, int64_t *n_nonempty, void *stream) {
    return rows_launch<false, true>(D, bs, nullptr, entries, ent_cap, Z, ldz, q, ldq, pe_tab_signed, pe_stat, nullptr, wfold_t,
                                    att, att_bias, ln_g, ln_b, n_counts, sel_ctl, pieces, units_cap, out, ldo, stream, perm,
                                    nullptr, n_nonempty, pair_tab, blk_cnt, pat_base, static_cast<const uint8_t *>(pat_grid),
                                    static_cast<const uint32_t *>(pat_sign), grid_n, grid_shift, grid_base, grid_ofs);
}

extern "C" int lpf_pair_attention_rows4_zbf16(int32_t D, int64_t bs, const void *pair_tab, const int32_t *blk_cnt,
                                              const void *entries, int64_t ent_cap, const void *Z_bf16, int64_t ldz,
                                              const float *q, int64_t ldq, const float *pe_tab_signed,
                                              const float *pe_stat, const float *pat_base, const void *pat_grid,
                                              const void *pat_sign, int32_t grid_n, int32_t grid_shift, int32_t grid_base,
                                              float grid_ofs,
                                              const float *wfold_t, const float *att, const float *att_bias,
                                              const float *ln_g, const float *ln_b, int32_t n_counts,
                                              const int64_t *sel_ctl, float *pieces, int64_t units_cap, float *out,
                                              int64_t ldo, int32_t *perm, int64_t *n_nonempty, void *stream) {
    return rows_launch<true, true>(D, bs, nullptr, entries, ent_cap, Z_bf16, ldz, q, ldq, pe_tab_signed, pe_stat, nullptr,
                                   wfold_t, att, att_bias, ln_g, ln_b, n_counts, sel_ctl, pieces, units_cap, out, ldo, stream,
                                   perm, nullptr, n_nonempty, pair_tab, blk_cnt, pat_base,
                                   static_cast<const uint8_t *>(pat_grid), static_cast<const uint32_t *>(pat_sign), grid_n,
                                   grid_shift, grid_base, grid_ofs);
}

/* floats of one piece record of lpf_pair_attention_rows_* (D accumulators, m, l, padded to whole 128-byte lines) */
extern "C" int64_t lpf_pair_rows_piece_floats(int32_t D) { return pr_piece_floats(D); }
