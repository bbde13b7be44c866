// Pairwise PPR-positional attention in one pass WITHOUT the D x D product per entry: the key projection of the
// positional encoding is evaluated through the ReLU activation pattern of its hidden layer.
//
// Reference: LinkAttention.message + PyG softmax + scatter-sum (src/modules/layers.py:193-224) with get_pos_encodings
// (src/models/link_transformer.py:182-211) folded in:
//     h_e = ReLU(y(pa,pb)) + ReLU(y(pb,pa)),  y_k(x,y) = r(x,y) (ta_k x + tc_k y + td_k) + beta_k     (closed-form LN)
//     k_e = Z[v_e] + Wfold_t h_e + bfold_t ;  s_e = att . leaky_relu(k_e * q_pair, 0.2) ;  segment softmax ;  sum alpha k
// pair_fused.hip spends 2 D^2 FLOP per entry on Wfold_t h_e (fp32 matrix cores, the dominant kernel of the step).
// But y_k is AFFINE in (r x, r y, r, 1), so for a fixed set S of active hidden units
//     sum_{k in S} Wfold[:,k] y_k  =  P_S (r x) + Q_S (r y) + R_S r + B_S          (four D-vectors that depend on S only)
// and the PPR values are small numbers: almost every entry has the activation pattern S0 of the point (0, 0), and one
// that does not differs from it in a few units.  With the four vectors of S0 precomputed (host, float64) the product is
//     Wfold h_e = P0 (r1 pa + r2 pb) + Q0 (r1 pb + r2 pa) + R0 (r1 + r2) + 2 B0 + sum_{k flipped} Wfold[:,k] |y_k|
// -- exact in real arithmetic for EVERY input (relu(y) - [k in S0] y = |y| on a flipped unit, whichever way it flipped);
// only the cost depends on the data: ~10 D FLOP per entry plus 2 D per flipped unit (collab-like bench: 0.66 flips per
// entry, 99th percentile 8, D = 128) instead of 2 D^2.  The kernel is then bound by the Z-row gather, not by the
// matrix pipe.  Worst case (every unit flips on every entry) it degenerates into a D x D product on the vector ALUs;
// the matrix-core kernel stays in the library for such weights (LinkTransformer.attention_impl).
//
// Layout: G = D/4 lanes own one entry -- lane j of the group holds features AND hidden units 4j .. 4j+3 --, a group
// walks one UNIT of 16 consecutive same-type entries with an online softmax and leaves exactly the records
// pair_fused.hip leaves (part[t][pair] for a segment inside one unit, boundary records otherwise), so the consumers
// (tail_chain.hip merge mode, pair_merge.hip) do not care which kernel ran.  64 / G units per wavefront at a time.
#include <type_traits>

#include "pe_common.h"

namespace {

constexpr uint32_t FL_PAIR_MASK = 0x7fffffffu;

struct FlipArgs {
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
    float *part;               // [3][bs][D+4]
    float *bnd;                // [3][units_cap][2][D+4]
    int64_t units_cap;
};

// Sum over the G lanes of a group, the same bits in every lane: DPP butterflies inside a row of 16 lanes (quad
// permutes, then the half-row and the row mirrored: after two steps a quad holds one value, so a mirror IS the xor
// partner), one swizzle across the rows, one pair of lane reads across the halves -- no LDS round trip per step
// (__shfl_xor compiles to ds_bpermute_b32: five dependent round trips per entry were a third of this kernel).
template <int CTRL>
__device__ __forceinline__ float fl_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int G>
__device__ __forceinline__ float fl_group_sum(float v) {
    v += fl_dpp<0xB1>(v);                   // quad_perm [1,0,3,2]
    v += fl_dpp<0x4E>(v);                   // quad_perm [2,3,0,1]
    if constexpr (G >= 8) v += fl_dpp<0x141>(v);    // row_half_mirror
    if constexpr (G >= 16) v += fl_dpp<0x140>(v);   // row_mirror
    if constexpr (G >= 32)
        v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401f));  // lane ^ 16
    if constexpr (G >= 64) {
        const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
        const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
        v = a + b;
    }
    return v;
}
// 1 / sqrt(var + eps) of the hidden layer's LayerNorm (closed form, pe_common.h) with the hardware reciprocal square root
__device__ __forceinline__ float fl_rstd(const PeStat &s, float x, float y) {
    const float var = s.c00 * x * x + s.c11 * y * y + s.cbb + 2.0f * (s.c01 * x * y + s.c0b * x + s.c1b * y);
    return __builtin_amdgcn_rsqf(fmaxf(var, 0.0f) + 1e-5f);
}

// G = D/4 lanes per entry; NTH threads per workgroup; WTL = how many types keep their Wfold^T in LDS (1: the one-hop
// type, which holds most of the entries -- and most of the flips: they sit in the ~10 % of the entries with a PPR value
// above ~0.03, typically six or seven units each; 2: the common-neighbour type as well; 3: all; 0: none, the columns
// come from L2).  A correction column read from LDS costs a dozen instructions; fetched from L2 it cost as much as the
// rest of the kernel (105 vs 55 us).
//
// The kernel is bound by vector-ALU issue, not by memory (without the Z and q loads: 73 instead of 84 us), so the walk
// is written for instruction count: what depends on the entry alone (both 1 / std) is computed ONCE per entry by one
// lane while the unit is set up and handed to the group through LDS, everything that comes in pairs is float2
// arithmetic (v_pk_fma_f32: the two argument orders of a hidden unit, feature pairs of the key), the softmax takes
// one exponential per entry (of the two factors exp(m - m'), exp(s - m') one is always exp(0)).
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int WTL>
__device__ __forceinline__ constexpr int fl_first_resident() { return WTL == 1 ? 1 : 0; }

// ZB: the node table Z is stored in bf16 (the bf16 throughput mode: 8 bytes per lane and entry instead of 16; widened to
// fp32 on arrival, everything else as in the fp32 kernel)
template <int G, int NTH, int WTL, bool ZB = false>
__global__ __launch_bounds__(NTH, NTH >= 512 ? 4 : 3) void pair_flip_kernel(const FlipArgs A) {
    using ZT = typename std::conditional<ZB, uint2, float4>::type;   // a lane's piece of a Z row as it travels
    constexpr int D = 4 * G, RS = D + 4, EPW = 64 / G, T_LO = fl_first_resident<WTL>();
    extern __shared__ __attribute__((aligned(16))) float4 fl_lds[];
    float4 *const ltab = fl_lds;                 // [4][3][G]: row j of hidden unit 4 lj + j, type t -> ((j * 3 + t) * G + lj)
    float4 *const lbase = fl_lds + 3 * D;        // [3][4][G]: P0, Q0, R0, C0 by feature quad
    float4 *const lwt = fl_lds + 6 * D;          // [WTL][D][G]: Wfold^T rows of the resident types
    int4 *const lrec = reinterpret_cast<int4 *>(fl_lds + 6 * D + WTL * D * G);   // [waves][EPW][16]: the units' records
    f32x2 *const lsc = reinterpret_cast<f32x2 *>(lrec + (NTH / 64) * EPW * 16);   // [waves][EPW][16]: (r1, r2) per entry
    const int lane = threadIdx.x & 63, grp = lane / G, lj = lane % G, off = 4 * lj;
    int64_t n[3], units[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        n[t] = A.type_ptr[(int64_t)t * (A.bs + 1) + A.bs];
        if (n[t] > A.ent_cap) n[t] = A.ent_cap;   // (overflow: flagged by the selection kernel, stay inside the region)
        units[t] = (n[t] + 15) >> 4;
    }
    const int64_t total_units = units[0] + units[1] + units[2];
    for (int i = threadIdx.x; i < 3 * D; i += NTH) {
        const int t = i / D, k = i % D;           // pe_tab[t][k] -> row k & 3 of the lane that owns unit k
        ltab[((k & 3) * 3 + t) * G + (k >> 2)] = reinterpret_cast<const float4 *>(A.pe_tab)[i];
        lbase[i] = reinterpret_cast<const float4 *>(A.base)[i];   // [t][4][D] floats = [t][4][G] float4
    }
    if constexpr (WTL > 0) {
        const float4 *src = reinterpret_cast<const float4 *>(A.wfoldT) + (int64_t)T_LO * D * G;
        for (int i = threadIdx.x; i < WTL * D * G; i += NTH) lwt[i] = src[i];
    }
#ifdef FL_STAMPS
    const uint64_t t_start = wall_clock64();
#endif
    int *const lticket = reinterpret_cast<int *>(lsc + (NTH / 64) * EPW * 16);
    if (threadIdx.x == 0) *lticket = 0;
    __syncthreads();
    const float4 at = *reinterpret_cast<const float4 *>(A.att + off);
    const f32x2 at01 = {at.x, at.y}, at23 = {at.z, at.w};

    // A workgroup owns every gridDim.x-th bundle of EPW units; INSIDE the workgroup they are handed out by a ticket in
    // LDS (the first round by position): what a unit costs depends on its flips -- a unit of strong entries takes three
    // or four times as long as one without -- and with a fixed stride per wavefront the slowest one finished long after
    // the average.  (One ticket for the whole grid in global memory was tried: ~12 k same-address atomics across the
    // eight XCDs serialise at ~20 ns each, the kernel took four times as long.)  The next ticket is drawn before the
    // current units are walked.
    auto draw = [&]() __attribute__((always_inline)) {
        int tk = 0;
        if (lane == 0) tk = atomicAdd(lticket, 1);
        return tk;
    };
    auto bundle = [&](int k) __attribute__((always_inline)) { return ((int64_t)k * gridDim.x + blockIdx.x) * EPW; };
    int tk_next = draw();
    for (int64_t u0 = bundle(threadIdx.x >> 6); u0 < total_units;) {
        const int64_t ug = u0 + grp;
        const bool live = ug < total_units;
        const int64_t uu = live ? ug : total_units - 1;
        const int t = uu < units[0] ? 0 : (uu < units[0] + units[1] ? 1 : 2);
        const int64_t U = uu - (t == 0 ? 0 : (t == 1 ? units[0] : units[0] + units[1]));
        const int64_t cnt = t == 0 ? n[0] : (t == 1 ? n[1] : n[2]);
        const int4 *ent = A.entries + (int64_t)t * A.ent_cap;
        const int64_t e0 = U * 16;
        const int nval = live ? (int)(cnt - e0 < 16 ? cnt - e0 : 16) : 0;   // >= 1 for a live unit
        const float4 *tabl0 = ltab + t * G + lj;      // rows j at tabl0[3 G j]
        const float4 *basel0 = lbase + t * D + lj;    // P0, Q0, R0, C0 at basel0[G v]
        const PeStat st = pe_load_stat(A.pe_stat, t);
        const float *wT = A.wfoldT + (int64_t)t * D * D + off;
        const bool wt_lds = t >= T_LO && t < T_LO + WTL;
        // (a group of a type that is not resident reads a resident column and throws it away: the address stays in LDS)
        const float4 *lw = lwt + (wt_lds ? (t - T_LO) * D * G : 0) + lj;
        const bool all_lds = WTL == 3 || __all(wt_lds);

        // neighbours of the unit: does its first entry start a segment, does its last one end one?
        auto rec_at = [&](int64_t i) __attribute__((always_inline)) {   // (clamped into the unit's valid entries)
            const int64_t e = e0 + (i < nval ? i : (nval > 0 ? nval - 1 : 0));
            return ent[live ? e : 0];
        };
        const int4 r_prev = ent[e0 > 0 ? e0 - 1 : 0], r_next = ent[(nval == 16 && e0 + 16 < cnt) ? e0 + 16 : e0];
        const int4 r_last = rec_at(15);
        const int prev_pair = e0 > 0 ? (int)((uint32_t)r_prev.x & FL_PAIR_MASK) : -1;
        const bool cont = nval == 16 && e0 + 16 < cnt &&
                          ((uint32_t)r_next.x & FL_PAIR_MASK) == ((uint32_t)r_last.x & FL_PAIR_MASK);
        float *const part_t = A.part + (int64_t)t * A.bs * RS;
        float *const bnd_u = A.bnd + (((int64_t)t * A.units_cap + U) * 2) * RS;

        float m = -INFINITY, l = 0.f;
        f32x2 o01 = {0.f, 0.f}, o23 = {0.f, 0.f};
        bool first = true;                       // no record flushed yet by this unit
        int cur_pair = 0, last_pair = prev_pair;
        bool st0 = false;
        auto flush = [&](int pair, bool cfront, bool cback) __attribute__((always_inline)) {
            float *dst = (cfront && cback) ? part_t + (int64_t)pair * RS : bnd_u + (cfront ? RS : 0);
            *reinterpret_cast<float4 *>(dst + off) = make_float4(o01.x, o01.y, o23.x, o23.y);
            if (lj == 0) *reinterpret_cast<float4 *>(dst + D) = make_float4(m, l, __int_as_float(pair), cback ? 0.f : 1.f);
        };
        auto z_row = [&](int node) __attribute__((always_inline)) {
            if constexpr (ZB)
                return *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(A.Z) +
                                                        (uint64_t)(uint32_t)node * A.ldz + off);
            else
                return *reinterpret_cast<const float4 *>(A.Z + (uint64_t)(uint32_t)node * A.ldz + off);
        };
        auto z_wide = [&](const ZT &z) __attribute__((always_inline)) {
            if constexpr (ZB)
                return make_float4(__uint_as_float(z.x << 16), __uint_as_float(z.x & 0xffff0000u),
                                   __uint_as_float(z.y << 16), __uint_as_float(z.y & 0xffff0000u));
            else
                return z;
        };
        auto q_row = [&](int pair) __attribute__((always_inline)) {
            return *reinterpret_cast<const float4 *>(A.q + (uint64_t)(uint32_t)pair * A.ldq + off);
        };
        // the walk, four entries at a time.  The unit's sixteen records go to LDS first (one 16-byte load per lane, one
        // round trip), so that the node ids of the NEXT batch are known while the current one is computed: its four Z
        // rows are requested a whole batch ahead and the only exposed round trips of a unit are the first two.  The
        // lane that fetched a record also computes the two 1 / std of its entry (under the first Z round trip).  The q
        // row rides one entry ahead of the arithmetic.
        int4 *const lr = lrec + ((threadIdx.x >> 6) * EPW + grp) * 16;
        f32x2 *const ls = lsc + ((threadIdx.x >> 6) * EPW + grp) * 16;
        int4 mine0 = make_int4(0, 0, 0, 0), mine1 = mine0;
        if (G >= 16) {
            if (lj < 16) lr[lj] = mine0 = rec_at(lj);
        } else {   // G = 8: two records per lane
            lr[lj] = mine0 = rec_at(lj);
            lr[lj + 8] = mine1 = rec_at(lj + 8);
        }
        int lro = 0;                 // (opaque zero: the record reads below must stay behind the stores above)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(lro) :: "memory");
        ZT za[4], zb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) za[u] = z_row(lr[lro + u].y);
        {
            const int4 r0 = lr[lro];
            cur_pair = (int)((uint32_t)r0.x & FL_PAIR_MASK);
            st0 = e0 == 0 || prev_pair != cur_pair;      // entry 0 starts a segment
        }
        float4 qc = q_row(cur_pair);
        if (G >= 16) {
            if (lj < 16) {
                const float pa = __int_as_float(mine0.z), pb = __int_as_float(mine0.w);
                ls[lj] = f32x2{fl_rstd(st, pa, pb), fl_rstd(st, pb, pa)};
            }
        } else {
            const float pa0 = __int_as_float(mine0.z), pb0 = __int_as_float(mine0.w);
            const float pa1 = __int_as_float(mine1.z), pb1 = __int_as_float(mine1.w);
            ls[lj] = f32x2{fl_rstd(st, pa0, pb0), fl_rstd(st, pb0, pa0)};
            ls[lj + 8] = f32x2{fl_rstd(st, pa1, pb1), fl_rstd(st, pb1, pa1)};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lro) :: "memory");

        // one entry: i = its position in the unit, zc = its Z row (this lane's four features)
        auto entry = [&](const int i, const ZT zraw) __attribute__((always_inline)) {
            const float4 zc = z_wide(zraw);
            const int4 rc = lr[lro + i];
            const f32x2 r12 = ls[lro + i];
            const int pair_n = (int)((uint32_t)lr[lro + (i < 15 ? i + 1 : 15)].x & FL_PAIR_MASK);
#ifdef FL_NOQ
            const float4 qn = make_float4(0.5f, 0.25f, 0.125f, 1.f);
#else
            const float4 qn = q_row(pair_n);
#endif
            const bool on = i < nval;
            const float pa = __int_as_float(rc.z), pb = __int_as_float(rc.w);
            const f32x2 pab = {pa, pb}, pba = {pb, pa};
            const int pair_i = (int)((uint32_t)rc.x & FL_PAIR_MASK);
            // (the table reads below do not depend on the entry: hide that from the optimiser, or it hoists all eight
            //  of them out of the loop and the registers are gone again)
            int toff = 0;
            asm volatile("" : "+v"(toff));
            const float4 *tabl = tabl0 + toff, *basel = basel0 + toff;
            // hidden layer: this lane's four units, both argument orders (.x: (pa, pb), .y: (pb, pa)).  The table rows
            // carry the sign of the unit's state at (0, 0), so z = (+-) y is negative exactly on the units that left
            // that pattern, and |y| = -z.
            f32x2 zz[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 tj = tabl[3 * G * j];
                zz[j] = r12 * (tj.x * pab + (tj.y * pba + tj.z)) + tj.w;
            }
            const float zmin = fminf(fminf(fminf(zz[0].x, zz[0].y), fminf(zz[1].x, zz[1].y)),
                                     fminf(fminf(zz[2].x, zz[2].y), fminf(zz[3].x, zz[3].y)));
            const bool fl = zmin < 0.f && on;
            const f32x2 cab = r12.x * pab + r12.y * pba;     // (r1 pa + r2 pb, r1 pb + r2 pa)
            const float cr = r12.x + r12.y;
            const float4 P0 = basel[0], Q0 = basel[G], R0 = basel[2 * G], C0 = basel[3 * G];
            f32x2 k01 = f32x2{zc.x, zc.y} +
                        (f32x2{P0.x, P0.y} * cab.x + (f32x2{Q0.x, Q0.y} * cab.y + (f32x2{R0.x, R0.y} * cr + f32x2{C0.x, C0.y})));
            f32x2 k23 = f32x2{zc.z, zc.w} +
                        (f32x2{P0.z, P0.w} * cab.x + (f32x2{Q0.z, Q0.w} * cab.y + (f32x2{R0.z, R0.w} * cr + f32x2{C0.z, C0.w})));
#ifdef FL_NOCORR
            if (false) {
#else
            if (__ballot(fl)) {
#endif
                // Some unit of some group left the pattern of (0, 0): every lane of that group owes Wfold[:, k] |y_k| for
                // it.  One pass over the eight (order, unit-of-the-lane) slots; the flipped lanes of a slot are taken one
                // at a time (scalar loop), the owner's |y| is read across the wave, the group's lanes add their piece of
                // the column -- from LDS when the type's table is resident (the other groups of the wave add zero times
                // whatever column they read).
                // (software-pipelined by one: a column is requested, then the PREVIOUS flip's column is added, so the
                //  scalar work of the next flip runs under the LDS round trip of this one)
                f32x2 wp01 = {0.f, 0.f}, wp23 = {0.f, 0.f};
                float vp = 0.f;
#pragma unroll
                for (int oj = 0; oj < 8; ++oj) {
                    const float zv = (oj & 4) ? zz[oj & 3].y : zz[oj & 3].x;
                    uint64_t bm = __ballot(zv < 0.f && on);
                    while (bm) {
                        const int b = __builtin_ctzll(bm);
                        bm &= bm - 1;
                        const float val = -__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zv), b));
                        const bool mine = b / G == grp;
                        const int kk = 4 * (b % G) + (oj & 3);     // (the same for every lane)
                        float4 w = lw[kk * G];
                        if (!all_lds) {
                            const float4 wg = *reinterpret_cast<const float4 *>(wT + (int64_t)kk * D);
                            w.x = wt_lds ? w.x : wg.x; w.y = wt_lds ? w.y : wg.y;
                            w.z = wt_lds ? w.z : wg.z; w.w = wt_lds ? w.w : wg.w;
                        }
                        k01 += wp01 * vp;
                        k23 += wp23 * vp;
                        wp01 = f32x2{w.x, w.y};
                        wp23 = f32x2{w.z, w.w};
                        vp = mine ? val : 0.f;
                    }
                }
                k01 += wp01 * vp;
                k23 += wp23 * vp;
            }
            // score of the entry: att . leaky_relu(k * q, 0.2), summed over the group's lanes
            const f32x2 x01 = k01 * f32x2{qc.x, qc.y}, x23 = k23 * f32x2{qc.z, qc.w};
            const f32x2 y01 = x01 * 0.2f, y23 = x23 * 0.2f;
            const f32x2 l01 = {fmaxf(x01.x, y01.x), fmaxf(x01.y, y01.y)}, l23 = {fmaxf(x23.x, y23.x), fmaxf(x23.y, y23.y)};
            const f32x2 sp = l01 * at01 + l23 * at23;
            const float s = fl_group_sum<G>(sp.x + sp.y);
            if (on) {
                if (i > 0 && pair_i != last_pair) {   // the previous entry closed a segment
                    flush(cur_pair, first ? st0 : true, true);
                    m = -INFINITY; l = 0.f;
                    o01 = f32x2{0.f, 0.f};
                    o23 = f32x2{0.f, 0.f};
                    first = false;
                }
                cur_pair = pair_i;
                // online softmax: exp(m - m') and exp(s - m') with m' = max(m, s) -- one of them is exp(0)
                const float d = s - m;                       // +inf on the first entry of a segment
                const float e = __expf(-fabsf(d));
                const bool up = d > 0.f;
                const float sca = up ? e : 1.f, w = up ? 1.f : e;
                l = fmaf(l, sca, w);
                o01 = o01 * sca + k01 * w;
                o23 = o23 * sca + k23 * w;
                m = fmaxf(m, s);
                last_pair = pair_i;
            }
            qc = qn;
            __builtin_amdgcn_sched_barrier(0);   // (left alone the scheduler hoists every load of the batch to its top)
        };
        // one batch: request the Z rows of the next one, then walk this one's four entries
        auto batch = [&](const int qt, const ZT (&zc4)[4], ZT (&zn4)[4]) __attribute__((always_inline)) {
            const int nb = qt < 3 ? 4 * qt + 4 : 12;        // (the last one re-requests itself: harmless)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#ifdef FL_NOZ
                zn4[u] = ZT{};
#else
                zn4[u] = z_row(lr[lro + nb + u].y);
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) entry(4 * qt + u, zc4[u]);
        };
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            batch(2 * h, za, zb);
            batch(2 * h + 1, zb, za);
        }
        if (nval > 0) flush(cur_pair, first ? st0 : true, !cont);
        u0 = bundle(NTH / 64 + __builtin_amdgcn_readfirstlane(tk_next));
        if (u0 < total_units) tk_next = draw();
    }
#ifdef FL_STAMPS
    // (tuning builds only: end time of every wavefront into the tail of the boundary-record buffer)
    if (lane == 0) {
        uint64_t *stamps = reinterpret_cast<uint64_t *>(A.bnd + (int64_t)3 * A.units_cap * 2 * RS) - (int64_t)gridDim.x * (NTH / 64) - 1;
        stamps[(int64_t)blockIdx.x * (NTH / 64) + (threadIdx.x >> 6)] = wall_clock64();
        if (blockIdx.x == 0 && threadIdx.x == 0) stamps[(int64_t)gridDim.x * (NTH / 64)] = t_start;
    }
#endif
}

}  // namespace

namespace {
template <bool ZB>
int flip_launch(int32_t D, int64_t bs, const int32_t *type_ptr, const void *entries, int64_t ent_cap, const void *Z,
                int64_t ldz, const float *q, int64_t ldq, const float *pe_tab_signed, const float *pe_stat,
                const float *base, const float *wfold_t, const float *att, float *part, float *bnd, int64_t units_cap,
                void *stream) {
    if (bs == 0) return LPF_OK;
    LPF_REQUIRE(bs > 0 && type_ptr && entries && ent_cap > 0 && Z && q && pe_tab_signed && pe_stat && base && wfold_t &&
                att && part && bnd && units_cap >= (ent_cap + 15) / 16);
    LPF_REQUIRE(ldz >= D && ldq >= D && ldz < (1ll << 31) && ldq < (1ll << 31) && (ldz & (ZB ? 7 : 3)) == 0 &&
                (ldq & 3) == 0 && lpf_aligned16(entries) && lpf_aligned16(Z) && lpf_aligned16(q) &&
                lpf_aligned16(pe_tab_signed) && lpf_aligned16(base) && lpf_aligned16(wfold_t) && lpf_aligned16(att) &&
                lpf_aligned16(part) && lpf_aligned16(bnd));
    const FlipArgs a{bs, type_ptr, static_cast<const int4 *>(entries), ent_cap, static_cast<const float *>(Z), (uint32_t)ldz,
                     q, (uint32_t)ldq, pe_tab_signed, pe_stat, base, wfold_t, att, part, bnd, units_cap};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int n_cu = lpf_cu_count();
    if (n_cu == 0) return LPF_ERR_NO_DEVICE;
    const int64_t max_units = 3 * ((ent_cap + 15) / 16) + 3;
    // persistent workgroups (they stride over the units they find): at D = 128 one of 1024 threads per CU around the
    // Wfold^T tables of two types (128 of its 152 KB of LDS), two of 512 threads below, three of 256 at D = 256
#define LPF_FLIP(GG, NTH, WTL, PER_CU)                                                                          \
    do {                                                                                                        \
        auto kern = pair_flip_kernel<GG, NTH, WTL, ZB>;                                                         \
        constexpr size_t lds = (size_t)(6 * 4 * GG + WTL * 4 * GG * GG + (NTH / 64) * (64 / GG) * 24 + 1) * sizeof(float4); \
        LPF_SET_MAX_LDS(kern, lds);                                                            \
        int64_t groups = (max_units + (NTH / 64) * (64 / GG) - 1) / ((NTH / 64) * (64 / GG));                   \
        if (groups > (int64_t)n_cu * PER_CU) groups = (int64_t)n_cu * PER_CU;                                   \
        hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(NTH), lds, s, a);                                 \
    } while (0)
    switch (D) {
        case 32: LPF_FLIP(8, 512, 3, 2); break;
        case 64: LPF_FLIP(16, 512, 3, 2); break;
#ifdef FL_CFG128      /* (tuning builds: other shapes of the D = 128 launch) */
#define LPF_FLIP_X(...) LPF_FLIP(__VA_ARGS__)
        case 128: LPF_FLIP_X(FL_CFG128); break;
#else
        case 128: LPF_FLIP(32, 1024, 2, 1); break;
#endif
        case 256: LPF_FLIP(64, 256, 0, 3); break;
        default: return LPF_ERR_UNSUPPORTED;
    }
#undef LPF_FLIP
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}
}  // namespace

extern "C" int lpf_pair_attention_flip_f32(int32_t D, int64_t bs, const int32_t *type_ptr, const void *entries,
                                           int64_t ent_cap, const float *Z, int64_t ldz, const float *q, int64_t ldq,
                                           const float *pe_tab_signed, const float *pe_stat, const float *base,
                                           const float *wfold_t, const float *att, float *part,
                                           float *bnd, int64_t units_cap, void *stream) {
    return flip_launch<false>(D, bs, type_ptr, entries, ent_cap, Z, ldz, q, ldq, pe_tab_signed, pe_stat, base, wfold_t, att,
                              part, bnd, units_cap, stream);
}

extern "C" int lpf_pair_attention_flip_zbf16(int32_t D, int64_t bs, const int32_t *type_ptr, const void *entries,
                                             int64_t ent_cap, const void *Z_bf16, int64_t ldz, const float *q, int64_t ldq,
                                             const float *pe_tab_signed, const float *pe_stat, const float *base,
                                             const float *wfold_t, const float *att, float *part, float *bnd,
                                             int64_t units_cap, void *stream) {
    return flip_launch<true>(D, bs, type_ptr, entries, ent_cap, Z_bf16, ldz, q, ldq, pe_tab_signed, pe_stat, base, wfold_t,
                             att, part, bnd, units_cap, stream);
}
