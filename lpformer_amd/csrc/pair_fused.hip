// Pairwise PPR-positional attention in ONE pass over the selected entries: score -> segment softmax -> weighted sum,
// every Z row gathered once, the per-entry key vector and the scores never written to memory.
//
// Reference: LinkAttention.message + PyG softmax + scatter-sum (src/modules/layers.py:193-224) with
// get_pos_encodings (src/models/link_transformer.py:182-211) folded in (algebra in DESIGN.md section 4):
//     k_e = Z[v_e] + Wfold_t h_e + bfold_t          s_e = att . leaky_relu(k_e * q[pair_e], 0.2)
//     alpha = softmax of s over the entries of a pair (max-shifted, denominator + 1e-16)
//     out_pair = sum_e alpha_e k_e + bias
// The selection kernels (select2.hip) leave per type t one dense region of {pair, node, pa, pb} records in which the
// entries of a pair are contiguous.  A wavefront takes a TILE of 32 consecutive same-type entries:
//   * v_mfma_f32_32x32x2_f32 with the ENTRIES AS ROWS: A operand = h_e generated in registers (first PE layer +
//     LayerNorm in closed form + ReLU, both argument orders), B operand = Wfold_t from a packed image resident in LDS
//     (shared by the workgroup's wavefronts).  The accumulators then hold k[entry][feature] with the FEATURE ON THE
//     LANE: Z and q rows are read as 128-byte lane-contiguous pieces, the score is one butterfly over 32 lanes, and
//     the weighted sum over the entries of a pair is an in-lane loop over registers.
//   * A-operand row r is fed with tile entry rho^-1(r), chosen so that lanes 0-31 own entries 0-15 and lanes 32-63
//     own entries 16-31 of the tile in register order: each half-wave is a UNIT of 16 consecutive entries that it
//     walks sequentially with an online softmax (running max / sum / weighted sum), flushing a record
//     {sum_e exp(s_e - m) k_e [D], m, l} whenever the pair changes.
//   * A pair's (type) segment that lies inside one unit is flushed straight to part[t][pair].  A segment that crosses
//     unit boundaries leaves one boundary record per unit it touches: slot 1 of the unit it starts in, slot 0 of every
//     unit after.  The tail kernel (tail_chain.hip, merge mode) finds a pair's records from its segment pointers,
//     merges them with the same rescaling as the online softmax, adds the bias and applies post_att_norm.
// Bound: fp32 MFMA (2 D^2 FLOP per entry); one Z row + one 16-byte record read per entry.
#include "pe_common.h"

// Tuning aids, compiled only with -DLPF_FUSED_STAMPS (make EXTRA=-DLPF_FUSED_STAMPS; tools/fused_stamps.py): s_memtime at
// the phase boundaries of the first eight tiles of 512 wavefronts, and the LPF_FUSED_DBG environment switches that
// ablate parts of the kernel (bit 0 no Z/q loads, 1 no softmax/flush, 2 no MFMA, 3 no butterfly, 5 stamps).  The
// default build has neither: PF_DBG() is the constant 0 and no environment variable is read.
#ifdef LPF_FUSED_STAMPS
#include <stdlib.h>
__device__ long long g_pf_stamps[4096 * 8];
#define PF_DBG(bits) (A.dbg & (bits))
#else
#define PF_DBG(bits) (0)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// wavefronts per workgroup sharing one LDS copy of Wfold_t = one workgroup per CU at the register budget of the tile
// code (NT = 1: 88 VGPRs -> 4 waves per SIMD; NT = 2: 142 -> 3; NT = 4: ~200 -> 2 -- at 3 waves per SIMD it spills 82
// registers to scratch, which cost more than half of the kernel's time); LDS: 16 NT KB of Z rows per wavefront
// NT = 8 (D = 256): 128 accumulator registers + a prefetched weight group -> ~300 VGPRs, ONE wave per SIMD (4 per
// workgroup), 32 KB of Z rows per wavefront.
template <int NT>
constexpr int pf_waves() { return NT == 1 ? 16 : (NT == 2 ? 12 : (NT == 4 ? 8 : 4)); }
constexpr uint32_t PF_PAIR_MASK = 0x7fffffffu;

struct FusedArgs {
    int64_t bs;
    const int32_t *type_ptr;   // [3][bs+1]
    const int4 *entries;       // [3][ent_cap]
    int64_t ent_cap;
    const float *Z; int64_t ldz;   // BF16 kernels: Z points at bf16 rows, ldz in bf16 elements
    const float *q; int64_t ldq;
    const float *pe_tab, *pe_stat, *wpk, *bfold, *att;
    float *part;               // [3][bs][D+4]
    float *bnd;                // [3][units_cap][2][D+4]
    int64_t units_cap;
    int dbg;                   // tuning aid (LPF_FUSED_STAMPS builds only)
};

template <int NT, bool BF16>
__device__ __forceinline__ void fused_tile(const FusedArgs &A, int t, int64_t idx, int64_t cnt, const float4 *wl,
                                           const float4 *tab, float *zbuf, int lane, int stamp_slot) {
    constexpr int D = 32 * NT, NSQ = D / 8, RS = D + 4;
    constexpr int ZB = BF16 ? 2 : 4;  // bytes per Z element
    const int col = lane & 31, lh = lane >> 5;
    // tile entry fed to A-operand row `col`: half 0 of the accumulator rows {0-3, 8-11, ...} <- entries 0..15
    const int ja = 16 * ((col >> 2) & 1) + 4 * (col >> 3) + (col & 3);
    const int4 *ent = A.entries + (int64_t)t * A.ent_cap;
    const int64_t e0 = idx * 32;
    const int64_t ea = e0 + ja;
    const bool valid_a = ea < cnt;
    int4 rec = make_int4(0, 0, 0, 0);
    int prev_pair = -1;
    if (valid_a) {
        rec = ent[ea];
        if (ea > 0) prev_pair = (int)((uint32_t)ent[ea - 1].x & PF_PAIR_MASK);
    }
    const int pair_a = (int)((uint32_t)rec.x & PF_PAIR_MASK), node_a = rec.y;
    const float pa = __int_as_float(rec.z), pb = __int_as_float(rec.w);
    const uint64_t sm = __ballot(valid_a && prev_pair != pair_a);  // bit r: the entry of row r starts a segment
    // does the segment of the tile's last entry continue in the next tile?
    bool tile_cont = false;
    if (e0 + 32 < cnt)
        tile_cont = ((uint32_t)ent[e0 + 32].x & PF_PAIR_MASK) == ((uint32_t)ent[e0 + 31].x & PF_PAIR_MASK);

#ifdef LPF_FUSED_STAMPS
    const bool stamp = PF_DBG(32) && lane == 0 && stamp_slot >= 0;
#define PF_STAMP(k) do { if (stamp) g_pf_stamps[stamp_slot * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
    (void)stamp_slot;
#define PF_STAMP(k) do { } while (0)
#endif
    PF_STAMP(0);
    const PeStat st = pe_load_stat(A.pe_stat, t);
    const float r_ab = pe_rstd(st, pa, pb), r_ba = pe_rstd(st, pb, pa);


    // The tile's 32 Z rows go straight to this wavefront's LDS buffer (LDS-DMA: no VGPR destination), in unit order
    // (row j = tile entry j): each instruction moves 64 x 16 bytes = RPI whole rows, lane l -> row m RPI + l / LPR,
    // 16-byte piece l % LPR.  They are issued before the MFMA loop and land while it runs (the loop's first weight
    // load waits for them -- hipcc's vmcnt is not selective --, the SIMD's other wavefront covers that); the epilogue
    // then reads Z from LDS.  Register staging of the same rows is what limited the first version of this kernel:
    // two rows in flight per lane at a time, every pair of entries waiting out a trip to the Infinity Cache.
    {
        constexpr int LPR = D * ZB / 16, RPI = 64 / LPR;  // lanes per row (16 bytes each), rows per instruction
        const char *zbase = reinterpret_cast<const char *>(A.Z);
#pragma unroll
        for (int mi = 0; mi < 32 / RPI; ++mi) {
            const int j = mi * RPI + lane / LPR;                                   // tile entry (unit order)
            const int src = 8 * ((j & 15) >> 2) + 4 * (j >> 4) + (j & 3);          // A row that carries entry j
            const int nd = __shfl(node_a, src, 64);
            const char *g = zbase + ((int64_t)nd * A.ldz) * ZB + 16 * (lane % LPR);
            __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void *)(zbuf + mi * 256), 16, 0, 0);
        }
    }

    // per-feature constants: bfold_t seeds the accumulators (k = Z + Wfold h + bfold comes out of the MFMA loop with
    // the bias already in), att is needed in the epilogue
    float at[NT];
    f32x16 acc[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const float bfc = A.bfold[t * D + 32 * c + col];
        at[c] = A.att[32 * c + col];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = bfc;
    }
    const float4 *tb = tab + t * D + lh * (D / 2);
    // (Letting the wavefronts of a SIMD take turns in the MFMA loop through an LDS token -- to keep one wave's
    // epilogue beside another's matrix loop -- measured slower: 203 vs 194 us.)
    PF_STAMP(1);
    if constexpr (!BF16 && NT == 8) {
        // D = 256: ONE wavefront per SIMD, so nobody covers the L2 round trip of a weight group (8 x 16 bytes per lane
        // and k-group).  The groups are double-buffered by hand: group k + 1 is requested before the 32 MFMAs of
        // group k.  Written with inline assembly because hipcc 7.2 sinks a C++ prefetch back to its use (the
        // load of the NEXT iteration's operands ends up at the top of that iteration, in front of an s_waitcnt: 138 k
        // instead of 66 k clocks per tile); the waits are counted by hand -- vmcnt(8) = "everything but the eight
        // newest requests has landed".  Older requests (the Z-row DMA, the records) are in front of them in the queue.
        typedef float f32x4n __attribute__((ext_vector_type(4)));  // (a native vector: inline-asm register operand)
        f32x4n w0[NT], w1[NT];
        auto issue = [&](f32x4n (&w)[NT], int sq) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < NT; ++c)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w[c]) : "v"(wl + (PF_DBG(64) ? 0 : (c * NSQ + sq) * 64)) : "memory");
        };
        auto group = [&](f32x4n (&w)[NT], int sq) __attribute__((always_inline)) {
            float h[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) h[u] = pe_hidden(tb[4 * sq + u], pa, pb, r_ab, r_ba);
            // (the group's registers pass THROUGH the wait: nothing that reads them can be scheduled in front of it)
            asm volatile("s_waitcnt vmcnt(8)"
                         : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7])
                         :: "memory");
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(h[0], w[c][0], acc[c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(h[1], w[c][1], acc[c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(h[2], w[c][2], acc[c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(h[3], w[c][3], acc[c], 0, 0, 0);
        };
        issue(w0, 0);
#pragma unroll 1
        for (int sq = 0; sq < NSQ; sq += 2) {
            issue(w1, sq + 1);
            group(w0, sq);
            issue(w0, sq + 2 < NSQ ? sq + 2 : sq);   // (the last one is a dummy: the count of requests in flight stays 8)
            group(w1, sq + 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if constexpr (!BF16) {
#pragma unroll 1
        for (int sq = 0; sq < (PF_DBG(4) ? 0 : NSQ); ++sq) {
            float4 wb[NT];  // (an explicit one-group-ahead prefetch of these measured slower: 198 vs 173 us)
#pragma unroll
            for (int c = 0; c < NT; ++c) wb[c] = wl[(c * NSQ + sq) * 64];
            float h[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) h[u] = pe_hidden(tb[4 * sq + u], pa, pb, r_ab, r_ba);
            // consecutive MFMAs go to different accumulators (no back-to-back dependency on one accumulator)
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(h[0], wb[c].x, acc[c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(h[1], wb[c].y, acc[c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(h[2], wb[c].z, acc[c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(h[3], wb[c].w, acc[c], 0, 0, 0);
        }
    } else {
        // bf16 throughput mode: v_mfma_f32_32x32x16_bf16, fp32 accumulate.  K-step s covers k = 16 s .. 16 s + 15;
        // lane (row, half) holds A[row][16 s + 8 half + j] = bf16(h_e[k]) and B[k][col] = bf16(Wfold_t[col][k]) from
        // the bf16 image (element (t, c, s, lane, j) = Wfold_t[32 c + (lane & 31)][16 s + 8 (lane >> 5) + j]).
        const float4 *tk = tab + t * D + 8 * lh;
#pragma unroll 1
        for (int ks = 0; ks < (PF_DBG(4) ? 0 : D / 16); ++ks) {
            bf16x8 wbb[NT];
#pragma unroll
            for (int c = 0; c < NT; ++c)
                wbb[c] = __builtin_bit_cast(bf16x8, wl[(c * (D / 16) + ks) * 64]);
            bf16x8 ha;
#pragma unroll
            for (int j = 0; j < 8; ++j) ha[j] = (__bf16)pe_hidden(tk[16 * ks + j], pa, pb, r_ab, r_ba);
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, wbb[c], acc[c], 0, 0, 0);
        }
    }
    // acc[c][i] = (Wfold_t h + bfold_t)[feature 32c + col] of unit entry i (tile entry 16 lh + i)

    // ---- scores of the unit's 16 entries.  Entries are taken two at a time (16 row pieces in flight per lane) with a
    // scheduling barrier in between: left to itself the scheduler hoists all 128 loads of the unit and spills.
    PF_STAMP(2);
    float sc[16];
    // q rows (global, mostly L1/L2 hits: consecutive entries share their pair) are requested one pair of entries
    // ahead of the arithmetic that uses them; the Z rows come from the LDS buffer
    float qn[2][NT];
    auto load_q = [&](int i0, float (&dst)[2][NT]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = i0 + u;
            const int r0 = 8 * (i >> 2) + (i & 3);  // A row that carried unit entry i of half 0 (half 1: r0 + 4)
            const int pr0 = __builtin_amdgcn_readlane(pair_a, r0), pr1 = __builtin_amdgcn_readlane(pair_a, r0 + 4);
            const float *qr = A.q + (int64_t)(lh ? pr1 : pr0) * A.ldq + col;
#pragma unroll
            for (int c = 0; c < NT; ++c) dst[u][c] = PF_DBG(1) ? 0.25f : qr[32 * c];
        }
    };
    load_q(0, qn);
#pragma unroll
    for (int i0 = 0; i0 < 16; i0 += 2) {
        float zv[2][NT], qv[2][NT];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int c = 0; c < NT; ++c) qv[u][c] = qn[u][c];
        if (i0 + 2 < 16) load_q(i0 + 2, qn);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = i0 + u;
#pragma unroll
            for (int c = 0; c < NT; ++c) {  // the staged row of unit entry i (LDS)
                if constexpr (BF16) {
                    const uint16_t zb = reinterpret_cast<const uint16_t *>(zbuf)[(16 * lh + i) * D + 32 * c + col];
                    zv[u][c] = PF_DBG(1) ? 0.5f : __uint_as_float((uint32_t)zb << 16);
                } else {
                    zv[u][c] = PF_DBG(1) ? 0.5f : zbuf[(16 * lh + i) * D + 32 * c + col];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = i0 + u;
            float p = 0.f;
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                const float k = acc[c][i] + zv[u][c];
                acc[c][i] = k;
                float x = k * qv[u][c];
                x = fmaxf(x, 0.2f * x);  // leaky_relu(x, 0.2)
                p = fmaf(x, at[c], p);
            }
            sc[i] = p;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    PF_STAMP(3);
    // Sum over the 32 feature lanes of the half, for 16 scores at once: a butterfly that halves the number of live
    // values at every step (a lane keeps the scores whose index bit k equals its lane bit k and hands the others to
    // its partner), 8 + 4 + 2 + 1 adds instead of 16 x 4, then one exchange with the other 16-lane row.  Lane l of
    // the half then holds the total of unit entry l & 15; every lane needs every total (they weight its own feature in
    // the softmax walk), so they are read back lane by lane.
    if (!PF_DBG(8)) {
        const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
        float r1[8], r2[4], r3[2];
#pragma unroll
        for (int j = 0; j < 8; ++j) {  // partner lane ^ 1 (quad_perm [1,0,3,2])
            const float keep = b0 ? sc[2 * j + 1] : sc[2 * j], send = b0 ? sc[2 * j] : sc[2 * j + 1];
            r1[j] = keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xf, 0xf, false));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // partner lane ^ 2 (quad_perm [2,3,0,1])
            const float keep = b1 ? r1[2 * j + 1] : r1[2 * j], send = b1 ? r1[2 * j] : r1[2 * j + 1];
            r2[j] = keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x4E, 0xf, 0xf, false));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {  // partner lane ^ 4: rotate by 12 for the banks with bit 2 clear, by 4 for the others
            const float keep = b2 ? r2[2 * j + 1] : r2[2 * j], send = b2 ? r2[2 * j] : r2[2 * j + 1];
            int got = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x12C, 0xf, 0x5, false);
            got = __builtin_amdgcn_update_dpp(got, __builtin_bit_cast(int, send), 0x124, 0xf, 0xa, false);
            r3[j] = keep + __builtin_bit_cast(float, got);
        }
        float r4;
        {   // partner lane ^ 8 (rotate by 8)
            const float keep = b3 ? r3[1] : r3[0], send = b3 ? r3[0] : r3[1];
            r4 = keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x128, 0xf, 0xf, false));
        }
        r4 += __shfl_xor(r4, 16, 64);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float t0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r4), i));
            const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r4), 32 + i));
            sc[i] = lh ? t1 : t0;
        }
    }

    PF_STAMP(4);
    // ---- online softmax over the unit's entries, one record per (pair) piece
    int64_t left = cnt - e0 - 16 * lh;
    const int nval = left >= 16 ? 16 : (left > 0 ? (int)left : 0);
    if (nval == 0) return;
    if (PF_DBG(2)) {  // (tuning aid: keep the scores alive, skip the softmax walk)
        float t_ = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t_ += sc[i] + acc[0][i];
        if (t_ == 12345.678f) A.part[0] = t_;
        return;
    }
    const uint32_t smh = (uint32_t)(sm >> (4 * lh));     // start bit of unit entry i at position 8 (i>>2) + (i&3)
    const bool st0 = smh & 1u;
    // the unit's last entry: does its segment continue in the next unit?
    bool cont = false;
    if (nval == 16) cont = lh == 0 ? ((cnt > e0 + 16) && !((sm >> 4) & 1ull)) : tile_cont;
    const int64_t U = 2 * idx + lh;
    float *const part_t = A.part + (int64_t)t * A.bs * RS;
    float *const bnd_u = A.bnd + (((int64_t)t * A.units_cap + U) * 2) * RS;

    float m = -INFINITY, l = 0.f, o[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) o[c] = 0.f;
    bool first = true;
    int cur_pair = 0;
    auto flush = [&](int pair, bool cfront, bool cback) __attribute__((always_inline)) {
        float *dst = (cfront && cback) ? part_t + (int64_t)pair * RS : bnd_u + (cfront ? RS : 0);
#pragma unroll
        for (int c = 0; c < NT; ++c) dst[32 * c + col] = o[c];
        if (col == 0)
            *reinterpret_cast<float4 *>(dst + D) = make_float4(m, l, __int_as_float(pair), cback ? 0.f : 1.f);
    };
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int pi0 = __builtin_amdgcn_readlane(pair_a, 8 * (i >> 2) + (i & 3));
        const int pi1 = __builtin_amdgcn_readlane(pair_a, 8 * (i >> 2) + 4 + (i & 3));
        const int pair_i = lh ? pi1 : pi0;
        if (i < nval) {
            const bool sti = (smh >> (8 * (i >> 2) + (i & 3))) & 1u;
            if (i > 0 && sti) {
                flush(cur_pair, first ? st0 : true, true);
                m = -INFINITY; l = 0.f;
#pragma unroll
                for (int c = 0; c < NT; ++c) o[c] = 0.f;
                first = false;
            }
            cur_pair = pair_i;
            const float mn = fmaxf(m, sc[i]);
            const float sca = __expf(m - mn), w = __expf(sc[i] - mn);
            l = fmaf(l, sca, w);
#pragma unroll
            for (int c = 0; c < NT; ++c) o[c] = fmaf(o[c], sca, w * acc[c][i]);
            m = mn;
        }
    }
    flush(cur_pair, first ? st0 : true, !cont);
    PF_STAMP(5);
#undef PF_STAMP
}

template <int NT, bool BF16>
__global__ __launch_bounds__(64 * pf_waves<NT>()) void pair_fused_kernel(const FusedArgs A) {
    constexpr int D = 32 * NT, NSQ = D / 8;
    constexpr int IMG = BF16 ? NT * (D / 16) * 64 : NT * NSQ * 64;  // 16-byte groups of the weight image per type
    constexpr int PF_WAVES = pf_waves<NT>();
    // LDS: the first PE layer's table, then one 32-row Z buffer per wavefront.  The packed Wfold_t image (B operand)
    // is streamed from L2: LDS cannot hold both it and the row buffers, and the rows are the part that pays.
    extern __shared__ __attribute__((aligned(16))) float4 pf_lds[];
    float4 *tab = pf_lds;
    for (int i = threadIdx.x; i < 3 * D; i += blockDim.x) tab[i] = reinterpret_cast<const float4 *>(A.pe_tab)[i];
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *zbuf = reinterpret_cast<float *>(pf_lds + 3 * D) + wave * (32 * D) / (BF16 ? 2 : 1);
    int64_t n[3], tiles[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        n[t] = A.type_ptr[(int64_t)t * (A.bs + 1) + A.bs];
        if (n[t] > A.ent_cap) n[t] = A.ent_cap;  // (overflow: flagged by the selection kernel, stay inside the region)
        tiles[t] = (n[t] + 31) >> 5;
    }
    const int64_t wave_id = (int64_t)blockIdx.x * PF_WAVES + wave, n_waves = (int64_t)gridDim.x * PF_WAVES;
    // (The two wavefronts of a SIMD -- waves w and w + PF_WAVES/2 -- run the same program and stay within a few thousand
    // clocks of each other, tile after tile.  Starting the second one late does not help, it only costs the delay:
    // 170 / 171 / 173 / 175 / 177 us for 0 / 6 k / 12 k / 18 k / 24 k clocks -- fp32 MFMA and vector instructions of
    // two wavefronts share the SIMD's ALUs, so being out of phase buys nothing.)
    for (int64_t tile = wave_id; tile < tiles[0] + tiles[1] + tiles[2]; tile += n_waves) {
        int t;
        int64_t idx;
        if (tile < tiles[0]) { t = 0; idx = tile; }
        else if (tile < tiles[0] + tiles[1]) { t = 1; idx = tile - tiles[0]; }
        else { t = 2; idx = tile - tiles[0] - tiles[1]; }
        const float4 *wp = reinterpret_cast<const float4 *>(A.wpk) + (int64_t)t * IMG + lane;
        const int64_t k_tile = (tile - wave_id) / n_waves;
        fused_tile<NT, BF16>(A, t, idx, n[t], wp, tab, zbuf, lane,
                             (blockIdx.x < 64 && k_tile < 8) ? (int)((blockIdx.x * PF_WAVES + wave) * 8 + k_tile) : -1);
    }
}

}  // namespace

namespace {

template <bool BF16>
int fused_launch(int32_t D, int64_t bs, const int32_t *type_ptr, const void *entries, int64_t ent_cap, const void *Z,
                 int64_t ldz, const float *q, int64_t ldq, const float *pe_tab, const float *pe_stat,
                 const void *wfold_packed, const float *bfold, const float *att, float *part, float *bnd,
                 int64_t units_cap, void *stream) {
    if (bs == 0) return LPF_OK;
    LPF_REQUIRE(bs > 0 && type_ptr && entries && ent_cap > 0 && Z && q && pe_tab && pe_stat && wfold_packed &&
                bfold && att && part && bnd && units_cap >= (ent_cap + 15) / 16);
    LPF_REQUIRE(ldz >= D && ldq >= D && lpf_aligned16(entries) && lpf_aligned16(pe_tab) &&
                lpf_aligned16(wfold_packed) && lpf_aligned16(part) && lpf_aligned16(bnd) && lpf_aligned16(Z) &&
                (ldz * (BF16 ? 2 : 4)) % 16 == 0);
#ifdef LPF_FUSED_STAMPS
    static int dbg = -1;
    if (dbg < 0) {
        const char *e = getenv("LPF_FUSED_DBG");
        dbg = e ? atoi(e) : 0;
    }
#else
    const int dbg = 0;
#endif
    FusedArgs a{bs, type_ptr, static_cast<const int4 *>(entries), ent_cap, static_cast<const float *>(Z), ldz, q, ldq,
                pe_tab, pe_stat, static_cast<const float *>(wfold_packed), bfold, att, part, bnd, units_cap, dbg};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int n_cu = lpf_cu_count();
    if (n_cu == 0) return LPF_ERR_NO_DEVICE;
    const int64_t tiles = (3 * ent_cap + 31) / 32 + 3;
#define LPF_FUSED(NT)                                                                                              \
    do {                                                                                                           \
        auto kern = pair_fused_kernel<NT, BF16>;                                                                   \
        constexpr int PF_WAVES = pf_waves<NT>();                                                                   \
        const size_t lds = (size_t)(3 * 32 * NT) * sizeof(float4) +                                                \
                           (size_t)PF_WAVES * 32 * 32 * NT * (BF16 ? 2 : 4);                                       \
        LPF_SET_MAX_LDS(kern, lds);                                                                                \
        static LpfPerDevice occ__; /* resident workgroups per CU, queried once per device */                       \
        const int per_cu = lpf_blocks_per_cu(occ__, reinterpret_cast<const void *>(kern), 64 * PF_WAVES, lds, 1);  \
        int64_t groups = (tiles + PF_WAVES - 1) / PF_WAVES;                                                        \
        if (groups > (int64_t)n_cu * per_cu) groups = (int64_t)n_cu * per_cu; /* persistent: one resident round */ \
        hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(64 * PF_WAVES), lds, s, a);                          \
    } while (0)
    switch (D) {
        case 32: LPF_FUSED(1); break;
        case 64: LPF_FUSED(2); break;
        case 128: LPF_FUSED(4); break;
        case 256: LPF_FUSED(8); break;
        default: return LPF_ERR_UNSUPPORTED;
    }
#undef LPF_FUSED
    // (the boundary records of segments that cross units are merged by the consumer, lpf_tail_chain_merge_f32, which
    // finds them from the segment pointers)
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

}  // namespace

extern "C" int lpf_pair_attention_fused_f32(int32_t D, int64_t bs, const int32_t *type_ptr, const void *entries,
                                            int64_t ent_cap, const float *Z, int64_t ldz, const float *q,
                                            int64_t ldq, const float *pe_tab, const float *pe_stat,
                                            const float *wfold_packed, const float *bfold, const float *att,
                                            float *part, float *bnd, int64_t units_cap,
                                            void *stream) {
    return fused_launch<false>(D, bs, type_ptr, entries, ent_cap, Z, ldz, q, ldq, pe_tab, pe_stat, wfold_packed, bfold,
                               att, part, bnd, units_cap, stream);
}

extern "C" int lpf_pair_attention_fused_bf16(int32_t D, int64_t bs, const int32_t *type_ptr, const void *entries,
                                             int64_t ent_cap, const void *Z_bf16, int64_t ldz, const float *q,
                                             int64_t ldq, const float *pe_tab, const float *pe_stat,
                                             const void *wfold_packed_bf16, const float *bfold, const float *att,
                                             float *part, float *bnd, int64_t units_cap,
                                             void *stream) {
    return fused_launch<true>(D, bs, type_ptr, entries, ent_cap, Z_bf16, ldz, q, ldq, pe_tab, pe_stat,
                              wfold_packed_bf16, bfold, att, part, bnd, units_cap, stream);
}

#ifdef LPF_FUSED_STAMPS
extern "C" int lpf_fused_debug_stamps(long long *dst_host, int64_t n) {
    if (n > 4096 * 8) n = 4096 * 8;
    return hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g_pf_stamps), (size_t)n * sizeof(long long)) == hipSuccess ? LPF_OK
                                                                                                                : LPF_ERR_LAUNCH;
}
#endif
