// Pairwise PPR-positional attention (src/modules/layers.py:161-224 + get_pos_encodings, link_transformer.py:182-211)
// restructured so that the per-entry key vector k_e is never written to memory (algebra in DESIGN.md):
//
//   phase A  lpf_pair_scores_f32          score_e = att . leaky_relu((Z[v_e] + Wfold_t h_e + bfold_t) * q[pair_e])
//            tiles of 32 same-type entries; h_e (first PE layer + LayerNorm + ReLU, both argument orders) is
//            generated in registers as the B operand of v_mfma_f32_32x32x2_f32, Wfold_t is the A operand from a
//            pre-packed image (one contiguous 1 KiB wave read per 4 MFMA steps) -- resident in LDS and shared by the
//            8 waves of a persistent workgroup when it fits (D <= 128), streamed from L2 otherwise; accumulators hold
//            k_e^T = [feature][entry]; the epilogue gathers Z / q rows in 16-byte pieces and reduces over features
//            in-lane (+ one cross-half shuffle).  Bound: fp32 MFMA (2*D*D FLOP per entry).
//   phase B  lpf_pair_softmax_gather_f32  per-pair segment softmax, then the alpha-weighted sums of Z rows and of
//            h_e per type; a group of D/4 lanes owns a pair (16-byte pieces of each row).  Bound: gather bandwidth.
//   phase C  is a plain GEMM (lpf_gemm_f32) on [sum alpha h_e | sum alpha | 1] with [Wfold | bfold | bias].
#include "pe_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SG_HEAVY = 64;        // pairs with more selected nodes than this get a whole workgroup in phase B
constexpr int SG_HEAVY_CHUNK = 16;  // entries per lane-group chunk in that kernel

// One tile of 32 same-type entries handled by one wavefront.  `wp` points at this lane's slice of the packed
// Wfold_t image (global memory or the workgroup's LDS copy): wp[(c*NSQ + sq)*64] is the A operand of step group sq for
// output-feature tile c.
template <int NT>
__device__ __forceinline__ void pair_scores_tile(
    int t, int64_t e, bool valid, int lh, const float4 *wp, const float4 *tab, const int32_t *__restrict__ sel_pair,
    const int32_t *__restrict__ sel_node, const float *__restrict__ sel_pa, const float *__restrict__ sel_pb,
    const float *__restrict__ Z, int64_t ldz, const float *__restrict__ q, int64_t ldq,
    const float *__restrict__ pe_stat, const float *__restrict__ bfold, const float *__restrict__ att,
    float *__restrict__ score) {
    constexpr int D = 32 * NT;
    constexpr int NSQ = D / 8;  // groups of 4 MFMA steps (each step consumes 2 values of k)
    const float pa = valid ? sel_pa[e] : 0.f, pb = valid ? sel_pb[e] : 0.f;
    const int32_t node = valid ? sel_node[e] : 0, pr = valid ? sel_pair[e] : 0;

    PeStat st;
    st.c00 = pe_stat[8 * t + 0]; st.c11 = pe_stat[8 * t + 1]; st.cbb = pe_stat[8 * t + 2];
    st.c01 = pe_stat[8 * t + 3]; st.c0b = pe_stat[8 * t + 4]; st.c1b = pe_stat[8 * t + 5];
    const float r_ab = pe_rstd(st, pa, pb), r_ba = pe_rstd(st, pb, pa);

    f32x16 acc[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    // Touch every 128-byte line of the entry's Z row and of its pair's q row now (one dword per line, the two lanes of
    // an entry split the lines): the rows are consumed only in the epilogue, and their HBM latency then overlaps the
    // MFMA loop instead of following it.  The values are kept (4 registers) so the loads cannot be dropped.
    const float *zrow = Z + (int64_t)node * ldz;
    const float *qrow = q + (int64_t)pr * ldq;
    constexpr int PF = NT >= 2 ? NT / 2 : 1;
    float pfz[PF], pfq[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        const int line = (lh * PF + i) % NT;
        pfz[i] = zrow[32 * line];
        pfq[i] = qrow[32 * line];
    }

    const float4 *tb = tab + t * D + lh * (D / 2);
    // No software prefetch of the A operands: one wavefront alone issues this MFMA at half rate (measured: 130
    // cycles per v_mfma_f32_32x32x2_f32 from one wave, 67 from two), so the kernel is built for >= 3 resident waves per
    // SIMD and lets the other waves cover the operand latency; registers are spent on occupancy instead.
#pragma unroll 1
    for (int sq = 0; sq < NSQ; ++sq) {
        float4 wa[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) wa[c] = wp[(c * NSQ + sq) * 64];
        float h[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) h[u] = pe_hidden(tb[4 * sq + u], pa, pb, r_ab, r_ba);
        // consecutive MFMAs go to different accumulators (no back-to-back dependency on one accumulator)
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[c].x, h[0], acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[c].y, h[1], acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[c].z, h[2], acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[c].w, h[3], acc[c], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) asm volatile("" ::"v"(pfz[i]), "v"(pfq[i]));  // (the line touches end here)
    // acc[c][4g+u] = (Wfold_t h_e)[feature 32c + 8g + 4*lh + u] for entry lj
    const float *bf = bfold + t * D;
    float part = 0.f;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f0 = 32 * c + 8 * g + 4 * lh;
            const float4 z4 = *reinterpret_cast<const float4 *>(zrow + f0);
            const float4 q4 = *reinterpret_cast<const float4 *>(qrow + f0);
            const float4 b4 = *reinterpret_cast<const float4 *>(bf + f0);
            const float4 a4 = *reinterpret_cast<const float4 *>(att + f0);
            const float zz[4] = {z4.x, z4.y, z4.z, z4.w}, qq[4] = {q4.x, q4.y, q4.z, q4.w};
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, aa[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float x = (acc[c][4 * g + u] + zz[u] + bb[u]) * qq[u];
                x = x > 0.f ? x : 0.2f * x;
                part = fmaf(x, aa[u], part);
            }
        }
    }
    part += __shfl_xor(part, 32, 64);
    if (valid && lh == 0) score[e] = part;
}

// Wfold streamed from global memory / L2 (packed image, L2-resident).  Register budget: 3 waves per SIMD without
// spilling for D <= 128 (a 4-wave budget spills ~150 MB per launch to scratch: measured slower).
template <int NT>
__global__ __launch_bounds__(256, (NT <= 4 ? 3 : 2)) void pair_scores_kernel(
    const int64_t *__restrict__ type_ptr, int64_t bs, const int32_t *__restrict__ sel_pair,
    const int32_t *__restrict__ sel_node, const float *__restrict__ sel_pa, const float *__restrict__ sel_pb,
    const float *__restrict__ Z, int64_t ldz, const float *__restrict__ q, int64_t ldq,
    const float *__restrict__ pe_tab, const float *__restrict__ pe_stat, const float *__restrict__ wpk,
    const float *__restrict__ bfold, const float *__restrict__ att, float *__restrict__ score) {
    constexpr int D = 32 * NT;
    constexpr int NSQ = D / 8;
    __shared__ float4 tab[3 * D];
    for (int i = threadIdx.x; i < 3 * D; i += blockDim.x) tab[i] = reinterpret_cast<const float4 *>(pe_tab)[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int lj = lane & 31, lh = lane >> 5;
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t n0 = type_ptr[bs], n1 = type_ptr[(bs + 1) + bs], n2 = type_ptr[2 * (bs + 1) + bs];
    const int64_t t0 = (n0 + 31) >> 5, t1 = (n1 + 31) >> 5, t2 = (n2 + 31) >> 5;

    for (int64_t tile = wave_id; tile < t0 + t1 + t2; tile += n_waves) {
        int t;
        int64_t idx, base, cnt;
        if (tile < t0) { t = 0; idx = tile; base = 0; cnt = n0; }
        else if (tile < t0 + t1) { t = 1; idx = tile - t0; base = n0; cnt = n1; }
        else { t = 2; idx = tile - t0 - t1; base = n0 + n1; cnt = n2; }
        const int64_t within = idx * 32 + lj;
        const float4 *wp = reinterpret_cast<const float4 *>(wpk) + (int64_t)t * NT * NSQ * 64 + lane;
        pair_scores_tile<NT>(t, base + within, within < cnt, lh, wp, tab, sel_pair, sel_node, sel_pa, sel_pb, Z, ldz, q,
                             ldq, pe_stat, bfold, att, score);
    }
}

// Same tiles, Wfold_t RESIDENT IN LDS (D <= 128: the packed image of one type is 4*D*D <= 64 KiB).  Sixteen waves
// share one copy; a workgroup walks groups of PSL_WAVES consecutive same-type tiles (tiles are type-major, so the groups a
// workgroup sees come in non-decreasing type order and the image is reloaded at most three times).  The A operands
// then arrive with LDS latency instead of L2 latency and the kernel's L2 reads drop by ~64 KiB per tile.  Sixteen
// waves per workgroup, one workgroup per CU: the same four waves per SIMD as two 8-wave workgroups, but 70 instead of
// 140 KiB of LDS, which the kernels of the other streams can use (76 -> 78 M pairs/s pipelined).
constexpr int PSL_WAVES = 16;

template <int NT>
__global__ __launch_bounds__(64 * PSL_WAVES, 4) void pair_scores_lds_kernel(
    const int64_t *__restrict__ type_ptr, int64_t bs, const int32_t *__restrict__ sel_pair,
    const int32_t *__restrict__ sel_node, const float *__restrict__ sel_pa, const float *__restrict__ sel_pb,
    const float *__restrict__ Z, int64_t ldz, const float *__restrict__ q, int64_t ldq,
    const float *__restrict__ pe_tab, const float *__restrict__ pe_stat, const float *__restrict__ wpk,
    const float *__restrict__ bfold, const float *__restrict__ att, float *__restrict__ score) {
    constexpr int D = 32 * NT;
    constexpr int NSQ = D / 8;
    constexpr int IMG = NT * NSQ * 64;  // float4 per type
    extern __shared__ __attribute__((aligned(16))) float4 psl_lds[];
    float4 *wl = psl_lds, *tab = psl_lds + IMG;
    for (int i = threadIdx.x; i < 3 * D; i += blockDim.x) tab[i] = reinterpret_cast<const float4 *>(pe_tab)[i];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lj = lane & 31, lh = lane >> 5;
    const int64_t n0 = type_ptr[bs], n1 = type_ptr[(bs + 1) + bs], n2 = type_ptr[2 * (bs + 1) + bs];
    const int64_t t0 = (n0 + 31) >> 5, t1 = (n1 + 31) >> 5, t2 = (n2 + 31) >> 5;          // tiles per type
    const int64_t g0 = (t0 + PSL_WAVES - 1) / PSL_WAVES, g1 = (t1 + PSL_WAVES - 1) / PSL_WAVES,
                  g2 = (t2 + PSL_WAVES - 1) / PSL_WAVES;                                   // tile groups per type
    int loaded = -1;
    for (int64_t g = blockIdx.x; g < g0 + g1 + g2; g += gridDim.x) {
        int t;
        int64_t idx, base, cnt, tiles;
        if (g < g0) { t = 0; idx = g * PSL_WAVES + wave; base = 0; cnt = n0; tiles = t0; }
        else if (g < g0 + g1) { t = 1; idx = (g - g0) * PSL_WAVES + wave; base = n0; cnt = n1; tiles = t1; }
        else { t = 2; idx = (g - g0 - g1) * PSL_WAVES + wave; base = n0 + n1; cnt = n2; tiles = t2; }
        if (t != loaded) {  // (the first pass also publishes tab)
            __syncthreads();
            const float4 *src = reinterpret_cast<const float4 *>(wpk) + (int64_t)t * IMG;
            for (int i = threadIdx.x; i < IMG; i += blockDim.x) wl[i] = src[i];
            loaded = t;
            __syncthreads();
        }
        if (idx < tiles) {
            const int64_t within = idx * 32 + lj;
            pair_scores_tile<NT>(t, base + within, within < cnt, lh, wl + lane, tab, sel_pair, sel_node, sel_pa, sel_pb,
                                 Z, ldz, q, ldq, pe_stat, bfold, att, score);
        }
    }
}

// One chunk of up to G consecutive entries (same pair, same type) handled by one lane group: lane i fetches the
// metadata of entry i (coalesced) and does the per-entry scalar math once (alpha, LayerNorm 1/std); then the entries
// are broadcast one by one while every lane gathers its 16 bytes of the Z row, four rows in flight.
template <int G, int CH = G>  // CH <= G: entries per chunk (lanes past CH idle while the metadata is fetched)
__device__ __forceinline__ void sg_chunk(const PeStat &st, const float4 (&k)[4], bool act, int off, int gbase, int lig,
                                         int remaining, int64_t e0, float m, float den,
                                         const int32_t *__restrict__ sel_node, const float *__restrict__ sel_pa,
                                         const float *__restrict__ sel_pb, const float *__restrict__ score,
                                         const float *__restrict__ Z, int64_t ldz, float *__restrict__ alpha_out,
                                         float4 &accz, float4 &acch, float &asum) {
    const int n_here = remaining < CH ? remaining : CH;
    const bool valid = lig < n_here;
    const int64_t e = e0 + lig;
    const float my_alpha = valid ? expf(score[e] - m) / den : 0.f;
    const float my_pa = valid ? sel_pa[e] : 0.f, my_pb = valid ? sel_pb[e] : 0.f;
    const int32_t my_node = valid ? sel_node[e] : 0;
    const float my_rab = pe_rstd(st, my_pa, my_pb), my_rba = pe_rstd(st, my_pb, my_pa);
    if (alpha_out && valid) alpha_out[e] = my_alpha;
    for (int j = 0; j < n_here; j += 4) {  // lanes past n_here carry alpha = 0, node = 0: harmless gathers
        float al[4], pa[4], pb[4], rab[4], rba[4];
        int32_t nd[4];
        float4 z[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int src = gbase + ((j + u) & (G - 1));
            al[u] = __shfl(my_alpha, src, 64);
            nd[u] = __shfl(my_node, src, 64);
            pa[u] = __shfl(my_pa, src, 64);
            pb[u] = __shfl(my_pb, src, 64);
            rab[u] = __shfl(my_rab, src, 64);
            rba[u] = __shfl(my_rba, src, 64);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            z[u] = act ? *reinterpret_cast<const float4 *>(Z + (int64_t)nd[u] * ldz + off)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            accz.x = fmaf(al[u], z[u].x, accz.x); accz.y = fmaf(al[u], z[u].y, accz.y);
            accz.z = fmaf(al[u], z[u].z, accz.z); accz.w = fmaf(al[u], z[u].w, accz.w);
            acch.x = fmaf(al[u], pe_hidden(k[0], pa[u], pb[u], rab[u], rba[u]), acch.x);
            acch.y = fmaf(al[u], pe_hidden(k[1], pa[u], pb[u], rab[u], rba[u]), acch.y);
            acch.z = fmaf(al[u], pe_hidden(k[2], pa[u], pb[u], rab[u], rba[u]), acch.z);
            acch.w = fmaf(al[u], pe_hidden(k[3], pa[u], pb[u], rab[u], rba[u]), acch.w);
            asum += al[u];
        }
    }
}

template <int G>
__global__ __launch_bounds__(256) void pair_softmax_gather_kernel(
    int D, int64_t bs, const int64_t *__restrict__ type_ptr, const int32_t *__restrict__ sel_node,
    const float *__restrict__ sel_pa, const float *__restrict__ sel_pb, const float *__restrict__ score,
    const float *__restrict__ Z, int64_t ldz, const float *__restrict__ pe_tab, const float *__restrict__ pe_stat,
    float *__restrict__ Gout, int64_t ldg, float *__restrict__ alpha_out, int32_t *__restrict__ heavy) {
    constexpr int RPW = 64 / G;
    const int lane = threadIdx.x & 63;
    const int grp = lane / G, lig = lane % G, gbase = grp * G;
    const int off = 4 * lig;
    const bool act = off < D;
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t tot0 = type_ptr[bs], tot1 = type_ptr[(bs + 1) + bs];
    const int64_t tbase[3] = {0, tot0, tot0 + tot1};

    for (int64_t p0 = wave_id * RPW; p0 < bs; p0 += n_waves * RPW) {
        const int64_t p = p0 + grp;
        const bool live = p < bs;
        int64_t beg[3];
        int cnt[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int64_t lo = live ? type_ptr[t * (bs + 1) + p] : 0, hi = live ? type_ptr[t * (bs + 1) + p + 1] : 0;
            beg[t] = tbase[t] + lo;
            cnt[t] = (int)(hi - lo);
        }
        // pairs with many selected nodes would serialise one lane group for a long time: they are left to
        // pair_softmax_gather_heavy_kernel (a whole workgroup per pair)
        const bool is_heavy = cnt[0] + cnt[1] + cnt[2] > SG_HEAVY;
        if (is_heavy) {
            if (lig == 0) heavy[1 + atomicAdd(&heavy[0], 1)] = (int32_t)p;  // any order: pairs are independent
            cnt[0] = cnt[1] = cnt[2] = 0;
        }
        // segment softmax statistics over all of the pair's entries (PyG softmax: shift by max, denom + 1e-16)
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < 3; ++t)
            for (int i = lig; i < cnt[t]; i += G) m = fmaxf(m, score[beg[t] + i]);
        m = lpf_group_max<G>(m);
        float den = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t)
            for (int i = lig; i < cnt[t]; i += G) den += expf(score[beg[t] + i] - m);
        den = lpf_group_sum<G>(den) + 1e-16f;

        float4 accz = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 acch[3];
        float asum[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            acch[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            asum[t] = 0.f;
            if (cnt[t] == 0) continue;
            PeStat st;
            st.c00 = pe_stat[8 * t + 0]; st.c11 = pe_stat[8 * t + 1]; st.cbb = pe_stat[8 * t + 2];
            st.c01 = pe_stat[8 * t + 3]; st.c0b = pe_stat[8 * t + 4]; st.c1b = pe_stat[8 * t + 5];
            float4 k[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                k[u] = act ? reinterpret_cast<const float4 *>(pe_tab)[t * D + off + u] : make_float4(0.f, 0.f, 0.f, 0.f);
            // chunks of G entries: lane i fetches the metadata of entry i (coalesced) and does the per-entry scalar
            // math once; the inner loop broadcasts entry by entry while every lane gathers its 16 bytes of the Z row
            for (int base = 0; base < cnt[t]; base += G)
                sg_chunk<G>(st, k, act, off, gbase, lig, cnt[t] - base, beg[t] + base, m, den, sel_node, sel_pa, sel_pb,
                            score, Z, ldz, alpha_out, accz, acch[t], asum[t]);
        }
        if (live && !is_heavy) {
            float *g = Gout + p * ldg;
            if (act) {
                *reinterpret_cast<float4 *>(g + off) = accz;
                *reinterpret_cast<float4 *>(g + D + off) = acch[0];
                *reinterpret_cast<float4 *>(g + 2 * D + off) = acch[1];
                *reinterpret_cast<float4 *>(g + 3 * D + off) = acch[2];
            }
            if (lig == 0) *reinterpret_cast<float4 *>(g + 4 * D) = make_float4(asum[0], asum[1], asum[2], 1.0f);
        }
    }
}

// Pairs with more than SG_HEAVY selected nodes: one 256-thread workgroup per pair (every other block exits at once).
// The NG = 256/G lane groups take entries round-robin, partial sums meet in LDS and are added in group order, so the
// result does not depend on scheduling.
template <int G>
__global__ __launch_bounds__(256) void pair_softmax_gather_heavy_kernel(
    int D, int64_t bs, const int64_t *__restrict__ type_ptr, const int32_t *__restrict__ sel_node,
    const float *__restrict__ sel_pa, const float *__restrict__ sel_pb, const float *__restrict__ score,
    const float *__restrict__ Z, int64_t ldz, const float *__restrict__ pe_tab, const float *__restrict__ pe_stat,
    float *__restrict__ Gout, int64_t ldg, float *__restrict__ alpha_out, const int32_t *__restrict__ heavy) {
    constexpr int NG = 256 / G;
    __shared__ float red[256];
    __shared__ float part[NG][4 * 4 * G + 4];  // per group: accz | acch[0..2] (4G floats each) | asum[0..2]
    const int64_t tot0 = type_ptr[bs], tot1 = type_ptr[(bs + 1) + bs];
    const int64_t tbase[3] = {0, tot0, tot0 + tot1};
    const int n_heavy = heavy[0];
  for (int hidx = blockIdx.x; hidx < n_heavy; hidx += gridDim.x) {  // list written by the light kernel
    const int64_t p = heavy[1 + hidx];
    int64_t beg[3];
    int cnt[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int64_t lo = type_ptr[t * (bs + 1) + p], hi = type_ptr[t * (bs + 1) + p + 1];
        beg[t] = tbase[t] + lo;
        cnt[t] = (int)(hi - lo);
    }
    __syncthreads();  // LDS reuse across list entries
    const int tid = threadIdx.x, grp = tid / G, lig = tid % G, off = 4 * lig;
    const int gbase = ((tid & 63) / G) * G;  // first lane of this group inside its wavefront
    const bool act = off < D;

    auto block_reduce = [&](float v, bool take_max) {  // wave butterfly, then the four wave results through LDS
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) {
            const float o = __shfl_xor(v, sft, 64);
            v = take_max ? fmaxf(v, o) : v + o;
        }
        __syncthreads();  // red[] free (previous use read by everyone)
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        const float r0 = red[0], r1 = red[1], r2 = red[2], r3 = red[3];
        return take_max ? fmaxf(fmaxf(r0, r1), fmaxf(r2, r3)) : (r0 + r1) + (r2 + r3);
    };
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < 3; ++t)
        for (int i = tid; i < cnt[t]; i += 256) m = fmaxf(m, score[beg[t] + i]);
    m = block_reduce(m, true);
    float den = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
        for (int i = tid; i < cnt[t]; i += 256) den += expf(score[beg[t] + i] - m);
    den = block_reduce(den, false) + 1e-16f;

    float4 accz = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acch[3];
    float asum[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        acch[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        asum[t] = 0.f;
        if (cnt[t] == 0) continue;
        PeStat st;
        st.c00 = pe_stat[8 * t + 0]; st.c11 = pe_stat[8 * t + 1]; st.cbb = pe_stat[8 * t + 2];
        st.c01 = pe_stat[8 * t + 3]; st.c0b = pe_stat[8 * t + 4]; st.c1b = pe_stat[8 * t + 5];
        float4 k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            k[u] = act ? reinterpret_cast<const float4 *>(pe_tab)[t * D + off + u] : make_float4(0.f, 0.f, 0.f, 0.f);
        // lane groups take chunks of SG_HEAVY_CHUNK consecutive entries round-robin (short chunks: a pair just above
        // the threshold still spreads over most of the workgroup's lane groups)
        for (int base = grp * SG_HEAVY_CHUNK; base < cnt[t]; base += NG * SG_HEAVY_CHUNK)
            sg_chunk<G, (SG_HEAVY_CHUNK < G ? SG_HEAVY_CHUNK : G)>(st, k, act, off, gbase, lig, cnt[t] - base,
                                                                 beg[t] + base, m, den, sel_node, sel_pa, sel_pb, score,
                                                                 Z, ldz, alpha_out, accz, acch[t], asum[t]);
    }
    float *mine = part[grp];
    *reinterpret_cast<float4 *>(mine + off) = accz;
#pragma unroll
    for (int t = 0; t < 3; ++t) *reinterpret_cast<float4 *>(mine + (1 + t) * 4 * G + off) = acch[t];
    if (lig == 0) {
        mine[16 * G + 0] = asum[0];
        mine[16 * G + 1] = asum[1];
        mine[16 * G + 2] = asum[2];
    }
    __syncthreads();
    if (grp == 0) {
        float *g = Gout + p * ldg;
        float4 s4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) s4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int gq = 0; gq < NG; ++gq) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(part[gq] + q * 4 * G + off);
                s4[q].x += v.x; s4[q].y += v.y; s4[q].z += v.z; s4[q].w += v.w;
            }
        }
        if (act) {
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(g + q * D + off) = s4[q];
        }
        if (lig == 0) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            for (int gq = 0; gq < NG; ++gq) {
                a0 += part[gq][16 * G + 0];
                a1 += part[gq][16 * G + 1];
                a2 += part[gq][16 * G + 2];
            }
            *reinterpret_cast<float4 *>(g + 4 * D) = make_float4(a0, a1, a2, 1.0f);
        }
    }
  }
}

}  // namespace

extern "C" int lpf_pair_scores_f32(int32_t D, const int64_t *type_ptr, int64_t bs, const int32_t *sel_pair,
                                   const int32_t *sel_node, const float *sel_pa, const float *sel_pb, const float *Z,
                                   int64_t ldz, const float *q, int64_t ldq, const float *pe_tab,
                                   const float *pe_stat, const float *wfold_packed, const float *bfold,
                                   const float *att, float *score, int64_t max_entries, void *stream) {
    if (bs == 0 || max_entries == 0) return LPF_OK;
    LPF_REQUIRE(bs > 0 && type_ptr && sel_pair && sel_node && sel_pa && sel_pb && Z && q && pe_tab && pe_stat &&
                wfold_packed && bfold && att && score && max_entries > 0);
    LPF_REQUIRE((ldz & 3) == 0 && (ldq & 3) == 0 && ldz >= D && ldq >= D && lpf_aligned16(Z) && lpf_aligned16(q) &&
                lpf_aligned16(pe_tab) && lpf_aligned16(wfold_packed) && lpf_aligned16(bfold) && lpf_aligned16(att));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the entry count lives on the device; size the grid from the host-side capacity and let waves stride over tiles
    int64_t tiles = (max_entries + 31) / 32 + 3;
    int64_t blocks = (tiles + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
#define LPF_SCORES_LDS(NT)                                                                                         \
    do {                                                                                                           \
        auto kern = pair_scores_lds_kernel<NT>;                                                                    \
        const size_t lds = (size_t)(NT * (32 * NT / 8) * 64 + 3 * 32 * NT) * sizeof(float4);                       \
        LPF_SET_MAX_LDS(kern, lds);                                                            \
        int64_t groups = (tiles + PSL_WAVES - 1) / PSL_WAVES + 3;                                                  \
        if (groups > 256) groups = 256; /* one 16-wave workgroup per CU: half the LDS of two 8-wave ones */       \
        hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(64 * PSL_WAVES), lds, s, type_ptr, bs, sel_pair,     \
                           sel_node, sel_pa, sel_pb, Z, ldz, q, ldq, pe_tab, pe_stat, wfold_packed, bfold, att,    \
                           score);                                                                                 \
    } while (0)
#define LPF_SCORES_LAUNCH(NT)                                                                                      \
    hipLaunchKernelGGL(pair_scores_kernel<NT>, dim3((unsigned)blocks), dim3(256), 0, s, type_ptr, bs, sel_pair,    \
                       sel_node, sel_pa, sel_pb, Z, ldz, q, ldq, pe_tab, pe_stat, wfold_packed, bfold, att, score)
    switch (D) {
        case 32: LPF_SCORES_LDS(1); break;
        case 64: LPF_SCORES_LDS(2); break;
        case 128: LPF_SCORES_LDS(4); break;
        case 256: LPF_SCORES_LAUNCH(8); break;
        default: return LPF_ERR_UNSUPPORTED;
    }
#undef LPF_SCORES_LAUNCH
#undef LPF_SCORES_LDS
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_pair_softmax_gather_f32(int32_t D, int64_t bs, const int64_t *type_ptr, const int32_t *sel_node,
                                           const float *sel_pa, const float *sel_pb, const float *score,
                                           const float *Z, int64_t ldz, const float *pe_tab, const float *pe_stat,
                                           float *G, int64_t ldg, float *alpha_out, int32_t *heavy_scratch,
                                           void *stream) {
    if (bs == 0) return LPF_OK;
    LPF_REQUIRE(bs > 0 && bs < (1ll << 31) && type_ptr && sel_node && sel_pa && sel_pb && score && Z && pe_tab &&
                pe_stat && G && heavy_scratch);
    if (D <= 0 || (D & 3) || D > 256) return LPF_ERR_UNSUPPORTED;
    LPF_REQUIRE((ldz & 3) == 0 && ldz >= D && (ldg & 3) == 0 && ldg >= 4 * D + 4 && lpf_aligned16(Z) &&
                lpf_aligned16(G) && lpf_aligned16(pe_tab));
    hipStream_t s = static_cast<hipStream_t>(stream);
    (void)hipMemsetAsync(heavy_scratch, 0, sizeof(int32_t), s);  // heavy-pair counter
    const int GG = D <= 64 ? 16 : (D <= 128 ? 32 : 64);
    int64_t blocks = (bs + (64 / GG) - 1) / (64 / GG);
    if (blocks > (1 << 22)) blocks = 1 << 22;
#define LPF_SG_LAUNCH(GV)                                                                                       \
    hipLaunchKernelGGL(pair_softmax_gather_kernel<GV>, dim3((unsigned)blocks), dim3(64), 0, s, D, bs, type_ptr, \
                       sel_node, sel_pa, sel_pb, score, Z, ldz, pe_tab, pe_stat, G, ldg, alpha_out, heavy_scratch)
    if (GG == 16) LPF_SG_LAUNCH(16);
    else if (GG == 32) LPF_SG_LAUNCH(32);
    else LPF_SG_LAUNCH(64);
#undef LPF_SG_LAUNCH
#define LPF_SGH_LAUNCH(GV)                                                                                          \
    hipLaunchKernelGGL(pair_softmax_gather_heavy_kernel<GV>, dim3(2048), dim3(256), 0, s, D, bs, type_ptr,         \
                       sel_node, sel_pa, sel_pb, score, Z, ldz, pe_tab, pe_stat, G, ldg, alpha_out, heavy_scratch)
    if (GG == 16) LPF_SGH_LAUNCH(16);
    else if (GG == 32) LPF_SGH_LAUNCH(32);
    else LPF_SGH_LAUNCH(64);
#undef LPF_SGH_LAUNCH
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}
