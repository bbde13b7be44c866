// The dense tail of the scoring path in ONE launch (LinkTransformer.score_pairs):
//
//   A  o   = LayerNorm_post( G[:, D:] Wcat^T + G[:, :D] )                attention output (layers.py:78)
//   B  r_p = ReLU(LayerNorm( W_p0 [o | counts] + b_p0 ))                 first layer of pairwise_lin
//   C  s   = w_s1 . ReLU( A_e r_e + A_p r_p + c ) + b_s1 ; sigmoid       score head with the boundary Linears folded
//
// Same machinery as dense_chain.hip (samples on the MFMA columns, a pair of wavefronts per 16 samples splitting the
// feature tiles, weights host-packed in A-operand order and staged global -> registers -> LDS one k-group ahead,
// double-buffered, one barrier per stage), with three stages chained through LDS: a stage's accumulators, written in
// accumulator layout, are the next stage's B operands.  Stage B appends one k-group read from global memory (the
// count features), stage C starts with k-groups read from global memory (r_e, produced on the side stream by the
// elementwise branch).  Against three separate launches this removes two launch ramps and the round trips of the
// attention output and of r_p through HBM.
#include <type_traits>

#include "lpf_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef LPF_TC_GROUPS
#define LPF_TC_GROUPS 4
#endif
constexpr int TC_GROUPS = LPF_TC_GROUPS;   // 16-sample groups per workgroup
constexpr int TC_WAVES = 2 * TC_GROUPS;    // a wave pair per group
constexpr int TC_THREADS = 64 * TC_WAVES;

struct TailArgs {
    int64_t M;
    const float *x; int64_t ldx; int KA;               // stage A input rows [M, KA]
    const float *addend; int64_t ldadd;                // [M, NA]
    const float *wA, *lnA_g, *lnA_b; int NA;
    const float *tail; int64_t ldtail;                 // [M, 4] appended to stage B's input (count features)
    const float *wB, *bB, *lnB_g, *lnB_b; int NB;
    const float *re; int64_t ldre;                     // [M, 16 * NGE] leading input of stage C
    const float *wC, *bC; int NC;
    const float *wdot, *bdot;
    float *logit, *prob;
    // merge mode (part != nullptr): stage A is not a GEMM but the merge of the per-type attention records written by
    // lpf_pair_attention_fused_f32 -- out = sum_t e^{m_t - M} acc_t / (sum_t e^{m_t - M} l_t + 1e-16) + bias -- and the
    // count features come from the int32 segment pointers of the selection
    const float *part;              // [3][M][NA + 4]: acc[NA], m, l, -, -
    const float *bnd;               // [3][units_cap][2][NA + 4]: boundary records of segments that cross 16-entry units
    int64_t units_cap;
    const int32_t *type_ptr;        // [3][M + 1]
    const float *att_bias;          // [NA]
    const int64_t *sel_ctl;         // selection control block: word 3 != 0 => the batch did not fit, scores = NaN
    int n_counts;
    // rows mode (rows != nullptr): stage A is already done -- lpf_pair_attention_rows_* left one finished row per pair,
    // [post_att_norm(attention output) (NA) | count features (4, zero padded)] -- and is a plain read
    const float *rows; int64_t ldrows;
    // rows mode, optional: the pairs are taken in the order perm[] (lpf_pair_attention_rows_perm_*: those with selected
    // nodes first, *n_full of them, the others behind).  A workgroup whose 64 pairs all lie behind *n_full skips stage
    // A, stage B and the r_p half of stage C: the pairwise branch of a pair without selected nodes is a constant, folded
    // into bC_empty on the host (fold.empty_pair_head_bias).
    const int32_t *perm;
    const int64_t *n_full;
    const float *bC_empty;
    // ... and a pair without selected nodes in a MIXED workgroup takes its (constant) row from here and zero counts: the
    // attention kernel that leaves the order does not write rows for such pairs
    const float *row_empty;         // [NA]
};

constexpr int tc_per_thread(int ntp) { return (ntp * 64 + TC_THREADS - 1) / TC_THREADS; }
// elements between two stages of a packed weight image (lpformer_amd/fold.py pack_dense: padded to 512 per stage)
constexpr int tc_stage_stride(int ntp) { return (ntp * 64 + 511) / 512 * 512; }

// weight elements: one per (tile, lane) and k-group -- four fp32 (f32x4) or, in the bf16 variant, four bf16 (uint2) of
// W[16 c + i][16 ks + 4 q .. + 3]
typedef short bf16x4_bits __attribute__((ext_vector_type(4)));
template <int P, int NTP, typename T>
__device__ __forceinline__ void tc_load(T (&r)[P], const float *packed, int stage, int tid) {
    const T *src = reinterpret_cast<const T *>(packed) + (int64_t)stage * tc_stage_stride(NTP);
#pragma unroll
    for (int e = 0; e < P; ++e) {
        const int i = e * TC_THREADS + tid;
#ifdef TC_NOWLOAD      /* (tuning builds: how much of the kernel is waiting for the weight stream?) */
        r[e] = T{};
        (void)src; (void)i;
#else
        r[e] = src[i < tc_stage_stride(NTP) ? i : 0];   // (the last strip of a thread may lie past the stage's padding)
#endif
    }
}
template <int P, typename T>
__device__ __forceinline__ void tc_store(const T (&r)[P], T *slab, int tid) {
#pragma unroll
    for (int e = 0; e < P; ++e) slab[e * TC_THREADS + tid] = r[e];
}

// acc[c] += W[16 (c0 + c) + i][k-group] * bv for this wave's TPW tiles; lw = slab + c0 * 64 + lane.  last = false (the
// same for the whole wavefront): the wave's last tile is padding -- all-zero weight rows -- and is left out.
template <int TPW>
__device__ __forceinline__ void tc_mfma(f32x4 (&acc)[TPW], const f32x4 *lw, const f32x4 bv, bool last = true) {
    f32x4 a[TPW];
#pragma unroll
    for (int c = 0; c < TPW; ++c) a[c] = lw[c * 64];
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // consecutive MFMAs go to different accumulators
#pragma unroll
        for (int c = 0; c < TPW - 1; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][u], bv[u], acc[c], 0, 0, 0);
        if (last) acc[TPW - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[TPW - 1][u], bv[u], acc[TPW - 1], 0, 0, 0);
    }
}
// bf16 variant: the four fp32 MFMAs of a k-group become ONE v_mfma_f32_16x16x16_bf16 -- a lane's four A values and
// four B values (k = 4 q .. 4 q + 3 of the group) are exactly that instruction's operands; the activations are rounded
// to bf16 here, the accumulation stays fp32.
template <int TPW>
__device__ __forceinline__ void tc_mfma(f32x4 (&acc)[TPW], const uint2 *lw, const f32x4 bv, bool last = true) {
    bf16x4_bits b;
#pragma unroll
    for (int u = 0; u < 4; ++u) b[u] = (short)lpf_f32_to_bf16(bv[u]);
#pragma unroll
    for (int c = 0; c < TPW; ++c) {
        if (c == TPW - 1 && !last) break;
        const uint2 w = lw[c * 64];
        const bf16x4_bits a = {(short)(w.x & 0xffffu), (short)(w.x >> 16), (short)(w.y & 0xffffu), (short)(w.y >> 16)};
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc[c], 0, 0, 0);
    }
}

__device__ __forceinline__ float tc_quad_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// LayerNorm over the n real features held by the wave pair (this wave: TPW tiles from feature fbase), in place
template <int TPW>
__device__ __forceinline__ void tc_layernorm(f32x4 (&acc)[TPW], int fbase, int n, const float *g, const float *b, int half,
                                             int q, float *my_x, const float *peer_x, bool relu) {
    float s1 = 0.f;
#pragma unroll
    for (int c = 0; c < TPW; ++c) s1 += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    s1 = tc_quad_sum(s1);  // padded features are exactly 0 and add nothing
    __syncthreads();       // exchange slots free
    if (q == 0) *my_x = s1;
    __syncthreads();
    const float mean = (half == 0 ? s1 + *peer_x : *peer_x + s1) / (float)n;  // same order in both waves
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < TPW; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = (fbase + 16 * c + r < n) ? acc[c][r] - mean : 0.f;
            s2 += d * d;
        }
    s2 = tc_quad_sum(s2);
    __syncthreads();
    if (q == 0) *my_x = s2;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((half == 0 ? s2 + *peer_x : *peer_x + s2) / (float)n + 1e-5f);
#pragma unroll
    for (int c = 0; c < TPW; ++c) {
        const f32x4 gg = *reinterpret_cast<const f32x4 *>(g + fbase + 16 * c);  // zero-padded: pads come out 0
        const f32x4 bb = *reinterpret_cast<const f32x4 *>(b + fbase + 16 * c);
        acc[c] = (acc[c] - mean) * rstd * gg + bb;
        if (relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[c][r] = fmaxf(acc[c][r], 0.f);
        }
    }
}

// (tuning builds, -DTC_STAMPS: wall-clock marks -- 100 MHz -- of lane 0 of every wavefront; tools/tail_stamps.py)
#ifdef TC_STAMPS
__device__ uint64_t *tc_stamp_buf = nullptr;
#define TC_STAMP(k) do { if (lane == 0) st_t[k] = wall_clock64(); } while (0)
#define TC_STAMPS_OUT(kind) do { if (lane == 0 && tc_stamp_buf) { uint64_t *o__ = tc_stamp_buf + ((int64_t)blockIdx.x * TC_WAVES + wave) * 8; \
        for (int k__ = 0; k__ < 7; ++k__) o__[k__] = st_t[k__]; o__[7] = (kind); } } while (0)
#else
#define TC_STAMP(k) do { } while (0)
#define TC_STAMPS_OUT(kind) do { } while (0)
#endif

// NTA / NTB / NTC: 16-feature tiles of the three stages' outputs (NTA = NGE = D/16)
template <int NTA, int NTB, int NTC>
struct TcShape {
    static constexpr int NTPA = (NTA + 1) & ~1, NTPB = (NTB + 1) & ~1, NTPC = (NTC + 1) & ~1;
    static constexpr int PA = tc_per_thread(NTPA), PB = tc_per_thread(NTPB), PC = tc_per_thread(NTPC);
    static constexpr int PM = PA > PB ? (PA > PC ? PA : PC) : (PB > PC ? PB : PC);
    static constexpr int SLAB = PM * TC_THREADS;
    static constexpr int HT = NTPA > NTPB ? NTPA : NTPB;  // hidden tiles kept per sample group
    static constexpr int HID = TC_GROUPS * HT * 64;
    static constexpr size_t BYTES = (size_t)(2 * SLAB + HID) * sizeof(f32x4) + TC_WAVES * 16 * sizeof(float);
};

// (D = 256: 32 output tiles of the score head -- 16 accumulators per lane in stage C alone --, one workgroup per CU with
//  twice the registers)
// WM: how the two GEMMs run -- 0 fp32 weights and MFMAs, 1 bf16 weights and activations (throughput mode)
template <int NTA, int NTB, int NTC, int WM = 0, bool ROWS = false>
__global__ __launch_bounds__(TC_THREADS, NTC >= 32 ? 2 : (TC_THREADS >= 512 ? 4 : 3)) void tail_chain_kernel(const TailArgs A) {
    using S = TcShape<NTA, NTB, NTC>;
    constexpr bool WB = WM != 0;
    // weight element: four fp32 or four bf16
    using WT = typename std::conditional<WM == 1, uint2, f32x4>::type;
    constexpr int NTPA = S::NTPA, NTPB = S::NTPB, NTPC = S::NTPC;
    constexpr int TPWA = NTPA / 2, TPWB = NTPB / 2, TPWC = NTPC / 2;
    constexpr int NGE = NTA;  // k-groups of r_e
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
    f32x4 *hid = lds + 2 * S::SLAB;
    float *xch = reinterpret_cast<float *>(hid + S::HID);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 1, half = wave & 1, q = lane >> 4, j = lane & 15;
    float *my_x = xch + wave * 16 + j;
    const float *peer_x = xch + (wave ^ 1) * 16 + j;
    f32x4 *my_hid = hid + (grp * S::HT) * 64 + lane;
    int buf = 0;
#ifdef TC_STAMPS
    uint64_t st_t[8] = {0};
    TC_STAMP(0);
#endif

    // (With an order the short workgroups are the LAST ones of the grid.  Dealing the tiles from both ends instead, a short
    // workgroup next to every full one on a CU, measured worse: collab-like 0.205 against 0.186 ms per pipelined step, the
    // launch alone unchanged at 55 us -- what the short workgroups buy is a grid that drains early for the next kernel.)
    const int64_t tile = blockIdx.x;
    const int64_t pos = tile * (16 * TC_GROUPS) + grp * 16 + j;
    const bool live = pos < A.M;
    bool lite = false;          // (rows mode with an order: every pair of this workgroup is one without selected nodes)
    int64_t m = pos;
    if constexpr (ROWS) {
        if (A.perm) {
            m = A.perm[live ? pos : A.M - 1];
            lite = tile * (16 * TC_GROUPS) >= *A.n_full;
        }
    }
    const int64_t mm = live ? m : (ROWS && A.perm ? m : A.M - 1);  // dead lanes compute on a valid row and store nothing
    bool no_sel = false;        // (rows mode with an order: this pair has no selected nodes)
    if constexpr (ROWS) no_sel = A.perm && A.row_empty && (live ? pos : A.M - 1) >= *A.n_full;

    if constexpr (ROWS) {
        if (lite) {
            // ---- 64 pairs without selected nodes: score = w_dot . ReLU(A_e r_e + bC_empty) + b_dot -- stage C over the
            //      r_e k-groups alone (35 % of the matrix work of a full workgroup), nothing else
            f32x4 acc[TPWC];
#pragma unroll
            for (int c = 0; c < TPWC; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            WT wr[S::PC];
            const float *rer = A.re + mm * A.ldre + 4 * q;
            tc_load<S::PC, NTPC>(wr, A.wC, 0, tid);
            f32x4 xr = *reinterpret_cast<const f32x4 *>(rer);
#pragma unroll 1
            for (int kg = 0; kg < NGE; ++kg) {
                const f32x4 bv = xr;
                WT *lw = reinterpret_cast<WT *>(lds) + buf * S::SLAB;
                tc_store<S::PC>(wr, lw, tid);
                __syncthreads();
                if (kg + 1 < NGE) {
                    tc_load<S::PC, NTPC>(wr, A.wC, kg + 1, tid);
                    xr = *reinterpret_cast<const f32x4 *>(rer + 16 * (kg + 1));
                }
                tc_mfma<TPWC>(acc, lw + (half * TPWC) * 64 + lane, bv);
                buf ^= 1;
            }
            const int fbase = 16 * half * TPWC + 4 * q;
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < TPWC; ++c) {
                const f32x4 b = *reinterpret_cast<const f32x4 *>(A.bC_empty + fbase + 16 * c);
                const f32x4 w = *reinterpret_cast<const f32x4 *>(A.wdot + fbase + 16 * c);
#pragma unroll
                for (int r = 0; r < 4; ++r) d = fmaf(fmaxf(acc[c][r] + b[r], 0.f), w[r], d);
            }
            d = tc_quad_sum(d);
            if (q == 0) *my_x = d;
            __syncthreads();
            if (live && q == 0 && half == 0) {
                d = d + *peer_x + A.bdot[0];
                if (A.sel_ctl && A.sel_ctl[3] != 0) d = __builtin_nanf("");
                if (A.logit) A.logit[m] = d;
                if (A.prob) A.prob[m] = 1.0f / (1.0f + expf(-d));
            }
            TC_STAMP(6);
            TC_STAMPS_OUT(1);
            return;
        }
    }
    // ------------------------------------------------------------------ stage A: attention output + post-norm
    f32x4 accA[TPWA];
#pragma unroll
    for (int c = 0; c < TPWA; ++c) accA[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int seg_cnt[3] = {0, 0, 0};
    if (!WB && !ROWS && A.part == nullptr) {
        const float *xa = A.x + mm * A.ldx;
        const int ngA = (A.KA + 15) >> 4;
        f32x4 wr[S::PA];
        tc_load<S::PA, NTPA>(wr, A.wA, 0, tid);  // (the GEMM form of stage A exists in fp32 only)
        int kk = 4 * q < A.KA ? 4 * q : A.KA - 4;  // clamped into the row; out-of-range groups are zeroed below
        f32x4 xr = *reinterpret_cast<const f32x4 *>(xa + kk);
#pragma unroll 1
        for (int kg = 0; kg < ngA; ++kg) {
            const f32x4 bv = (16 * kg + 4 * q < A.KA) ? xr : (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 *lw = lds + buf * S::SLAB;
            tc_store<S::PA>(wr, lw, tid);
            __syncthreads();
            if (kg + 1 < ngA) {  // next k-group's operands fly while this one's MFMAs run
                tc_load<S::PA, NTPA>(wr, A.wA, kg + 1, tid);
                kk = 16 * (kg + 1) + 4 * q;
                kk = kk < A.KA ? kk : A.KA - 4;
                xr = *reinterpret_cast<const f32x4 *>(xa + kk);
            }
            tc_mfma<TPWA>(accA, lw + (half * TPWA) * 64 + lane, bv);
            buf ^= 1;
        }
    }
    WT wrB[S::PB];
    tc_load<S::PB, NTPB>(wrB, A.wB, 0, tid);  // stage B's first weights fly during the epilogue
    if constexpr (ROWS) {
        const int fbase = 16 * half * TPWA + 4 * q;
        const float *row = no_sel ? A.row_empty : A.rows + mm * A.ldrows;
#pragma unroll
        for (int c = 0; c < TPWA; ++c) {
            const int f0 = fbase + 16 * c;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(row + (f0 < A.NA ? f0 : 0));
            accA[c] = f0 < A.NA ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    } else if (A.part == nullptr) {
        const int fbase = 16 * half * TPWA + 4 * q;
        f32x4 ad[TPWA];
#pragma unroll
        for (int c = 0; c < TPWA; ++c) {
            const int f0 = fbase + 16 * c;
            ad[c] = *reinterpret_cast<const f32x4 *>(A.addend + mm * A.ldadd + (f0 < A.NA ? f0 : 0));
        }
#pragma unroll
        for (int c = 0; c < TPWA; ++c) accA[c] += (fbase + 16 * c < A.NA) ? ad[c] : (f32x4){0.f, 0.f, 0.f, 0.f};
    } else {
        // Merge of this pair's attention records (online-softmax states {acc, m, l}).  A (pair, type) segment
        // [lo, hi) of the type's entry region that lies inside one 16-entry unit left ONE record in part[t][pair];
        // one that crosses units left a boundary record per unit it touches -- the head in slot 1 of its first unit,
        // then slot 0 of every following unit (pair_fused.hip) -- whose addresses follow from lo and hi alone.
        // All addresses are known before the first record arrives, so the reads go out in batches (the three segment
        // pointers, the three first records, then the remaining boundary records two or four at a time): the record region is
        // hundreds of MB touched sparsely, every read is a long-latency miss, and there are only two wavefronts per
        // SIMD here to hide it.
        const int fbase = 16 * half * TPWA + 4 * q;
        const int64_t rs = A.NA + 4;
        float mx = -INFINITY, den = 0.f;
        f32x4 v[TPWA];
#pragma unroll
        for (int c = 0; c < TPWA; ++c) v[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto fetch = [&](const float *rec, f32x4 &h, f32x4 *b) __attribute__((always_inline)) {
            h = *reinterpret_cast<const f32x4 *>(rec + A.NA);
#pragma unroll
            for (int c = 0; c < TPWA; ++c) {
                const int f0 = fbase + 16 * c;
                b[c] = *reinterpret_cast<const f32x4 *>(rec + (f0 < A.NA ? f0 : 0));
            }
        };
        auto merge = [&](const f32x4 &h, const f32x4 *b) __attribute__((always_inline)) {
            const float mn = fmaxf(mx, h[0]);
            const float sa = __expf(mx - mn), sb = __expf(h[0] - mn);
            den = fmaf(den, sa, h[1] * sb);
#pragma unroll
            for (int c = 0; c < TPWA; ++c) v[c] = v[c] * sa + b[c] * sb;
            mx = mn;
        };
        int lo[3], hi[3], n_p[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int32_t *tp = A.type_ptr + (int64_t)t * (A.M + 1) + mm;
            lo[t] = tp[0];
            hi[t] = tp[1];
        }
        {
            f32x4 h0[3], b0[3][TPWA];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                seg_cnt[t] = hi[t] - lo[t];
                // (a segment past the region: sel_ctl[3] is set and the row becomes NaN below)
                const bool some = hi[t] > lo[t] && ((hi[t] - 1) >> 4) < A.units_cap;
                n_p[t] = some ? ((hi[t] - 1) >> 4) - (lo[t] >> 4) + 1 : 0;
                const float *rec = n_p[t] > 1 ? A.bnd + ((((int64_t)t * A.units_cap + (lo[t] >> 4)) * 2) + 1) * rs
                                              : A.part + ((int64_t)t * A.M + mm) * rs;
                h0[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < TPWA; ++c) b0[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (n_p[t] > 0) fetch(rec, h0[t], b0[t]);  // (most pairs have no entry of some type: nothing to read)
            }
#pragma unroll
            for (int t = 0; t < 3; ++t)
                if (n_p[t] > 0) merge(h0[t], b0[t]);
        }
        const int r0 = n_p[0] > 1 ? n_p[0] - 1 : 0, r1 = n_p[1] > 1 ? n_p[1] - 1 : 0, r2 = n_p[2] > 1 ? n_p[2] - 1 : 0;
        const int n_rest = r0 + r1 + r2;
        constexpr int NB = TPWA >= 4 ? 2 : 4;  // (registers: the kernel is held to 128)
#pragma unroll 1
        for (int k0 = 0; k0 < n_rest; k0 += NB) {
            f32x4 hh[NB], bb[NB][TPWA];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int k = k0 + i < n_rest ? k0 + i : n_rest - 1;
                const int t = k < r0 ? 0 : (k < r0 + r1 ? 1 : 2);
                const int jj = k - (t == 0 ? 0 : (t == 1 ? r0 : r0 + r1));
                const int ut = t == 0 ? lo[0] >> 4 : (t == 1 ? lo[1] >> 4 : lo[2] >> 4);
                fetch(A.bnd + (((int64_t)t * A.units_cap + ut + 1 + jj) * 2) * rs, hh[i], bb[i]);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i)
                if (k0 + i < n_rest) merge(hh[i], bb[i]);
        }
        const float inv = 1.0f / (den + 1e-16f);
#pragma unroll
        for (int c = 0; c < TPWA; ++c) {
            const int f0 = fbase + 16 * c;
            if (f0 < A.NA) accA[c] = v[c] * inv + *reinterpret_cast<const f32x4 *>(A.att_bias + f0);
        }
    }
    {
        const int fbase = 16 * half * TPWA + 4 * q;
        if constexpr (!ROWS) tc_layernorm<TPWA>(accA, fbase, A.NA, A.lnA_g, A.lnA_b, half, q, my_x, peer_x, false);
#pragma unroll
        for (int c = 0; c < TPWA; ++c) my_hid[(half * TPWA + c) * 64] = accA[c];
    }

    TC_STAMP(1);
    // ------------------------------------------------------------------ stage B: first layer of pairwise_lin
    f32x4 accB[TPWB];
#pragma unroll
    for (int c = 0; c < TPWB; ++c) accB[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        // the appended k-group: the count features (4 floats per sample) in lane quarter 0, zeros elsewhere
        f32x4 tailv = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (q == 0) {
            if constexpr (ROWS) {
                if (!no_sel) tailv = *reinterpret_cast<const f32x4 *>(A.rows + mm * A.ldrows + A.NA);
            } else if (A.part == nullptr) {
                tailv = *reinterpret_cast<const f32x4 *>(A.tail + mm * A.ldtail);
            } else {  // get_structure_cnts (link_transformer.py:340-356): n_cn, n_1hop, [n_non1hop,] n_cn + n_1hop
                const float n0 = (float)seg_cnt[0], n1 = (float)seg_cnt[1], n2 = (float)seg_cnt[2];
                tailv = A.n_counts == 4 ? (f32x4){n0, n1, n2, n0 + n1}
                                         : (A.n_counts == 3 ? (f32x4){n0, n1, n0 + n1, 0.f} : (f32x4){n0, 0.f, 0.f, 0.f});
            }
        }
#pragma unroll
        for (int kg = 0; kg < NTPA + 1; ++kg) {
            WT *lw = reinterpret_cast<WT *>(lds) + buf * S::SLAB;
            tc_store<S::PB>(wrB, lw, tid);
            __syncthreads();  // (kg == 0: also publishes stage A's hidden tiles)
            if (kg + 1 < NTPA + 1) tc_load<S::PB, NTPB>(wrB, A.wB, kg + 1, tid);
            const f32x4 bv = kg < NTPA ? my_hid[(kg < NTPA ? kg : 0) * 64] : tailv;
            // (NTB odd: tile NTPB - 1, the second wave's last one, is padding -- its accumulators stay 0)
            tc_mfma<TPWB>(accB, lw + (half * TPWB) * 64 + lane, bv, !((NTB & 1) && half == 1));
            buf ^= 1;
        }
    }
    TC_STAMP(2);
    WT wrC[S::PC];
    tc_load<S::PC, NTPC>(wrC, A.wC, 0, tid);
    const float *rer = A.re + mm * A.ldre + 4 * q;
    f32x4 xr = *reinterpret_cast<const f32x4 *>(rer);  // stage C's first input group
    {
        const int fbase = 16 * half * TPWB + 4 * q;
#pragma unroll
        for (int c = 0; c < TPWB; ++c) accB[c] += *reinterpret_cast<const f32x4 *>(A.bB + fbase + 16 * c);
        tc_layernorm<TPWB>(accB, fbase, A.NB, A.lnB_g, A.lnB_b, half, q, my_x, peer_x, true);
        // (the barriers inside the LayerNorm exchange come after every wave's last read of stage A's tiles)
#pragma unroll
        for (int c = 0; c < TPWB; ++c) my_hid[(half * TPWB + c) * 64] = accB[c];
    }

    TC_STAMP(3);
    // ------------------------------------------------------------------ stage C: folded score head
    f32x4 accC[TPWC];
#pragma unroll
    for (int c = 0; c < TPWC; ++c) accC[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int kg = 0; kg < NGE; ++kg) {  // r_e, read from global memory
        const f32x4 bv = xr;
        WT *lw = reinterpret_cast<WT *>(lds) + buf * S::SLAB;
        tc_store<S::PC>(wrC, lw, tid);
        __syncthreads();
        tc_load<S::PC, NTPC>(wrC, A.wC, kg + 1, tid);  // (stage NGE exists: the r_p groups follow)
        if (kg + 1 < NGE) xr = *reinterpret_cast<const f32x4 *>(rer + 16 * (kg + 1));
        tc_mfma<TPWC>(accC, lw + (half * TPWC) * 64 + lane, bv);
        buf ^= 1;
    }
    TC_STAMP(4);
#pragma unroll
    for (int kg = 0; kg < NTB; ++kg) {  // r_p, straight from LDS (its padding tile, all zeros, is no k-group worth running)
        WT *lw = reinterpret_cast<WT *>(lds) + buf * S::SLAB;
        tc_store<S::PC>(wrC, lw, tid);
        __syncthreads();  // (kg == 0: also publishes stage B's hidden tiles)
        if (kg + 1 < NTB) tc_load<S::PC, NTPC>(wrC, A.wC, NGE + kg + 1, tid);
        tc_mfma<TPWC>(accC, lw + (half * TPWC) * 64 + lane, my_hid[kg * 64]);
        buf ^= 1;
    }
    TC_STAMP(5);
    {
        const int fbase = 16 * half * TPWC + 4 * q;
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < TPWC; ++c) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(A.bC + fbase + 16 * c);
            const f32x4 w = *reinterpret_cast<const f32x4 *>(A.wdot + fbase + 16 * c);  // zero-padded
#pragma unroll
            for (int r = 0; r < 4; ++r) d = fmaf(fmaxf(accC[c][r] + b[r], 0.f), w[r], d);
        }
        d = tc_quad_sum(d);
        __syncthreads();  // exchange slots free (the LayerNorm exchanges have been read by everyone)
        if (q == 0) *my_x = d;
        __syncthreads();
        if (live && q == 0 && half == 0) {
            d = d + *peer_x + A.bdot[0];
            if (A.sel_ctl && A.sel_ctl[3] != 0) d = __builtin_nanf("");  // the batch did not fit the selection workspace
            if (A.logit) A.logit[m] = d;
            if (A.prob) A.prob[m] = 1.0f / (1.0f + expf(-d));
        }
    }
    TC_STAMP(6);
    TC_STAMPS_OUT(0);
}

template <int NTA, int NTB, int NTC, int WM = 0, bool ROWS = false>
int tc_launch(const TailArgs &a, hipStream_t s) {
    constexpr size_t lds = TcShape<NTA, NTB, NTC>::BYTES;
    auto kern = tail_chain_kernel<NTA, NTB, NTC, WM, ROWS>;
    LPF_SET_MAX_LDS(kern, lds);  // (per instantiation and device; the attribute is sticky)
    const int64_t blocks = (a.M + 16 * TC_GROUPS - 1) / (16 * TC_GROUPS);
    if (blocks > 0x7fffffff) return LPF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(TC_THREADS), lds, s, a);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

}  // namespace

#ifdef TC_STAMPS
extern "C" int lpf_tail_chain_set_stamps(void *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(tc_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? LPF_OK : LPF_ERR_LAUNCH;
}
#endif


extern "C" int lpf_tail_chain_f32(int64_t M, int32_t D, int32_t n_counts, const float *G, int64_t ldg,
                                  const float *wA_packed, const float *lnA_g, const float *lnA_b, const float *counts,
                                  int64_t ldc, const float *wB_packed, const float *bB, const float *lnB_g,
                                  const float *lnB_b, const float *r_e, int64_t ldre, const float *wC_packed,
                                  const float *bC, const float *w_dot, const float *b_dot, float *logit, float *prob,
                                  void *stream) {
    if (M == 0) return LPF_OK;
    LPF_REQUIRE(M > 0 && G && wA_packed && lnA_g && lnA_b && counts && wB_packed && bB && lnB_g && lnB_b && r_e &&
                wC_packed && bC && w_dot && b_dot && (logit || prob));
    LPF_REQUIRE(n_counts >= 1 && n_counts <= 4 && (ldg & 3) == 0 && ldg >= 4 * D + 4 && (ldc & 3) == 0 && ldc >= 4 &&
                (ldre & 3) == 0 && ldre >= D);
    LPF_REQUIRE(lpf_aligned16(G) && lpf_aligned16(counts) && lpf_aligned16(r_e) && lpf_aligned16(wA_packed) &&
                lpf_aligned16(wB_packed) && lpf_aligned16(wC_packed) && lpf_aligned16(lnA_g) && lpf_aligned16(lnA_b) &&
                lpf_aligned16(bB) && lpf_aligned16(lnB_g) && lpf_aligned16(lnB_b) && lpf_aligned16(bC) &&
                lpf_aligned16(w_dot));
    TailArgs a{M, G + D, ldg, 3 * D + 4, G, ldg, wA_packed, lnA_g, lnA_b, D, counts, ldc, wB_packed, bB, lnB_g,
               lnB_b, D + n_counts, r_e, ldre, wC_packed, bC, 2 * D, w_dot, b_dot, logit, prob,
               nullptr, nullptr, 0, nullptr, nullptr, nullptr, n_counts, nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (D) {
        case 32: return tc_launch<2, 3, 4>(a, s);
        case 64: return tc_launch<4, 5, 8>(a, s);
        case 128: return tc_launch<8, 9, 16>(a, s);
        default: return LPF_ERR_UNSUPPORTED;  // D = 256: 32-tile score head, the per-layer chains are used instead
    }
}

extern "C" int lpf_tail_chain_merge_f32(int64_t M, int32_t D, int32_t n_counts, const float *part, const float *bnd,
                                        int64_t units_cap, const int32_t *type_ptr, const float *att_bias, const float *lnA_g,
                                        const float *lnA_b, const float *wB_packed, const float *bB, const float *lnB_g,
                                        const float *lnB_b, const float *r_e, int64_t ldre, const float *wC_packed,
                                        const float *bC, const float *w_dot, const float *b_dot,
                                        const int64_t *sel_ctl, float *logit, float *prob, void *stream) {
    if (M == 0) return LPF_OK;
    LPF_REQUIRE(M > 0 && part && bnd && units_cap > 0 && lpf_aligned16(bnd) && type_ptr && att_bias && lnA_g && lnA_b && wB_packed && bB && lnB_g && lnB_b && r_e &&
                wC_packed && bC && w_dot && b_dot && (logit || prob));
    LPF_REQUIRE((n_counts == 1 || n_counts == 3 || n_counts == 4) && (ldre & 3) == 0 && ldre >= D);
    LPF_REQUIRE(lpf_aligned16(part) && lpf_aligned16(att_bias) && lpf_aligned16(r_e) && lpf_aligned16(wB_packed) &&
                lpf_aligned16(wC_packed) && lpf_aligned16(lnA_g) && lpf_aligned16(lnA_b) && lpf_aligned16(bB) &&
                lpf_aligned16(lnB_g) && lpf_aligned16(lnB_b) && lpf_aligned16(bC) && lpf_aligned16(w_dot));
    TailArgs a{M, nullptr, 0, 0, nullptr, 0, nullptr, lnA_g, lnA_b, D, nullptr, 0, wB_packed, bB, lnB_g,
               lnB_b, D + n_counts, r_e, ldre, wC_packed, bC, 2 * D, w_dot, b_dot, logit, prob,
               part, bnd, units_cap, type_ptr, att_bias, sel_ctl, n_counts, nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (D) {
        case 32: return tc_launch<2, 3, 4>(a, s);
        case 64: return tc_launch<4, 5, 8>(a, s);
        case 128: return tc_launch<8, 9, 16>(a, s);
        default: return LPF_ERR_UNSUPPORTED;
    }
}

extern "C" int lpf_tail_chain_merge_bf16(int64_t M, int32_t D, int32_t n_counts, const float *part, const float *bnd,
                                         int64_t units_cap, const int32_t *type_ptr, const float *att_bias,
                                         const float *lnA_g, const float *lnA_b, const void *wB_packed_bf16,
                                         const float *bB, const float *lnB_g, const float *lnB_b, const float *r_e,
                                         int64_t ldre, const void *wC_packed_bf16, const float *bC, const float *w_dot,
                                         const float *b_dot, const int64_t *sel_ctl, float *logit, float *prob,
                                         void *stream) {
    if (M == 0) return LPF_OK;
    LPF_REQUIRE(M > 0 && part && bnd && units_cap > 0 && lpf_aligned16(bnd) && type_ptr && att_bias && lnA_g && lnA_b &&
                wB_packed_bf16 && bB && lnB_g && lnB_b && r_e && wC_packed_bf16 && bC && w_dot && b_dot &&
                (logit || prob));
    LPF_REQUIRE((n_counts == 1 || n_counts == 3 || n_counts == 4) && (ldre & 3) == 0 && ldre >= D);
    LPF_REQUIRE(lpf_aligned16(part) && lpf_aligned16(att_bias) && lpf_aligned16(r_e) && lpf_aligned16(wB_packed_bf16) &&
                lpf_aligned16(wC_packed_bf16) && lpf_aligned16(lnA_g) && lpf_aligned16(lnA_b) && lpf_aligned16(bB) &&
                lpf_aligned16(lnB_g) && lpf_aligned16(lnB_b) && lpf_aligned16(bC) && lpf_aligned16(w_dot));
    TailArgs a{M, nullptr, 0, 0, nullptr, 0, nullptr, lnA_g, lnA_b, D, nullptr, 0,
               static_cast<const float *>(wB_packed_bf16), bB, lnB_g, lnB_b, D + n_counts, r_e, ldre,
               static_cast<const float *>(wC_packed_bf16), bC, 2 * D, w_dot, b_dot, logit, prob,
               part, bnd, units_cap, type_ptr, att_bias, sel_ctl, n_counts, nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (D) {
        case 32: return tc_launch<2, 3, 4, 1>(a, s);
        case 64: return tc_launch<4, 5, 8, 1>(a, s);
        case 128: return tc_launch<8, 9, 16, 1>(a, s);
        default: return LPF_ERR_UNSUPPORTED;
    }
}

// Rows mode: stage A was done by lpf_pair_attention_rows_* (one finished row per pair: post_att_norm(attention output)
// followed by the count features, zero padded to 4); the tail is the two GEMMs, their LayerNorm and the score.
namespace {
template <int WM>
int tc_rows(int64_t M, int32_t D, int32_t n_counts, const float *rows, int64_t ldrows, const void *wB, const float *bB,
            const float *lnB_g, const float *lnB_b, const float *r_e, int64_t ldre, const void *wC, const float *bC,
            const float *w_dot, const float *b_dot, const int64_t *sel_ctl, float *logit, float *prob, void *stream,
            const int32_t *perm = nullptr, const int64_t *n_full = nullptr, const float *bC_empty = nullptr,
            const float *row_empty = nullptr) {
    if (M == 0) return LPF_OK;
    LPF_REQUIRE(!perm || (n_full && bC_empty && lpf_aligned16(bC_empty)));
    LPF_REQUIRE(!row_empty || (perm && lpf_aligned16(row_empty)));
    LPF_REQUIRE(M > 0 && rows && wB && bB && lnB_g && lnB_b && r_e && wC && bC && w_dot && b_dot && (logit || prob));
    LPF_REQUIRE((n_counts == 1 || n_counts == 3 || n_counts == 4) && (ldre & 3) == 0 && ldre >= D && (ldrows & 3) == 0 &&
                ldrows >= D + 4);
    LPF_REQUIRE(lpf_aligned16(rows) && lpf_aligned16(r_e) && lpf_aligned16(wB) && lpf_aligned16(wC) && lpf_aligned16(bB) &&
                lpf_aligned16(lnB_g) && lpf_aligned16(lnB_b) && lpf_aligned16(bC) && lpf_aligned16(w_dot));
    TailArgs a{M, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, D, nullptr, 0, static_cast<const float *>(wB), bB,
               lnB_g, lnB_b, D + n_counts, r_e, ldre, static_cast<const float *>(wC), bC, 2 * D, w_dot, b_dot, logit, prob,
               nullptr, nullptr, 0, nullptr, nullptr, sel_ctl, n_counts, rows, ldrows, perm, n_full, bC_empty, row_empty};
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (D) {
        case 32: return tc_launch<2, 3, 4, WM, true>(a, s);
        case 64: return tc_launch<4, 5, 8, WM, true>(a, s);
        case 128: return tc_launch<8, 9, 16, WM, true>(a, s);
        case 256: return tc_launch<16, 17, 32, WM, true>(a, s);
        default: return LPF_ERR_UNSUPPORTED;
    }
}
}  // namespace

extern "C" int lpf_tail_chain_rows_f32(int64_t M, int32_t D, int32_t n_counts, const float *rows, int64_t ldrows,
                                       const float *wB_packed, const float *bB, const float *lnB_g, const float *lnB_b,
                                       const float *r_e, int64_t ldre, const float *wC_packed, const float *bC,
                                       const float *w_dot, const float *b_dot, const int64_t *sel_ctl, float *logit,
                                       float *prob, void *stream) {
    return tc_rows<0>(M, D, n_counts, rows, ldrows, wB_packed, bB, lnB_g, lnB_b, r_e, ldre, wC_packed, bC, w_dot, b_dot,
                          sel_ctl, logit, prob, stream);
}

extern "C" int lpf_tail_chain_rows_bf16(int64_t M, int32_t D, int32_t n_counts, const float *rows, int64_t ldrows,
                                        const void *wB_packed_bf16, const float *bB, const float *lnB_g, const float *lnB_b,
                                        const float *r_e, int64_t ldre, const void *wC_packed_bf16, const float *bC,
                                        const float *w_dot, const float *b_dot, const int64_t *sel_ctl, float *logit,
                                        float *prob, void *stream) {
    return tc_rows<1>(M, D, n_counts, rows, ldrows, wB_packed_bf16, bB, lnB_g, lnB_b, r_e, ldre, wC_packed_bf16, bC, w_dot,
                         b_dot, sel_ctl, logit, prob, stream);
}

/* lpf_tail_chain_rows_* taking the pairs in the order lpf_pair_attention_rows_perm_* left: perm int32[M], *n_full = how
 * many of them (the first ones) have selected nodes.  Workgroups whose 64 pairs all lie behind *n_full compute
 * score = w_dot . ReLU(A_e r_e + bC_empty) + b_dot only -- bC_empty [2D] = bC + A_p r_p0 with r_p0 the (constant) hidden
 * activation of pairwise_lin for a pair without selected nodes (lpformer_amd/fold.py empty_pair_head_bias). */
extern "C" int lpf_tail_chain_rows_perm_f32(int64_t M, int32_t D, int32_t n_counts, const float *rows, int64_t ldrows,
                                            const float *wB_packed, const float *bB, const float *lnB_g,
                                            const float *lnB_b, const float *r_e, int64_t ldre, const float *wC_packed,
                                            const float *bC, const float *w_dot, const float *b_dot,
                                            const int64_t *sel_ctl, const int32_t *perm, const int64_t *n_full,
                                            const float *bC_empty, const float *row_empty, float *logit, float *prob,
                                            void *stream) {
    LPF_REQUIRE(perm && n_full && bC_empty);
    return tc_rows<0>(M, D, n_counts, rows, ldrows, wB_packed, bB, lnB_g, lnB_b, r_e, ldre, wC_packed, bC, w_dot, b_dot,
                          sel_ctl, logit, prob, stream, perm, n_full, bC_empty, row_empty);
}

extern "C" int lpf_tail_chain_rows_perm_bf16(int64_t M, int32_t D, int32_t n_counts, const float *rows, int64_t ldrows,
                                             const void *wB_packed_bf16, const float *bB, const float *lnB_g,
                                             const float *lnB_b, const float *r_e, int64_t ldre,
                                             const void *wC_packed_bf16, const float *bC, const float *w_dot,
                                             const float *b_dot, const int64_t *sel_ctl, const int32_t *perm,
                                             const int64_t *n_full, const float *bC_empty, const float *row_empty,
                                             float *logit, float *prob, void *stream) {
    LPF_REQUIRE(perm && n_full && bC_empty);
    return tc_rows<1>(M, D, n_counts, rows, ldrows, wB_packed_bf16, bB, lnB_g, lnB_b, r_e, ldre, wC_packed_bf16, bC, w_dot,
                         b_dot, sel_ctl, logit, prob, stream, perm, n_full, bC_empty, row_empty);
}
