// Fused dense chains of the pair stage on the fp32 matrix cores: one launch replaces (gather) + Linear + LayerNorm +
// ReLU + Linear (+ dot + sigmoid) -- elementwise_lin, pairwise_lin, the q projection, the attention-output projection
// with its post-norm, and the mlp_score head (reference: src/models/other_models.py:125-138,173-179,
// src/modules/layers.py:78,212-215, src/models/link_transformer.py:101-102,177).
//
// Orientation: SAMPLES ON LANES.  v_mfma_f32_16x16x4_f32 computes D[i][j] += sum_k A[i][k] B[k][j] with lane
// l = (q = l>>4, j = l&15) holding B[k=q][j] and, in the accumulator, D[4q + r][j] (r = register 0..3).  Here i is an
// output feature, j one of 16 samples.  A group of 16 samples is owned by a PAIR of wavefronts, each computing half of
// the output-feature tiles of both layers (a batch of 32,768 samples is only 2,048 sample groups = 2 per SIMD, and a
// SIMD needs >= 2 wavefronts *issuing* fp32 MFMAs to reach full rate; the pair doubles the wavefronts
// and halves their registers, so 4 per SIMD are resident and barrier / load stalls of one are covered by the others):
//   * layer 1's B operand is the sample's own input row (the four lanes of a sample read 64 contiguous bytes per
//     k-group; or the product / sum of two gathered rows),
//   * LayerNorm over features is an in-lane reduction, two cross-lane steps and one exchange with the partner wave,
//   * the hidden activations go to LDS in exactly the accumulator layout, which IS the B-operand layout of layer 2
//     (k-group t of layer 2 = hidden tile t), so layer 2 reads them back with one 16-byte LDS load per k-group,
//   * a 1-wide second layer (the score head) is an in-lane dot product plus the same partner exchange.
// Weights are the A operand: host-packed in MFMA order (layout below), staged global -> registers -> LDS one stage
// ahead of the MFMAs that consume them (double-buffered LDS, one barrier per stage) and shared by the 8 wavefronts of
// a workgroup.  Tile counts are template constants (no guards inside the MFMA streams); shapes without an
// instantiation return LPF_ERR_UNSUPPORTED and the host uses the unfused kernels.
#include "lpf_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef LPF_DC_GROUPS   /* (tuning aid: 16-sample groups per workgroup.  2 is valid where a weight stage -- padded to 512
                          float4 by fold.pack_dense -- is a whole number of 256-thread passes: D >= 128) */
#define LPF_DC_GROUPS 4
#endif
constexpr int DC_GROUPS = LPF_DC_GROUPS;     // 16-sample groups per workgroup
constexpr int DC_WAVES = 2 * DC_GROUPS;      // wavefronts per workgroup (a pair per group)
constexpr int DC_THREADS = 64 * DC_WAVES;

struct DenseChainArgs {
    int64_t M;
    int in_mode;                 // 0: rows of X; 1: X[a] * X[b]; 2: X[a] + X[b]
    const float *X; int64_t ldx;
    const int64_t *batch; int64_t batch_ld; int64_t n_rows;
    int K1;                      // logical input width
    const float *w1p; int N1;    // packed layer-1 weights, logical output width
    const float *b1; const float *addend; int64_t ldadd;
    const float *ln_g; const float *ln_b;
    uint32_t flags;              // LPF_FLAG_RELU after (LN of) layer 1
    const float *w2p; int N2;    // packed layer-2 weights (NULL: single layer); N2 == 1: dot mode, w2p = plain vector
    const float *b2;
    float *out; int64_t ldo;     // [M, N2] (or [M, N1] for a single layer); dot mode: logit[M] (may be NULL)
    float *prob;                 // dot mode: sigmoid(logit) (may be NULL)
    // gather modes, optional: a SECOND table read with the same ids -- side[m] = S[a] + S[b] (lpf_dense_chain_side_f32:
    // the attention's per-pair query beside the elementwise branch, one launch for both)
    const float *side_tab; int64_t ld_side_tab; int side_dim;
    float *side_out; int64_t ld_side_out;
};

// k-groups (16 input features) per pipeline stage; fold.py::dense_stage_groups mirrors this.  One: with batches
// pipelined over streams the LDS a workgroup holds matters more than the barriers it saves (measured: 4 / 2 / 1
// k-groups per stage for the single-layer chains give 72.8 / 75.7 / 76.0 M pairs/s; the q projection alone drops from
// 0.033 to 0.020 ms).  The kernel stays templated on it.
constexpr int dc_groups(int ntp1, int ntp2) {
    (void)ntp1;
    (void)ntp2;
    return 1;
}

// Weight image (either layer), tiles padded to an even count NTP: "k-group" ks holds the A operands of the four MFMA
// steps that consume input features 16 ks .. 16 ks + 15 (lane q supplying B values 16 ks + 4 q + u): float4
// (c, lane = 16q + i) of the group = W[16c + i][16 ks + 4q + 0..3].  A stage = G consecutive k-groups (sq, c, lane),
// zero padded to a whole number of float4 per thread (P * 512), so staging is branch-free; missing k-groups of the
// last stage and the padding tile are zeros.
//
// Pipeline per stage: [regs -> LDS buffer b] barrier [issue global loads of stage s+1 into regs] [MFMAs of stage s
// from buffer b]; buffers alternate, so one barrier per stage is enough (a wave can only reach the write of buffer b
// for stage s+2 after every wave has passed the barrier of stage s+1, i.e. finished reading b for stage s).
constexpr int dc_per_thread(int ntp, int g) { return (ntp * g * 64 + DC_THREADS - 1) / DC_THREADS; }

template <int P>
__device__ __forceinline__ void dc_stage_load(f32x4 (&r)[P], const float *packed, int stage, int tid) {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(packed) + (int64_t)stage * (P * DC_THREADS);
#pragma unroll
    for (int e = 0; e < P; ++e) r[e] = src[e * DC_THREADS + tid];
}

template <int P>
__device__ __forceinline__ void dc_stage_store(const f32x4 (&r)[P], f32x4 *slab, int tid) {
#pragma unroll
    for (int e = 0; e < P; ++e) slab[e * DC_THREADS + tid] = r[e];
}

// acc[cc] += sum over GG k-groups of W[16 (c0 + cc) + i][k] * B[k][j] for this wave's TPW tiles; lw points at the
// wave's first tile of k-group 0 (+ lane); consecutive k-groups are NTP tiles apart.  The operand blocks are walked in
// pairs with the next pair's LDS reads issued before the current pair's eight MFMAs, and the two accumulators of a
// pair alternate so consecutive MFMAs are independent.
template <int TPW, int NTP, int G, int GG>
__device__ __forceinline__ void dc_mfma_n(f32x4 (&acc)[TPW], const f32x4 *lw, const f32x4 (&bv)[G]) {
    constexpr int T = GG * TPW;  // operand blocks, block t = (sq = t / TPW, cc = t % TPW)
    auto at = [&](int t) -> const f32x4 & { return lw[((t / TPW) * NTP + (t % TPW)) * 64]; };
    f32x4 n0 = at(0), n1 = T > 1 ? at(1) : at(0);
#pragma unroll
    for (int t = 0; t < T; t += 2) {
        const f32x4 a0 = n0, a1 = n1;
        if (t + 2 < T) n0 = at(t + 2);
        if (t + 3 < T) n1 = at(t + 3);
        const int s0 = t / TPW, c0 = t % TPW, s1 = (t + 1) / TPW, c1 = (t + 1) % TPW;
        if (t + 1 < T) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[c0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], bv[s0][u], acc[c0], 0, 0, 0);
                acc[c1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], bv[s1][u], acc[c1], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                acc[c0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], bv[s0][u], acc[c0], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch distance at one pair (register budget: 128)
    }
}

template <int TPW, int NTP, int G>
__device__ __forceinline__ void dc_mfma(f32x4 (&acc)[TPW], const f32x4 *lw, const f32x4 (&bv)[G], int cnt) {
    if (cnt == G) {
        dc_mfma_n<TPW, NTP, G, G>(acc, lw, bv);
    } else {  // ragged last stage
        if constexpr (G == 4) {
            if (cnt == 3) dc_mfma_n<TPW, NTP, G, 3>(acc, lw, bv);
            else if (cnt == 2) dc_mfma_n<TPW, NTP, G, 2>(acc, lw, bv);
            else dc_mfma_n<TPW, NTP, G, 1>(acc, lw, bv);
        } else if constexpr (G == 2) {
            dc_mfma_n<TPW, NTP, G, 1>(acc, lw, bv);
        }
    }
}

__device__ __forceinline__ float dc_quad_sum(float v) {  // sum over the 4 lanes (q = 0..3) that share a sample
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// this lane's raw input values of one stage: k-group g reads k = 16 g + 4 q + (0..3) of row a (and row b)
template <int G, int MODE>
__device__ __forceinline__ void dc_input_load(f32x4 (&a)[G], f32x4 (&b)[G], const DenseChainArgs &A, const float *xa,
                                              const float *xb, int g0, int q) {
#pragma unroll
    for (int sq = 0; sq < G; ++sq) {  // addresses clamped into the row; out-of-range values are zeroed in combine
        const int k = 16 * (g0 + sq) + 4 * q;
        const int kk = k < A.K1 ? k : A.K1 - 4;
        a[sq] = *reinterpret_cast<const f32x4 *>(xa + kk);
        if constexpr (MODE != 0) b[sq] = *reinterpret_cast<const f32x4 *>(xb + kk);
    }
}

template <int G, int MODE>
__device__ __forceinline__ void dc_input_combine(const f32x4 (&a)[G], const f32x4 (&b)[G], const DenseChainArgs &A,
                                                 int g0, int q, f32x4 (&bv)[G]) {
#pragma unroll
    for (int sq = 0; sq < G; ++sq) {
        const bool in = 16 * (g0 + sq) + 4 * q < A.K1;  // K1 % 4 == 0: a float4 is all-in or all-out
        f32x4 o = a[sq];
        if constexpr (MODE == 1) o = o * b[sq];
        if constexpr (MODE == 2) o = o + b[sq];
        bv[sq] = in ? o : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

// LDS carve-up (float4 units): [2 weight buffers][hidden exchange: DC_GROUPS * NTP1 * 64][partner scalars]
template <int NT1, int NT2, int G>
struct DcLds {
    static constexpr int NTP1 = (NT1 + 1) & ~1, NTP2 = (NT2 + 1) & ~1;
    static constexpr int P1 = dc_per_thread(NTP1, G), P2 = dc_per_thread(NTP2 ? NTP2 : 2, G);
    static constexpr int SLAB = (P1 > P2 ? P1 : P2) * DC_THREADS;
    static constexpr int HID = NT2 > 0 ? DC_GROUPS * NTP1 * 64 : 0;
    static constexpr int XCH = DC_WAVES * 16 / 4;  // one float per (wave, sample)
    static constexpr size_t BYTES = (size_t)(2 * SLAB + HID + XCH) * sizeof(f32x4);
};

// two workgroups per CU (4 wavefronts per SIMD, <= 128 VGPRs) whenever their LDS fits twice
template <int NT1, int NT2, int G>
constexpr int dc_min_waves() { return 2 * DcLds<NT1, NT2, G>::BYTES <= 160 * 1024 ? 4 : 2; }

template <int NT1, int NT2, int G, int MODE, int MINW>
__global__ __launch_bounds__(DC_THREADS, MINW) void dense_chain_kernel(const DenseChainArgs A) {
    using L = DcLds<NT1, NT2, G>;
    constexpr int NTP1 = L::NTP1, NTP2 = L::NTP2, TPW1 = NTP1 / 2, TPW2 = NTP2 / 2;
    constexpr int P1 = L::P1, P2 = L::P2, SLAB = L::SLAB;
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
    f32x4 *hid = lds + 2 * SLAB;
    float *xch = reinterpret_cast<float *>(lds + 2 * SLAB + L::HID);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 1, half = wave & 1;   // sample group of the workgroup; which half of the feature tiles
    const int q = lane >> 4, j = lane & 15;
    const int ng1 = (A.K1 + 15) >> 4;             // k-groups of layer 1
    constexpr int NS2 = (NTP1 + G - 1) / G;       // stages of layer 2 (k-groups = hidden tiles)
    float *my_x = xch + wave * 16 + j;            // partner exchange slots (one float per sample)
    const float *peer_x = xch + (wave ^ 1) * 16 + j;
    int buf = 0;

    {   // one block of 64 samples per workgroup (no persistent loop: loop-invariant operand addresses hoisted out of
        // it cost ~60 registers, and the weight staging is per 64 samples either way)
        const int64_t m0 = (int64_t)blockIdx.x * (16 * DC_GROUPS);
        const int64_t m = m0 + grp * 16 + j;
        const bool live = m < A.M;
        const int64_t mm = live ? m : A.M - 1;  // clamp: dead lanes compute on a valid row and store nothing
        int64_t ra = mm, rb = 0;
        if constexpr (MODE != 0) {
            ra = A.batch[mm];
            rb = A.batch[A.batch_ld + mm];
            // ids outside the table read row 0 (the selection kernel reports them; nothing is read out of bounds)
            if ((uint64_t)ra >= (uint64_t)A.n_rows) ra = 0;
            if ((uint64_t)rb >= (uint64_t)A.n_rows) rb = 0;
        }
        const float *xa = A.X + ra * A.ldx, *xb = A.X + rb * A.ldx;
        // side gather (lpf_dense_chain_side_f32): the eight threads of a sample (two wavefronts x four quads) share its
        // row S[a] + S[b] (that order, = lpf_pair_gather_f32).  Requested here, in front of the first operands; added and
        // stored behind the first stage's input wait, which they have passed by then (same queue, issued earlier).
        constexpr int SIDE_U = 4;
        f32x4 sva[SIDE_U], svb[SIDE_U];
        const bool side = MODE != 0 && A.side_out != nullptr;
        const int side_n4 = A.side_dim >> 2, side_f0 = 4 * half + q;
        auto side_load = [&](int f0) __attribute__((always_inline)) {
            const float *sa = A.side_tab + ra * A.ld_side_tab, *sb = A.side_tab + rb * A.ld_side_tab;
#pragma unroll
            for (int u = 0; u < SIDE_U; ++u) {
                const int f = f0 + 8 * u < side_n4 ? f0 + 8 * u : side_f0;
                sva[u] = *reinterpret_cast<const f32x4 *>(sa + 4 * f);
                svb[u] = *reinterpret_cast<const f32x4 *>(sb + 4 * f);
            }
        };
        auto side_store = [&](int f0) __attribute__((always_inline)) {
            float *so = A.side_out + mm * A.ld_side_out;
#pragma unroll
            for (int u = 0; u < SIDE_U; ++u)
                if (live && f0 + 8 * u < side_n4) *reinterpret_cast<f32x4 *>(so + 4 * (f0 + 8 * u)) = sva[u] + svb[u];
        };
        if (side && side_f0 < side_n4) side_load(side_f0);

        // ---------------- layer 1: this wave's tiles half*TPW1 .. half*TPW1 + TPW1 - 1
        f32x4 acc1[TPW1];
#pragma unroll
        for (int c = 0; c < TPW1; ++c) acc1[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            f32x4 wr[P1], xra[G], xrb[G];
            dc_stage_load<P1>(wr, A.w1p, 0, tid);
            dc_input_load<G, MODE>(xra, xrb, A, xa, xb, 0, q);
#pragma unroll 1
            for (int g0 = 0; g0 < ng1; g0 += G) {
                const int cnt = ng1 - g0 < G ? ng1 - g0 : G;
                f32x4 bv[G];
                dc_input_combine<G, MODE>(xra, xrb, A, g0, q, bv);
                if (side && g0 == 0) {
                    if (side_f0 < side_n4) side_store(side_f0);
                    for (int f0 = side_f0 + 8 * SIDE_U; f0 < side_n4; f0 += 8 * SIDE_U) {   // (tables wider than 128)
                        side_load(f0);
                        side_store(f0);
                    }
                }
                f32x4 *lw = lds + buf * SLAB;
                dc_stage_store<P1>(wr, lw, tid);
                __syncthreads();
                if (g0 + G < ng1) {  // next stage's operands fly while this stage's MFMAs run
                    dc_stage_load<P1>(wr, A.w1p, g0 / G + 1, tid);
                    dc_input_load<G, MODE>(xra, xrb, A, xa, xb, g0 + G, q);
                }
                dc_mfma<TPW1, NTP1, G>(acc1, lw + (half * TPW1) * 64 + lane, bv, cnt);
                buf ^= 1;
            }
        }
        // layer 2's first weight stage flies during the epilogue
        f32x4 w2r[P2];
        if constexpr (NT2 > 0) dc_stage_load<P2>(w2r, A.w2p, 0, tid);

        // epilogue 1: bias (+ addend) -> LayerNorm over the N1 real features -> ReLU
        const int fbase = 16 * half * TPW1 + 4 * q;  // first feature of this lane's register quad in tile 0
#pragma unroll
        for (int c = 0; c < TPW1; ++c)
            acc1[c] += *reinterpret_cast<const f32x4 *>(A.b1 + fbase + 16 * c);  // padded with zeros by the host
        if (A.addend) {  // all pieces in flight together; columns clamped into the row, pads add nothing
            f32x4 ad[TPW1];
#pragma unroll
            for (int c = 0; c < TPW1; ++c) {
                const int f0 = fbase + 16 * c;
                ad[c] = *reinterpret_cast<const f32x4 *>(A.addend + mm * A.ldadd + (f0 < A.N1 ? f0 : 0));
            }
#pragma unroll
            for (int c = 0; c < TPW1; ++c)
                acc1[c] += (fbase + 16 * c < A.N1) ? ad[c] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (A.ln_g) {
            float s1 = 0.f;
#pragma unroll
            for (int c = 0; c < TPW1; ++c) s1 += acc1[c][0] + acc1[c][1] + acc1[c][2] + acc1[c][3];
            s1 = dc_quad_sum(s1);  // padded features are exactly 0 and add nothing
            if (q == 0) *my_x = s1;
            __syncthreads();
            const float mean = (half == 0 ? s1 + *peer_x : *peer_x + s1) / (float)A.N1;  // same order in both waves
            __syncthreads();       // slots free for the second exchange
            float s2 = 0.f;
#pragma unroll
            for (int c = 0; c < TPW1; ++c) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = (fbase + 16 * c + r < A.N1) ? acc1[c][r] - mean : 0.f;
                    s2 += d * d;
                }
            }
            s2 = dc_quad_sum(s2);
            if (q == 0) *my_x = s2;
            __syncthreads();
            const float rstd = 1.0f / sqrtf((half == 0 ? s2 + *peer_x : *peer_x + s2) / (float)A.N1 + 1e-5f);
#pragma unroll
            for (int c = 0; c < TPW1; ++c) {
                const f32x4 g = *reinterpret_cast<const f32x4 *>(A.ln_g + fbase + 16 * c);  // zero-padded: pads -> 0
                const f32x4 be = *reinterpret_cast<const f32x4 *>(A.ln_b + fbase + 16 * c);
                acc1[c] = (acc1[c] - mean) * rstd * g + be;
            }
        }
        if (A.flags & LPF_FLAG_RELU) {
#pragma unroll
            for (int c = 0; c < TPW1; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc1[c][r] = fmaxf(acc1[c][r], 0.f);
        }

        if constexpr (NT2 == 0) {
            if (!A.w2p) {  // single layer: store [M, N1]
#pragma unroll
                for (int c = 0; c < TPW1; ++c) {
                    const int f0 = fbase + 16 * c;
                    if (live && f0 < A.N1) *reinterpret_cast<f32x4 *>(A.out + m * A.ldo + f0) = acc1[c];
                }
            } else {  // 1-wide second layer: logit = w2 . hidden + b2
                float d = 0.f;
#pragma unroll
                for (int c = 0; c < TPW1; ++c) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(A.w2p + fbase + 16 * c);  // zero-padded
                    d = fmaf(acc1[c][0], w[0], fmaf(acc1[c][1], w[1], fmaf(acc1[c][2], w[2], fmaf(acc1[c][3], w[3], d))));
                }
                d = dc_quad_sum(d);
                __syncthreads();  // (the LayerNorm exchange, if any, has been read by everyone)
                if (q == 0) *my_x = d;
                __syncthreads();
                if (live && q == 0 && half == 0) {
                    d = d + *peer_x + A.b2[0];
                    if (A.out) A.out[m] = d;
                    if (A.prob) A.prob[m] = 1.0f / (1.0f + expf(-d));
                }
            }
        } else {
            // ---------------- layer 2: hidden tile t (all NTP1 of the group) is k-group t, read back from LDS in the
            // accumulator layout, which is the B-operand layout
            f32x4 *my_hid = hid + (grp * NTP1) * 64 + lane;
#pragma unroll
            for (int c = 0; c < TPW1; ++c) my_hid[(half * TPW1 + c) * 64] = acc1[c];
            f32x4 acc2[TPW2];
#pragma unroll
            for (int c = 0; c < TPW2; ++c) acc2[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < NS2; ++st) {
                const int cnt = NTP1 - st * G < G ? NTP1 - st * G : G;
                f32x4 *lw = lds + buf * SLAB;
                dc_stage_store<P2>(w2r, lw, tid);
                __syncthreads();  // (stage 0: also publishes the hidden tiles)
                if (st + 1 < NS2) dc_stage_load<P2>(w2r, A.w2p, st + 1, tid);
                f32x4 bv[G];
#pragma unroll
                for (int sq = 0; sq < G; ++sq)
                    bv[sq] = st * G + sq < NTP1 ? my_hid[(st * G + sq < NTP1 ? st * G + sq : 0) * 64]
                                                : (f32x4){0.f, 0.f, 0.f, 0.f};
                dc_mfma<TPW2, NTP2, G>(acc2, lw + (half * TPW2) * 64 + lane, bv, cnt);
                buf ^= 1;
            }
            const int fb2 = 16 * half * TPW2 + 4 * q;
#pragma unroll
            for (int c = 0; c < TPW2; ++c) {
                const int f0 = fb2 + 16 * c;
                if (live && f0 < A.N2)
                    *reinterpret_cast<f32x4 *>(A.out + m * A.ldo + f0) =
                        acc2[c] + *reinterpret_cast<const f32x4 *>(A.b2 + f0);
            }
        }
    }
}

template <int NT1, int NT2, int MODE>
int dc_launch(const DenseChainArgs &a, hipStream_t s) {
    constexpr int NTP1 = (NT1 + 1) & ~1, NTP2 = (NT2 + 1) & ~1;
    constexpr int G = dc_groups(NTP1, NTP2);
    constexpr size_t lds = DcLds<NT1, NT2, G>::BYTES;
    auto kern = dense_chain_kernel<NT1, NT2, G, MODE, dc_min_waves<NT1, NT2, G>()>;
    LPF_SET_MAX_LDS(kern, lds);  // (per instantiation and device; the attribute is sticky)
    const int64_t blocks = (a.M + 16 * DC_GROUPS - 1) / (16 * DC_GROUPS);
    if (blocks > 0x7fffffff) return LPF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(DC_THREADS), lds, s, a);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

}  // namespace

namespace {
int dc_entry(int64_t M, int32_t in_mode, const float *X, int64_t ldx, const int64_t *batch, int64_t batch_ld,
             int64_t n_rows, int32_t K1, const float *w1_packed, int32_t N1, const float *b1, const float *addend,
             int64_t ldadd, const float *ln_g, const float *ln_b, uint32_t flags, const float *w2_packed, int32_t N2,
             const float *b2, float *out, int64_t ldo, float *prob, const float *side_tab, int64_t ld_side_tab,
             int32_t side_dim, float *side_out, int64_t ld_side_out, void *stream) {
    if (M == 0) return LPF_OK;
    LPF_REQUIRE(M > 0 && X && w1_packed && b1 && K1 > 0 && N1 > 0 && (ldx & 3) == 0 && lpf_aligned16(X) &&
                lpf_aligned16(w1_packed) && lpf_aligned16(b1));
    LPF_REQUIRE(in_mode >= 0 && in_mode <= 2 && (in_mode == 0 || (batch && batch_ld >= M && n_rows > 0)));
    LPF_REQUIRE((K1 & 3) == 0 && ldx >= K1);
    LPF_REQUIRE((!ln_g) == (!ln_b) && (!ln_g || (lpf_aligned16(ln_g) && lpf_aligned16(ln_b))));
    LPF_REQUIRE(!addend || ((ldadd & 3) == 0 && lpf_aligned16(addend) && ldadd >= N1 && (N1 & 3) == 0));
    const bool two = w2_packed != nullptr;
    const bool dot = two && N2 == 1;
    LPF_REQUIRE(!two || (b2 && lpf_aligned16(w2_packed) && N2 > 0));
    LPF_REQUIRE(dot ? (out || prob) : (out && (ldo & 3) == 0 && lpf_aligned16(out)));
    LPF_REQUIRE(dot || ((two ? N2 : N1) & 3) == 0);
    LPF_REQUIRE(dot || ldo >= (two ? N2 : N1));
    LPF_REQUIRE(dot || !two || lpf_aligned16(b2));
    const int nt1 = (N1 + 15) / 16, nt2 = (two && !dot) ? (N2 + 15) / 16 : 0;
    LPF_REQUIRE(!side_out || (in_mode != 0 && side_tab && side_dim > 0 && (side_dim & 3) == 0 && ld_side_tab >= side_dim &&
                              ld_side_out >= side_dim && (ld_side_tab & 3) == 0 && (ld_side_out & 3) == 0 &&
                              lpf_aligned16(side_tab) && lpf_aligned16(side_out)));
    DenseChainArgs a{M, in_mode, X, ldx, batch, batch_ld, n_rows, K1, w1_packed, N1, b1, addend, ldadd, ln_g, ln_b, flags,
                     w2_packed, N2, b2, out, ldo, prob, side_tab, ld_side_tab, side_dim, side_out, ld_side_out};
    hipStream_t s = static_cast<hipStream_t>(stream);
    // in_mode 1 (gather-multiply) is built for the square two-layer chains (elementwise_lin) and the single-layer
    // ones (its first layer alone), in_mode 2 (gather-add) for the single-layer ones; plain rows for everything
#define DC_CASE(T1, T2, MODE) \
    if (nt1 == T1 && nt2 == T2 && in_mode == MODE) return dc_launch<T1, T2, MODE>(a, s)
#define DC_SINGLE(T1) DC_CASE(T1, 0, 0); DC_CASE(T1, 0, 1); DC_CASE(T1, 0, 2)
#define DC_SQUARE(T1) DC_CASE(T1, T1, 0); DC_CASE(T1, T1, 1)
    DC_SINGLE(2);    // hidden 32
    DC_SINGLE(4);    // hidden 64: q projection, attention output, score head
    DC_SINGLE(8);    // hidden 128
    DC_SINGLE(16);   // hidden 256
    DC_CASE(32, 0, 0);  // hidden 512 (score head of a 256-wide model)
    DC_SQUARE(2);
    DC_SQUARE(4);    // elementwise_lin 64 -> 64 -> 64
    DC_SQUARE(8);
    DC_SQUARE(16);
    DC_CASE(3, 0, 0);   // first layer of pairwise_lin alone (hidden layer kept for the folded score head)
    DC_CASE(5, 0, 0);
    DC_CASE(9, 0, 0);
    DC_CASE(17, 0, 0);
    DC_CASE(3, 2, 0);   // pairwise_lin: (D + counts) -> (D + counts) -> D
    DC_CASE(5, 4, 0);
    DC_CASE(9, 8, 0);
    DC_CASE(17, 16, 0);
#undef DC_CASE
#undef DC_SINGLE
#undef DC_SQUARE
    return LPF_ERR_UNSUPPORTED;
}
}  // namespace

extern "C" int lpf_dense_chain_f32(int64_t M, int32_t in_mode, const float *X, int64_t ldx, const int64_t *batch,
                                   int64_t batch_ld, int64_t n_rows, int32_t K1, const float *w1_packed, int32_t N1, const float *b1,
                                   const float *addend, int64_t ldadd, const float *ln_g, const float *ln_b,
                                   uint32_t flags, const float *w2_packed, int32_t N2, const float *b2, float *out,
                                   int64_t ldo, float *prob, void *stream) {
    return dc_entry(M, in_mode, X, ldx, batch, batch_ld, n_rows, K1, w1_packed, N1, b1, addend, ldadd, ln_g, ln_b, flags,
                    w2_packed, N2, b2, out, ldo, prob, nullptr, 0, 0, nullptr, 0, stream);
}

/* lpf_dense_chain_f32 in a gather mode (in_mode 1 or 2) that ALSO leaves side_out[m, :side_dim] = S[a_m] + S[b_m] for a
 * second table S [n_rows, ld_side_tab] read with the same ids -- lpf_pair_gather_f32(sum) without a launch of its own:
 * the attention's per-pair query q = lin_l(x_a) + lin_l(x_b) from the per-node table lin_l(X) (layers.py:212-215) beside
 * the elementwise branch of the same batch.  side_dim % 4 == 0; rows 16-byte aligned. */
extern "C" int lpf_dense_chain_side_f32(int64_t M, int32_t in_mode, const float *X, int64_t ldx, const int64_t *batch,
                                        int64_t batch_ld, int64_t n_rows, int32_t K1, const float *w1_packed, int32_t N1,
                                        const float *b1, const float *addend, int64_t ldadd, const float *ln_g,
                                        const float *ln_b, uint32_t flags, const float *w2_packed, int32_t N2,
                                        const float *b2, float *out, int64_t ldo, float *prob, const float *side_tab,
                                        int64_t ld_side_tab, int32_t side_dim, float *side_out, int64_t ld_side_out,
                                        void *stream) {
    LPF_REQUIRE(side_tab && side_out && in_mode != 0);
    return dc_entry(M, in_mode, X, ldx, batch, batch_ld, n_rows, K1, w1_packed, N1, b1, addend, ldadd, ln_g, ln_b, flags,
                    w2_packed, N2, b2, out, ldo, prob, side_tab, ld_side_tab, side_dim, side_out, ld_side_out, stream);
}
