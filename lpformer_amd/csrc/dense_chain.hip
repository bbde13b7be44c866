// Fused dense chains of the pair stage on the fp32 matrix cores: one launch replaces (gather) + Linear + LayerNorm +
// ReLU + Linear (+ dot + sigmoid) -- elementwise_lin, pairwise_lin, the q projection, the attention-output projection
// with its post-norm, and the mlp_score head (reference: src/models/other_models.py:125-138,173-179,
// src/modules/layers.py:78,212-215, src/models/link_transformer.py:101-102,177).
//
// Orientation: SAMPLES ON LANES.  v_mfma_f32_16x16x4_f32 computes D[i][j] += sum_k A[i][k] B[k][j] with lane
// l = (q = l>>4, j = l&15) holding B[k=q][j] and, in the accumulator, D[4q + r][j] (r = register 0..3).  Here i is an
// output feature, j one of 16 samples, so a wavefront owns 16 samples and
//   * layer 1's B operand is the sample's own input row (the four lanes of a sample read 64 contiguous bytes per step
//     group; or the product / sum of two gathered rows),
//   * LayerNorm over features is an in-lane reduction plus two cross-lane steps,
//   * layer 2's B operand for step (feature tile c, register r) IS accumulator register acc1[c][r] of the same lane:
//     the hidden activations never leave the registers,
//   * a 1-wide second layer (the score head) is an in-lane dot product.
// Weights are the A operand: host-packed in MFMA order (layout below), staged global -> registers -> LDS one stage
// ahead of the MFMAs that consume them (double-buffered LDS, one barrier per stage) and shared by the 4 wavefronts of
// a workgroup; two workgroups per CU keep two waves per SIMD resident (one wave alone issues fp32 MFMAs at half rate,
// DESIGN.md 5.1).  Tile counts are template constants (exact, no guards inside the MFMA streams); shapes without an
// instantiation return LPF_ERR_UNSUPPORTED and the host uses the unfused kernels.
#include "lpf_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DC_WAVES = 4;      // wavefronts per workgroup

struct DenseChainArgs {
    int64_t M;
    int in_mode;                 // 0: rows of X; 1: X[a] * X[b]; 2: X[a] + X[b]
    const float *X; int64_t ldx;
    const int64_t *batch; int64_t batch_ld;
    int K1;                      // logical input width
    const float *w1p; int N1;    // packed layer-1 weights, logical output width
    const float *b1; const float *addend; int64_t ldadd;
    const float *ln_g; const float *ln_b;
    uint32_t flags;              // LPF_FLAG_RELU after (LN of) layer 1
    const float *w2p; int N2;    // packed layer-2 weights (NULL: single layer); N2 == 1: dot mode, w2p = plain vector
    const float *b2;
    float *out; int64_t ldo;     // [M, N2] (or [M, N1] for a single layer); dot mode: logit[M] (may be NULL)
    float *prob;                 // dot mode: sigmoid(logit) (may be NULL)
};

// Weight image (either layer): "k-group" ks holds the A operands of the four MFMA steps that consume input features
// 16 ks .. 16 ks + 15 (lane q supplying B values 16 ks + 4 q + u): float4 (c, lane = 16q + i) of the group =
// W[16c + i][16 ks + 4q + 0..3].  A stage = G consecutive k-groups (sq, c, lane), zero padded to a whole number of
// float4 per thread (P * 256), so staging is branch-free; missing k-groups of the last stage are zeros.
//
// Pipeline per stage: [regs -> LDS buffer b] barrier [issue global loads of stage s+1 into regs] [MFMAs of stage s
// from buffer b]; buffers alternate, so one barrier per stage is enough (a wave can only reach the write of buffer b
// for stage s+2 after every wave has passed the barrier of stage s+1, i.e. finished reading b for stage s).
template <int NT, int G>
constexpr int dc_per_thread() { return (NT * G * 64 + 64 * DC_WAVES - 1) / (64 * DC_WAVES); }

template <int NT, int G, int P>
__device__ __forceinline__ void dc_stage_load(f32x4 (&r)[P], const float *packed, int stage, int tid) {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(packed) + (int64_t)stage * (P * 64 * DC_WAVES);
#pragma unroll
    for (int e = 0; e < P; ++e) r[e] = src[e * 64 * DC_WAVES + tid];
}

template <int NT, int G, int P>
__device__ __forceinline__ void dc_stage_store(const f32x4 (&r)[P], f32x4 *slab, int tid) {
#pragma unroll
    for (int e = 0; e < P; ++e) slab[e * 64 * DC_WAVES + tid] = r[e];
}

// acc[c] += sum over GG k-groups of W[16c+i][k] * B[k][j]; bv[sq]: the lane's four B values of k-group sq.
// The (k-group, tile) operand blocks are walked in pairs with the next pair's LDS reads issued before the current
// pair's eight MFMAs, and the two accumulators of a pair alternate so consecutive MFMAs are independent.
template <int NT, int G, int GG>
__device__ __forceinline__ void dc_mfma_n(f32x4 (&acc)[NT], const f32x4 *lw, int lane, const f32x4 (&bv)[G]) {
    constexpr int T = GG * NT;  // operand blocks, block t = (sq = t / NT, c = t % NT)
    const f32x4 *p = lw + lane;
    f32x4 n0 = p[0], n1 = T > 1 ? p[64] : p[0];
#pragma unroll
    for (int t = 0; t < T; t += 2) {
        const f32x4 a0 = n0, a1 = n1;
        if (t + 2 < T) n0 = p[(t + 2) * 64];
        if (t + 3 < T) n1 = p[(t + 3) * 64];
        const int s0 = t / NT, c0 = t % NT, s1 = (t + 1) / NT, c1 = (t + 1) % NT;
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
    }
}

template <int NT, int G>
__device__ __forceinline__ void dc_mfma(f32x4 (&acc)[NT], const f32x4 *lw, int lane, const f32x4 (&bv)[G], int cnt) {
    if (cnt == G) {
        dc_mfma_n<NT, G, G>(acc, lw, lane, bv);
    } else {  // ragged last stage
        if constexpr (G == 4) {
            if (cnt == 3) dc_mfma_n<NT, G, 3>(acc, lw, lane, bv);
            else if (cnt == 2) dc_mfma_n<NT, G, 2>(acc, lw, lane, bv);
            else dc_mfma_n<NT, G, 1>(acc, lw, lane, bv);
        } else {
            dc_mfma_n<NT, G, 1>(acc, lw, lane, bv);
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

template <int NT1, int NT2, int G, int MODE>
__global__ __launch_bounds__(64 * DC_WAVES) void dense_chain_kernel(const DenseChainArgs A) {
    constexpr int P1 = dc_per_thread<NT1, G>(), P2 = dc_per_thread<(NT2 ? NT2 : 1), G>();
    constexpr int SLAB = (P1 > P2 ? P1 : P2) * 64 * DC_WAVES;         // float4 per LDS buffer
    extern __shared__ __attribute__((aligned(16))) f32x4 slab[];     // 2 buffers
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int ng1 = (A.K1 + 15) >> 4;             // k-groups of layer 1
    constexpr int NS2 = (NT1 + G - 1) / G;        // stages of layer 2 (k-groups = hidden tiles)
    int buf = 0;

#pragma unroll 1
    for (int64_t m0 = (int64_t)blockIdx.x * (16 * DC_WAVES); m0 < A.M; m0 += (int64_t)gridDim.x * (16 * DC_WAVES)) {
        const int64_t m = m0 + wave * 16 + j;
        const bool live = m < A.M;
        const int64_t mm = live ? m : A.M - 1;  // clamp: dead lanes compute on a valid row and store nothing
        int64_t ra = mm, rb = 0;
        if constexpr (MODE != 0) {
            ra = A.batch[mm];
            rb = A.batch[A.batch_ld + mm];
        }
        const float *xa = A.X + ra * A.ldx, *xb = A.X + rb * A.ldx;

        // ---------------- layer 1
        f32x4 acc1[NT1];
#pragma unroll
        for (int c = 0; c < NT1; ++c) acc1[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            f32x4 wr[P1], xra[G], xrb[G];
            dc_stage_load<NT1, G>(wr, A.w1p, 0, tid);
            dc_input_load<G, MODE>(xra, xrb, A, xa, xb, 0, q);
#pragma unroll 1
            for (int g0 = 0; g0 < ng1; g0 += G) {
                const int cnt = ng1 - g0 < G ? ng1 - g0 : G;
                f32x4 bv[G];
                dc_input_combine<G, MODE>(xra, xrb, A, g0, q, bv);
                f32x4 *lw = slab + buf * SLAB;
                dc_stage_store<NT1, G>(wr, lw, tid);
                __syncthreads();
                if (g0 + G < ng1) {  // next stage's operands fly while this stage's MFMAs run
                    dc_stage_load<NT1, G>(wr, A.w1p, g0 / G + 1, tid);
                    dc_input_load<G, MODE>(xra, xrb, A, xa, xb, g0 + G, q);
                }
                dc_mfma<NT1, G>(acc1, lw, lane, bv, cnt);
                buf ^= 1;
            }
        }
        // layer 2's first weight stage flies during the epilogue
        f32x4 w2r[P2];
        if constexpr (NT2 > 0) dc_stage_load<NT2, G>(w2r, A.w2p, 0, tid);

        // epilogue 1: bias (+ addend) -> LayerNorm over the N1 real features -> ReLU
        float s1 = 0.f;
#pragma unroll
        for (int c = 0; c < NT1; ++c) {
            const int f0 = 16 * c + 4 * q;
            const float4 b = *reinterpret_cast<const float4 *>(A.b1 + f0);  // padded with zeros by the host
            acc1[c][0] += b.x; acc1[c][1] += b.y; acc1[c][2] += b.z; acc1[c][3] += b.w;
        }
        if (A.addend) {  // all rows' pieces in flight together; columns clamped into the row, pads add nothing
            f32x4 ad[NT1];
#pragma unroll
            for (int c = 0; c < NT1; ++c) {
                const int f0 = 16 * c + 4 * q;
                ad[c] = *reinterpret_cast<const f32x4 *>(A.addend + mm * A.ldadd + (f0 < A.N1 ? f0 : 0));
            }
#pragma unroll
            for (int c = 0; c < NT1; ++c)
                acc1[c] += (16 * c + 4 * q < A.N1) ? ad[c] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int c = 0; c < NT1; ++c) s1 += acc1[c][0] + acc1[c][1] + acc1[c][2] + acc1[c][3];
        if (A.ln_g) {
            const float mean = dc_quad_sum(s1) / (float)A.N1;  // padded features are exactly 0 and add nothing
            float s2 = 0.f;
#pragma unroll
            for (int c = 0; c < NT1; ++c) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = (16 * c + 4 * q + r < A.N1) ? acc1[c][r] - mean : 0.f;
                    s2 += d * d;
                }
            }
            const float rstd = 1.0f / sqrtf(dc_quad_sum(s2) / (float)A.N1 + 1e-5f);
#pragma unroll
            for (int c = 0; c < NT1; ++c) {
                const int f0 = 16 * c + 4 * q;
                const float4 g = *reinterpret_cast<const float4 *>(A.ln_g + f0);  // zero-padded: pads come out 0
                const float4 be = *reinterpret_cast<const float4 *>(A.ln_b + f0);
                acc1[c][0] = (acc1[c][0] - mean) * rstd * g.x + be.x;
                acc1[c][1] = (acc1[c][1] - mean) * rstd * g.y + be.y;
                acc1[c][2] = (acc1[c][2] - mean) * rstd * g.z + be.z;
                acc1[c][3] = (acc1[c][3] - mean) * rstd * g.w + be.w;
            }
        }
        if (A.flags & LPF_FLAG_RELU) {
#pragma unroll
            for (int c = 0; c < NT1; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc1[c][r] = fmaxf(acc1[c][r], 0.f);
        }

        if constexpr (NT2 == 0) {
            if (!A.w2p) {  // single layer: store [M, N1]
#pragma unroll
                for (int c = 0; c < NT1; ++c) {
                    const int f0 = 16 * c + 4 * q;
                    if (live && f0 < A.N1)
                        *reinterpret_cast<float4 *>(A.out + m * A.ldo + f0) =
                            make_float4(acc1[c][0], acc1[c][1], acc1[c][2], acc1[c][3]);
                }
            } else {  // 1-wide second layer: logit = w2 . hidden + b2
                float d = 0.f;
#pragma unroll
                for (int c = 0; c < NT1; ++c) {
                    const float4 w = *reinterpret_cast<const float4 *>(A.w2p + 16 * c + 4 * q);  // zero-padded
                    d = fmaf(acc1[c][0], w.x, fmaf(acc1[c][1], w.y, fmaf(acc1[c][2], w.z, fmaf(acc1[c][3], w.w, d))));
                }
                d = dc_quad_sum(d) + A.b2[0];
                if (live && q == 0) {
                    if (A.out) A.out[m] = d;
                    if (A.prob) A.prob[m] = 1.0f / (1.0f + expf(-d));
                }
            }
        } else {
            // ---------------- layer 2: the B operand of k-group t (hidden tile t) is acc1[t] itself
            f32x4 acc2[NT2];
#pragma unroll
            for (int c = 0; c < NT2; ++c) acc2[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < NS2; ++st) {
                const int cnt = NT1 - st * G < G ? NT1 - st * G : G;
                f32x4 bv[G];
#pragma unroll
                for (int sq = 0; sq < G; ++sq) {
                    const int t = st * G + sq;
                    bv[sq] = t < NT1 ? acc1[t < NT1 ? t : 0] : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                f32x4 *lw = slab + buf * SLAB;
                dc_stage_store<NT2, G>(w2r, lw, tid);
                __syncthreads();
                if (st + 1 < NS2) dc_stage_load<NT2, G>(w2r, A.w2p, st + 1, tid);
                dc_mfma<NT2, G>(acc2, lw, lane, bv, cnt);
                buf ^= 1;
            }
#pragma unroll
            for (int c = 0; c < NT2; ++c) {
                const int f0 = 16 * c + 4 * q;
                if (live && f0 < A.N2) {
                    const float4 b = *reinterpret_cast<const float4 *>(A.b2 + f0);
                    *reinterpret_cast<float4 *>(A.out + m * A.ldo + f0) =
                        make_float4(acc2[c][0] + b.x, acc2[c][1] + b.y, acc2[c][2] + b.z, acc2[c][3] + b.w);
                }
            }
        }
    }
}

template <int NT1, int NT2, int MODE>
int dc_launch(const DenseChainArgs &a, hipStream_t s) {
    constexpr int slab_tiles = NT1 > NT2 ? NT1 : NT2;
    constexpr int G = slab_tiles <= 9 ? 4 : 2;  // k-groups per stage: 64 (narrow layers) or 32 input features
    constexpr int P1 = dc_per_thread<NT1, G>(), P2 = dc_per_thread<(NT2 ? NT2 : 1), G>();
    constexpr size_t lds = 2 * (size_t)(P1 > P2 ? P1 : P2) * 64 * DC_WAVES * sizeof(float4);
    auto kern = dense_chain_kernel<NT1, NT2, G, MODE>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            lpf_set_hip_error(e);
            return LPF_ERR_LAUNCH;
        }
    }
    int64_t blocks = (a.M + 16 * DC_WAVES - 1) / (16 * DC_WAVES);
    if (blocks > 2048) blocks = 2048;  // persistent beyond eight workgroups per CU
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * DC_WAVES), lds, s, a);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

}  // namespace

extern "C" int lpf_dense_chain_f32(int64_t M, int32_t in_mode, const float *X, int64_t ldx, const int64_t *batch,
                                   int64_t batch_ld, int32_t K1, const float *w1_packed, int32_t N1, const float *b1,
                                   const float *addend, int64_t ldadd, const float *ln_g, const float *ln_b,
                                   uint32_t flags, const float *w2_packed, int32_t N2, const float *b2, float *out,
                                   int64_t ldo, float *prob, void *stream) {
    if (M == 0) return LPF_OK;
    LPF_REQUIRE(M > 0 && X && w1_packed && b1 && K1 > 0 && N1 > 0 && (ldx & 3) == 0 && lpf_aligned16(X) &&
                lpf_aligned16(w1_packed) && lpf_aligned16(b1));
    LPF_REQUIRE(in_mode >= 0 && in_mode <= 2 && (in_mode == 0 || (batch && batch_ld >= M)));
    LPF_REQUIRE((K1 & 3) == 0 && ldx >= K1);
    LPF_REQUIRE((!ln_g) == (!ln_b) && (!ln_g || (lpf_aligned16(ln_g) && lpf_aligned16(ln_b))));
    LPF_REQUIRE(!addend || ((ldadd & 3) == 0 && lpf_aligned16(addend) && ldadd >= N1 && (N1 & 3) == 0));
    const bool two = w2_packed != nullptr;
    const bool dot = two && N2 == 1;
    LPF_REQUIRE(!two || (b2 && lpf_aligned16(w2_packed) && N2 > 0));
    LPF_REQUIRE(dot ? (out || prob) : (out && (ldo & 3) == 0 && lpf_aligned16(out)));
    LPF_REQUIRE(dot || ((two ? N2 : N1) & 3) == 0);
    LPF_REQUIRE(dot || ldo >= (two ? N2 : N1));
    const int nt1 = (N1 + 15) / 16, nt2 = (two && !dot) ? (N2 + 15) / 16 : 0;
    DenseChainArgs a{M, in_mode, X, ldx, batch, batch_ld, K1, w1_packed, N1, b1, addend, ldadd, ln_g, ln_b, flags,
                     w2_packed, N2, b2, out, ldo, prob};
    hipStream_t s = static_cast<hipStream_t>(stream);
    // in_mode 1 (gather-multiply) is built for the square two-layer chains (elementwise_lin), in_mode 2 (gather-add)
    // for the single-layer ones (the q projection); plain rows for everything
#define DC_CASE(T1, T2, MODE) \
    if (nt1 == T1 && nt2 == T2 && in_mode == MODE) return dc_launch<T1, T2, MODE>(a, s)
#define DC_SINGLE(T1) DC_CASE(T1, 0, 0); DC_CASE(T1, 0, 2)
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
    DC_CASE(3, 2, 0);   // pairwise_lin: (D + counts) -> (D + counts) -> D
    DC_CASE(5, 4, 0);
    DC_CASE(9, 8, 0);
    DC_CASE(17, 16, 0);
#undef DC_CASE
#undef DC_SINGLE
#undef DC_SQUARE
    return LPF_ERR_UNSUPPORTED;
}
