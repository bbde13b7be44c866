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
// Weights are the A operand: host-packed in MFMA order (both layers: float4 (kc, c, sq, lane = 16q + i) holds
// W[16c + i][64kc + 16sq + 4q + 0..3]), staged through LDS in K chunks of 64 and shared by the 4 wavefronts of a
// workgroup; two or more workgroups per CU overlap one's staging barriers with the other's MFMAs (one wave alone issues
// fp32 MFMAs at half rate).  Tile counts are template constants (exact, no guards inside the MFMA
// streams); shapes without an instantiation return LPF_ERR_UNSUPPORTED and the host uses the unfused kernels.
#include "lpf_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DC_KC = 64;        // K chunk (input features per staged weight slab)
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

// One K chunk of one layer for one wavefront: acc[c] += sum over the chunk's k of W[16c+i][k] * B[k][j].
// lw: this workgroup's LDS slab, float4 element (c*4 + sq)*64 + lane; bv[4*sq + u]: the lane's B value of step (sq,u);
// nsq: step groups of the chunk that hold real k (4 except in a ragged last chunk).
template <int NT>
__device__ __forceinline__ void dc_chunk(f32x4 (&acc)[NT], const float4 *lw, int lane, const float (&bv)[16], int nsq) {
#pragma unroll
    for (int sq = 0; sq < 4; ++sq) {
        if (sq < nsq) {
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                const float4 a = lw[(c * 4 + sq) * 64 + lane];
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bv[4 * sq + 0], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bv[4 * sq + 1], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bv[4 * sq + 2], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bv[4 * sq + 3], acc[c], 0, 0, 0);
            }
        }
    }
}

__device__ __forceinline__ float dc_quad_sum(float v) {  // sum over the 4 lanes (q = 0..3) that share a sample
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

template <int NT>
__device__ __forceinline__ void dc_stage(float4 *slab, const float *packed, int kc, int tid) {
    const float4 *src = reinterpret_cast<const float4 *>(packed) + (int64_t)kc * NT * 256;
#pragma unroll
    for (int e = 0; e < NT * 256 / (64 * DC_WAVES); ++e) slab[e * 64 * DC_WAVES + tid] = src[e * 64 * DC_WAVES + tid];
}

template <int NT1, int NT2>
__global__ __launch_bounds__(64 * DC_WAVES) void dense_chain_kernel(const DenseChainArgs A) {
    extern __shared__ __attribute__((aligned(16))) float4 slab[];  // max(NT1, NT2) * 4 * 64 float4
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int nkc1 = (A.K1 + DC_KC - 1) / DC_KC;
    constexpr int NKC2 = (NT1 + 3) / 4;  // hidden features in chunks of 64 (= 4 tiles)

#pragma unroll 1
    for (int64_t m0 = (int64_t)blockIdx.x * (16 * DC_WAVES); m0 < A.M; m0 += (int64_t)gridDim.x * (16 * DC_WAVES)) {
        const int64_t m = m0 + wave * 16 + j;
        const bool live = m < A.M;
        const int64_t mm = live ? m : A.M - 1;  // clamp: dead lanes compute on a valid row and store nothing
        int64_t ra = mm, rb = 0;
        if (A.in_mode != 0) {
            ra = A.batch[mm];
            rb = A.batch[A.batch_ld + mm];
        }
        const float *xa = A.X + ra * A.ldx, *xb = A.X + rb * A.ldx;

        // ---------------- layer 1
        f32x4 acc1[NT1];
#pragma unroll
        for (int c = 0; c < NT1; ++c) acc1[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int kc = 0; kc < nkc1; ++kc) {
            // this lane's 16 input values of the chunk: step group sq reads k = 64 kc + 16 sq + 4 q + (0..3)
            float bv[16];
#pragma unroll
            for (int sq = 0; sq < 4; ++sq) {
                const int k = kc * DC_KC + 16 * sq + 4 * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < A.K1) {  // K1 % 4 == 0: a float4 is all-in or all-out
                    v = *reinterpret_cast<const float4 *>(xa + k);
                    if (A.in_mode != 0) {
                        const float4 w = *reinterpret_cast<const float4 *>(xb + k);
                        if (A.in_mode == 1) { v.x *= w.x; v.y *= w.y; v.z *= w.z; v.w *= w.w; }
                        else { v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
                    }
                }
                bv[4 * sq + 0] = v.x; bv[4 * sq + 1] = v.y; bv[4 * sq + 2] = v.z; bv[4 * sq + 3] = v.w;
            }
            __syncthreads();  // the slab is free (previous chunk consumed by every wave)
            dc_stage<NT1>(slab, A.w1p, kc, tid);
            __syncthreads();
            const int rem = A.K1 - kc * DC_KC;
            dc_chunk<NT1>(acc1, slab, lane, bv, rem >= DC_KC ? 4 : (rem + 15) >> 4);
        }
        // epilogue 1: bias (+ addend) -> LayerNorm over the N1 real features -> ReLU
        float s1 = 0.f;
#pragma unroll
        for (int c = 0; c < NT1; ++c) {
            const int f0 = 16 * c + 4 * q;
            const float4 b = *reinterpret_cast<const float4 *>(A.b1 + f0);  // padded with zeros by the host
            acc1[c][0] += b.x; acc1[c][1] += b.y; acc1[c][2] += b.z; acc1[c][3] += b.w;
            if (A.addend && f0 < A.N1) {
                const float4 ad = *reinterpret_cast<const float4 *>(A.addend + mm * A.ldadd + f0);
                acc1[c][0] += ad.x; acc1[c][1] += ad.y; acc1[c][2] += ad.z; acc1[c][3] += ad.w;
            }
            s1 += acc1[c][0] + acc1[c][1] + acc1[c][2] + acc1[c][3];
        }
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
            // ---------------- layer 2: the B operand of step (hidden tile t, register r) is acc1[t][r] itself
            f32x4 acc2[NT2];
#pragma unroll
            for (int c = 0; c < NT2; ++c) acc2[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < NKC2; ++kc) {
                float bv[16];
#pragma unroll
                for (int sq = 0; sq < 4; ++sq) {
                    f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (4 * kc + sq < NT1) t = acc1[4 * kc + sq < NT1 ? 4 * kc + sq : 0];
                    bv[4 * sq + 0] = t[0]; bv[4 * sq + 1] = t[1]; bv[4 * sq + 2] = t[2]; bv[4 * sq + 3] = t[3];
                }
                __syncthreads();
                dc_stage<NT2>(slab, A.w2p, kc, tid);
                __syncthreads();
                dc_chunk<NT2>(acc2, slab, lane, bv, NT1 - 4 * kc >= 4 ? 4 : NT1 - 4 * kc);
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

template <int NT1, int NT2>
int dc_launch(const DenseChainArgs &a, hipStream_t s) {
    constexpr int slab_tiles = NT1 > NT2 ? NT1 : NT2;
    constexpr size_t lds = (size_t)slab_tiles * 256 * sizeof(float4);
    auto kern = dense_chain_kernel<NT1, NT2>;
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
#define DC_CASE(T1, T2) \
    if (nt1 == T1 && nt2 == T2) return dc_launch<T1, T2>(a, s)
    DC_CASE(2, 0);   // hidden 32
    DC_CASE(4, 0);   // hidden 64: q projection, attention output, score head
    DC_CASE(8, 0);   // hidden 128
    DC_CASE(16, 0);  // hidden 256
    DC_CASE(32, 0);  // hidden 512 (score head of a 256-wide model)
    DC_CASE(2, 2);
    DC_CASE(4, 4);   // elementwise_lin 64 -> 64 -> 64
    DC_CASE(8, 8);
    DC_CASE(16, 16);
    DC_CASE(3, 2);   // pairwise_lin: (D + counts) -> (D + counts) -> D
    DC_CASE(5, 4);
    DC_CASE(9, 8);
    DC_CASE(17, 16);
#undef DC_CASE
    return LPF_ERR_UNSUPPORTED;
}
