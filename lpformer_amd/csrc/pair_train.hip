// Training-step kernels of the pair stage: positional-encoding hidden layer, per-pair attention with the state a
// backward pass needs, and their gradients.
//
// Reference: the forward of get_pos_encodings (src/models/link_transformer.py:182-211) and LinkAttention.message +
// PyG softmax + scatter-sum (src/modules/layers.py:193-224) as run by the training step (src/train/train_model.py:59-77);
// the gradients are what torch autograd derives from that forward.  Algebra (unfused here):
//     h_e   = ReLU(LN_t(W1_t [pa,pb] + b1_t)) + ReLU(LN_t(W1_t [pb,pa] + b1_t))                (pe_hidden_*)
//     kp_e  = Wfold_t h_e + bfold_t                                   (lpf_gemm_f32 over the entries of type t)
//     k_e   = Z[v_e] + kp_e ;  s_e = att . leaky_relu(k_e * q_p, 0.2)
//     alpha = softmax of s over the entries of pair p (max-shifted, denominator + 1e-16) ;  o_p = sum alpha_e k_e + bias
// The inference kernels (pair_fused.hip) never materialise h, kp or s; a backward pass needs them, so the training
// forward writes H and KP (entry-major [n, D]) and the raw scores, and the backward recomputes k_e from Z and KP.
// Backward of the attention, per pair p with upstream gradient do_p (c_p = do_p . (o_p - bias) = sum_e alpha_e dalpha_e):
//     dalpha_e = do_p . k_e ;  ds_e = alpha_e (dalpha_e - c_p)
//     dk_e = alpha_e do_p + ds_e att * lrelu'(k_e q_p) * q_p          -> dK[e] (entry-major); dZ[v] = sum of dk_e over node v's entries
//     dq_p = sum_e ds_e att * lrelu'(k_e q_p) * k_e ;  datt = sum ds_e lrelu(k_e q_p) ;  dbias = sum_p do_p
// then dH_t = dK_t Wfold_t, dWfold_t = dK_t^T H_t (lpf_gemm_f32 / lpf_gemm_tn_f32), dbfold_t = column sums of dK_t, and
// the LayerNorm / first-layer gradients of the hidden layer (pe_hidden_bwd).
//
// Layout: entries sorted by (type, pair): e_pair / e_node / e_pa / e_pb [n], seg[3][bs+1] = first entry of pair p's
// type-t segment (global entry indices; seg[t][bs] = end of type t).  G = D/4 lanes own one entry row (pe_hidden) or
// one pair (attention), 16 bytes per lane and access.  Column sums over entries / pairs go through per-block partials
// that a second small kernel adds in block order; dZ is summed run by run of the entries sorted by node
// (segment_sum_kernel; float atomics only when the caller passes a dZ to the backward kernel itself): every gradient of
// this stage is deterministic.  (lpf_pair_scatter_add_f32, the endpoint scatter with atomics, is what a caller
// without a sorted endpoint list gets; the training step sums by node there too.)
#include "lpf_common.h"

namespace {

constexpr int PT_THREADS = 256;
#ifndef LPF_PT_U
#define LPF_PT_U 4
#endif
constexpr int PT_U = LPF_PT_U;      // entries of a pair requested per trip of the attention kernels

template <int G>
__device__ __forceinline__ float4 pt_ln_relu(const float4 u, const float4 g, const float4 b, int D, float &mean,
                                             float &rstd) {
    mean = lpf_group_sum<G>(u.x + u.y + u.z + u.w) / (float)D;
    const float4 d = make_float4(u.x - mean, u.y - mean, u.z - mean, u.w - mean);
    rstd = 1.0f / sqrtf(lpf_group_sum<G>(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) / (float)D + 1e-5f);
    return make_float4(fmaxf(d.x * rstd * g.x + b.x, 0.f), fmaxf(d.y * rstd * g.y + b.y, 0.f),
                       fmaxf(d.z * rstd * g.z + b.z, 0.f), fmaxf(d.w * rstd * g.w + b.w, 0.f));
}

struct PeParams {   // one type's first layer: w1 [D][2] row-major, b1 / gamma / beta [D]
    const float *w1, *b1, *gam, *bet;
};

// ------------------------------------------------------------------------------------------- pe hidden, forward
template <int G>
__global__ __launch_bounds__(PT_THREADS) void pe_hidden_fwd_kernel(int64_t lo, int64_t hi, int D, PeParams P,
                                                                   const float *__restrict__ pa,
                                                                   const float *__restrict__ pb, float *__restrict__ H,
                                                                   int64_t ldh) {
    constexpr int EPW = 64 / G;
    const int lane = threadIdx.x & 63, grp = lane / G, off = 4 * (lane % G);
    const int64_t wave_id = (int64_t)blockIdx.x * (PT_THREADS / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (PT_THREADS / 64);
    float4 w0, w1;
    {   // this lane's four features of the first layer
        const float4 a = *reinterpret_cast<const float4 *>(P.w1 + 2 * off), b = *reinterpret_cast<const float4 *>(P.w1 + 2 * off + 4);
        w0 = make_float4(a.x, a.z, b.x, b.z);
        w1 = make_float4(a.y, a.w, b.y, b.w);
    }
    const float4 bb = *reinterpret_cast<const float4 *>(P.b1 + off);
    const float4 gg = *reinterpret_cast<const float4 *>(P.gam + off), be = *reinterpret_cast<const float4 *>(P.bet + off);
    for (int64_t e0 = lo + wave_id * EPW; e0 < hi; e0 += n_waves * EPW) {
        const int64_t e = e0 + grp;
        const bool live = e < hi;
        const float x = live ? pa[e] : 0.f, y = live ? pb[e] : 0.f;
        float m, r;
        const float4 uab = make_float4(fmaf(w0.x, x, fmaf(w1.x, y, bb.x)), fmaf(w0.y, x, fmaf(w1.y, y, bb.y)),
                                       fmaf(w0.z, x, fmaf(w1.z, y, bb.z)), fmaf(w0.w, x, fmaf(w1.w, y, bb.w)));
        const float4 uba = make_float4(fmaf(w0.x, y, fmaf(w1.x, x, bb.x)), fmaf(w0.y, y, fmaf(w1.y, x, bb.y)),
                                       fmaf(w0.z, y, fmaf(w1.z, x, bb.z)), fmaf(w0.w, y, fmaf(w1.w, x, bb.w)));
        const float4 h1 = pt_ln_relu<G>(uab, gg, be, D, m, r), h2 = pt_ln_relu<G>(uba, gg, be, D, m, r);
        if (live)
            *reinterpret_cast<float4 *>(H + e * ldh + off) = make_float4(h1.x + h2.x, h1.y + h2.y, h1.z + h2.z, h1.w + h2.w);
    }
}

// ------------------------------------------------------------------------------------------- pe hidden, backward
// partial[block][5][D]: dw1[:,0], dw1[:,1], db1, dgamma, dbeta summed over the block's entries
template <int G>
__global__ __launch_bounds__(PT_THREADS) void pe_hidden_bwd_kernel(int64_t lo, int64_t hi, int D, PeParams P,
                                                                   const float *__restrict__ pa,
                                                                   const float *__restrict__ pb,
                                                                   const float *__restrict__ dH, int64_t ldh,
                                                                   float *__restrict__ partial) {
    constexpr int EPW = 64 / G;
    __shared__ float red[PT_THREADS / 64][5][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane / G, off = 4 * (lane % G);
    const int64_t wave_id = (int64_t)blockIdx.x * (PT_THREADS / 64) + wave;
    const int64_t n_waves = (int64_t)gridDim.x * (PT_THREADS / 64);
    float4 w0, w1;
    {
        const float4 a = *reinterpret_cast<const float4 *>(P.w1 + 2 * off), b = *reinterpret_cast<const float4 *>(P.w1 + 2 * off + 4);
        w0 = make_float4(a.x, a.z, b.x, b.z);
        w1 = make_float4(a.y, a.w, b.y, b.w);
    }
    const float4 bb = *reinterpret_cast<const float4 *>(P.b1 + off);
    const float4 gg = *reinterpret_cast<const float4 *>(P.gam + off), be = *reinterpret_cast<const float4 *>(P.bet + off);
    float acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int64_t e0 = lo + wave_id * EPW; e0 < hi; e0 += n_waves * EPW) {
        const int64_t e = e0 + grp;
        const bool live = e < hi;
        const float x = live ? pa[e] : 0.f, y = live ? pb[e] : 0.f;
        const float4 dh4 = live ? *reinterpret_cast<const float4 *>(dH + e * ldh + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float dh[4] = {dh4.x, dh4.y, dh4.z, dh4.w};
        const float w0a[4] = {w0.x, w0.y, w0.z, w0.w}, w1a[4] = {w1.x, w1.y, w1.z, w1.w};
        const float ba[4] = {bb.x, bb.y, bb.z, bb.w}, ga[4] = {gg.x, gg.y, gg.z, gg.w}, bea[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
        for (int ord = 0; ord < 2; ++ord) {          // g([pa, pb]) and g([pb, pa])
            const float xx = ord ? y : x, yy = ord ? x : y;
            float u[4], s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { u[j] = fmaf(w0a[j], xx, fmaf(w1a[j], yy, ba[j])); s += u[j]; }
            const float mean = lpf_group_sum<G>(s) / (float)D;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { u[j] -= mean; q += u[j] * u[j]; }
            const float rstd = 1.0f / sqrtf(lpf_group_sum<G>(q) / (float)D + 1e-5f);
            float dxh[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = u[j] * rstd;                       // normalised value
                const float dy = (xh * ga[j] + bea[j] > 0.f) ? dh[j] : 0.f;  // ReLU
                acc[3][j] += dy * xh;                               // dgamma
                acc[4][j] += dy;                                    // dbeta
                dxh[j] = dy * ga[j];
                u[j] = xh;
                s1 += dxh[j];
                s2 += dxh[j] * xh;
            }
            s1 = lpf_group_sum<G>(s1) / (float)D;
            s2 = lpf_group_sum<G>(s2) / (float)D;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float du = rstd * (dxh[j] - s1 - u[j] * s2);  // LayerNorm backward
                acc[0][j] += du * xx;
                acc[1][j] += du * yy;
                acc[2][j] += du;
            }
        }
    }
    // groups of a wave, then waves of the block, in a fixed order
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[i][j];
#pragma unroll
            for (int m = G; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
            if (grp == 0) red[wave][i][off + j] = v;
        }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 5 * D; idx += PT_THREADS) {
        const int i = idx / D, c = idx % D;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < PT_THREADS / 64; ++w) v += red[w][i][c];
        partial[((int64_t)blockIdx.x * 5 + i) * D + c] = v;
    }
}

// out[i][c] = sum over blocks (in block order) of partial[block][i][c]
__global__ __launch_bounds__(256) void partial_sum_kernel(int64_t n_blocks, int64_t width, const float *__restrict__ partial,
                                                          float *__restrict__ out) {
    // 64 columns per workgroup, four row strips of the partials each (fixed strip -> fixed order of additions)
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, strip = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + cl;
    float v = 0.f;
    if (c < width)
        for (int64_t b = strip; b < n_blocks; b += 4) v += partial[b * width + c];
    red[strip][cl] = v;
    __syncthreads();
    if (strip == 0 && c < width) out[c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// partial[block][c] = sum of the block's rows of x[:, c]  (block b owns rows b, b + n_blocks, ... by waves)
__global__ __launch_bounds__(PT_THREADS) void colsum_partial_kernel(int64_t M, int D, const float *__restrict__ x,
                                                                    int64_t ldx, float *__restrict__ partial) {
    __shared__ float red[PT_THREADS / 64][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wave_id = (int64_t)blockIdx.x * (PT_THREADS / 64) + wave;
    const int64_t n_waves = (int64_t)gridDim.x * (PT_THREADS / 64);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    const int off = 4 * lane;
    if (off < D)
        for (int64_t r = wave_id; r < M; r += n_waves) {
            const float4 v = *reinterpret_cast<const float4 *>(x + r * ldx + off);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    if (off < D) { red[wave][off] = a.x; red[wave][off + 1] = a.y; red[wave][off + 2] = a.z; red[wave][off + 3] = a.w; }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += PT_THREADS) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < PT_THREADS / 64; ++w) v += red[w][c];
        partial[(int64_t)blockIdx.x * D + c] = v;
    }
}

// ------------------------------------------------------------------------------------------- attention, forward
struct AttnArgs {
    int64_t bs, n;
    int D;
    const int64_t *seg;        // [3][bs+1]
    const int32_t *e_node;     // [n]
    const float *Z; int64_t ldz;
    const float *KP; int64_t ldk;
    const float *q; int64_t ldq;
    const float *att, *bias;
    float *out; int64_t ldo;   // [bs][D]: sum alpha k + bias
    float *score;              // [n] raw scores s_e
    float *pmax, *pinv;        // [bs]: segment max, 1 / (sum exp + 1e-16)   (0 for a pair without entries)
    // backward only
    const float *dout; int64_t lddo;
    float *dK; int64_t lddk;   // [n][D]
    float *dZ; int64_t lddz;   // [N][D], accumulated with float atomics (zeroed by the caller)
    float *dq; int64_t lddq;   // [bs][D]
    float *partial;            // [blocks][2][D]: datt, dbias
};

__device__ __forceinline__ float pt_dot4(const float4 a, const float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float pt_lrelu(float x) { return fmaxf(x, 0.2f * x); }

template <int G>
__global__ __launch_bounds__(PT_THREADS) void pair_attn_train_fwd_kernel(const AttnArgs A) {
    constexpr int PPW = 64 / G;
    const int lane = threadIdx.x & 63, grp = lane / G, off = 4 * (lane % G);
    const int64_t wave_id = (int64_t)blockIdx.x * (PT_THREADS / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (PT_THREADS / 64);
    const float4 at = *reinterpret_cast<const float4 *>(A.att + off), bi = *reinterpret_cast<const float4 *>(A.bias + off);
    for (int64_t p0 = wave_id * PPW; p0 < A.bs; p0 += n_waves * PPW) {
        const int64_t p = p0 + grp;
        const bool live = p < A.bs;
        const int64_t pp = live ? p : A.bs - 1;
        const float4 qv = *reinterpret_cast<const float4 *>(A.q + pp * A.ldq + off);
        float m = -INFINITY, l = 0.f;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        // the three type segments of the pair, one entry at a time (the groups of a wave run different trip counts:
        // the loop bound is taken over the wave so that the group reductions stay convergent)
#pragma unroll 1
        for (int t = 0; t < 3; ++t) {
            const int64_t lo = A.seg[(int64_t)t * (A.bs + 1) + pp], hi = live ? A.seg[(int64_t)t * (A.bs + 1) + pp + 1] : lo;
            int64_t cnt = hi - lo, cmax = cnt;
#pragma unroll
            for (int d = G; d < 64; d <<= 1) { const int64_t o2 = __shfl_xor(cmax, d, 64); cmax = o2 > cmax ? o2 : cmax; }
            // PT_U entries per trip: their node ids, then their Z and KP rows, are all requested before the first one is
            // used (one entry per trip was a chain of two dependent trips to memory per entry: 155 us for the 92 k
            // entries of a collab-like training batch); the arithmetic runs entry by entry in the same order as before
            for (int64_t i0 = 0; i0 < cmax; i0 += PT_U) {
                int32_t v[PT_U];
                float4 z[PT_U], kp[PT_U];
#pragma unroll
                for (int u = 0; u < PT_U; ++u) v[u] = i0 + u < cnt ? A.e_node[lo + i0 + u] : 0;
#pragma unroll
                for (int u = 0; u < PT_U; ++u) {
                    const int64_t e = i0 + u < cnt ? lo + i0 + u : 0;
                    z[u] = *reinterpret_cast<const float4 *>(A.Z + (int64_t)v[u] * A.ldz + off);
                    kp[u] = *reinterpret_cast<const float4 *>(A.KP + e * A.ldk + off);
                }
#pragma unroll
                for (int u = 0; u < PT_U; ++u) {
                    if (i0 + u >= cmax) break;          // (uniform over the wave)
                    const bool on = i0 + u < cnt;
                    const int64_t e = lo + i0 + u;
                    const float4 k = make_float4(z[u].x + kp[u].x, z[u].y + kp[u].y, z[u].z + kp[u].z, z[u].w + kp[u].w);
                    const float s = lpf_group_sum<G>(at.x * pt_lrelu(k.x * qv.x) + at.y * pt_lrelu(k.y * qv.y) +
                                                     at.z * pt_lrelu(k.z * qv.z) + at.w * pt_lrelu(k.w * qv.w));
                    if (on) {
                        if (off == 0) A.score[e] = s;
                        const float mn = fmaxf(m, s);
                        const float sa = __expf(m - mn), w = __expf(s - mn);
                        l = fmaf(l, sa, w);
                        o = make_float4(fmaf(o.x, sa, w * k.x), fmaf(o.y, sa, w * k.y), fmaf(o.z, sa, w * k.z), fmaf(o.w, sa, w * k.w));
                        m = mn;
                    }
                }
            }
        }
        if (live) {
            const float inv = l > 0.f ? 1.0f / (l + 1e-16f) : 0.f;
            *reinterpret_cast<float4 *>(A.out + p * A.ldo + off) =
                make_float4(fmaf(o.x, inv, bi.x), fmaf(o.y, inv, bi.y), fmaf(o.z, inv, bi.z), fmaf(o.w, inv, bi.w));
            if (off == 0) { A.pmax[p] = m; A.pinv[p] = inv; }
        }
    }
}

// ------------------------------------------------------------------------------------------- attention, backward
template <int G>
__global__ __launch_bounds__(PT_THREADS) void pair_attn_train_bwd_kernel(const AttnArgs A) {
    constexpr int PPW = 64 / G;
    __shared__ float red[PT_THREADS / 64][2][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane / G, off = 4 * (lane % G);
    const int64_t wave_id = (int64_t)blockIdx.x * (PT_THREADS / 64) + wave;
    const int64_t n_waves = (int64_t)gridDim.x * (PT_THREADS / 64);
    const float4 at = *reinterpret_cast<const float4 *>(A.att + off), bi = *reinterpret_cast<const float4 *>(A.bias + off);
    float4 datt = make_float4(0.f, 0.f, 0.f, 0.f), dbias = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t p0 = wave_id * PPW; p0 < A.bs; p0 += n_waves * PPW) {
        const int64_t p = p0 + grp;
        const bool live = p < A.bs;
        const int64_t pp = live ? p : A.bs - 1;
        const float4 qv = *reinterpret_cast<const float4 *>(A.q + pp * A.ldq + off);
        float4 go = *reinterpret_cast<const float4 *>(A.dout + pp * A.lddo + off);
        if (!live) go = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 ov = *reinterpret_cast<const float4 *>(A.out + pp * A.ldo + off);
        const float pm = A.pmax[pp], pinv = A.pinv[pp];
        // c = do . (o - bias) = sum_e alpha_e (do . k_e)
        const float c = lpf_group_sum<G>(go.x * (ov.x - bi.x) + go.y * (ov.y - bi.y) + go.z * (ov.z - bi.z) + go.w * (ov.w - bi.w));
        dbias.x += go.x; dbias.y += go.y; dbias.z += go.z; dbias.w += go.w;
        float4 gq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
        for (int t = 0; t < 3; ++t) {
            const int64_t lo = A.seg[(int64_t)t * (A.bs + 1) + pp], hi = live ? A.seg[(int64_t)t * (A.bs + 1) + pp + 1] : lo;
            int64_t cnt = hi - lo, cmax = cnt;
#pragma unroll
            for (int d = G; d < 64; d <<= 1) { const int64_t o2 = __shfl_xor(cmax, d, 64); cmax = o2 > cmax ? o2 : cmax; }
            for (int64_t i0 = 0; i0 < cmax; i0 += PT_U) {       // (PT_U entries requested per trip, as in the forward)
                int32_t vv[PT_U];
                float4 zz[PT_U], kpp[PT_U];
                float sc[PT_U];
#pragma unroll
                for (int u = 0; u < PT_U; ++u) {
                    const bool in = i0 + u < cnt;
                    vv[u] = in ? A.e_node[lo + i0 + u] : 0;
                    sc[u] = in ? A.score[lo + i0 + u] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < PT_U; ++u) {
                    const int64_t e = i0 + u < cnt ? lo + i0 + u : 0;
                    zz[u] = *reinterpret_cast<const float4 *>(A.Z + (int64_t)vv[u] * A.ldz + off);
                    kpp[u] = *reinterpret_cast<const float4 *>(A.KP + e * A.ldk + off);
                }
#pragma unroll
                for (int u = 0; u < PT_U; ++u) {
                if (i0 + u >= cmax) break;              // (uniform over the wave)
                const bool on = i0 + u < cnt;
                const int64_t e = lo + i0 + u;
                const int32_t v = vv[u];
                const float4 z = zz[u], kp = kpp[u];
                const float4 k = make_float4(z.x + kp.x, z.y + kp.y, z.z + kp.z, z.w + kp.w);
                const float dalpha = lpf_group_sum<G>(pt_dot4(go, k));
                if (on) {
                    const float alpha = __expf(sc[u] - pm) * pinv;
                    const float ds = alpha * (dalpha - c);
                    const float xk[4] = {k.x * qv.x, k.y * qv.y, k.z * qv.z, k.w * qv.w};
                    const float ka[4] = {k.x, k.y, k.z, k.w}, qa[4] = {qv.x, qv.y, qv.z, qv.w};
                    const float aa[4] = {at.x, at.y, at.z, at.w}, ga[4] = {go.x, go.y, go.z, go.w};
                    float dk[4], dqv[4], da[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float slope = xk[j] > 0.f ? 1.0f : 0.2f;
                        const float g1 = ds * aa[j] * slope;
                        dk[j] = fmaf(alpha, ga[j], g1 * qa[j]);
                        dqv[j] = g1 * ka[j];
                        da[j] = ds * pt_lrelu(xk[j]);
                    }
                    gq.x += dqv[0]; gq.y += dqv[1]; gq.z += dqv[2]; gq.w += dqv[3];
                    datt.x += da[0]; datt.y += da[1]; datt.z += da[2]; datt.w += da[3];
                    *reinterpret_cast<float4 *>(A.dK + e * A.lddk + off) = make_float4(dk[0], dk[1], dk[2], dk[3]);
                    if (A.dZ) {   // (dZ = nullptr: the caller sums dK by node itself, lpf_segment_rows_sum_f32)
                        float *dz = A.dZ + (int64_t)v * A.lddz + off;
                        unsafeAtomicAdd(dz + 0, dk[0]); unsafeAtomicAdd(dz + 1, dk[1]);
                        unsafeAtomicAdd(dz + 2, dk[2]); unsafeAtomicAdd(dz + 3, dk[3]);
                    }
                }
                }
            }
        }
        if (live) *reinterpret_cast<float4 *>(A.dq + p * A.lddq + off) = gq;
    }
    // datt / dbias: groups of the wave, waves of the block, then one partial row per block
    const float vals[2][4] = {{datt.x, datt.y, datt.z, datt.w}, {dbias.x, dbias.y, dbias.z, dbias.w}};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = vals[i][j];
#pragma unroll
            for (int m = G; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
            if (grp == 0) red[wave][i][off + j] = v;
        }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 2 * A.D; idx += PT_THREADS) {
        const int i = idx / A.D, c2 = idx % A.D;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < PT_THREADS / 64; ++w) v += red[w][i][c2];
        A.partial[((int64_t)blockIdx.x * 2 + i) * A.D + c2] = v;
    }
}

// ------------------------------------------------------------------------------------------- endpoint scatter-add
// Gradient of lpf_pair_gather_f32: dX[a] += dsum + dmul * X[b], dX[b] += dsum + dmul * X[a]   (float atomics)
template <int G>
__global__ __launch_bounds__(PT_THREADS) void pair_scatter_kernel(int64_t bs, int D, const int64_t *__restrict__ batch,
                                                                  int64_t batch_ld, int64_t n_rows,
                                                                  const float *__restrict__ X, int64_t ldx,
                                                                  const float *__restrict__ dmul, int64_t ldm,
                                                                  const float *__restrict__ dsum, int64_t lds,
                                                                  float *__restrict__ dX, int64_t lddx) {
    constexpr int RPW = 64 / G;
    const int lane = threadIdx.x & 63, grp = lane / G, off = 4 * (lane % G);
    const int64_t wave_id = (int64_t)blockIdx.x * (PT_THREADS / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (PT_THREADS / 64);
    for (int64_t p = wave_id * RPW + grp; p < bs; p += n_waves * RPW) {
        int64_t a = batch[p], b = batch[batch_ld + p];
        if ((uint64_t)a >= (uint64_t)n_rows || (uint64_t)b >= (uint64_t)n_rows) continue;  // (flagged by the selection)
        float4 ga = make_float4(0.f, 0.f, 0.f, 0.f), gb = ga;
        if (dsum) { ga = *reinterpret_cast<const float4 *>(dsum + p * lds + off); gb = ga; }
        if (dmul) {
            const float4 g = *reinterpret_cast<const float4 *>(dmul + p * ldm + off);
            const float4 xa = *reinterpret_cast<const float4 *>(X + a * ldx + off), xb = *reinterpret_cast<const float4 *>(X + b * ldx + off);
            ga.x += g.x * xb.x; ga.y += g.y * xb.y; ga.z += g.z * xb.z; ga.w += g.w * xb.w;
            gb.x += g.x * xa.x; gb.y += g.y * xa.y; gb.z += g.z * xa.z; gb.w += g.w * xa.w;
        }
        float *da = dX + a * lddx + off, *db = dX + b * lddx + off;
        unsafeAtomicAdd(da + 0, ga.x); unsafeAtomicAdd(da + 1, ga.y); unsafeAtomicAdd(da + 2, ga.z); unsafeAtomicAdd(da + 3, ga.w);
        unsafeAtomicAdd(db + 0, gb.x); unsafeAtomicAdd(db + 1, gb.y); unsafeAtomicAdd(db + 2, gb.z); unsafeAtomicAdd(db + 3, gb.w);
    }
}

// ------------------------------------------------------------------------------------------- rows summed by key
// dst[key] = sum of src[order[j]] over the run of equal keys in the SORTED key list (j ascending: deterministic) -- the
// gradient of a row gather, Z[node_e] -> dZ[v] = sum_{e: node_e = v} dK[e], without atomics: 4 D float atomics per entry
// were 210 of the attention backward's 375 us on a collab-like training batch.  One lane group per position of the sorted
// list; a position that does not start a run leaves at once, the head of a run walks it, four rows requested per trip.
template <int G>
__global__ __launch_bounds__(PT_THREADS) void segment_sum_kernel(int64_t n, const int32_t *__restrict__ keys,
                                                                 const int64_t *__restrict__ order,
                                                                 const float *__restrict__ src, int64_t lds,
                                                                 float *__restrict__ dst, int64_t ldd) {
    const int lane = threadIdx.x & 63, grp = lane / G, off = 4 * (lane % G);
    const int64_t i = ((int64_t)blockIdx.x * (PT_THREADS / 64) + (threadIdx.x >> 6)) * (64 / G) + grp;
    if (i >= n) return;
    const int32_t key = keys[i];
    if (i > 0 && keys[i - 1] == key) return;       // not the head of its run
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t j0 = i; j0 < n && keys[j0] == key; j0 += 4) {
        float4 v[4];
        bool in[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            in[u] = j0 + u < n && keys[j0 + u] == key;
            v[u] = in[u] ? *reinterpret_cast<const float4 *>(src + order[j0 + u] * lds + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (in[u]) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        if (!in[3]) break;
    }
    *reinterpret_cast<float4 *>(dst + (int64_t)key * ldd + off) = acc;
}

inline int pt_blocks(int64_t units, int per_block) {
    int64_t b = (units + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 512) b = 512;   // two resident workgroups per CU; the final pass adds the partial rows serially
    return (int)b;
}

#define PT_DISPATCH(D, CALL)                       \
    switch (D) {                                   \
        case 32: { constexpr int GG = 8; CALL; } break;   \
        case 64: { constexpr int GG = 16; CALL; } break;  \
        case 128: { constexpr int GG = 32; CALL; } break; \
        case 256: { constexpr int GG = 64; CALL; } break; \
        default: return LPF_ERR_UNSUPPORTED;       \
    }

}  // namespace

extern "C" int64_t lpf_train_partial_blocks(int64_t units) { (void)units; return 512; }  // the launch cap of pt_blocks

extern "C" int lpf_pe_hidden_fwd_f32(int64_t n_entries, int32_t D, const float *w1, const float *b1, const float *gamma,
                                     const float *beta, const float *pa, const float *pb, float *H, int64_t ldh,
                                     void *stream) {
    if (n_entries == 0) return LPF_OK;
    LPF_REQUIRE(n_entries > 0 && w1 && b1 && gamma && beta && pa && pb && H && ldh >= D && (ldh & 3) == 0 &&
                lpf_aligned16(w1) && lpf_aligned16(b1) && lpf_aligned16(gamma) && lpf_aligned16(beta) && lpf_aligned16(H));
    const PeParams P{w1, b1, gamma, beta};
    hipStream_t s = static_cast<hipStream_t>(stream);
    PT_DISPATCH(D, hipLaunchKernelGGL(pe_hidden_fwd_kernel<GG>, dim3(pt_blocks(n_entries, 64)), dim3(PT_THREADS), 0, s,
                                      (int64_t)0, n_entries, (int)D, P, pa, pb, H, ldh));
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_pe_hidden_bwd_f32(int64_t n_entries, int32_t D, const float *w1, const float *b1, const float *gamma,
                                     const float *beta, const float *pa, const float *pb, const float *dH, int64_t ldh,
                                     float *grads, float *workspace, void *stream) {
    LPF_REQUIRE(n_entries >= 0 && w1 && b1 && gamma && beta && grads && workspace && (n_entries == 0 || (pa && pb && dH)) &&
                ldh >= D && (ldh & 3) == 0 && lpf_aligned16(w1) && lpf_aligned16(b1) && lpf_aligned16(gamma) &&
                lpf_aligned16(beta) && (n_entries == 0 || lpf_aligned16(dH)));
    const PeParams P{w1, b1, gamma, beta};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nb = pt_blocks(n_entries, 64);
    PT_DISPATCH(D, hipLaunchKernelGGL(pe_hidden_bwd_kernel<GG>, dim3(nb), dim3(PT_THREADS), 0, s, (int64_t)0, n_entries,
                                      (int)D, P, pa, pb, dH, ldh, workspace));
    hipLaunchKernelGGL(partial_sum_kernel, dim3((5 * D + 63) / 64), dim3(256), 0, s, (int64_t)nb, (int64_t)5 * D,
                       workspace, grads);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_colsum_f32(int64_t M, int32_t D, const float *x, int64_t ldx, float *out, float *workspace,
                              void *stream) {
    LPF_REQUIRE(M >= 0 && D > 0 && D <= 256 && (D & 3) == 0 && out && workspace && (M == 0 || (x && lpf_aligned16(x))) &&
                ldx >= D && (ldx & 3) == 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nb = pt_blocks(M, 256);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb), dim3(PT_THREADS), 0, s, M, (int)D, x, ldx, workspace);
    hipLaunchKernelGGL(partial_sum_kernel, dim3((D + 63) / 64), dim3(256), 0, s, (int64_t)nb, (int64_t)D, workspace, out);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_pair_attention_train_fwd_f32(int64_t bs, int64_t n_entries, int32_t D, const int64_t *seg,
                                                const int32_t *e_node, const float *Z, int64_t ldz, const float *KP,
                                                int64_t ldk, const float *q, int64_t ldq, const float *att,
                                                const float *bias, float *out, int64_t ldo, float *score, float *pmax,
                                                float *pinv, void *stream) {
    if (bs == 0) return LPF_OK;
    LPF_REQUIRE(bs > 0 && n_entries >= 0 && seg && Z && q && att && bias && out && pmax && pinv &&
                (n_entries == 0 || (e_node && KP && score)) && ldz >= D && ldk >= D && ldq >= D && ldo >= D &&
                ((ldz | ldk | ldq | ldo) & 3) == 0 && lpf_aligned16(Z) && lpf_aligned16(q) && lpf_aligned16(att) &&
                lpf_aligned16(bias) && lpf_aligned16(out) && (n_entries == 0 || lpf_aligned16(KP)));
    AttnArgs a{};
    a.bs = bs; a.n = n_entries; a.D = D; a.seg = seg; a.e_node = e_node; a.Z = Z; a.ldz = ldz;
    a.KP = KP ? KP : Z; a.ldk = ldk; a.q = q; a.ldq = ldq; a.att = att; a.bias = bias; a.out = out; a.ldo = ldo;
    a.score = score; a.pmax = pmax; a.pinv = pinv;
    hipStream_t s = static_cast<hipStream_t>(stream);
    PT_DISPATCH(D, hipLaunchKernelGGL(pair_attn_train_fwd_kernel<GG>, dim3(pt_blocks(bs, 4 * (64 / GG))), dim3(PT_THREADS),
                                      0, s, a));
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_pair_attention_train_bwd_f32(int64_t bs, int64_t n_entries, int32_t D, const int64_t *seg,
                                                const int32_t *e_node, const float *Z, int64_t ldz, const float *KP,
                                                int64_t ldk, const float *q, int64_t ldq, const float *att,
                                                const float *bias, const float *out, int64_t ldo, const float *score,
                                                const float *pmax, const float *pinv, const float *dout, int64_t lddo,
                                                float *dK, int64_t lddk, float *dZ, int64_t lddz, float *dq, int64_t lddq,
                                                float *datt_dbias, float *workspace, void *stream) {
    if (bs == 0) return LPF_OK;
    LPF_REQUIRE(bs > 0 && n_entries >= 0 && seg && Z && q && att && bias && out && pmax && pinv && dout && dq &&
                datt_dbias && workspace && (n_entries == 0 || (e_node && KP && score && dK)) && ldz >= D && ldk >= D &&
                ldq >= D && ldo >= D && lddo >= D && lddk >= D && lddz >= D && lddq >= D &&
                ((ldz | ldk | ldq | ldo | lddo | lddk | lddz | lddq) & 3) == 0 && lpf_aligned16(Z) && lpf_aligned16(q) &&
                lpf_aligned16(att) && lpf_aligned16(bias) && lpf_aligned16(out) && lpf_aligned16(dout) &&
                (!dZ || lpf_aligned16(dZ)) && lpf_aligned16(dq) &&
                (n_entries == 0 || (lpf_aligned16(KP) && lpf_aligned16(dK))));
    AttnArgs a{};
    a.bs = bs; a.n = n_entries; a.D = D; a.seg = seg; a.e_node = e_node; a.Z = Z; a.ldz = ldz;
    a.KP = KP ? KP : Z; a.ldk = ldk; a.q = q; a.ldq = ldq; a.att = att; a.bias = bias;
    a.out = const_cast<float *>(out); a.ldo = ldo; a.score = const_cast<float *>(score);
    a.pmax = const_cast<float *>(pmax); a.pinv = const_cast<float *>(pinv);
    a.dout = dout; a.lddo = lddo; a.dK = dK; a.lddk = lddk; a.dZ = dZ; a.lddz = lddz; a.dq = dq; a.lddq = lddq;
    a.partial = workspace;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int nb = 1;
    PT_DISPATCH(D, { nb = pt_blocks(bs, 4 * (64 / GG));
                     hipLaunchKernelGGL(pair_attn_train_bwd_kernel<GG>, dim3(nb), dim3(PT_THREADS), 0, s, a); });
    hipLaunchKernelGGL(partial_sum_kernel, dim3((2 * D + 63) / 64), dim3(256), 0, s, (int64_t)nb, (int64_t)2 * D,
                       workspace, datt_dbias);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_pair_scatter_add_f32(int64_t bs, int32_t D, const int64_t *batch, int64_t batch_ld, int64_t n_rows,
                                        const float *X, int64_t ldx, const float *dmul, int64_t ldm, const float *dsum,
                                        int64_t lds, float *dX, int64_t lddx, void *stream) {
    if (bs == 0) return LPF_OK;
    LPF_REQUIRE(bs > 0 && batch && batch_ld >= bs && n_rows > 0 && dX && (dmul || dsum) && (!dmul || (X && ldm >= D)) &&
                (!dsum || lds >= D) && lddx >= D && ((ldx | ldm | lds | lddx) & 3) == 0 && lpf_aligned16(dX) &&
                (!X || lpf_aligned16(X)) && (!dmul || lpf_aligned16(dmul)) && (!dsum || lpf_aligned16(dsum)));
    hipStream_t s = static_cast<hipStream_t>(stream);
    PT_DISPATCH(D, hipLaunchKernelGGL(pair_scatter_kernel<GG>, dim3(pt_blocks(bs, 4 * (64 / GG))), dim3(PT_THREADS), 0, s,
                                      bs, (int)D, batch, batch_ld, n_rows, X, ldx, dmul, ldm, dsum, lds, dX, lddx));
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_segment_rows_sum_f32(int64_t n, int32_t D, const int32_t *keys_sorted, const int64_t *order,
                                        const float *src, int64_t lds, float *dst, int64_t ldd, void *stream) {
    if (n == 0) return LPF_OK;
    LPF_REQUIRE(n > 0 && keys_sorted && order && src && dst && lds >= D && ldd >= D && ((lds | ldd) & 3) == 0 &&
                lpf_aligned16(src) && lpf_aligned16(dst));
    hipStream_t s = static_cast<hipStream_t>(stream);
    PT_DISPATCH(D, {
        const int per_block = (PT_THREADS / 64) * (64 / GG);
        hipLaunchKernelGGL(segment_sum_kernel<GG>, dim3((unsigned)((n + per_block - 1) / per_block)), dim3(PT_THREADS), 0, s,
                           n, keys_sorted, order, src, lds, dst, ldd);
    });
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}
