// Row-wise helpers of the pair stage: LayerNorm(+ReLU), endpoint gather (product and sum), score dot + sigmoid.
// All are streaming kernels bound by HBM bandwidth; one wavefront (or a 16/32-lane group) owns a row, 16-byte
// accesses wherever the row length allows it, reductions by xor butterflies.
#include "lpf_common.h"

namespace {

constexpr int LN_MAX_PER_LANE = 16;  // D <= 1024

__global__ __launch_bounds__(256) void layernorm_kernel(int64_t M, int D, const float *__restrict__ x, int64_t ldx,
                                                        const float *__restrict__ g, const float *__restrict__ b,
                                                        float *__restrict__ y, int64_t ldy, uint32_t flags) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const bool relu = flags & LPF_FLAG_RELU;
    for (int64_t row = wave_id; row < M; row += n_waves) {
        float v[LN_MAX_PER_LANE];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAX_PER_LANE; ++j) {
            const int c = lane + 64 * j;
            v[j] = (c < D) ? x[row * ldx + c] : 0.f;
            s += v[j];
        }
        float mean = 0.f, rstd = 1.f;
        if (g) {
            mean = lpf_group_sum<64>(s) / (float)D;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < LN_MAX_PER_LANE; ++j) {
                const int c = lane + 64 * j;
                const float d = (c < D) ? v[j] - mean : 0.f;
                q += d * d;
            }
            rstd = 1.0f / sqrtf(lpf_group_sum<64>(q) / (float)D + 1e-5f);
        }
#pragma unroll
        for (int j = 0; j < LN_MAX_PER_LANE; ++j) {
            const int c = lane + 64 * j;
            if (c < D) {
                float o = g ? (v[j] - mean) * rstd * g[c] + b[c] : v[j];
                if (relu) o = fmaxf(o, 0.f);
                y[row * ldy + c] = o;
            }
        }
    }
}

template <int G>
__global__ __launch_bounds__(256) void pair_gather_kernel(int64_t bs, int D, const int64_t *__restrict__ batch,
                                                          int64_t batch_ld, int64_t n_rows,
                                                          const float *__restrict__ X, int64_t ldx,
                                                          float *__restrict__ mul, int64_t ldm,
                                                          float *__restrict__ sum, int64_t lds) {
    constexpr int RPW = 64 / G;
    const int lane = threadIdx.x & 63;
    const int grp = lane / G, off = 4 * (lane % G);
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t k0 = wave_id * RPW; k0 < bs; k0 += n_waves * RPW) {
        const int64_t k = k0 + grp;
        if (k >= bs || off >= D) continue;
        int64_t a = batch[k], b = batch[batch_ld + k];
        // ids outside the table read row 0 (the selection kernel reports them; nothing is read out of bounds)
        if ((uint64_t)a >= (uint64_t)n_rows) a = 0;
        if ((uint64_t)b >= (uint64_t)n_rows) b = 0;
        const float4 xa = *reinterpret_cast<const float4 *>(X + a * ldx + off);
        const float4 xb = *reinterpret_cast<const float4 *>(X + b * ldx + off);
        if (mul)
            *reinterpret_cast<float4 *>(mul + k * ldm + off) =
                make_float4(xa.x * xb.x, xa.y * xb.y, xa.z * xb.z, xa.w * xb.w);
        if (sum)
            *reinterpret_cast<float4 *>(sum + k * lds + off) =
                make_float4(xa.x + xb.x, xa.y + xb.y, xa.z + xb.z, xa.w + xb.w);
    }
}

__global__ __launch_bounds__(256) void rowdot_sigmoid_kernel(int64_t M, int K, const float *__restrict__ A,
                                                             int64_t lda, const float *__restrict__ w, float bias,
                                                             float *__restrict__ logit, float *__restrict__ prob) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t row = wave_id; row < M; row += n_waves) {
        float s = 0.f;
        for (int c = lane; c < K; c += 64) s = fmaf(A[row * lda + c], w[c], s);
        s = lpf_group_sum<64>(s) + bias;
        if (lane == 0) {
            if (logit) logit[row] = s;
            if (prob) prob[row] = 1.0f / (1.0f + expf(-s));
        }
    }
}

inline unsigned grid_for_rows(int64_t rows_per_block_units) {
    int64_t blocks = rows_per_block_units;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

extern "C" int lpf_layernorm_f32(int64_t M, int32_t D, const float *x, int64_t ldx, const float *g, const float *b,
                                 float *y, int64_t ldy, uint32_t flags, void *stream) {
    if (M == 0) return LPF_OK;
    LPF_REQUIRE(M > 0 && x && y && ldx >= D && ldy >= D && (!g) == (!b));
    if (D <= 0 || D > 64 * LN_MAX_PER_LANE) return LPF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(layernorm_kernel, dim3(grid_for_rows((M + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), M, D, x, ldx, g, b, y, ldy, flags);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_pair_gather_f32(int64_t bs, int32_t D, const int64_t *batch, int64_t batch_ld, int64_t n_rows,
                                   const float *X, int64_t ldx, float *mul, int64_t ldm, float *sum, int64_t lds,
                                   void *stream) {
    if (bs == 0) return LPF_OK;
    LPF_REQUIRE(bs > 0 && batch && X && batch_ld >= bs && n_rows > 0);
    if (D <= 0 || (D & 3) || D > 256) return LPF_ERR_UNSUPPORTED;
    LPF_REQUIRE((ldx & 3) == 0 && ldx >= D && lpf_aligned16(X));
    LPF_REQUIRE(!mul || ((ldm & 3) == 0 && ldm >= D && lpf_aligned16(mul)));
    LPF_REQUIRE(!sum || ((lds & 3) == 0 && lds >= D && lpf_aligned16(sum)));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int G = D <= 64 ? 16 : (D <= 128 ? 32 : 64);
    const unsigned grid = grid_for_rows((bs + 4 * (64 / G) - 1) / (4 * (64 / G)));
    if (G == 16)
        hipLaunchKernelGGL(pair_gather_kernel<16>, dim3(grid), dim3(256), 0, s, bs, D, batch, batch_ld, n_rows, X, ldx, mul,
                           ldm, sum, lds);
    else if (G == 32)
        hipLaunchKernelGGL(pair_gather_kernel<32>, dim3(grid), dim3(256), 0, s, bs, D, batch, batch_ld, n_rows, X, ldx, mul,
                           ldm, sum, lds);
    else
        hipLaunchKernelGGL(pair_gather_kernel<64>, dim3(grid), dim3(256), 0, s, bs, D, batch, batch_ld, n_rows, X, ldx, mul,
                           ldm, sum, lds);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_rowdot_sigmoid_f32(int64_t M, int32_t K, const float *A, int64_t lda, const float *w, float b,
                                      float *logit, float *prob, void *stream) {
    if (M == 0) return LPF_OK;
    LPF_REQUIRE(M > 0 && K > 0 && A && w && lda >= K && (logit || prob));
    hipLaunchKernelGGL(rowdot_sigmoid_kernel, dim3(grid_for_rows((M + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), M, K, A, lda, w, b, logit, prob);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward
// (the training step, src/train/train_model.py:59-77 through autograd; forward = lpf_layernorm_f32)
//   xhat = (x - mean) rstd;  g = dy * gamma;  dx = rstd (g - mean(g) - xhat mean(g xhat));
//   dgamma = sum_rows dy xhat;  dbeta = sum_rows dy
// Row statistics are recomputed from x (two reads of a row that is in registers anyway: nothing is saved by the
// forward).  G = D/4 lanes per row; a lane keeps the column sums of its four features over the rows it sees, the
// groups of a workgroup meet in LDS, every workgroup writes one partial row pair and a second kernel adds them in
// workgroup order (deterministic).  Bound: HBM (x, dy read once, dx written once).
namespace {

// RELU variant (lpf_layernorm_relu_bwd_f32): the forward was y = ReLU(LN(x)), so dy only counts where gamma xhat + beta
// > 0, and a third partial carries the column sums of dx -- the gradient of the bias that was added in front of the
// LayerNorm (the GCN layer's `conv.bias`, other_models.py:66-69).
// DROP (lpf_layernorm_relu_drop_bwd_f32): the forward was y = dropout(ReLU(LN(x))) with the in-kernel mask of the fused
// GCN layer (csrc/gcn_fused.hip, lpf_common.h lpf_drop_bits): dy is scaled by 1 / (1 - p) where (row, feature) was
// kept and dropped where it was not, recomputed from the seed.
template <int G, bool RELU, bool DROP = false>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(int64_t M, int D, const float *__restrict__ x, int64_t ldx,
                                                            const float *__restrict__ dy, int64_t ldy,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float *__restrict__ dx,
                                                            int64_t lddx, float *__restrict__ part,
                                                            uint32_t drop_thresh = 0, float drop_scale = 1.f,
                                                            uint64_t drop_seed = 0) {
    constexpr int NG = 256 / G, NP = RELU ? 3 : 2;
    __shared__ float4 red[NP][NG][G];
    const int tid = threadIdx.x, grp = tid / G, lig = tid % G;
    const int off = 4 * lig;
    const bool act = off < D;
    float4 gm = make_float4(0.f, 0.f, 0.f, 0.f), bt = gm;
    if (act) gm = *reinterpret_cast<const float4 *>(gamma + off);
    if (RELU && act) bt = *reinterpret_cast<const float4 *>(beta + off);
    float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sb = sg, sx = sg;
    const int64_t g_id = (int64_t)blockIdx.x * NG + grp, n_g = (int64_t)gridDim.x * NG;
    for (int64_t r0 = 0; r0 < M; r0 += n_g) {  // (every group runs the same number of rounds: the shuffles stay converged)
        const int64_t r = r0 + g_id;
        const bool live = r < M && act;
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), dv = xv;
        if (live) {
            xv = *reinterpret_cast<const float4 *>(x + r * ldx + off);
            dv = *reinterpret_cast<const float4 *>(dy + r * ldy + off);
            if (DROP) {
                const uint32_t rk = lpf_drop_row_key(r, drop_seed);
                dv.x = lpf_drop_bits(rk, off + 0, drop_seed) >= drop_thresh ? dv.x * drop_scale : 0.f;
                dv.y = lpf_drop_bits(rk, off + 1, drop_seed) >= drop_thresh ? dv.y * drop_scale : 0.f;
                dv.z = lpf_drop_bits(rk, off + 2, drop_seed) >= drop_thresh ? dv.z * drop_scale : 0.f;
                dv.w = lpf_drop_bits(rk, off + 3, drop_seed) >= drop_thresh ? dv.w * drop_scale : 0.f;
            }
        }
        const float mean = lpf_group_sum<G>(xv.x + xv.y + xv.z + xv.w) / (float)D;
        float4 c = make_float4(xv.x - mean, xv.y - mean, xv.z - mean, xv.w - mean);
        if (!live) c = make_float4(0.f, 0.f, 0.f, 0.f);
        const float var = lpf_group_sum<G>(c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w) / (float)D;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        const float4 xh = make_float4(c.x * rstd, c.y * rstd, c.z * rstd, c.w * rstd);
        if (RELU) {   // the gradient passes only where the forward's ReLU did
            if (!(fmaf(gm.x, xh.x, bt.x) > 0.f)) dv.x = 0.f;
            if (!(fmaf(gm.y, xh.y, bt.y) > 0.f)) dv.y = 0.f;
            if (!(fmaf(gm.z, xh.z, bt.z) > 0.f)) dv.z = 0.f;
            if (!(fmaf(gm.w, xh.w, bt.w) > 0.f)) dv.w = 0.f;
        }
        const float4 gv = make_float4(dv.x * gm.x, dv.y * gm.y, dv.z * gm.z, dv.w * gm.w);
        const float m1 = lpf_group_sum<G>(gv.x + gv.y + gv.z + gv.w) / (float)D;
        const float m2 = lpf_group_sum<G>(gv.x * xh.x + gv.y * xh.y + gv.z * xh.z + gv.w * xh.w) / (float)D;
        if (live) {
            const float4 dxv = make_float4(rstd * (gv.x - m1 - xh.x * m2), rstd * (gv.y - m1 - xh.y * m2),
                                           rstd * (gv.z - m1 - xh.z * m2), rstd * (gv.w - m1 - xh.w * m2));
            *reinterpret_cast<float4 *>(dx + r * lddx + off) = dxv;
            if (RELU) { sx.x += dxv.x; sx.y += dxv.y; sx.z += dxv.z; sx.w += dxv.w; }
            sg.x += dv.x * xh.x; sg.y += dv.y * xh.y; sg.z += dv.z * xh.z; sg.w += dv.w * xh.w;
            sb.x += dv.x; sb.y += dv.y; sb.z += dv.z; sb.w += dv.w;
        }
    }
    red[0][grp][lig] = sg;
    red[1][grp][lig] = sb;
    if (RELU) red[NP - 1][grp][lig] = sx;
    __syncthreads();
    if (grp == 0 && act) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            float4 s = red[k][0][lig];
            for (int g = 1; g < NG; ++g) {
                const float4 v = red[k][g][lig];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            *reinterpret_cast<float4 *>(part + ((int64_t)blockIdx.x * NP + k) * D + off) = s;
        }
    }
}

// 16 columns x 16 slices of the workgroup list per workgroup: a slice adds its workgroups in order, the slices meet in
// LDS and are added in slice order (deterministic; one thread per column walking a thousand partials was the slowest
// kernel of the training step)
__global__ __launch_bounds__(256) void layernorm_bwd_reduce_kernel(int D, int np, int blocks,
                                                                   const float *__restrict__ part,
                                                                   float *__restrict__ dgamma,
                                                                   float *__restrict__ dbeta,
                                                                   float *__restrict__ dxsum) {
    __shared__ float red[16][17];
    const int colx = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + colx;  // column of the [np D] row group
    float s = 0.f;
    if (c < np * D) {
        const int k = c / D, f = c % D;
        for (int b = slice; b < blocks; b += 16) s += part[((int64_t)b * np + k) * D + f];
    }
    red[slice][colx] = s;
    __syncthreads();
    if (slice == 0 && c < np * D) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][colx];
        (c / D == 0 ? dgamma : (c / D == 1 ? dbeta : dxsum))[c % D] = t;
    }
}

constexpr int LN_BWD_BLOCKS = 1024;

}  // namespace

extern "C" int64_t lpf_layernorm_bwd_workspace_floats(int32_t D) { return (int64_t)LN_BWD_BLOCKS * 3 * (D > 0 ? D : 0); }

namespace {
template <bool RELU, bool DROP = false>
int ln_bwd_launch(int64_t M, int32_t D, const float *x, int64_t ldx, const float *dy, int64_t ldy, const float *gamma,
                  const float *beta, float *dx, int64_t lddx, float *dgamma, float *dbeta, float *dxsum,
                  float *workspace, void *stream, float drop_p = 0.f, uint64_t drop_seed = 0) {
    if (D <= 0 || (D & 3) || D > 256) return LPF_ERR_UNSUPPORTED;
    LPF_REQUIRE(M >= 0 && gamma && dgamma && dbeta && workspace && lpf_aligned16(gamma) && lpf_aligned16(workspace) &&
                (!RELU || (beta && dxsum && lpf_aligned16(beta))));
    LPF_REQUIRE(drop_p >= 0.f && drop_p < 1.f);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int blocks = 0;
    if (M > 0) {
        LPF_REQUIRE(x && dy && dx && ldx >= D && ldy >= D && lddx >= D && ((ldx | ldy | lddx) & 3) == 0 &&
                    lpf_aligned16(x) && lpf_aligned16(dy) && lpf_aligned16(dx));
        const int G = D <= 32 ? 8 : (D <= 64 ? 16 : (D <= 128 ? 32 : 64));
        const int64_t want = (M + 256 / G - 1) / (256 / G);
        blocks = (int)(want < LN_BWD_BLOCKS ? want : LN_BWD_BLOCKS);
#define LPF_LNB(GG) hipLaunchKernelGGL((layernorm_bwd_kernel<GG, RELU, DROP>), dim3(blocks), dim3(256), 0, s, M, D, x, ldx, \
                                       dy, ldy, gamma, beta, dx, lddx, workspace, lpf_drop_threshold(drop_p),             \
                                       1.0f / (1.0f - drop_p), drop_seed)
        switch (G) {
            case 8: LPF_LNB(8); break;
            case 16: LPF_LNB(16); break;
            case 32: LPF_LNB(32); break;
            default: LPF_LNB(64); break;
        }
#undef LPF_LNB
    }
    const int np = RELU ? 3 : 2;
    hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3((np * D + 15) / 16), dim3(256), 0, s, D, np, blocks, workspace,
                       dgamma, dbeta, dxsum);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}
}  // namespace

extern "C" int lpf_layernorm_bwd_f32(int64_t M, int32_t D, const float *x, int64_t ldx, const float *dy, int64_t ldy,
                                     const float *gamma, float *dx, int64_t lddx, float *dgamma, float *dbeta,
                                     float *workspace, void *stream) {
    return ln_bwd_launch<false>(M, D, x, ldx, dy, ldy, gamma, nullptr, dx, lddx, dgamma, dbeta, nullptr, workspace, stream);
}

extern "C" int lpf_layernorm_relu_bwd_f32(int64_t M, int32_t D, const float *x, int64_t ldx, const float *dy, int64_t ldy,
                                          const float *gamma, const float *beta, float *dx, int64_t lddx, float *dgamma,
                                          float *dbeta, float *dxsum, float *workspace, void *stream) {
    return ln_bwd_launch<true>(M, D, x, ldx, dy, ldy, gamma, beta, dx, lddx, dgamma, dbeta, dxsum, workspace, stream);
}

extern "C" int lpf_layernorm_relu_drop_bwd_f32(int64_t M, int32_t D, const float *x, int64_t ldx, const float *dy,
                                               int64_t ldy, const float *gamma, const float *beta, float drop_p,
                                               uint64_t drop_seed, float *dx, int64_t lddx, float *dgamma, float *dbeta,
                                               float *dxsum, float *workspace, void *stream) {
    if (drop_p == 0.f)
        return ln_bwd_launch<true>(M, D, x, ldx, dy, ldy, gamma, beta, dx, lddx, dgamma, dbeta, dxsum, workspace, stream);
    return ln_bwd_launch<true, true>(M, D, x, ldx, dy, ldy, gamma, beta, dx, lddx, dgamma, dbeta, dxsum, workspace, stream,
                                     drop_p, drop_seed);
}
