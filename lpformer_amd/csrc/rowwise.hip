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
