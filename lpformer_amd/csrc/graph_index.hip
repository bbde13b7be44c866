// Per-model indexes over the PPR matrix, built on the device (host twins in lpformer_amd/graph.py):
//   T0 / P1 : the PPR rows restricted to the entries that can pass the >1-hop / one-hop test of the reference
//             (src/models/link_transformer.py:241-250, 464-478; same fp32 +1-1 round trip, contraction off),
//   selfp   : selfp[e] = P[i, j] for every adjacency entry e = (i, j), 0 where nothing is stored.
// One wavefront per row; filtered rows keep their column order (ballot + prefix), so they stay sorted.
#include "lpf_common.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ bool ppr_passes(float p, int mode, float theta) {
    const float s = __fsub_rn(__fadd_rn(p, 1.0f), 1.0f);
    return mode == 0 ? (p > 0.f && s >= theta) : (s >= theta);
}

__global__ __launch_bounds__(256) void ppr_filter_count_kernel(int64_t n, const int64_t *__restrict__ rowptr,
                                                               const float *__restrict__ val, int mode, float theta,
                                                               int64_t *__restrict__ out_len) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
    int64_t cnt = 0;
    for (int64_t e = e0; e < e1; e += 64) {
        const bool keep = (e + lane < e1) && ppr_passes(val[e + lane], mode, theta);
        cnt += __popcll(__ballot(keep));
    }
    if (lane == 0) out_len[row] = cnt;
}

__global__ __launch_bounds__(256) void ppr_filter_fill_kernel(int64_t n, const int64_t *__restrict__ rowptr,
                                                              const int32_t *__restrict__ col,
                                                              const float *__restrict__ val, int mode, float theta,
                                                              const int64_t *__restrict__ out_rowptr,
                                                              int32_t *__restrict__ out_col,
                                                              float *__restrict__ out_val) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
    int64_t dst = out_rowptr[row];
    for (int64_t e = e0; e < e1; e += 64) {
        const bool in = e + lane < e1;
        const float p = in ? val[e + lane] : 0.f;
        const bool keep = in && ppr_passes(p, mode, theta);
        const uint64_t m = __ballot(keep);
        if (keep) {
            const int64_t o = dst + __popcll(m & ((1ull << lane) - 1ull));
            out_col[o] = col[e + lane];
            out_val[o] = p;
        }
        dst += __popcll(m);
    }
}

__global__ __launch_bounds__(256) void self_ppr_kernel(int64_t n, const int64_t *__restrict__ adj_rowptr,
                                                       const int32_t *__restrict__ adj_col,
                                                       const int64_t *__restrict__ ppr_rowptr,
                                                       const int32_t *__restrict__ ppr_col,
                                                       const float *__restrict__ ppr_val, float *__restrict__ selfp) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const int64_t a0 = adj_rowptr[row], a1 = adj_rowptr[row + 1];
    const int64_t p0 = ppr_rowptr[row], p1 = ppr_rowptr[row + 1];
    for (int64_t e = a0 + lane; e < a1; e += 64) {
        const int32_t key = adj_col[e];
        const int64_t i = lpf_lower_bound(ppr_col, p0, p1, key);
        selfp[e] = (i < p1 && ppr_col[i] == key) ? ppr_val[i] : 0.f;
    }
}

// out[q] = M[rows[q], cols[q]] of a CSR matrix with sorted columns, 0 where nothing is stored: one thread per query
__global__ __launch_bounds__(256) void csr_lookup_kernel(int64_t nq, int64_t n, const int64_t *__restrict__ rows,
                                                         const int64_t *__restrict__ cols,
                                                         const int64_t *__restrict__ rowptr,
                                                         const int32_t *__restrict__ col, const float *__restrict__ val,
                                                         float *__restrict__ out) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const int64_t r = rows[q], c = cols[q];
    float v = 0.f;
    if ((uint64_t)r < (uint64_t)n && (uint64_t)c < (uint64_t)n) {
        const int64_t p0 = rowptr[r], p1 = rowptr[r + 1];
        const int64_t i = lpf_lower_bound(col, p0, p1, (int32_t)c);
        if (i < p1 && col[i] == (int32_t)c) v = val[i];
    }
    out[q] = v;
}

}  // namespace

/* out[q] = M[rows[q], cols[q]] (0 where nothing is stored; ids outside [0, n) give 0) for a CSR matrix with sorted
 * columns: the raw PPR values of a handful of (endpoint, node) pairs -- the entries whose type changes when the training
 * loop removes the batch's positive edges from the typing adjacency (src/train/train_model.py:40-46 ->
 * src/models/link_transformer.py:229-237,290-291: their values are re-derived with the other type's round trip). */
extern "C" int lpf_csr_lookup_f32(int64_t nq, int64_t n, const int64_t *rows, const int64_t *cols, const int64_t *rowptr,
                                  const int32_t *col, const float *val, float *out, void *stream) {
    if (nq == 0) return LPF_OK;
    LPF_REQUIRE(nq > 0 && n > 0 && rows && cols && rowptr && col && val && out);
    hipLaunchKernelGGL(csr_lookup_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), nq, n, rows, cols, rowptr, col, val, out);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_ppr_filter_count(int64_t n, const int64_t *rowptr, const float *val, int32_t mode, float theta,
                                    int64_t *out_len, void *stream) {
    if (n == 0) return LPF_OK;
    LPF_REQUIRE(n > 0 && rowptr && val && out_len && (mode == 0 || mode == 1));
    hipLaunchKernelGGL(ppr_filter_count_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), n, rowptr, val, mode, theta, out_len);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_ppr_filter_fill(int64_t n, const int64_t *rowptr, const int32_t *col, const float *val,
                                   int32_t mode, float theta, const int64_t *out_rowptr, int32_t *out_col,
                                   float *out_val, void *stream) {
    if (n == 0) return LPF_OK;
    LPF_REQUIRE(n > 0 && rowptr && col && val && out_rowptr && out_col && out_val && (mode == 0 || mode == 1));
    hipLaunchKernelGGL(ppr_filter_fill_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), n, rowptr, col, val, mode, theta, out_rowptr, out_col,
                       out_val);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_self_ppr(int64_t n, const int64_t *adj_rowptr, const int32_t *adj_col, const int64_t *ppr_rowptr,
                            const int32_t *ppr_col, const float *ppr_val, float *selfp, void *stream) {
    if (n == 0) return LPF_OK;
    LPF_REQUIRE(n > 0 && adj_rowptr && adj_col && ppr_rowptr && ppr_col && ppr_val && selfp);
    hipLaunchKernelGGL(self_ppr_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       n, adj_rowptr, adj_col, ppr_rowptr, ppr_col, ppr_val, selfp);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}
