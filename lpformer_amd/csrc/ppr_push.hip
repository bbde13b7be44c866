// Device-side PPR producer: approximate personalised PageRank for every source node on the GPU,
// bit-identical to the reference's numba kernel `calc_ppr` + `create_sparse_ppr_matrix`
// (src/util/calc_ppr_scores.py:136-192, 221-241) and to the host twin in host_ppr.cpp.
//
// The push for ONE source is inherently sequential: the work stack is LIFO, residuals are accumulated in float64 in
// the order the stack dictates, and the selected index sets downstream depend on every bit of it.  What is parallel:
//   * sources are independent: one wavefront per source, thousands of wavefronts resident, dynamic assignment;
//   * inside one pop, the neighbour list of the popped node: each lane takes one neighbour (distinct nodes, so the
//     read-modify-writes of r[] never collide), qualifying neighbours are appended to the stack in lane (= CSR)
//     order through a ballot, which is exactly the order the sequential loop pushes them in.
// State is MI355X-sized rather than hash-based: every wavefront owns DENSE epoch-stamped arrays over all N nodes
// (16 B per node for r/on-stack, 16 B for p; the stamp is the source id, so nothing is cleared between sources) --
// 32 N bytes per wavefront, several GB in total, which 288 GB of HBM has room for and which makes a push one
// 16-byte access with no probing.  The stack / touched lists are bounded by 1/(alpha*eps) entries (every pop moves
// at least alpha*eps of the unit mass into p).
// Finished rows go to a pool in completion order (one atomic bump per row); lpf_ppr_pack_csr sorts each row by
// column (rocPRIM segmented radix sort: a plain library sort, not part of the scoring path) and packs the CSR.
#include <string.h>

#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "lpf_common.h"

#pragma clang fp contract(off)  // float64 expressions must round exactly like the sequential reference

namespace {

struct alignas(16) PprR {
    double r;
    int32_t stamp;     // == source id when r / on_stack are valid for the current source
    int32_t on_stack;
};
struct alignas(16) PprP {
    double p;
    int32_t stamp;
    int32_t pad;
};

__device__ __forceinline__ int64_t wave_bcast_i64(int64_t v) {
    const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffff));
    const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return ((int64_t)hi << 32) | (uint32_t)lo;
}

// A stack entry carries the node's row start and degree (both are loaded anyway when the node is pushed), so a pop can
// request the node's state and its neighbour list in the same round trip; the entry pushed last also stays in
// registers, so the common "pop what was just pushed" needs no load at all.  Per pop: two dependent memory round trips
// (state + neighbours, then the neighbours' state) instead of four.
struct alignas(16) PprEntry {
    int32_t node, deg;
    int64_t e0;
};

__global__ __launch_bounds__(256) void ppr_push_kernel(
    int64_t n, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, double alpha, double alpha_eps,
    PprR *__restrict__ r_all, PprP *__restrict__ p_all, PprEntry *__restrict__ stack_all,
    int32_t *__restrict__ touched_all, int64_t list_cap, unsigned long long *__restrict__ counters,
    int32_t *__restrict__ pool_col, float *__restrict__ pool_val, int64_t pool_cap, int64_t *__restrict__ row_off,
    int32_t *__restrict__ row_len) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    PprR *R = r_all + wave_id * n;
    PprP *P = p_all + wave_id * n;
    PprEntry *stack = stack_all + wave_id * list_cap;
    int32_t *touched = touched_all + wave_id * list_cap;
    const uint64_t lanes_lt = (1ull << lane) - 1ull;

    while (true) {
        int64_t src64 = 0;
        if (lane == 0) src64 = (int64_t)atomicAdd(&counters[0], 1ull);
        src64 = wave_bcast_i64(src64);
        if (src64 >= n) break;
        const int32_t src = (int32_t)src64;
        const int64_t s0 = rowptr[src], s1 = rowptr[src + 1];
        if (lane == 0) {  // p = {src: 0.0}; r = {src: alpha}; q = [src]
            P[src] = PprP{0.0, src, 0};
            touched[0] = src;
            R[src] = PprR{alpha, src, 1};
        }
        int64_t sp = 1, nt = 1;  // stack[0 .. sp); `top` caches the last entry (the source's own entry exists only there)
        PprEntry top{src, (int32_t)(s1 - s0), s0};
        bool top_valid = true, overflow = false;
        __threadfence_block();
        while (sp > 0) {
            --sp;
            const PprEntry cur = top_valid ? top : stack[sp];  // q.pop(): LIFO
            top_valid = false;
            const int32_t u = cur.node;
            const int64_t e0 = cur.e0, e1 = cur.e0 + cur.deg;
            const PprR ru = R[u];
            const PprP pu = P[u];
            int32_t v0 = (lane < cur.deg) ? col[e0 + lane] : 0;  // first 64 neighbours ride along with the state
            const double res = (ru.stamp == src) ? ru.r : 0.0;
            const bool seen = pu.stamp == src;
            if (lane == 0) {
                P[u] = PprP{seen ? pu.p + res : res, src, 0};
                if (!seen && nt < list_cap) touched[nt] = u;
                R[u] = PprR{0.0, src, 0};
            }
            if (!seen) {
                if (nt >= list_cap) overflow = true;
                ++nt;
            }
            const double push = (1.0 - alpha) * res / (double)cur.deg;  // same expression and order as the reference
            __threadfence_block();  // r[u] = 0 is in place before a self-loop edge reads it
            for (int64_t e = e0; e < e1; e += 64) {
                const bool act = e + lane < e1;
                const int32_t v = (e == e0) ? v0 : (act ? col[e + lane] : 0);
                bool q = false;
                PprEntry mine{v, 0, 0};
                if (act) {
                    const PprR rv = R[v];
                    const int64_t b0 = rowptr[v], b1 = rowptr[v + 1];
                    mine.deg = (int32_t)(b1 - b0);
                    mine.e0 = b0;
                    const bool cur_src = rv.stamp == src;
                    const double val = cur_src ? rv.r + push : push;
                    const bool was_on = cur_src && rv.on_stack != 0;
                    q = (val >= alpha_eps * (double)mine.deg) && !was_on;
                    R[v] = PprR{val, src, (was_on || q) ? 1 : 0};
                }
                const uint64_t m = __ballot(q);
                if (m) {
                    const int64_t dst = sp + __popcll(m & lanes_lt);
                    if (q && dst < list_cap) stack[dst] = mine;
                    sp += __popcll(m);
                    const int last = 63 - __builtin_clzll(m);  // the entry pushed last is the next one popped
                    top.node = __builtin_amdgcn_readlane(mine.node, last);
                    top.deg = __builtin_amdgcn_readlane(mine.deg, last);
                    top.e0 = ((int64_t)__builtin_amdgcn_readlane((int)(mine.e0 >> 32), last) << 32) |
                             (uint32_t)__builtin_amdgcn_readlane((int)(mine.e0 & 0xffffffff), last);
                    top_valid = true;
                    if (sp > list_cap) {  // cannot happen for alpha*eps > 0 (mass bound); never write out of bounds
                        overflow = true;
                        sp = list_cap;
                        top_valid = false;
                    }
                }
            }
            __threadfence_block();  // pushes (written by other lanes) are visible to later pops
        }
        // emit: keys of p in insertion order, values rounded to fp32 (torch.Tensor(list) in the reference)
        int64_t off = 0;
        if (lane == 0) off = (int64_t)atomicAdd(&counters[1], (unsigned long long)nt);
        off = wave_bcast_i64(off);
        if (lane == 0) {
            row_off[src] = off;
            row_len[src] = (int32_t)nt;
            if (overflow) atomicAdd(&counters[2], 1ull);
        }
        if (off + nt <= pool_cap && !overflow) {
            for (int64_t i = lane; i < nt; i += 64) {
                const int32_t v = touched[i];
                pool_col[off + i] = v;
                pool_val[off + i] = (float)P[v].p;
            }
        }
    }
}

__global__ __launch_bounds__(256) void ppr_row_end_kernel(int64_t n, const int64_t *__restrict__ row_off,
                                                          const int32_t *__restrict__ row_len,
                                                          int64_t *__restrict__ row_end,
                                                          int64_t *__restrict__ len64) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        row_end[i] = row_off[i] + row_len[i];
        len64[i] = row_len[i];
    }
}

// one wavefront per row: copy the sorted segment to its place in the CSR
__global__ __launch_bounds__(256) void ppr_pack_kernel(int64_t n, const int64_t *__restrict__ row_off,
                                                       const int32_t *__restrict__ row_len,
                                                       const int64_t *__restrict__ out_rowptr,
                                                       const int32_t *__restrict__ scol, const float *__restrict__ sval,
                                                       int32_t *__restrict__ out_col, float *__restrict__ out_val) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const int64_t src = row_off[row], dst = out_rowptr[row];
    const int len = row_len[row];
    for (int i = lane; i < len; i += 64) {
        out_col[dst + i] = scol[src + i];
        out_val[dst + i] = sval[src + i];
    }
}

constexpr int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

int64_t ppr_list_cap(int64_t n, double alpha, double eps) {
    const double bound = 1.0 / (alpha * eps) + 2.0;  // pops (hence touched nodes and live stack entries) per source
    return (bound < (double)n) ? (int64_t)bound : n;
}

}  // namespace

extern "C" int64_t lpf_ppr_push_workspace_bytes(int64_t n, int64_t n_waves, double alpha, double eps) {
    if (n <= 0 || n_waves <= 0 || !(alpha > 0.0) || !(eps > 0.0)) return 0;
    const int64_t cap = ppr_list_cap(n, alpha, eps);
    return align256(n_waves * n * (int64_t)sizeof(PprR)) + align256(n_waves * n * (int64_t)sizeof(PprP)) +
           align256(n_waves * cap * (int64_t)sizeof(PprEntry)) + align256(n_waves * cap * 4) + 256;
}

extern "C" int lpf_ppr_push_f64(int64_t n, const int64_t *rowptr, const int32_t *col, double alpha, double eps,
                                int64_t n_waves, void *workspace, int64_t workspace_bytes, int32_t *pool_col,
                                float *pool_val, int64_t pool_capacity, int64_t *row_off, int32_t *row_len,
                                int64_t *counters, void *stream) {
    if (n == 0) return LPF_OK;
    LPF_REQUIRE(n > 0 && n < (1ll << 31) && rowptr && col && alpha > 0.0 && alpha < 1.0 && eps > 0.0 && n_waves > 0 &&
                (n_waves & 3) == 0 && workspace && pool_col && pool_val && pool_capacity >= 0 && row_off && row_len &&
                counters && lpf_aligned16(workspace));
    LPF_REQUIRE(workspace_bytes >= lpf_ppr_push_workspace_bytes(n, n_waves, alpha, eps));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t cap = ppr_list_cap(n, alpha, eps);
    char *w = static_cast<char *>(workspace);
    PprR *R = reinterpret_cast<PprR *>(w);
    w += align256(n_waves * n * (int64_t)sizeof(PprR));
    PprP *P = reinterpret_cast<PprP *>(w);
    w += align256(n_waves * n * (int64_t)sizeof(PprP));
    PprEntry *stack = reinterpret_cast<PprEntry *>(w);
    w += align256(n_waves * cap * (int64_t)sizeof(PprEntry));
    int32_t *touched = reinterpret_cast<int32_t *>(w);
    // stamps = -1 (no source has that id): 0xFF bytes over the two dense state arrays
    if (hipMemsetAsync(R, 0xFF, (size_t)(n_waves * n) * sizeof(PprR), s) != hipSuccess ||
        hipMemsetAsync(P, 0xFF, (size_t)(n_waves * n) * sizeof(PprP), s) != hipSuccess ||
        hipMemsetAsync(counters, 0, 4 * sizeof(int64_t), s) != hipSuccess)
        return LPF_ERR_LAUNCH;
    hipLaunchKernelGGL(ppr_push_kernel, dim3((unsigned)(n_waves / 4)), dim3(256), 0, s, n, rowptr, col, alpha,
                       alpha * eps, R, P, stack, touched, cap, reinterpret_cast<unsigned long long *>(counters),
                       pool_col, pool_val, pool_capacity, row_off, row_len);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int64_t lpf_ppr_pack_workspace_bytes(int64_t n, int64_t nnz) {
    if (n <= 0 || nnz < 0 || nnz >= (1ll << 32)) return 0;
    size_t sort_bytes = 0, scan_bytes = 0;
    (void)rocprim::segmented_radix_sort_pairs(nullptr, sort_bytes, (const int32_t *)nullptr, (int32_t *)nullptr,
                                        (const float *)nullptr, (float *)nullptr, (unsigned)nnz, (unsigned)n,
                                        (const int64_t *)nullptr, (const int64_t *)nullptr, 0, 32);
    (void)rocprim::inclusive_scan(nullptr, scan_bytes, (const int64_t *)nullptr, (int64_t *)nullptr, (size_t)n,
                            rocprim::plus<int64_t>());
    const int64_t tmp = (int64_t)(sort_bytes > scan_bytes ? sort_bytes : scan_bytes);
    return 2 * align256(nnz * 4) + 2 * align256(n * 8) + align256(tmp) + 256;
}

extern "C" int lpf_ppr_pack_csr(int64_t n, const int64_t *row_off, const int32_t *row_len, const int32_t *pool_col,
                                const float *pool_val, int64_t nnz, int64_t *out_rowptr, int32_t *out_col,
                                float *out_val, void *workspace, int64_t workspace_bytes, void *stream) {
    LPF_REQUIRE(n >= 0 && out_rowptr);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) {
        (void)hipMemsetAsync(out_rowptr, 0, sizeof(int64_t), s);
        return LPF_OK;
    }
    LPF_REQUIRE(row_off && row_len && pool_col && pool_val && nnz >= 0 && nnz < (1ll << 32) && out_col && out_val &&
                workspace && n < (1ll << 31));
    LPF_REQUIRE(workspace_bytes >= lpf_ppr_pack_workspace_bytes(n, nnz));
    char *w = static_cast<char *>(workspace);
    int32_t *scol = reinterpret_cast<int32_t *>(w);
    w += align256(nnz * 4);
    float *sval = reinterpret_cast<float *>(w);
    w += align256(nnz * 4);
    int64_t *row_end = reinterpret_cast<int64_t *>(w);
    w += align256(n * 8);
    int64_t *len64 = reinterpret_cast<int64_t *>(w);
    w += align256(n * 8);
    size_t tmp_bytes = (size_t)(workspace_bytes - (w - static_cast<char *>(workspace)));
    hipLaunchKernelGGL(ppr_row_end_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, row_off, row_len,
                       row_end, len64);
    // out_rowptr[0] = 0, out_rowptr[1..n] = inclusive scan of the row lengths
    if (hipMemsetAsync(out_rowptr, 0, sizeof(int64_t), s) != hipSuccess) return LPF_ERR_LAUNCH;
    size_t scan_bytes = tmp_bytes;
    if (rocprim::inclusive_scan(w, scan_bytes, len64, out_rowptr + 1, (size_t)n, rocprim::plus<int64_t>(), s) !=
        hipSuccess)
        return LPF_ERR_LAUNCH;
    if (nnz > 0) {
        size_t sort_bytes = tmp_bytes;
        if (rocprim::segmented_radix_sort_pairs(w, sort_bytes, pool_col, scol, pool_val, sval, (unsigned)nnz,
                                                (unsigned)n, row_off, static_cast<const int64_t *>(row_end), 0, 32,
                                                s) != hipSuccess)
            return LPF_ERR_LAUNCH;
        hipLaunchKernelGGL(ppr_pack_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, n, row_off, row_len,
                           out_rowptr, scol, sval, out_col, out_val);
    }
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}
