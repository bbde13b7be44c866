// Host-side PPR producer (liblpformer_host.so): approximate personalised PageRank for every source node.
//
// Behavioural twin of the reference's numba kernel `calc_ppr` (src/util/calc_ppr_scores.py:136-192) followed by
// `create_sparse_ppr_matrix` (:221-241): Andersen push with a LIFO work stack, float64 arithmetic evaluated in the
// same order, membership test before every push, values rounded to fp32 and rows sorted by column.  Because the
// selected index sets downstream must be bit-exact, nothing here may re-order the floating-point operations:
// compile without -ffast-math / FMA contraction.
//
// Data structures are MI355X-host friendly rather than dict-based: each OpenMP thread owns dense epoch-stamped
// arrays (p, r, "is on the stack") of N entries, so a push is O(1) with no hashing, and appends its finished rows to
// a private arena; the arenas are stitched into one CSR at the end.
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "../../include/lpformer_hip.h"

namespace {

struct RowRef {
    int64_t row;
    int64_t offset;  // into the owning thread's arena
    int32_t len;
};

struct Arena {
    std::vector<int32_t> col;
    std::vector<float> val;
    std::vector<RowRef> rows;
};

}  // namespace

extern "C" int lpf_ppr_push_cpu(int64_t n, const int64_t *indptr, const int32_t *indices, double alpha, double eps,
                                int64_t *out_rowptr, int32_t **out_col, float **out_val, int32_t num_threads) {
    if (n < 0 || !indptr || (!indices && n > 0 && indptr[n] > 0) || !out_rowptr || !out_col || !out_val)
        return LPF_ERR_INVALID;
    *out_col = nullptr;
    *out_val = nullptr;
    if (n >= (1ll << 31)) return LPF_ERR_UNSUPPORTED;
    const int nt = num_threads > 0 ? num_threads : omp_get_max_threads();
    const double alpha_eps = alpha * eps;
    std::vector<Arena> arenas((size_t)nt);

#pragma omp parallel num_threads(nt)
    {
        Arena &ar = arenas[(size_t)omp_get_thread_num()];
        std::vector<double> p((size_t)n), r((size_t)n);
        std::vector<int32_t> p_stamp((size_t)n, -1), r_stamp((size_t)n, -1);
        std::vector<uint8_t> on_stack((size_t)n, 0);
        std::vector<int32_t> stack, touched;
        std::vector<std::pair<int32_t, float>> rowbuf;

#pragma omp for schedule(dynamic, 32)
        for (int64_t src64 = 0; src64 < n; ++src64) {
            const int32_t src = (int32_t)src64;
            touched.clear();
            stack.clear();
            p[src] = 0.0;  // p = {src: 0.0}
            p_stamp[src] = src;
            touched.push_back(src);
            r[src] = alpha;  // r = {src: alpha}
            r_stamp[src] = src;
            stack.push_back(src);
            on_stack[src] = 1;
            while (!stack.empty()) {
                const int32_t u = stack.back();  // q.pop(): LIFO
                stack.pop_back();
                on_stack[u] = 0;
                const double res = (r_stamp[u] == src) ? r[u] : 0.0;
                if (p_stamp[u] == src) {
                    p[u] += res;
                } else {
                    p[u] = res;
                    p_stamp[u] = src;
                    touched.push_back(u);
                }
                r[u] = 0.0;
                r_stamp[u] = src;
                const int64_t e0 = indptr[u], e1 = indptr[u + 1];
                const double deg_u = (double)(e1 - e0);
                for (int64_t e = e0; e < e1; ++e) {
                    const int32_t v = indices[e];
                    const double push = (1.0 - alpha) * res / deg_u;  // same expression, same order, every edge
                    if (r_stamp[v] == src) {
                        r[v] += push;
                    } else {
                        r[v] = push;
                        r_stamp[v] = src;
                    }
                    const double deg_v = (double)(indptr[v + 1] - indptr[v]);
                    if (r[v] >= alpha_eps * deg_v && !on_stack[v]) {
                        stack.push_back(v);
                        on_stack[v] = 1;
                    }
                }
            }
            rowbuf.clear();
            for (int32_t v : touched) rowbuf.emplace_back(v, (float)p[v]);  // torch.Tensor(list): f64 -> f32
            std::sort(rowbuf.begin(), rowbuf.end(),
                      [](const std::pair<int32_t, float> &x, const std::pair<int32_t, float> &y) { return x.first < y.first; });
            ar.rows.push_back({src64, (int64_t)ar.col.size(), (int32_t)rowbuf.size()});
            for (const auto &kv : rowbuf) {
                ar.col.push_back(kv.first);
                ar.val.push_back(kv.second);
            }
        }
    }

    std::fill(out_rowptr, out_rowptr + n + 1, (int64_t)0);
    for (const Arena &ar : arenas)
        for (const RowRef &rr : ar.rows) out_rowptr[rr.row + 1] = rr.len;
    for (int64_t i = 0; i < n; ++i) out_rowptr[i + 1] += out_rowptr[i];
    const int64_t nnz = out_rowptr[n];
    int32_t *col = (int32_t *)malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(nnz, 1));
    float *val = (float *)malloc(sizeof(float) * (size_t)std::max<int64_t>(nnz, 1));
    if (!col || !val) {
        free(col);
        free(val);
        return LPF_ERR_INVALID;
    }
#pragma omp parallel for num_threads(nt) schedule(static)
    for (int t = 0; t < nt; ++t) {
        const Arena &ar = arenas[(size_t)t];
        for (const RowRef &rr : ar.rows) {
            memcpy(col + out_rowptr[rr.row], ar.col.data() + rr.offset, sizeof(int32_t) * (size_t)rr.len);
            memcpy(val + out_rowptr[rr.row], ar.val.data() + rr.offset, sizeof(float) * (size_t)rr.len);
        }
    }
    *out_col = col;
    *out_val = val;
    return LPF_OK;
}

extern "C" void lpf_host_free(void *p) { free(p); }

extern "C" int lpf_host_abi_version(void) { return LPF_ABI_VERSION; }
