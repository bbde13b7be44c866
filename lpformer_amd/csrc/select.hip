// PPR-thresholded node selection for a batch of candidate pairs (integer / bit-exact part of the pair stage).
//
// The reference builds BS x N sparse COO temporaries and coalesces (sorts) them seven times per batch
// (src/models/link_transformer.py:214-319,434-481).  Here every pair is handled by one wavefront working directly
// on CSR rows with sorted columns:
//   pass A  lanes walk N(a) (coalesced), each lane binary-searches its node in N(b) (type 2 = CN, else 1-hop) and in
//           the PPR rows of a and b, applies the reference's fp32 round trip and threshold, and the survivors are
//           ballot-compacted -- in ascending node order -- into the CN run and the first 1-hop run;
//   pass B  the same for N(b) \ N(a) (second 1-hop run; the two runs are merged by lpf_select_compact);
//   pass T  lanes walk the shorter of the two >1-hop candidate rows (the PPR row, or a per-threshold prefiltered
//           copy of it), look the node up in the other row and in both adjacency rows, and emit the >1-hop run.
// No LDS, no atomics, no sorting: order comes from the CSR order.  Traffic is row reads (coalesced) plus binary-search
// probes that hit L2 -- the kernel is bound by memory latency/bandwidth, not arithmetic.
#include "lpf_common.h"

// the reference's fp32 round trip must be evaluated op by op: no fused multiply-add in this file
#pragma clang fp contract(off)

namespace {

// fl32((fl32(fl32(p*t)+t)-t)/t) for t in {1,2}, without letting the compiler contract or re-associate anything.
// p*1, p*2, x/1 and x/2 are exact in binary fp32, so only the add and the subtract round.
__device__ __forceinline__ float ppr_round_trip(float p, bool two) {
    if (two) return 0.5f * __fsub_rn(__fadd_rn(p * 2.0f, 2.0f), 2.0f);
    return __fsub_rn(__fadd_rn(p, 1.0f), 1.0f);
}

__device__ __forceinline__ bool csr_contains(const int32_t *__restrict__ col, int64_t lo, int64_t hi, int32_t key) {
    const int64_t i = lpf_lower_bound(col, lo, hi, key);
    return i < hi && col[i] == key;
}

// value stored at (row, key) or 0 when absent (a sparse entry that is not stored reads as 0)
__device__ __forceinline__ float csr_value(const int32_t *__restrict__ col, const float *__restrict__ val, int64_t lo,
                                           int64_t hi, int32_t key, bool *found) {
    const int64_t i = lpf_lower_bound(col, lo, hi, key);
    const bool f = i < hi && col[i] == key;
    *found = f;
    return f ? val[i] : 0.0f;
}

__device__ __forceinline__ int lanes_below(uint64_t mask, int lane) {
    return __popcll(mask & ((1ull << lane) - 1ull));
}

// Block-wide sum of one int64 per thread (256 threads); result valid in every thread.
__device__ __forceinline__ int64_t block_sum_i64(int64_t v, int64_t *lds4) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

__global__ __launch_bounds__(256) void select_bound_kernel(int64_t bs, const int64_t *__restrict__ batch,
                                                           int64_t batch_ld, const int64_t *__restrict__ adj_rowptr,
                                                           const int64_t *__restrict__ t0_rowptr,
                                                           int64_t *__restrict__ stage_off,
                                                           int64_t *__restrict__ blk) {
    __shared__ int64_t red[4];
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) stage_off[0] = 0;
    int64_t ub = 0;
    if (k < bs) {
        const int64_t a = batch[k], b = batch[batch_ld + k];
        const int64_t dA = adj_rowptr[a + 1] - adj_rowptr[a], dB = adj_rowptr[b + 1] - adj_rowptr[b];
        ub = 2 * dA + dB;
        if (t0_rowptr) {
            const int64_t ha = t0_rowptr[a + 1] - t0_rowptr[a], hb = t0_rowptr[b + 1] - t0_rowptr[b];
            ub += ha < hb ? ha : hb;
        }
        stage_off[k + 1] = ub;
    }
    const int64_t tot = block_sum_i64(ub, red);
    if (threadIdx.x == 0) blk[blockIdx.x] = tot;
}

// Second half of a two-kernel scan.  The producer kernel (256 threads, one element per thread) has written its
// values to data[q*stride + 1 + i] and the sum of every 256-element block to blk[q*nb + block]; here block B adds
// the sums of the blocks before it to an in-block inclusive scan.  Fully parallel: grid = nb blocks.
template <int NSEQ>
__global__ __launch_bounds__(256) void scan_blocks_kernel(int64_t n, int64_t *__restrict__ data, int64_t stride,
                                                          const int64_t *__restrict__ blk, int64_t nb) {
    __shared__ int64_t red[4];
    __shared__ int64_t wtot[NSEQ][4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t B = blockIdx.x, i = B * 256 + t;
#pragma unroll
    for (int q = 0; q < NSEQ; ++q) {
        int64_t pre = 0;
        for (int64_t j = t; j < B; j += 256) pre += blk[q * nb + j];
        pre = block_sum_i64(pre, red);
        int64_t x = (i < n) ? data[q * stride + 1 + i] : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wtot[q][wave] = x;
        __syncthreads();
        for (int w = 0; w < wave; ++w) pre += wtot[q][w];
        if (i < n) data[q * stride + 1 + i] = x + pre;
    }
}

// A sorted int32 row that was staged into LDS when it fits (fast probes) and is searched in global memory otherwise.
struct SortedRow {
    const int32_t *lds;            // nullptr when the row did not fit
    const int32_t *__restrict__ g; // global column array
    int64_t lo;                    // row start in g
    int n;                         // row length
};

// index of `key` inside the row, or -1
__device__ __forceinline__ int row_find(const SortedRow &r, int32_t key) {
    int lo = 0, hi = r.n;
    if (r.lds) {
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (r.lds[mid] < key) lo = mid + 1; else hi = mid;
        }
        return (lo < r.n && r.lds[lo] == key) ? lo : -1;
    }
    const int32_t *__restrict__ a = r.g + r.lo;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < r.n && a[lo] == key) ? lo : -1;
}

constexpr int SEL_CAP_PPR = 1024;  // PPR row columns staged per endpoint (4 KiB each)
constexpr int SEL_CAP_ADJ = 512;   // adjacency row staged per endpoint (2 KiB each)

// One 64-lane workgroup (= one wavefront) per pair; 12 KiB of LDS holds the four sorted rows that get probed.
__global__ __launch_bounds__(64) void select_nodes_kernel(
    int64_t bs, const int64_t *__restrict__ batch, int64_t batch_ld, const int64_t *__restrict__ adj_rowptr,
    const int32_t *__restrict__ adj_col, const int64_t *__restrict__ adjx_rowptr, const int32_t *__restrict__ adjx_col,
    const int64_t *__restrict__ ppr_rowptr, const int32_t *__restrict__ ppr_col, const float *__restrict__ ppr_val,
    const int64_t *__restrict__ t0_rowptr, const int32_t *__restrict__ t0_col, const float *__restrict__ t0_val,
    float th_cn, float th_1, float th_n, const int64_t *__restrict__ stage_off, int32_t *__restrict__ stage_node,
    float *__restrict__ stage_pa, float *__restrict__ stage_pb, int32_t *__restrict__ stage_cnt) {
    __shared__ int32_t s_pa[SEL_CAP_PPR], s_pb[SEL_CAP_PPR], s_a[SEL_CAP_ADJ], s_b[SEL_CAP_ADJ];
    const int lane = threadIdx.x;
    const bool same_adj = (adjx_rowptr == adj_rowptr) && (adjx_col == adj_col);

    for (int64_t p = blockIdx.x; p < bs; p += gridDim.x) {
        const int64_t a = batch[p], b = batch[batch_ld + p];
        const int64_t ra0 = adj_rowptr[a], ra1 = adj_rowptr[a + 1];
        const int64_t rb0 = adj_rowptr[b], rb1 = adj_rowptr[b + 1];
        const int64_t pa0 = ppr_rowptr[a], pa1 = ppr_rowptr[a + 1];
        const int64_t pb0 = ppr_rowptr[b], pb1 = ppr_rowptr[b + 1];
        const int64_t dA = ra1 - ra0, dB = rb1 - rb0;
        const int64_t s = stage_off[p];
        const int64_t cn_base = s, l1_base = s + dA, l2_base = s + 2 * dA, t0_base = s + 2 * dA + dB;
        int n_cn = 0, n_l1 = 0, n_l2 = 0, n_t0 = 0;

        // ---- stage the probed rows (coalesced 4-byte reads, all in flight together)
        SortedRow rowA{nullptr, adj_col, ra0, (int)dA}, rowB{nullptr, adj_col, rb0, (int)dB};
        SortedRow rowPa{nullptr, ppr_col, pa0, (int)(pa1 - pa0)}, rowPb{nullptr, ppr_col, pb0, (int)(pb1 - pb0)};
        __syncthreads();  // previous pair's probes are done before the rows are overwritten
        if (rowA.n <= SEL_CAP_ADJ) { for (int i = lane; i < rowA.n; i += 64) s_a[i] = adj_col[ra0 + i]; rowA.lds = s_a; }
        if (rowB.n <= SEL_CAP_ADJ) { for (int i = lane; i < rowB.n; i += 64) s_b[i] = adj_col[rb0 + i]; rowB.lds = s_b; }
        if (rowPa.n <= SEL_CAP_PPR) { for (int i = lane; i < rowPa.n; i += 64) s_pa[i] = ppr_col[pa0 + i]; rowPa.lds = s_pa; }
        if (rowPb.n <= SEL_CAP_PPR) { for (int i = lane; i < rowPb.n; i += 64) s_pb[i] = ppr_col[pb0 + i]; rowPb.lds = s_pb; }
        __syncthreads();

        // ---- pass A: every neighbour of a
        for (int64_t i0 = 0; i0 < dA; i0 += 64) {
            const int64_t i = i0 + lane;
            const bool valid = i < dA;
            int32_t x = 0;
            bool in_b = false, keep = false;
            float va = 0.f, vb = 0.f;
            if (valid) {
                x = rowA.lds ? s_a[i] : adj_col[ra0 + i];
                in_b = row_find(rowB, x) >= 0;
                const int ia = row_find(rowPa, x), ib = row_find(rowPb, x);
                va = ppr_round_trip(ia >= 0 ? ppr_val[pa0 + ia] : 0.0f, in_b);
                vb = ppr_round_trip(ib >= 0 ? ppr_val[pb0 + ib] : 0.0f, in_b);
                const float th = in_b ? th_cn : th_1;
                keep = (va >= th) && (vb >= th);
            }
            const uint64_t m_cn = __ballot(keep && in_b), m_l1 = __ballot(keep && !in_b);
            if (keep) {
                const int64_t dst = in_b ? cn_base + n_cn + lanes_below(m_cn, lane)
                                         : l1_base + n_l1 + lanes_below(m_l1, lane);
                stage_node[dst] = x;
                stage_pa[dst] = va;
                stage_pb[dst] = vb;
            }
            n_cn += __popcll(m_cn);
            n_l1 += __popcll(m_l1);
        }
        // ---- pass B: neighbours of b that are not neighbours of a (always type 1)
        for (int64_t j0 = 0; j0 < dB; j0 += 64) {
            const int64_t j = j0 + lane;
            int32_t y = 0;
            bool keep = false;
            float va = 0.f, vb = 0.f;
            if (j < dB) {
                y = rowB.lds ? s_b[j] : adj_col[rb0 + j];
                if (row_find(rowA, y) < 0) {
                    const int ia = row_find(rowPa, y), ib = row_find(rowPb, y);
                    va = ppr_round_trip(ia >= 0 ? ppr_val[pa0 + ia] : 0.0f, false);
                    vb = ppr_round_trip(ib >= 0 ? ppr_val[pb0 + ib] : 0.0f, false);
                    keep = (va >= th_1) && (vb >= th_1);
                }
            }
            const uint64_t m = __ballot(keep);
            if (keep) {
                const int64_t dst = l2_base + n_l2 + lanes_below(m, lane);
                stage_node[dst] = y;
                stage_pa[dst] = va;
                stage_pb[dst] = vb;
            }
            n_l2 += __popcll(m);
        }
        // ---- pass T: >1-hop nodes = stored in both PPR rows, adjacent to neither endpoint (unmasked adjacency)
        if (t0_rowptr) {
            const int64_t ta0 = t0_rowptr[a], ta1 = t0_rowptr[a + 1];
            const int64_t tb0 = t0_rowptr[b], tb1 = t0_rowptr[b + 1];
            const bool walk_a = (ta1 - ta0) <= (tb1 - tb0);  // walk the shorter row, probe the longer
            const int64_t w0 = walk_a ? ta0 : tb0, w1 = walk_a ? ta1 : tb1;
            const int64_t o0 = walk_a ? tb0 : ta0, o1 = walk_a ? tb1 : ta1;
            SortedRow rowXa = rowA, rowXb = rowB;
            if (!same_adj) {
                rowXa = SortedRow{nullptr, adjx_col, adjx_rowptr[a], (int)(adjx_rowptr[a + 1] - adjx_rowptr[a])};
                rowXb = SortedRow{nullptr, adjx_col, adjx_rowptr[b], (int)(adjx_rowptr[b + 1] - adjx_rowptr[b])};
            }
            for (int64_t i0 = w0; i0 < w1; i0 += 64) {
                const int64_t i = i0 + lane;
                int32_t v = 0;
                bool keep = false;
                float sa = 0.f, sb = 0.f;
                if (i < w1) {
                    v = t0_col[i];
                    const float pw = t0_val[i];
                    const float sw = __fsub_rn(__fadd_rn(pw, 1.0f), 1.0f);
                    if (pw > 0.f && sw >= th_n) {
                        bool f;
                        const float po = csr_value(t0_col, t0_val, o0, o1, v, &f);
                        const float so = __fsub_rn(__fadd_rn(po, 1.0f), 1.0f);
                        if (f && po > 0.f && so >= th_n && row_find(rowXa, v) < 0 && row_find(rowXb, v) < 0) {
                            keep = true;
                            sa = walk_a ? sw : so;
                            sb = walk_a ? so : sw;
                        }
                    }
                }
                const uint64_t m = __ballot(keep);
                if (keep) {
                    const int64_t dst = t0_base + n_t0 + lanes_below(m, lane);
                    stage_node[dst] = v;
                    stage_pa[dst] = sa;
                    stage_pb[dst] = sb;
                }
                n_t0 += __popcll(m);
            }
        }
        if (lane == 0) {
            stage_cnt[4 * p + 0] = n_cn;
            stage_cnt[4 * p + 1] = n_l1;
            stage_cnt[4 * p + 2] = n_l2;
            stage_cnt[4 * p + 3] = n_t0;
        }
    }
}

// counts -> type_ptr rows (offset by one, ready for scan_blocks_kernel) + block sums + float count features
__global__ __launch_bounds__(256) void select_counts_kernel(int64_t bs, const int32_t *__restrict__ stage_cnt,
                                                            int64_t *__restrict__ type_ptr,
                                                            float *__restrict__ counts_f, int64_t ldc, int want_t0,
                                                            int64_t *__restrict__ blk, int64_t nb) {
    __shared__ int64_t red[4];
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) {
        type_ptr[0] = 0;
        type_ptr[bs + 1] = 0;
        type_ptr[2 * (bs + 1)] = 0;
    }
    int n_cn = 0, n_1 = 0, n_0 = 0;
    if (k < bs) {
        const int4 c4 = *reinterpret_cast<const int4 *>(stage_cnt + 4 * k);
        n_cn = c4.x; n_1 = c4.y + c4.z; n_0 = c4.w;
        type_ptr[k + 1] = n_cn;
        type_ptr[(bs + 1) + k + 1] = n_1;
        type_ptr[2 * (bs + 1) + k + 1] = n_0;
        if (counts_f) {
            float *c = counts_f + k * ldc;
            c[0] = (float)n_cn;
            c[1] = (float)n_1;
            if (want_t0) {
                c[2] = (float)n_0;
                c[3] = (float)(n_cn + n_1);
            } else {
                c[2] = (float)(n_cn + n_1);
            }
        }
    }
    const int64_t s0 = block_sum_i64(n_cn, red), s1 = block_sum_i64(n_1, red), s2 = block_sum_i64(n_0, red);
    if (threadIdx.x == 0) {
        blk[blockIdx.x] = s0;
        blk[nb + blockIdx.x] = s1;
        blk[2 * nb + blockIdx.x] = s2;
    }
}

__global__ __launch_bounds__(256) void select_compact_kernel(
    int64_t bs, const int64_t *__restrict__ batch, int64_t batch_ld, const int64_t *__restrict__ adj_rowptr,
    const int64_t *__restrict__ stage_off, const int32_t *__restrict__ stage_node, const float *__restrict__ stage_pa,
    const float *__restrict__ stage_pb, const int32_t *__restrict__ stage_cnt, const int64_t *__restrict__ type_ptr,
    int32_t *__restrict__ sel_pair, int32_t *__restrict__ sel_node, float *__restrict__ sel_pa,
    float *__restrict__ sel_pb) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t tot_cn = type_ptr[bs], tot_1 = type_ptr[(bs + 1) + bs];
    for (int64_t p = wave_id; p < bs; p += n_waves) {
        const int64_t a = batch[p], b = batch[batch_ld + p];
        const int64_t dA = adj_rowptr[a + 1] - adj_rowptr[a], dB = adj_rowptr[b + 1] - adj_rowptr[b];
        const int64_t s = stage_off[p];
        const int n_cn = stage_cnt[4 * p], n_l1 = stage_cnt[4 * p + 1], n_l2 = stage_cnt[4 * p + 2],
                  n_t0 = stage_cnt[4 * p + 3];
        const int64_t d_cn = type_ptr[p], d_1 = tot_cn + type_ptr[(bs + 1) + p],
                      d_0 = tot_cn + tot_1 + type_ptr[2 * (bs + 1) + p];
        const int64_t l1 = s + dA, l2 = s + 2 * dA, t0 = s + 2 * dA + dB;
        for (int i = lane; i < n_cn; i += 64) {
            sel_pair[d_cn + i] = (int32_t)p;
            sel_node[d_cn + i] = stage_node[s + i];
            sel_pa[d_cn + i] = stage_pa[s + i];
            sel_pb[d_cn + i] = stage_pb[s + i];
        }
        // merge the two sorted, disjoint 1-hop runs: final rank = own index + rank in the other run
        for (int i = lane; i < n_l1; i += 64) {
            const int32_t x = stage_node[l1 + i];
            const int64_t dst = d_1 + i + (lpf_lower_bound(stage_node, l2, l2 + n_l2, x) - l2);
            sel_pair[dst] = (int32_t)p;
            sel_node[dst] = x;
            sel_pa[dst] = stage_pa[l1 + i];
            sel_pb[dst] = stage_pb[l1 + i];
        }
        for (int j = lane; j < n_l2; j += 64) {
            const int32_t y = stage_node[l2 + j];
            const int64_t dst = d_1 + j + (lpf_lower_bound(stage_node, l1, l1 + n_l1, y) - l1);
            sel_pair[dst] = (int32_t)p;
            sel_node[dst] = y;
            sel_pa[dst] = stage_pa[l2 + j];
            sel_pb[dst] = stage_pb[l2 + j];
        }
        for (int i = lane; i < n_t0; i += 64) {
            sel_pair[d_0 + i] = (int32_t)p;
            sel_node[d_0 + i] = stage_node[t0 + i];
            sel_pa[d_0 + i] = stage_pa[t0 + i];
            sel_pb[d_0 + i] = stage_pb[t0 + i];
        }
    }
}

inline unsigned wave_grid(int64_t n_items) {  // 4 waves per block, grid-stride past ~32 blocks per CU
    int64_t blocks = (n_items + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

extern "C" int lpf_select_bound(int64_t bs, const int64_t *batch, int64_t batch_ld, const int64_t *adj_rowptr,
                                const int64_t *t0_rowptr, int64_t *stage_off, int64_t *scratch, void *stream) {
    LPF_REQUIRE(bs >= 0 && stage_off && scratch);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (bs == 0) {
        (void)hipMemsetAsync(stage_off, 0, sizeof(int64_t), s);
        return LPF_OK;
    }
    LPF_REQUIRE(batch && adj_rowptr && batch_ld >= bs);
    const int64_t nb = (bs + 255) / 256;
    hipLaunchKernelGGL(select_bound_kernel, dim3((unsigned)nb), dim3(256), 0, s, bs, batch, batch_ld, adj_rowptr,
                       t0_rowptr, stage_off, scratch);
    hipLaunchKernelGGL(scan_blocks_kernel<1>, dim3((unsigned)nb), dim3(256), 0, s, bs, stage_off, (int64_t)0, scratch,
                       nb);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_select_nodes(int64_t bs, const int64_t *batch, int64_t batch_ld, const int64_t *adj_rowptr,
                                const int32_t *adj_col, const int64_t *adjx_rowptr, const int32_t *adjx_col,
                                const int64_t *ppr_rowptr, const int32_t *ppr_col, const float *ppr_val,
                                const int64_t *t0_rowptr, const int32_t *t0_col, const float *t0_val, float th_cn,
                                float th_1hop, float th_non1hop, const int64_t *stage_off, int32_t *stage_node,
                                float *stage_pa, float *stage_pb, int32_t *stage_cnt, void *stream) {
    if (bs == 0) return LPF_OK;
    LPF_REQUIRE(bs > 0 && batch && batch_ld >= bs && adj_rowptr && adj_col && ppr_rowptr && ppr_col && ppr_val);
    LPF_REQUIRE(stage_off && stage_node && stage_pa && stage_pb && stage_cnt);
    LPF_REQUIRE(!t0_rowptr || (t0_col && t0_val && adjx_rowptr && adjx_col));
    const unsigned sel_blocks = (unsigned)(bs < 256 * 64 ? bs : 256 * 64);  // one wave per block, grid-stride
    hipLaunchKernelGGL(select_nodes_kernel, dim3(sel_blocks), dim3(64), 0, static_cast<hipStream_t>(stream), bs,
                       batch, batch_ld, adj_rowptr, adj_col, adjx_rowptr, adjx_col, ppr_rowptr, ppr_col, ppr_val,
                       t0_rowptr, t0_col, t0_val, th_cn, th_1hop, th_non1hop, stage_off, stage_node, stage_pa,
                       stage_pb, stage_cnt);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_select_scan(int64_t bs, const int32_t *stage_cnt, int64_t *type_ptr, float *counts_f, int64_t ldc,
                               int32_t want_t0, int64_t *scratch, void *stream) {
    LPF_REQUIRE(bs >= 0 && type_ptr && scratch);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (bs == 0) {
        (void)hipMemsetAsync(type_ptr, 0, 3 * sizeof(int64_t), s);
        return LPF_OK;
    }
    LPF_REQUIRE(stage_cnt && (!counts_f || ldc >= (want_t0 ? 4 : 3)));
    const int64_t nb = (bs + 255) / 256;
    hipLaunchKernelGGL(select_counts_kernel, dim3((unsigned)nb), dim3(256), 0, s, bs, stage_cnt, type_ptr, counts_f,
                       ldc, (int)want_t0, scratch, nb);
    hipLaunchKernelGGL(scan_blocks_kernel<3>, dim3((unsigned)nb), dim3(256), 0, s, bs, type_ptr, bs + 1, scratch, nb);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_select_compact(int64_t bs, const int64_t *batch, int64_t batch_ld, const int64_t *adj_rowptr,
                                  const int64_t *stage_off, const int32_t *stage_node, const float *stage_pa,
                                  const float *stage_pb, const int32_t *stage_cnt, const int64_t *type_ptr,
                                  int32_t *sel_pair, int32_t *sel_node, float *sel_pa, float *sel_pb, void *stream) {
    if (bs == 0) return LPF_OK;
    LPF_REQUIRE(bs > 0 && bs < (1ll << 31) && batch && batch_ld >= bs && adj_rowptr && stage_off && stage_node &&
                stage_pa && stage_pb && stage_cnt && type_ptr && sel_pair && sel_node && sel_pa && sel_pb);
    hipLaunchKernelGGL(select_compact_kernel, dim3(wave_grid(bs)), dim3(256), 0, static_cast<hipStream_t>(stream), bs,
                       batch, batch_ld, adj_rowptr, stage_off, stage_node, stage_pa, stage_pb, stage_cnt, type_ptr,
                       sel_pair, sel_node, sel_pa, sel_pb);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}
