// GCN aggregation on gfx950: CSR SpMM with the whole per-layer epilogue fused, plus the one-time GCN normalisation.
//
// Layout: a row of D fp32 features is covered by G = D/4 lanes (16 B per lane, so one row read is one or two full
// 128-byte lines per group); a 64-lane wave therefore works on 64/G rows at once (D=64 -> 4 rows, D=128 -> 2,
// D=256 -> 1).  A group reads G edge (col, weight) pairs with one coalesced load, then broadcasts them lane by lane
// (ds_bpermute) and gathers the neighbour rows four at a time so four 16-byte gathers are always in flight.  The
// accumulator never leaves registers: bias, LayerNorm (group-wide butterfly reduction), ReLU, residual and the
// final `gnn_norm` LayerNorm are applied before the single store.  Bound: HBM / L2 gather bandwidth.
#include "lpf_common.h"

namespace {

// A lane holds V consecutive float4 (V = 1: fp32 table, 16 bytes gathered per lane; V = 2: bf16 table, 8 bf16 = 16
// bytes gathered per lane, so a row needs half the lanes and a wavefront works on twice the rows).
template <int G, int V>
__device__ __forceinline__ void group_layernorm(float4 (&y)[V], bool act, int D, const float *__restrict__ g,
                                                const float *__restrict__ b, int off) {
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v) s += act ? (y[v].x + y[v].y + y[v].z + y[v].w) : 0.f;
    const float mean = lpf_group_sum<G>(s) / (float)D;
    float q = 0.f;
    float4 d[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        d[v] = make_float4(y[v].x - mean, y[v].y - mean, y[v].z - mean, y[v].w - mean);
        q += act ? (d[v].x * d[v].x + d[v].y * d[v].y + d[v].z * d[v].z + d[v].w * d[v].w) : 0.f;
    }
    const float var = lpf_group_sum<G>(q) / (float)D;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    if (act) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float4 gg = *reinterpret_cast<const float4 *>(g + off + 4 * v);
            const float4 bb = *reinterpret_cast<const float4 *>(b + off + 4 * v);
            y[v].x = d[v].x * rstd * gg.x + bb.x;
            y[v].y = d[v].y * rstd * gg.y + bb.y;
            y[v].z = d[v].z * rstd * gg.z + bb.z;
            y[v].w = d[v].w * rstd * gg.w + bb.w;
        }
    }
}

constexpr int SPMM_LONG = 128;  // rows with more stored entries than this are left to spmm_long_rows_kernel

// acc += sum over edges [e0, e1) taken in chunks of G starting at e0 + first_chunk*G with stride chunk_stride*G:
// a chunk is one coalesced (col, weight) read by the group, then lane-by-lane broadcasts with four 16-byte neighbour
// gathers in flight.  (V = 2: the gathered table H holds bf16 rows -- half the gather bytes, the bound of this kernel;
// the sum stays fp32.)
template <int G, int V, bool HB>
__device__ __forceinline__ void spmm_accumulate(float4 (&acc)[V], int64_t e0, int64_t e1, int first_chunk,
                                                int chunk_stride, const int32_t *__restrict__ col,
                                                const float *__restrict__ w, const float *__restrict__ H, int64_t ldh,
                                                int off, bool act, int gbase, int lig) {
    for (int64_t e = e0 + (int64_t)first_chunk * G; e < e1; e += (int64_t)chunk_stride * G) {
        const int64_t mine = e + lig;
        int32_t c = 0;
        float wv = 0.f;
        if (mine < e1) {
            c = col[mine];
            wv = w[mine];
        }
        const int cnt = (int)((e1 - e) < G ? (e1 - e) : G);
        for (int t = 0; t < cnt; t += 4) {  // lanes past `cnt` carry (col 0, weight 0): harmless gathers
            int32_t cc[4];
            float ww[4];
            float4 h[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                cc[u] = __shfl(c, gbase + ((t + u) & (G - 1)), 64);
                ww[u] = __shfl(wv, gbase + ((t + u) & (G - 1)), 64);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if constexpr (HB) {
                    static_assert(V == 2, "a bf16 row piece of 16 bytes is two float4 accumulators");
                    uint4 b = make_uint4(0u, 0u, 0u, 0u);
                    if (act) b = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(H) +
                                                                  (int64_t)cc[u] * ldh + off);
                    h[u][0] = make_float4(__uint_as_float(b.x << 16), __uint_as_float(b.x & 0xffff0000u),
                                          __uint_as_float(b.y << 16), __uint_as_float(b.y & 0xffff0000u));
                    h[u][1] = make_float4(__uint_as_float(b.z << 16), __uint_as_float(b.z & 0xffff0000u),
                                          __uint_as_float(b.w << 16), __uint_as_float(b.w & 0xffff0000u));
                } else {
#pragma unroll
                    for (int v = 0; v < V; ++v)
                        h[u][v] = act ? *reinterpret_cast<const float4 *>(H + (int64_t)cc[u] * ldh + off + 4 * v)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    acc[v].x = fmaf(ww[u], h[u][v].x, acc[v].x);
                    acc[v].y = fmaf(ww[u], h[u][v].y, acc[v].y);
                    acc[v].z = fmaf(ww[u], h[u][v].z, acc[v].z);
                    acc[v].w = fmaf(ww[u], h[u][v].w, acc[v].w);
                }
        }
    }
}

// fused epilogue (GCN.forward lines after conv(); propagate's gnn_norm for the last layer); result stored by the group
template <int G, int V>
__device__ __forceinline__ void spmm_epilogue(float4 (&y)[V], int64_t row, bool live, bool act, int D, int off,
                                              float *__restrict__ out, int64_t ldo, const float *__restrict__ bias,
                                              const float *__restrict__ ln_g, const float *__restrict__ ln_b,
                                              const float *__restrict__ residual, int64_t ldr,
                                              const float *__restrict__ ln2_g, const float *__restrict__ ln2_b,
                                              uint32_t flags) {
    if (bias && act) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float4 bv = *reinterpret_cast<const float4 *>(bias + off + 4 * v);
            y[v].x += bv.x; y[v].y += bv.y; y[v].z += bv.z; y[v].w += bv.w;
        }
    }
    if (ln_g) group_layernorm<G, V>(y, act, D, ln_g, ln_b, off);
    if (flags & LPF_FLAG_RELU) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            y[v].x = fmaxf(y[v].x, 0.f); y[v].y = fmaxf(y[v].y, 0.f);
            y[v].z = fmaxf(y[v].z, 0.f); y[v].w = fmaxf(y[v].w, 0.f);
        }
    }
    if (residual && act && live) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float4 rv = *reinterpret_cast<const float4 *>(residual + row * ldr + off + 4 * v);
            y[v].x += rv.x; y[v].y += rv.y; y[v].z += rv.z; y[v].w += rv.w;
        }
    }
    if (ln2_g) group_layernorm<G, V>(y, act, D, ln2_g, ln2_b, off);
    if (act && live) {
#pragma unroll
        for (int v = 0; v < V; ++v) *reinterpret_cast<float4 *>(out + row * ldo + off + 4 * v) = y[v];
    }
}

template <int G, int V, bool HB>
__global__ __launch_bounds__(256) void spmm_csr_kernel(int64_t n, int D, const int64_t *__restrict__ rowptr,
                                                       const int32_t *__restrict__ col, const float *__restrict__ w,
                                                       const float *__restrict__ H, int64_t ldh,
                                                       float *__restrict__ out, int64_t ldo,
                                                       const float *__restrict__ bias, const float *__restrict__ ln_g,
                                                       const float *__restrict__ ln_b,
                                                       const float *__restrict__ residual, int64_t ldr,
                                                       const float *__restrict__ ln2_g,
                                                       const float *__restrict__ ln2_b, uint32_t flags,
                                                       int skip_long) {
    constexpr int RPW = 64 / G;
    const int lane = threadIdx.x & 63;
    const int grp = lane / G, lig = lane % G;
    const int gbase = grp * G;
    const int off = 4 * V * lig;
    const bool act = off < D;
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);

    for (int64_t row0 = wave_id * RPW; row0 < n; row0 += n_waves * RPW) {
        const int64_t row = row0 + grp;
        bool live = row < n;
        int64_t e0 = 0, e1 = 0;
        if (live) {
            e0 = rowptr[row];
            e1 = rowptr[row + 1];
        }
        if (skip_long && e1 - e0 > SPMM_LONG) {  // a hub row would hold this group for a long time
            live = false;
            e1 = e0;
        }
        float4 acc[V];
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        spmm_accumulate<G, V, HB>(acc, e0, e1, 0, 1, col, w, H, ldh, off, act, gbase, lig);
        spmm_epilogue<G, V>(acc, row, live, act, D, off, out, ldo, bias, ln_g, ln_b, residual, ldr, ln2_g, ln2_b,
                            flags);
    }
}

// Hub rows (more than SPMM_LONG entries): one 256-thread workgroup per row.  The NG = 256/G lane groups take chunks of
// G edges round-robin, their partial sums meet in LDS and are added in group order (deterministic), then group 0 runs
// the same fused epilogue.
template <int G, int V, bool HB>
__global__ __launch_bounds__(256) void spmm_long_rows_kernel(
    const int32_t *__restrict__ long_rows, int D, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ w, const float *__restrict__ H, int64_t ldh, float *__restrict__ out, int64_t ldo,
    const float *__restrict__ bias, const float *__restrict__ ln_g, const float *__restrict__ ln_b,
    const float *__restrict__ residual, int64_t ldr, const float *__restrict__ ln2_g, const float *__restrict__ ln2_b,
    uint32_t flags) {
    constexpr int NG = 256 / G;
    __shared__ float4 part[NG][G][V];
    const int tid = threadIdx.x, grp = tid / G, lig = tid % G;
    const int gbase = ((tid & 63) / G) * G;
    const int off = 4 * V * lig;
    const bool act = off < D;
    const int64_t row = long_rows[blockIdx.x];
    const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
    float4 acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    spmm_accumulate<G, V, HB>(acc, e0, e1, grp, NG, col, w, H, ldh, off, act, gbase, lig);
#pragma unroll
    for (int v = 0; v < V; ++v) part[grp][lig][v] = acc[v];
    __syncthreads();
    if (grp == 0) {
        float4 y[V];
#pragma unroll
        for (int v = 0; v < V; ++v) y[v] = part[0][lig][v];
        for (int g = 1; g < NG; ++g) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const float4 x = part[g][lig][v];
                y[v].x += x.x; y[v].y += x.y; y[v].z += x.z; y[v].w += x.w;
            }
        }
        spmm_epilogue<G, V>(y, row, true, act, D, off, out, ldo, bias, ln_g, ln_b, residual, ldr, ln2_g, ln2_b, flags);
    }
}

// Slices of hub rows for gcn_fused.hip: workgroup p sums the stored entries [parts[2p], parts[2p+1]) of one row -- its
// 256 / G lane groups take chunks of G entries round-robin, the partial sums are added in group order -- and writes the
// plain sum (no epilogue) to row p of a compact table.
// HB: the table holds bf16 rows in gcn_fused.hip's permuted order (element 32 i + 8 q + 4 h + u = feature
// 16 (2 i + h) + 4 q + u); the sums are elementwise, and are stored as fp32 in NORMAL order.
template <int G, int V, bool HB = false>
__global__ __launch_bounds__(256) void spmm_row_parts_kernel(const int64_t *__restrict__ parts, int D,
                                                             const int32_t *__restrict__ col,
                                                             const float *__restrict__ w, const float *__restrict__ H,
                                                             int64_t ldh, float *__restrict__ out) {
    constexpr int NG = 256 / G;
    __shared__ float4 part[NG][G][V];
    const int tid = threadIdx.x, grp = tid / G, lig = tid % G;
    const int gbase = ((tid & 63) / G) * G;
    const int off = 4 * V * lig;
    const bool act = off < D;
    const int64_t e0 = parts[2 * (int64_t)blockIdx.x], e1 = parts[2 * (int64_t)blockIdx.x + 1];
    float4 acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    spmm_accumulate<G, V, HB>(acc, e0, e1, grp, NG, col, w, H, ldh, off, act, gbase, lig);
#pragma unroll
    for (int v = 0; v < V; ++v) part[grp][lig][v] = acc[v];
    __syncthreads();
    if (grp == 0 && act) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float4 y = part[0][lig][v];
            for (int g = 1; g < NG; ++g) {
                const float4 x = part[g][lig][v];
                y.x += x.x; y.y += x.y; y.z += x.z; y.w += x.w;
            }
            // (HB: lane lig holds elements 8 lig .. + 7 = i = lig / 4, q = lig % 4, h = v)
            const int dst = HB ? 32 * (lig >> 2) + 16 * v + 4 * (lig & 3) : off + 4 * v;
            *reinterpret_cast<float4 *>(out + (int64_t)blockIdx.x * D + dst) = y;
        }
    }
}

// deg^-1/2 with the diagonal forced to weight 1 (torch_sparse.fill_diag semantics).  One WAVEFRONT per row, lanes strided
// over the row's entries, the 64 partial sums added in a fixed butterfly order (deterministic).  (Until round 6 one
// THREAD per row: the launch lasted as long as the longest hub row read one entry at a time -- 0.41 + 0.63 ms for the two
// kernels on the collab-like graph, paid per batch by a training loop that overrides the propagation matrix.)
__global__ __launch_bounds__(256) void gcn_deg_kernel(int64_t n, const int64_t *__restrict__ rowptr,
                                                      const int32_t *__restrict__ col, const float *__restrict__ w,
                                                      float *__restrict__ dis) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t lo = rowptr[i], hi = rowptr[i + 1];
    float deg = 0.f;
    for (int64_t e = lo + lane; e < hi; e += 64) deg += (col[e] == i) ? 1.0f : (w ? w[e] : 1.0f);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) deg += __shfl_xor(deg, m, 64);
    if (lane == 0) {
        const float d = powf(deg, -0.5f);
        dis[i] = isinf(d) ? 0.f : d;
    }
}

__global__ __launch_bounds__(256) void gcn_scale_kernel(int64_t n, const int64_t *__restrict__ rowptr,
                                                        const int32_t *__restrict__ col, const float *__restrict__ w,
                                                        const float *__restrict__ dis, float *__restrict__ w_out) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float di = dis[i];
    const int64_t lo = rowptr[i], hi = rowptr[i + 1];
    for (int64_t e = lo + lane; e < hi; e += 64) {
        const int32_t c = col[e];
        const float v = (c == i) ? 1.0f : (w ? w[e] : 1.0f);
        w_out[e] = (v * di) * dis[c];  // same association as gcn_norm: (w * dis[row]) * dis[col]
    }
}

}  // namespace

extern "C" int lpf_gcn_norm_csr(int64_t n, const int64_t *rowptr, const int32_t *col, const float *w_in,
                                float *w_out, float *dis_tmp, void *stream) {
    if (n == 0) return LPF_OK;
    LPF_REQUIRE(n > 0 && rowptr && col && w_out && dis_tmp);
    hipStream_t s = static_cast<hipStream_t>(stream);
    LPF_REQUIRE(n < (1ll << 33));
    const unsigned blocks = (unsigned)((n + 3) / 4);     // four rows (wavefronts) per workgroup
    hipLaunchKernelGGL(gcn_deg_kernel, dim3(blocks), dim3(256), 0, s, n, rowptr, col, w_in, dis_tmp);
    hipLaunchKernelGGL(gcn_scale_kernel, dim3(blocks), dim3(256), 0, s, n, rowptr, col, w_in, dis_tmp, w_out);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

namespace {
template <bool HB>
int spmm_launch(int64_t n, int32_t D, const int64_t *rowptr, const int32_t *col, const float *w, const float *H,
                int64_t ldh, float *out, int64_t ldo, const float *bias, const float *ln_g, const float *ln_b,
                const float *residual, int64_t ldr, const float *ln2_g, const float *ln2_b, uint32_t flags,
                const int32_t *long_rows, int64_t n_long, void *stream) {
    if (n == 0) return LPF_OK;
    LPF_REQUIRE(n_long >= 0 && (n_long == 0 || long_rows) && n_long < (1ll << 31));
    LPF_REQUIRE(n > 0 && rowptr && col && w && H && out);
    if (D <= 0 || (D & (HB ? 7 : 3)) || D > 256) return LPF_ERR_UNSUPPORTED;
    LPF_REQUIRE(ldh >= D && ldo >= D && (ldh & 3) == 0 && (ldo & 3) == 0 && lpf_aligned16(H) && lpf_aligned16(out));
    LPF_REQUIRE((!ln_g) == (!ln_b) && (!ln2_g) == (!ln2_b));
    LPF_REQUIRE(!residual || ((ldr & 3) == 0 && ldr >= D && lpf_aligned16(residual)));
    LPF_REQUIRE(!bias || lpf_aligned16(bias));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // float4 per lane: 2 whenever D allows it -- the bf16 table then gives 16 gathered bytes per lane, the fp32 table 32
    // (twice the rows per wavefront: 235 -> 219 us per layer on the collab-like graph; -DLPF_SPMM_V=1 builds the
    // one-float4 shape for comparison)
#ifndef LPF_SPMM_V
#define LPF_SPMM_V 2
#endif
    constexpr int v_f32 = LPF_SPMM_V;
    const int V = HB ? 2 : (((D & 7) || D > 128) ? 1 : v_f32);  // (D = 256, hub rows only: one float4 measured faster)
    const int G = (D <= 64 ? 16 : (D <= 128 ? 32 : 64)) / V;
    const int rpw = 64 / G;
    int64_t blocks = (n + 4 * rpw - 1) / (4 * rpw);
    if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond ~32 blocks per CU
#define LPF_SPMM_LAUNCH(GG, VV)                                                                                    \
    do {                                                                                                           \
        hipLaunchKernelGGL((spmm_csr_kernel<GG, VV, HB>), dim3((unsigned)blocks), dim3(256), 0, s, n, D, rowptr,   \
                           col, w, H, ldh, out, ldo, bias, ln_g, ln_b, residual, ldr, ln2_g, ln2_b, flags,         \
                           long_rows ? 1 : 0);                                                                     \
        if (n_long > 0)                                                                                            \
            hipLaunchKernelGGL((spmm_long_rows_kernel<GG, VV, HB>), dim3((unsigned)n_long), dim3(256), 0, s,       \
                               long_rows, D, rowptr, col, w, H, ldh, out, ldo, bias, ln_g, ln_b, residual, ldr,    \
                               ln2_g, ln2_b, flags);                                                               \
    } while (0)
    if (V == 2) {
        if (G == 8) LPF_SPMM_LAUNCH(8, 2);
        else if (G == 16) LPF_SPMM_LAUNCH(16, 2);
        else LPF_SPMM_LAUNCH(32, 2);
    } else {
        if constexpr (!HB) {
            if (G == 16) LPF_SPMM_LAUNCH(16, 1);
            else if (G == 32) LPF_SPMM_LAUNCH(32, 1);
            else LPF_SPMM_LAUNCH(64, 1);
        }
    }
#undef LPF_SPMM_LAUNCH
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}
}  // namespace

extern "C" int lpf_spmm_csr_f32(int64_t n, int32_t D, const int64_t *rowptr, const int32_t *col, const float *w,
                                const float *H, int64_t ldh, float *out, int64_t ldo, const float *bias,
                                const float *ln_g, const float *ln_b, const float *residual, int64_t ldr,
                                const float *ln2_g, const float *ln2_b, uint32_t flags, const int32_t *long_rows,
                                int64_t n_long, void *stream) {
    return spmm_launch<false>(n, D, rowptr, col, w, H, ldh, out, ldo, bias, ln_g, ln_b, residual, ldr, ln2_g, ln2_b,
                              flags, long_rows, n_long, stream);
}

extern "C" int lpf_spmm_row_parts_f32(int32_t D, const int64_t *parts, int64_t n_parts, const int32_t *col, const float *w,
                                      const float *H, int64_t ldh, float *out, void *stream) {
    if (n_parts == 0) return LPF_OK;
    LPF_REQUIRE(n_parts > 0 && n_parts < (1ll << 31) && parts && col && w && H && out);
    if (D <= 0 || (D & 7) || D > 128) return LPF_ERR_UNSUPPORTED;
    LPF_REQUIRE(ldh >= D && (ldh & 3) == 0 && lpf_aligned16(H) && lpf_aligned16(out));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (D <= 64)
        hipLaunchKernelGGL((spmm_row_parts_kernel<8, 2>), dim3((unsigned)n_parts), dim3(256), 0, s, parts, D, col, w, H,
                           ldh, out);
    else
        hipLaunchKernelGGL((spmm_row_parts_kernel<16, 2>), dim3((unsigned)n_parts), dim3(256), 0, s, parts, D, col, w, H,
                           ldh, out);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_spmm_row_parts_bf16p(int32_t D, const int64_t *parts, int64_t n_parts, const int32_t *col,
                                        const float *w, const void *H_bf16p, int64_t ldh, float *out, void *stream) {
    if (n_parts == 0) return LPF_OK;
    LPF_REQUIRE(n_parts > 0 && n_parts < (1ll << 31) && parts && col && w && H_bf16p && out);
    if (D != 64 && D != 128) return LPF_ERR_UNSUPPORTED;
    LPF_REQUIRE(ldh >= D && (ldh & 7) == 0 && lpf_aligned16(H_bf16p) && lpf_aligned16(out));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float *H = static_cast<const float *>(H_bf16p);
    if (D == 64)
        hipLaunchKernelGGL((spmm_row_parts_kernel<8, 2, true>), dim3((unsigned)n_parts), dim3(256), 0, s, parts, D, col, w,
                           H, ldh, out);
    else
        hipLaunchKernelGGL((spmm_row_parts_kernel<16, 2, true>), dim3((unsigned)n_parts), dim3(256), 0, s, parts, D, col,
                           w, H, ldh, out);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}

extern "C" int lpf_spmm_csr_bf16(int64_t n, int32_t D, const int64_t *rowptr, const int32_t *col, const float *w,
                                 const void *H_bf16, int64_t ldh, float *out, int64_t ldo, const float *bias,
                                 const float *ln_g, const float *ln_b, const float *residual, int64_t ldr,
                                 const float *ln2_g, const float *ln2_b, uint32_t flags, const int32_t *long_rows,
                                 int64_t n_long, void *stream) {
    LPF_REQUIRE((ldh & 7) == 0);  // 16-byte aligned bf16 rows
    return spmm_launch<true>(n, D, rowptr, col, w, static_cast<const float *>(H_bf16), ldh, out, ldo, bias, ln_g, ln_b,
                             residual, ldr, ln2_g, ln2_b, flags, long_rows, n_long, stream);
}
