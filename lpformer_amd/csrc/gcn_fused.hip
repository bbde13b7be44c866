// One GCN layer in ONE launch, square case (in = out = D <= 128): aggregate first, transform in registers.
//
// Reference: GCN.forward (src/models/other_models.py:61-76) -> PyG GCNConv: lin(x), then propagate with the gcn_norm
// weights, + bias; LayerNorm / ReLU / residual after it; LinkTransformer.propagate's gnn_norm for the last layer
// (src/models/link_transformer.py:110-129).  A_hat (X W^T) = (A_hat X) W^T, so the layer is
//     out[r] = epilogue( (sum_e w_e X[col_e]) W^T )
// and the two launches of the unfused path (gemm_f32.hip: N x D x D product, 118 us on the collab-like graph, then
// spmm_csr.hip: 215 us) with the N x D round trip between them become one gather-bound kernel whose matrix work hides
// under the gather (different rounding order than transform-then-aggregate; same order in every row, every launch).
//
// Layout.  v_mfma_f32_16x16x4_f32 with the NODES ON THE COLUMNS: a wavefront owns a tile of 16 output rows, lane
// (j = lane % 16, q = lane / 16) accumulates row j's input features 16 g + 4 q + 0..3 (g = 0 .. D/16 - 1) -- NT float4
// per lane, each neighbour row fetched as 16-byte pieces, 64 contiguous bytes per (row, g) -- and those accumulators
// ARE the B operands of the product (k = 16 g + 4 q + u for the u-th MFMA of k-group g, the packing of
// lpformer_amd/fold.py pack_dense): no transposition, no LDS staging of activations.  W^T sits packed in LDS (NT
// stages x 8 KB), shared by the workgroup's wavefronts, which otherwise never meet: after the first tile they drift
// apart, some gathering while others multiply.  The product leaves lane (j, q) with out[j][16 c + 4 q + 0..3] for
// c = 0 .. NT-1: the epilogue's LayerNorm is an in-lane sum plus two cross-lane steps, loads and stores are float4.
//
// The 16 rows of a tile advance through their edge lists in lockstep, so tiles are cut from a DEGREE-SORTED row order
// (lpformer_amd/graph.py fused_row_order; any permutation is correct, this one keeps the 16 rows equally long).  Hub
// rows (> 64 entries) are cut into slices of 256 entries that spmm_row_parts_kernel (spmm_csr.hip) sums into a
// compact table, one workgroup per slice; here a hub row is a row whose "neighbours" are its slices, weight 1.
//
// HB (the bf16-table encoder mode): the gathered table holds bf16 rows in the PERMUTED order this kernel's lanes want
// -- element 32 i + 8 q + 4 h + u = feature 16 (2 i + h) + 4 q + u, so a lane's 16-byte load yields two whole
// k-groups -- which is also the order a lane can store its own results in with 16-byte writes: the epilogue writes the
// fp32 rows (normal order: residuals, the final output) AND, on request, that bf16 image for the next layer to gather.
#include <type_traits>

#include "lpf_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GcnFusedArgs {
    int64_t n_tiles;
    const int32_t *row_order;    // [16 n_tiles]: row id >= 0; -1 = padding; <= -2: hub number -2 - v
    int64_t row_base;            // out / residual hold rows row_base ..
    const int64_t *rowptr;
    const int32_t *col;
    const float *w;
    const float *H; int64_t ldh;
    const float *wp;             // pack_dense(W, 1): [NT stages][512] float4
    float *out; int64_t ldo;
    const float *bias, *ln_g, *ln_b;
    const float *residual; int64_t ldr;
    const float *ln2_g, *ln2_b;
    uint32_t flags;
    const int32_t *hubs;         // [n_hub][3]: row id, first slice, number of slices
    const float *t_parts;        // [n_slices][D]: the slices' sums
    float *pre; int64_t ldpre;   // optional: the pre-normalisation rows (product + bias), for a LayerNorm backward
    uint16_t *out_b; int64_t ldob;   // optional (HB): the result rows as permuted bf16 (the next layer's table)
    float *agg; int64_t ldagg;   // optional: the AGGREGATED rows (sum_e w_e H[col_e], before the product), for dW = dU^T agg
    uint32_t drop_thresh;        // training: dropout behind the ReLU (lpf_common.h lpf_drop_bits; 0 = none)
    float drop_scale;            // 1 / (1 - p)
    uint64_t drop_seed;
};

constexpr int GF_THREADS = 512;   // threads of a workgroup
#ifndef LPF_GF_NB
#define LPF_GF_NB 4
#endif
constexpr int GF_NB = LPF_GF_NB;  // neighbours of EACH row of a pair requested per step (x D/32 float4 per lane)
// workgroups per CU (they share nothing but the CU): at D = 128 ONE, i.e. two wavefronts per SIMD with 256 registers
// each -- 32 float4 of neighbour rows in flight per lane beside the accumulators (measured per layer on the
// collab-like graph: 2 workgroups x 2 neighbours 230 us, 1 x 4: 217, 1 x 5: 216, 256 threads 2 x 4: 227,
// 768 threads 1 x 3: 222, 1,024 threads 1 x 2: 231); two below, where the rows are short
template <int NT> constexpr int gf_per_cu() { return NT == 8 ? 1 : 2; }
constexpr int GF_STAGE = 512;    // float4 per packed stage (pack_dense pads a stage to 512)

__device__ __forceinline__ float gf_row_sum(float v) {   // over the four lanes (q) of a row
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

__device__ __forceinline__ float gf_partner(float v) {   // the value lane ^ 1 holds (DPP quad_perm [1,0,3,2])
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
}

template <int NT>
__device__ __forceinline__ void gf_layernorm(f32x4 (&y)[NT], const float *g, const float *b, int q) {
    constexpr float inv_d = 1.0f / (16 * NT);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NT; ++c) s += (y[c][0] + y[c][1]) + (y[c][2] + y[c][3]);
    const float mean = gf_row_sum(s) * inv_d;
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = y[c][r] - mean;
            s2 = fmaf(d, d, s2);
        }
    const float rstd = 1.0f / sqrtf(gf_row_sum(s2) * inv_d + 1e-5f);
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const f32x4 gg = *reinterpret_cast<const f32x4 *>(g + 16 * c + 4 * q);
        const f32x4 bb = *reinterpret_cast<const f32x4 *>(b + 16 * c + 4 * q);
        y[c] = (y[c] - mean) * rstd * gg + bb;
    }
}

// TRAIN: the training launches (lpf_gcn_layer_fused_train_f32) -- the aggregated rows written out, dropout behind the
// ReLU; an instantiation of its own so that the inference kernels keep their registers.
template <int NT, bool HB, bool TRAIN>
__global__ __launch_bounds__(GF_THREADS, GF_THREADS * gf_per_cu<NT>() / 256) void gcn_fused_kernel(const GcnFusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) f32x4 gf_lds[];
    f32x4 *const lw = gf_lds;                                       // [NT][GF_STAGE]
    int *const lticket = reinterpret_cast<int *>(gf_lds + NT * GF_STAGE);
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    for (int i = threadIdx.x; i < NT * GF_STAGE; i += GF_THREADS) lw[i] = reinterpret_cast<const f32x4 *>(A.wp)[i];
    if (threadIdx.x == 0) *lticket = 0;
    __syncthreads();

    // tiles: a workgroup owns every gridDim.x-th one, its wavefronts draw them from a ticket in LDS (first round by
    // position), the next ticket is drawn before the current tile is worked on
    auto draw = [&]() __attribute__((always_inline)) {
        int tk = 0;
        if (lane == 0) tk = atomicAdd(lticket, 1);
        return tk;
    };
    auto tile_of = [&](int k) __attribute__((always_inline)) { return (int64_t)k * gridDim.x + blockIdx.x; };
    int tk_next = draw();
    for (int64_t tile = tile_of(threadIdx.x >> 6); tile < A.n_tiles;) {
        // Rows are gathered in PAIRS: lanes (j, q) and (j ^ 1, q) together fetch whole 128-byte lines -- of row
        // X = the even lane's row, then of row Y = the odd lane's -- the even lane the first 64 bytes of a line
        // (k-group 2L), the odd lane the second (k-group 2L + 1).  With every lane fetching only its own row a request
        // covered 16 rows x 64 bytes, half a cache line each, and the fabric moved 6.0 TB/s where the 256-byte pieces of
        // spmm_csr.hip get 7.1.  Each lane therefore accumulates half of the k-groups of BOTH rows and the two lanes
        // swap the foreign halves (one DPP move per register) before the product.
        struct RowSrc {
            int64_t e;            // first entry of the row's list in col / w (ordinary rows)
            int n_total;          // entries, or slices of a hub row
            int32_t hub_c;        // first slice of a hub row
            bool hub;
            const char *tab;      // table the entries point into (+ this lane's 16 bytes inside a 128-byte line)
            int64_t ldb;          // bytes per table row
        };
        const int odd = j & 1;
        auto source = [&](int code, int64_t &row_out) __attribute__((always_inline)) {
            RowSrc r;
            const bool lv = code != -1;
            r.hub = code < -1;
            r.e = 0;
            r.n_total = 0;
            r.hub_c = 0;
            int64_t row = lv ? code : 0;
            if (r.hub) {
                const int32_t *hb = A.hubs + 3 * (int64_t)(-2 - code);
                row = hb[0];
                r.hub_c = hb[1];
                r.n_total = hb[2];
            } else if (lv) {
                r.e = A.rowptr[row];
                r.n_total = (int)(A.rowptr[row + 1] - r.e);
            }
            // (slice sums are fp32 rows in normal order whatever the table is)
            r.tab = reinterpret_cast<const char *>(r.hub ? A.t_parts : A.H) + 64 * odd + 16 * q;
            r.ldb = r.hub ? 64 * NT : A.ldh * (HB ? 2 : 4);
            row_out = row;
            return r;
        };
        const int roX = A.row_order[tile * 16 + (j & ~1)], roY = A.row_order[tile * 16 + (j | 1)];
        int64_t rowX, rowY;
        const RowSrc X = source(roX, rowX), Y = source(roY, rowY);
        const int ro = odd ? roY : roX;
        const bool live = ro != -1;
        const int64_t row = odd ? rowY : rowX;      // the row this lane multiplies, normalises and stores

        // The gather of one tile from a table of fp32 rows (B16 = false: a lane's 16 bytes of a line are ONE k-group,
        // g = 2 l + odd) or of permuted bf16 rows (B16 = true: TWO k-groups, g = 4 l + 2 odd + h); then the swap: a
        // lane keeps what it gathered of its own row and takes the other half from its partner.
        f32x4 acc[NT];
        auto gather = [&](auto kind) __attribute__((always_inline)) {
            constexpr bool B16 = decltype(kind)::value;
            constexpr int KP = B16 ? 2 : 1;              // k-groups a lane gets from its 16 bytes of a line
            constexpr int NL = NT / (2 * KP);            // 128-byte lines of a row
            f32x4 accX[NL][KP], accY[NL][KP];
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
                for (int h = 0; h < KP; ++h) accX[l][h] = accY[l][h] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifndef GF_NOGATHER
            // NBP neighbours of each of the two rows per step; the (col, weight) pairs of the NEXT step are requested
            // before this step's rows are added.  Entries past the end of a row are (row 0, weight 0).
            auto edge = [&](const RowSrc &r, int k, int32_t &c, float &wv) __attribute__((always_inline)) {
                c = 0;
                wv = 0.f;
                if (k < r.n_total) {
                    if (r.hub) {
                        c = r.hub_c + k;
                        wv = 1.0f;
                    } else {
                        c = A.col[r.e + k];
                        wv = A.w[r.e + k];
                    }
                }
            };
            auto widen = [&](const uint4 raw, f32x4 (&v)[KP]) __attribute__((always_inline)) {
                if constexpr (B16) {
                    v[0] = (f32x4){__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u),
                                   __uint_as_float(raw.y << 16), __uint_as_float(raw.y & 0xffff0000u)};
                    v[1] = (f32x4){__uint_as_float(raw.z << 16), __uint_as_float(raw.z & 0xffff0000u),
                                   __uint_as_float(raw.w << 16), __uint_as_float(raw.w & 0xffff0000u)};
                } else {
                    v[0] = (f32x4){__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z),
                                   __uint_as_float(raw.w)};
                }
            };
            constexpr int NBP = NT == 8 ? GF_NB : GF_NB / 2;   // neighbours per step and row (bf16 table: 6 measured slower than 4)
            int32_t cx[NBP], cy[NBP];
            float wx[NBP], wy[NBP];
#pragma unroll
            for (int i = 0; i < NBP; ++i) {
                edge(X, i, cx[i], wx[i]);
                edge(Y, i, cy[i], wy[i]);
            }
            const int n_max = X.n_total > Y.n_total ? X.n_total : Y.n_total;
            for (int k = 0; __any(k < n_max); k += NBP) {
                uint4 hx[NBP][NL], hy[NBP][NL];
                float ux[NBP], uy[NBP];
#pragma unroll
                for (int i = 0; i < NBP; ++i) {
                    const char *px = X.tab + (int64_t)cx[i] * X.ldb, *py = Y.tab + (int64_t)cy[i] * Y.ldb;
                    ux[i] = wx[i];
                    uy[i] = wy[i];
#pragma unroll
                    for (int l = 0; l < NL; ++l) hx[i][l] = *reinterpret_cast<const uint4 *>(px + 128 * l);
#pragma unroll
                    for (int l = 0; l < NL; ++l) hy[i][l] = *reinterpret_cast<const uint4 *>(py + 128 * l);
                }
#pragma unroll
                for (int i = 0; i < NBP; ++i) {
                    edge(X, k + NBP + i, cx[i], wx[i]);
                    edge(Y, k + NBP + i, cy[i], wy[i]);
                }
#pragma unroll
                for (int i = 0; i < NBP; ++i) {
#pragma unroll
                    for (int l = 0; l < NL; ++l) {
                        f32x4 vx[KP], vy[KP];
                        widen(hx[i][l], vx);
                        widen(hy[i][l], vy);
#pragma unroll
                        for (int h = 0; h < KP; ++h) {
                            accX[l][h] += vx[h] * ux[i];
                            accY[l][h] += vy[h] * uy[i];
                        }
                    }
                }
            }
#endif
            // an even lane keeps accX (its own row) and needs its partner's accX; an odd lane keeps accY and needs its
            // partner's accY: both send what they hold of the OTHER row
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
                for (int h = 0; h < KP; ++h) {
                    const f32x4 send = odd ? accX[l][h] : accY[l][h];
                    const float s0 = send[0], s1 = send[1], s2 = send[2], s3 = send[3];
                    const f32x4 recv = {gf_partner(s0), gf_partner(s1), gf_partner(s2), gf_partner(s3)};
                    const int g_even = 2 * KP * l + h, g_odd = 2 * KP * l + KP + h;     // k-groups of the two halves
                    acc[g_even] = odd ? recv : accX[l][h];
                    acc[g_odd] = odd ? accY[l][h] : recv;
                }
        };
        if constexpr (HB) {
            // (hub rows come first in the order, in tiles of their own: graph.fused_row_order pads them to 16)
            if (__any(X.hub || Y.hub)) gather(std::false_type{});
            else gather(std::true_type{});
        } else {
            gather(std::false_type{});
        }

        const int64_t orow = row - A.row_base;
        if constexpr (TRAIN) {
            if (A.agg && live) {     // (the training forward: the weight gradient of the layer is dU^T agg)
                float *gp = A.agg + orow * A.ldagg + 4 * q;
#pragma unroll
                for (int g = 0; g < NT; ++g) *reinterpret_cast<f32x4 *>(gp + 16 * g) = acc[g];
            }
        }

        // out^T tile = W . acc: NT output tiles x NT k-groups x 4 MFMAs
        f32x4 y[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) y[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef GF_NOMFMA
#pragma unroll
        for (int c = 0; c < NT; ++c) y[c] = acc[c];
#else
#pragma unroll
        for (int g = 0; g < NT; ++g) {
            const f32x4 *lg = lw + g * GF_STAGE + lane;
#pragma unroll
            for (int c0_ = 0; c0_ < NT; c0_ += 4) {      // four tiles' operands at a time
                f32x4 a[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] = (c0_ + c < NT) ? lg[(c0_ + c) * 64] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c0_ + c < NT)
                            y[c0_ + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][u], acc[g][u], y[c0_ + c], 0, 0, 0);
                // (keeps the operand reads of later blocks behind these MFMAs: hoisted together they spill)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif

        // epilogue: + bias, LayerNorm, ReLU, + residual, gnn_norm; lane (j, q) holds out[row j][16 c + 4 q + 0..3]
        if (A.bias) {
#pragma unroll
            for (int c = 0; c < NT; ++c) y[c] += *reinterpret_cast<const f32x4 *>(A.bias + 16 * c + 4 * q);
        }
        if (A.pre && live) {
            float *pp = A.pre + orow * A.ldpre + 4 * q;
#pragma unroll
            for (int c = 0; c < NT; ++c) *reinterpret_cast<f32x4 *>(pp + 16 * c) = y[c];
        }
        if (A.ln_g) gf_layernorm<NT>(y, A.ln_g, A.ln_b, q);
        if (A.flags & LPF_FLAG_RELU) {
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) y[c][r] = fmaxf(y[c][r], 0.f);
        }
        if constexpr (TRAIN) {
            if (A.drop_thresh) {     // (training: F.dropout behind the layer, other_models.py:69; the mask is a function of
                                     //  (seed, row, feature) that the LayerNorm/ReLU backward recomputes)
                const uint32_t rk = lpf_drop_row_key(row, A.drop_seed);
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        y[c][r] = lpf_drop_bits(rk, 16 * c + 4 * q + r, A.drop_seed) >= A.drop_thresh ? y[c][r] * A.drop_scale : 0.f;
            }
        }
        if (A.residual && live) {
            const float *rp = A.residual + orow * A.ldr + 4 * q;
#pragma unroll
            for (int c = 0; c < NT; ++c) y[c] += *reinterpret_cast<const f32x4 *>(rp + 16 * c);
        }
        if (A.ln2_g) gf_layernorm<NT>(y, A.ln2_g, A.ln2_b, q);
        if (live) {
            float *op = A.out + orow * A.ldo + 4 * q;
#pragma unroll
            for (int c = 0; c < NT; ++c) *reinterpret_cast<f32x4 *>(op + 16 * c) = y[c];
            if constexpr (HB) {
                if (A.out_b) {      // elements 32 i + 8 q .. + 7 = features 16 (2 i) + 4 q + u, 16 (2 i + 1) + 4 q + u
                    uint16_t *ob = A.out_b + orow * A.ldob + 8 * q;
#pragma unroll
                    for (int i = 0; i < NT / 2; ++i) {
                        const f32x4 lo = y[2 * i], hi = y[2 * i + 1];
                        uint4 pk;
                        pk.x = (uint32_t)lpf_f32_to_bf16(lo[0]) | ((uint32_t)lpf_f32_to_bf16(lo[1]) << 16);
                        pk.y = (uint32_t)lpf_f32_to_bf16(lo[2]) | ((uint32_t)lpf_f32_to_bf16(lo[3]) << 16);
                        pk.z = (uint32_t)lpf_f32_to_bf16(hi[0]) | ((uint32_t)lpf_f32_to_bf16(hi[1]) << 16);
                        pk.w = (uint32_t)lpf_f32_to_bf16(hi[2]) | ((uint32_t)lpf_f32_to_bf16(hi[3]) << 16);
                        *reinterpret_cast<uint4 *>(ob + 32 * i) = pk;
                    }
                }
            }
        }
        tile = tile_of(GF_THREADS / 64 + __builtin_amdgcn_readfirstlane(tk_next));
        if (tile < A.n_tiles) tk_next = draw();
    }
}

}  // namespace

namespace {
template <bool HB, bool TRAIN = false>
int gf_launch(int32_t D, int64_t n_tiles, const int32_t *row_order, int64_t row_base, const int64_t *rowptr,
              const int32_t *col, const float *w, const void *H, int64_t ldh, const float *w_packed, float *out,
              int64_t ldo, const float *bias, const float *ln_g, const float *ln_b, const float *residual, int64_t ldr,
              const float *ln2_g, const float *ln2_b, uint32_t flags, const int32_t *hubs, const float *t_parts,
              float *pre_out, int64_t ldpre, void *out_b, int64_t ldob, float *agg_out, int64_t ldagg, float drop_p,
              uint64_t drop_seed, void *stream) {
    if (n_tiles == 0) return LPF_OK;
    LPF_REQUIRE(n_tiles > 0 && row_order && rowptr && col && w && H && w_packed && out);
    if (D != 32 && D != 64 && D != 128) return LPF_ERR_UNSUPPORTED;
    if (HB && D == 32) return LPF_ERR_UNSUPPORTED;      // (a bf16 row of 64 bytes is half a line)
    LPF_REQUIRE(ldh >= D && ldo >= D && (ldh & (HB ? 7 : 3)) == 0 && (ldo & 3) == 0 && lpf_aligned16(H) &&
                lpf_aligned16(out) && lpf_aligned16(w_packed));
    LPF_REQUIRE((!ln_g) == (!ln_b) && (!ln2_g) == (!ln2_b) && (!hubs) == (!t_parts));
    LPF_REQUIRE(!pre_out || (ldpre >= D && (ldpre & 3) == 0 && lpf_aligned16(pre_out)));
    LPF_REQUIRE(!out_b || (HB && ldob >= D && (ldob & 7) == 0 && lpf_aligned16(out_b)));
    LPF_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || TRAIN));
    LPF_REQUIRE(!agg_out || (TRAIN && ldagg >= D && (ldagg & 3) == 0 && lpf_aligned16(agg_out)));
    LPF_REQUIRE(!residual || ((ldr & 3) == 0 && ldr >= D && lpf_aligned16(residual)));
    LPF_REQUIRE((!bias || lpf_aligned16(bias)) && (!ln_g || (lpf_aligned16(ln_g) && lpf_aligned16(ln_b))) &&
                (!ln2_g || (lpf_aligned16(ln2_g) && lpf_aligned16(ln2_b))) && (!t_parts || lpf_aligned16(t_parts)));
    const GcnFusedArgs a{n_tiles, row_order, row_base, rowptr, col, w, static_cast<const float *>(H), ldh, w_packed, out,
                         ldo, bias, ln_g, ln_b, residual, ldr, ln2_g, ln2_b, flags, hubs, t_parts, pre_out, ldpre,
                         static_cast<uint16_t *>(out_b), ldob, agg_out, ldagg, lpf_drop_threshold(drop_p),
                         1.0f / (1.0f - drop_p), drop_seed};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int n_cu = lpf_cu_count();
    if (n_cu == 0) return LPF_ERR_NO_DEVICE;
    // persistent workgroups
    const int64_t want = (n_tiles + GF_THREADS / 64 - 1) / (GF_THREADS / 64);
#define LPF_GF(NT)                                                                                                  \
    do {                                                                                                            \
        auto kern = gcn_fused_kernel<NT, HB, TRAIN>;                                                                       \
        constexpr size_t lds = (size_t)(NT * GF_STAGE + 1) * sizeof(f32x4);                                         \
        LPF_SET_MAX_LDS(kern, lds);                                                                                 \
        const int64_t cap = (int64_t)gf_per_cu<NT>() * n_cu;                                                       \
        hipLaunchKernelGGL(kern, dim3((unsigned)(want < cap ? want : cap)), dim3(GF_THREADS), lds, s, a);           \
    } while (0)
    switch (D) {
        case 32:
            if constexpr (!HB) LPF_GF(2);
            break;
        case 64: LPF_GF(4); break;
        default: LPF_GF(8); break;
    }
#undef LPF_GF
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}
}  // namespace

extern "C" int lpf_gcn_layer_fused_f32(int32_t D, int64_t n_tiles, const int32_t *row_order, int64_t row_base,
                                       const int64_t *rowptr, const int32_t *col, const float *w, const float *H,
                                       int64_t ldh, const float *w_packed, float *out, int64_t ldo, const float *bias,
                                       const float *ln_g, const float *ln_b, const float *residual, int64_t ldr,
                                       const float *ln2_g, const float *ln2_b, uint32_t flags,
                                       const int32_t *hubs, const float *t_parts, float *pre_out, int64_t ldpre,
                                       void *stream) {
    return gf_launch<false>(D, n_tiles, row_order, row_base, rowptr, col, w, H, ldh, w_packed, out, ldo, bias, ln_g, ln_b,
                            residual, ldr, ln2_g, ln2_b, flags, hubs, t_parts, pre_out, ldpre, nullptr, 0, nullptr, 0, 0.f, 0, stream);
}

extern "C" int lpf_gcn_layer_fused_train_f32(int32_t D, int64_t n_tiles, const int32_t *row_order, int64_t row_base,
                                             const int64_t *rowptr, const int32_t *col, const float *w, const float *H,
                                             int64_t ldh, const float *w_packed, float *out, int64_t ldo,
                                             const float *bias, const float *ln_g, const float *ln_b,
                                             const float *residual, int64_t ldr, uint32_t flags, const int32_t *hubs,
                                             const float *t_parts, float *pre_out, int64_t ldpre, float *agg_out,
                                             int64_t ldagg, float drop_p, uint64_t drop_seed, void *stream) {
    return gf_launch<false, true>(D, n_tiles, row_order, row_base, rowptr, col, w, H, ldh, w_packed, out, ldo, bias, ln_g, ln_b,
                            residual, ldr, nullptr, nullptr, flags, hubs, t_parts, pre_out, ldpre, nullptr, 0, agg_out,
                            ldagg, drop_p, drop_seed, stream);
}

extern "C" int lpf_gcn_layer_fused_bf16(int32_t D, int64_t n_tiles, const int32_t *row_order, int64_t row_base,
                                        const int64_t *rowptr, const int32_t *col, const float *w, const void *H_bf16p,
                                        int64_t ldh, const float *w_packed, float *out, int64_t ldo, const float *bias,
                                        const float *ln_g, const float *ln_b, const float *residual, int64_t ldr,
                                        const float *ln2_g, const float *ln2_b, uint32_t flags, const int32_t *hubs,
                                        const float *t_parts, void *out_bf16p, int64_t ldob, void *stream) {
    return gf_launch<true>(D, n_tiles, row_order, row_base, rowptr, col, w, H_bf16p, ldh, w_packed, out, ldo, bias, ln_g,
                           ln_b, residual, ldr, ln2_g, ln2_b, flags, hubs, t_parts, nullptr, 0, out_bf16p, ldob, nullptr, 0, 0.f, 0, stream);
}
