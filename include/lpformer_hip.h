/*
 * lpformer_hip.h -- C ABI of the MI355X (gfx950) LPFormer link-scoring library.
 *
 * Two shared objects export these symbols:
 *   liblpformer_hip.so   (hipcc, --offload-arch=gfx950)  every entry point taking a stream
 *   liblpformer_host.so  (g++ -fopenmp)                   lpf_ppr_push_cpu, lpf_host_free, lpf_host_abi_version
 *
 * The reference (HarryShomer/LPFormer) is 100 % Python and has no FFI of its own; each entry point
 * below therefore cites the reference Python it replaces (paths relative to the reference repo root)
 * and INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every buffer (device pointers unless a name ends in
 *     _host); no allocation, no exceptions, no hidden state: re-entrant per stream.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).  Nothing synchronises the host
 *     (graph-capturable) -- except the *_host functions, which run on the calling thread(s).
 *   - return value: LPF_OK or a negative LPF_ERR_* code; lpf_strerror() names it.
 *   - CSR graphs: rowptr int64[n+1], col int32[nnz] sorted ascending inside each row, no duplicates.
 *   - feature matrices are row-major fp32 with an explicit leading dimension (in elements).
 *   - fp32 rows handed to the GEMM must be 16-byte aligned with ld % 4 == 0.
 */
#ifndef LPFORMER_HIP_H
#define LPFORMER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LPF_OK 0
#define LPF_ERR_INVALID (-1)     /* bad argument (null pointer, size, alignment)        */
#define LPF_ERR_UNSUPPORTED (-2) /* shape outside what the kernels are built for        */
#define LPF_ERR_LAUNCH (-3)      /* hipLaunch / runtime error (see lpf_last_hip_error)  */
#define LPF_ERR_NO_DEVICE (-4)   /* no gfx950 device visible                            */

#define LPF_ABI_VERSION 9

/* GEMM / row-wise epilogue flags */
#define LPF_FLAG_RELU 1u

int lpf_abi_version(void);
const char *lpf_strerror(int code);
/* Last HIP runtime error string seen by this library on the calling thread ("" if none). */
const char *lpf_last_hip_error(void);
/* Fills cu_count / lds_bytes_per_cu / wave_size of the current device; LPF_ERR_NO_DEVICE if none. */
int lpf_device_info(int *cu_count, int *lds_bytes_per_cu, int *wave_size, char *arch_name, int arch_name_len);

/* ------------------------------------------------------------------------------------------------
 * Encoder (reference: src/models/other_models.py:61-76 GCN.forward -> torch_geometric GCNConv;
 *          src/models/link_transformer.py:110-129 propagate)
 * ---------------------------------------------------------------------------------------------- */

/* GCN symmetric normalisation on a CSR that already CONTAINS every diagonal entry
 * (replaces torch_geometric.nn.conv.gcn_conv.gcn_norm, called from other_models.py:35,66):
 *   w'_ij = (i==j ? 1 : w_ij);  deg_i = sum_j w'_ij;  w_out_ij = deg_i^-1/2 * w'_ij * deg_j^-1/2 (inf -> 0).
 * w_in may be NULL (all ones).  dis_tmp: n floats of scratch (receives deg^-1/2). */
int lpf_gcn_norm_csr(int64_t n, const int64_t *rowptr, const int32_t *col, const float *w_in,
                     float *w_out, float *dis_tmp, void *stream);

/* out[i,:] = epilogue( sum_j w_ij * H[col_ij,:] )   -- GCNConv.propagate + everything up to the next layer
 * (other_models.py:66-74 and, for the last layer, link_transformer.py:127):
 *   y = acc + bias;  if ln_g: y = LN(y; ln_g, ln_b);  if RELU: y = max(y,0);
 *   if residual: y = residual[i,:] + y;  if ln2_g: y = LN(y; ln2_g, ln2_b)
 * D % 4 == 0, D <= 256.  bias, ln_g, ln_b, residual, ln2_g, ln2_b may each be NULL.
 * long_rows (optional, NULL/0 = none): int32[n_long] list of EVERY row with more than LPF_SPMM_LONG_ROW stored entries
 * (row ids relative to `rowptr`); those hub rows get a whole workgroup each instead of one lane group. */
#define LPF_SPMM_LONG_ROW 128
int lpf_spmm_csr_f32(int64_t n, int32_t D, const int64_t *rowptr, const int32_t *col, const float *w,
                     const float *H, int64_t ldh, float *out, int64_t ldo, const float *bias,
                     const float *ln_g, const float *ln_b, const float *residual, int64_t ldr,
                     const float *ln2_g, const float *ln2_b, uint32_t flags, const int32_t *long_rows,
                     int64_t n_long, void *stream);

/* One GCN layer in one launch, square case (in = out = D in {32, 64, 128}): aggregate first, transform in registers
 * (gcn_fused.hip).  Replaces lpf_gemm_f32 + lpf_spmm_csr_f32 for such a layer (other_models.py:61-76 -> GCNConv):
 *   out[r - row_base] = epilogue( (sum_e w_e H[col_e]) W^T ),  epilogue = + bias, LayerNorm, ReLU, + residual, gnn_norm
 * as lpf_spmm_csr_f32 (same flags).  row_order int32[16 n_tiles]: the rows to produce, in the order they are worked on
 * -- any order is correct; lpformer_amd/graph.py fused_row_order sorts by degree so that the 16 rows of a tile, which
 * advance in lockstep, are equally long -- with -1 = padding and v <= -2 = hub number (-2 - v): hubs int32[n_hub][3] =
 * (row id, first slice, number of slices), the hub row's entry list being replaced by rows first .. first + number - 1
 * of t_parts (float[n_slices][D], lpf_spmm_row_parts_f32), each with weight 1.  rowptr is indexed by the global row
 * id; out / residual hold rows row_base...  w_packed = the weight image of lpformer_amd/fold.py pack_dense(W, 1)
 * (W = GCNConv.lin.weight, [D, D]).  pre_out (optional, training): receives the rows before the LayerNorm (product +
 * bias), what a LayerNorm backward needs.  Rounding order differs from transform-then-aggregate (A (X W^T) vs
 * (A X) W^T); identical from launch to launch. */
int lpf_gcn_layer_fused_f32(int32_t D, int64_t n_tiles, const int32_t *row_order, int64_t row_base,
                            const int64_t *rowptr, const int32_t *col, const float *w, const float *H, int64_t ldh,
                            const float *w_packed, float *out, int64_t ldo, const float *bias, const float *ln_g,
                            const float *ln_b, const float *residual, int64_t ldr, const float *ln2_g,
                            const float *ln2_b, uint32_t flags, const int32_t *hubs, const float *t_parts,
                            float *pre_out, int64_t ldpre, void *stream);
/* The layer as the TRAINING forward and backward launch it (lpformer_amd/train.py GcnFusedFn; reference: the autograd
 * graph of GCNConv + LayerNorm + ReLU + dropout + residual, src/models/other_models.py:61-76): no second LayerNorm, and
 *   agg_out (optional, float[n][ldagg]) receives the AGGREGATED rows sum_e w_e H[col_e] the product is taken of -- what
 *     the weight gradient dW = dU^T agg needs;
 *   drop_p > 0: F.dropout behind the ReLU (other_models.py:69), in the kernel -- element (row, feature) is kept and scaled
 *     by 1 / (1 - drop_p) iff a counter-based hash of (drop_seed, row, feature) passes p (csrc/lpf_common.h
 *     lpf_drop_bits: no mask tensor; lpf_layernorm_relu_drop_bwd_f32 recomputes it); then residual (optional) is added.
 * out = residual + dropout(ReLU(LN((A H) W^T + bias))).  The backward launches it once more over the transposed graph
 * with the transposed weight image, no epilogue, residual = the gradient that arrived through the skip connection:
 * dX = dOut + (A^T dU) W. */
int lpf_gcn_layer_fused_train_f32(int32_t D, int64_t n_tiles, const int32_t *row_order, int64_t row_base,
                                  const int64_t *rowptr, const int32_t *col, const float *w, const float *H, int64_t ldh,
                                  const float *w_packed, float *out, int64_t ldo, const float *bias, const float *ln_g,
                                  const float *ln_b, const float *residual, int64_t ldr, uint32_t flags,
                                  const int32_t *hubs, const float *t_parts, float *pre_out, int64_t ldpre,
                                  float *agg_out, int64_t ldagg, float drop_p, uint64_t drop_seed, void *stream);
/* The same layer gathering from a bf16 table (the bf16-table encoder mode; D = 64 or 128).  H_bf16p: uint16 rows, ldh in
 * elements (a multiple of 8), in the PERMUTED order  element 32 i + 8 q + 4 h + u = feature 16 (2 i + h) + 4 q + u
 * (i < D/32, q < 4, h < 2, u < 4) -- a lane's 16-byte load then holds two whole 16-feature k-groups.  out receives the
 * fp32 rows in normal order; out_bf16p (optional, ldob in elements) the same rows as permuted bf16, i.e. the next
 * layer's table.  Hub rows must sit in tiles of their own (fused_row_order(..., pad_hubs=True)); their slices come
 * from lpf_spmm_row_parts_bf16p, which reads the permuted table and writes fp32 sums in normal order. */
int lpf_gcn_layer_fused_bf16(int32_t D, int64_t n_tiles, const int32_t *row_order, int64_t row_base,
                             const int64_t *rowptr, const int32_t *col, const float *w, const void *H_bf16p,
                             int64_t ldh, const float *w_packed, float *out, int64_t ldo, const float *bias,
                             const float *ln_g, const float *ln_b, const float *residual, int64_t ldr,
                             const float *ln2_g, const float *ln2_b, uint32_t flags, const int32_t *hubs,
                             const float *t_parts, void *out_bf16p, int64_t ldob, void *stream);
int lpf_spmm_row_parts_bf16p(int32_t D, const int64_t *parts, int64_t n_parts, const int32_t *col, const float *w,
                             const void *H_bf16p, int64_t ldh, float *out, void *stream);

/* Sums of slices of (hub) rows: out[p][:D] = sum over the stored entries e in [parts[2p], parts[2p+1]) of w_e H[col_e]
 * (one workgroup per slice, partial sums added in a fixed order).  D a multiple of 8, <= 128. */
int lpf_spmm_row_parts_f32(int32_t D, const int64_t *parts, int64_t n_parts, const int32_t *col, const float *w,
                           const float *H, int64_t ldh, float *out, void *stream);

/* bf16 THROUGHPUT MODE of the aggregation (BASELINE.json config 2 names bf16 storage; SURVEY 8b lpf_spmm_csr_bf16):
 * the gathered table H holds bf16 rows (ldh in bf16 elements, rows 16-byte aligned) -- half the gather bytes, which
 * are what bounds this kernel --, the sum over neighbours, the epilogue and the output stay fp32 (D % 8 == 0: a lane
 * gathers 8 bf16 = 16 bytes, so a row needs half the lanes of the fp32 kernel).  The table comes
 * from lpf_gemm_f32_out_bf16 (the layer's X W^T, rounded to nearest even once). */
int lpf_spmm_csr_bf16(int64_t n, int32_t D, const int64_t *rowptr, const int32_t *col, const float *w,
                      const void *H_bf16, int64_t ldh, float *out, int64_t ldo, const float *bias,
                      const float *ln_g, const float *ln_b, const float *residual, int64_t ldr,
                      const float *ln2_g, const float *ln2_b, uint32_t flags, const int32_t *long_rows,
                      int64_t n_long, void *stream);

/* C[M,N] = A[M,K] * W[N,K]^T (+ bias[N]) (+ addend[M,N]) (ReLU)   -- nn.Linear / PyG Linear:
 * GCNConv.lin (other_models.py:66), lin_l / node half of lin_r (src/modules/layers.py:206-214),
 * MLP linears (other_models.py:125-138), mlp_score (other_models.py:173-179).
 * fp32 MFMA (v_mfma_f32_32x32x2_f32), exact fp32 accumulate.  lda, ldw % 4 == 0; bias/addend may be NULL. */
int lpf_gemm_f32(int64_t M, int32_t N, int32_t K, const float *A, int64_t lda, const float *W, int64_t ldw,
                 const float *bias, const float *addend, int64_t ldadd, float *C, int64_t ldc,
                 uint32_t flags, void *stream);
/* Same product, C stored as bf16 (ldc in bf16 elements). */
int lpf_gemm_f32_out_bf16(int64_t M, int32_t N, int32_t K, const float *A, int64_t lda, const float *W, int64_t ldw,
                          const float *bias, const float *addend, int64_t ldadd, void *C_bf16, int64_t ldc,
                          uint32_t flags, void *stream);

/* C[N,K] = A[M,N]^T * B[M,K]   -- the weight gradient of a Linear layer (dW = dY^T X; the training step,
 * src/train/train_model.py:59-77 through autograd).  The reduction over the M rows is split into chunks whose partial
 * products are added in chunk order (deterministic).  workspace: lpf_gemm_tn_workspace_floats(M, N, K) floats. */
int64_t lpf_gemm_tn_workspace_floats(int64_t M, int32_t N, int32_t K);
int lpf_gemm_tn_f32(int64_t M, int32_t N, int32_t K, const float *A, int64_t lda, const float *B, int64_t ldb,
                    float *C, int64_t ldc, float *workspace, void *stream);
/* The same product with the column sums of A beside it: colsum[n] = sum_m A[m][n] -- the bias gradient of the Linear
 * layer whose weight gradient C is (dY^T X and dY^T 1 in one pass over dY; replaces a lpf_colsum_f32 launch pair per
 * layer of the training step).  Same workspace. */
int lpf_gemm_tn_colsum_f32(int64_t M, int32_t N, int32_t K, const float *A, int64_t lda, const float *B, int64_t ldb,
                           float *C, int64_t ldc, float *colsum, float *workspace, void *stream);

/* y[i,:] = LN(x[i,:]; g, b) (then ReLU if flagged), in place allowed (nn.LayerNorm eps 1e-5, biased variance).
 * other_models.py:131-132, layers.py:78.  D <= 1024. g/b NULL -> plain ReLU / identity. */
int lpf_layernorm_f32(int64_t M, int32_t D, const float *x, int64_t ldx, const float *g, const float *b,
                      float *y, int64_t ldy, uint32_t flags, void *stream);

/* Backward of lpf_layernorm_f32 (no ReLU): dx, dgamma, dbeta from x, dy, gamma (row statistics recomputed from x;
 * column sums added in a fixed order).  D % 4 == 0, D <= 256; workspace: lpf_layernorm_bwd_workspace_floats(D) floats.
 * The training step (src/train/train_model.py:59-77 through autograd). */
int64_t lpf_layernorm_bwd_workspace_floats(int32_t D);
/* The same for y = ReLU(LN(x)) (the GCN layer's epilogue, other_models.py:66-69, and the MLPs' hidden layers, :131-133):
 * dy counts only where gamma xhat + beta > 0; dxsum[D] = column sums of dx, the gradient of a bias added in front of
 * the LayerNorm.  Same workspace. */
int lpf_layernorm_relu_bwd_f32(int64_t M, int32_t D, const float *x, int64_t ldx, const float *dy, int64_t ldy,
                               const float *gamma, const float *beta, float *dx, int64_t lddx, float *dgamma,
                               float *dbeta, float *dxsum, float *workspace, void *stream);
/* The same backward behind the in-kernel dropout of lpf_gcn_layer_fused_train_f32: the forward was
 * y = dropout(ReLU(LN(x))) with (drop_p, drop_seed); dy is scaled by 1 / (1 - drop_p) where (row, feature) was kept and
 * dropped where it was not -- the mask is recomputed from the seed (rows are numbered from 0). */
int lpf_layernorm_relu_drop_bwd_f32(int64_t M, int32_t D, const float *x, int64_t ldx, const float *dy, int64_t ldy,
                                    const float *gamma, const float *beta, float drop_p, uint64_t drop_seed, float *dx,
                                    int64_t lddx, float *dgamma, float *dbeta, float *dxsum, float *workspace,
                                    void *stream);
int lpf_layernorm_bwd_f32(int64_t M, int32_t D, const float *x, int64_t ldx, const float *dy, int64_t ldy,
                          const float *gamma, float *dx, int64_t lddx, float *dgamma, float *dbeta, float *workspace,
                          void *stream);

/* ------------------------------------------------------------------------------------------------
 * Pair stage (reference: src/models/link_transformer.py:132-178 calc_pairwise and helpers)
 * ---------------------------------------------------------------------------------------------- */

/* mul[k,:] = X[a_k,:] * X[b_k,:]  and  sum[k,:] = X[a_k,:] + X[b_k,:]   (link_transformer.py:101-102,143).
 * With X = Y := X_node W_l^T + b_l (one GEMM per encoder output) the sum IS the attention query of the pair,
 * lin_l(xa) + lin_l(xb) (layers.py:212-215): no per-pair GEMM.
 * batch: int64 [2, bs] row-major (row 0 = a, row 1 = b) with row stride `batch_ld`.  mul or sum may be NULL.
 * n_rows: rows of X; an id outside [0, n_rows) reads row 0 instead (lpf_select_plan raises LPF_SELECT_ERR_NODE_RANGE
 * for such a batch; the reference raises an index error at link_transformer.py:101). */
int lpf_pair_gather_f32(int64_t bs, int32_t D, const int64_t *batch, int64_t batch_ld, int64_t n_rows, const float *X,
                        int64_t ldx, float *mul, int64_t ldm, float *sum, int64_t lds, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Selection, GENERAL path (select2.hip; a caller-supplied typing adjacency -- for the model's own adjacency see
 * lpf_select3_* below): two launches, nothing read back by the host.
 * compute_node_mask + get_ppr_vals + get_non_1hop_ppr (link_transformer.py:214-319,434-481), eval mode; bit-exact.
 * The candidates of the batch form one flat slot space -- pair k owns N(a_k) | N(b_k) | the shorter of the two T0
 * rows, at least one slot -- cut into work items of LPF_SELECT_ITEM slots; one thread per slot.
 *
 *   ctl   int64[LPF_SELECT_CTL_WORDS], zero-initialised once by the caller, then owned by the library; ONE batch size
 *         per control block (the launch number is derived from the plan ticket):
 *         [0] slots of the batch  [1] work items  [2] item ticket  [3] STICKY error bits (LPF_SELECT_ERR_*; the caller
 *         clears them after handling)  [4..6] selected entries per type (cn, 1-hop, >1-hop)  [7] plan ticket
 *         [8] launch number (tags the chained-scan words: those of earlier launches read as "not ready", nothing is
 *         ever cleared and no per-launch value comes from the host, so the two launches replay from a captured graph)
 *   desc  128 bytes per pair;  offs int64[bs+1];  item_pair int32[item_cap];  plan_lb uint64[lpf_select_plan_blocks(bs)]
 *   run_lb uint64[3*item_cap];  type_ptr int32[3*(bs+1)];  entries 16 bytes x 3 x ent_cap
 *   val_*  the rows PPR values are looked up in: the raw PPR matrix, plain sorted CSR (val_rowptr / val_col / val_val;
 *          val_len NULL).  adj_selfp and val_cv must be NULL: round 2's indexed form of this call (self-PPR aligned
 *          with the adjacency + a hashed one-hop index) was superseded by lpf_select3_* and returns LPF_ERR_UNSUPPORTED
 *   BLOCKED index (the T0 rows): every row padded to a multiple of 16 entries (column INT32_MAX, value 0), row
 *          pointers counting padded entries, t0_len[i] = real entries of row i, t0_cv = interleaved {int32 column,
 *          float value} pairs (a 16-entry block = one aligned 128-byte line), t0_skip[b] = last column of block b
 *          (+ 40 spare entries): a lookup reads the row's skip entries, then one block that carries the value.
 *   HASHED index (the P1 rows with adj_selfp): row i owns val_len[i] buckets of 8 {column, value} entries (val_cv,
 *          64 aligned bytes per bucket, unused entries column INT32_MAX), val_rowptr counting entries
 *          (8 per bucket); entry (i, c) sits in bucket  mulhi_u32(c * 2654435761 mod 2^32, val_len[i])  and no
 *          bucket overflows (the builder grows a row's bucket count until that holds): ONE memory round trip per
 *          lookup.  Without adj_selfp (adjacency override) val_col / val_val are the raw PPR rows, plain sorted CSR.
 *   adjx_* UNMASKED adjacency for the >1-hop exclusion (link_transformer.py:443); NULL = the typing adjacency
 * Result: per type t a dense region entries[t*ent_cap ..) of {pair | from_N(b) << 31, node, pa, pb} records ordered by
 * (pair, candidate slot); segment of pair k = [type_ptr[t*(bs+1)+k], type_ptr[t*(bs+1)+k+1]); the one-hop segment
 * lists the kept nodes of N(a), then those of N(b).  Entries past ent_cap are dropped and LPF_SELECT_ERR_ENTRY_CAP is
 * raised; consumers clamp to ent_cap.  grid_blocks: persistent workgroups of the run kernel (0 = default).
 * mode_cn (lpf_select_run): mask mode "cn" (link_transformer.py:39-44,232-247) -- common neighbours only, their round
 * trip with t = 1, thresh_cn filter; pass t0_* = NULL with it (no >1-hop nodes). */
#define LPF_SELECT_ITEM 1024
#define LPF_SELECT_CTL_WORDS 16
#define LPF_SELECT_ERR_NODE_RANGE 1 /* a node id of the batch is outside [0, n_nodes): the pair was treated as empty */
#define LPF_SELECT_ERR_ITEM_CAP 2   /* more work items than item_cap: the batch was not processed completely      */
#define LPF_SELECT_ERR_ENTRY_CAP 4  /* more selected entries of one type than ent_cap                              */
int64_t lpf_select_plan_blocks(int64_t bs);
int lpf_select_plan(int64_t bs, const int64_t *batch, int64_t batch_ld, int64_t n_nodes, const int64_t *adj_rowptr,
                    const int64_t *val_rowptr, const int64_t *t0_rowptr, const int64_t *adjx_rowptr,
                    const int32_t *val_len, const int32_t *t0_len, void *desc, int64_t *offs, int32_t *item_pair,
                    int64_t item_cap, int64_t *ctl, uint64_t *plan_lb, void *stream);
int lpf_select_run(int64_t bs, const void *desc, const int64_t *offs, const int32_t *item_pair, int64_t item_cap,
                   int64_t *ctl, uint64_t *run_lb, const int32_t *adj_col, const float *adj_selfp,
                   const int32_t *adjx_col, const int32_t *val_col, const float *val_val, const void *val_cv,
                   const void *t0_cv, const int32_t *t0_skip, float th_cn, float th_1hop,
                   float th_non1hop, int32_t mode_cn, int32_t *type_ptr, void *entries, int64_t ent_cap,
                   int32_t grid_blocks, void *stream);
/* Selection over the per-model WALK INDEXES (select3.hip) -- the evaluation path: the typing adjacency is the model's
 * own adj_mask / full_adj_mask (link_transformer.py:226-227 with adj=None), mask modes "all", "1-hop" and "cn"
 * (:39-44).  Same control block, workspaces, result layout and error bits as lpf_select_plan / lpf_select_run above
 * (which remain the path for a caller-supplied adjacency override); desc is 128 bytes per pair, a pair owns at least
 * 16 slots.  Every selected set is an intersection evaluated from its shorter side with one hashed look-up per
 * candidate.  Index arrays, built once per (adjacency, PPR matrix, thresholds) by
 * lpformer_amd/graph.py build_walk_index; "cv" = interleaved {int32 node, fp32 value bits} pairs:
 *   node_rec 64 bytes per node: int64 {adj0, a10, px0, t00, u0} element offsets of the node's rows in adj_cv, a1_cv,
 *            px_cv, t0_cv (entries) and u_cv (entries, 8 per bucket), int32 {deg, n_a1, n_px, n_t0, u_buckets, 0}
 *   adj_cv   the adjacency rows with selfp[e] = P[i, j] (0 where the PPR matrix stores nothing)
 *   a1_cv    the adjacency entries whose own value passes the one-hop test fl32(fl32(p+1)-1) >= theta_1
 *   px_cv    the PPR entries (i, v), v NOT adjacent to i, that pass the weaker of the one-hop / >1-hop tests
 *   t0_cv    the px entries with p > 0 and fl32(fl32(p+1)-1) >= theta_n (NULL: no >1-hop nodes -- modes "1-hop", "cn")
 *   u_cv     HASHED union of a node's adjacency row and its px row: row i owns u_buckets[i] buckets of 8 entries
 *            (64 aligned bytes, unused entries node INT32_MAX), entry (i, v) in bucket
 *            mulhi_u32(v * 2654435761 mod 2^32, u_buckets[i]), value = P[i, v] with the SIGN BIT set when v is
 *            adjacent to i; no bucket overflows (the builder grows a row's bucket count until that holds)
 *   mini     uint32 [n_nodes][32]: a fixed 1,024-bit absence filter of every union row -- key v sets bits h & 31 and
 *            (h >> 5) & 31 of word h >> 27, h = mix32(v ^ 0x9E3779B9) (graph.py mini_filters / bloom_hash).  The run
 *            kernel stages the filters of an item's endpoints in LDS and fetches a bucket only for candidates that
 *            pass: random reads are what bounds it (profiles/r04_random_read_probe.txt)
 * mode_cn: mask mode "cn" -- common neighbours only, round trip with t = 1, thresh_cn filter (:232-247).
 * use_px:  0 when theta_1 <= 0 (absent PPR entries then pass the one-hop test and px rows cannot stand in for them). */
int lpf_select3_plan(int64_t bs, const int64_t *batch, int64_t batch_ld, int64_t n_nodes, const void *node_rec,
                     const void *adj_cv, const void *a1_cv, const void *px_cv, const void *t0_cv, int32_t mode_cn,
                     int32_t use_px, void *desc, int64_t *offs, int32_t *item_pair, int64_t item_cap, int64_t *ctl,
                     uint64_t *plan_lb, void *stream);
int lpf_select3_run(int64_t bs, const void *desc, const int64_t *offs, const int32_t *item_pair, int64_t item_cap,
                    int64_t *ctl, uint64_t *run_lb, const void *u_cv, const void *mini, float th_cn, float th_1hop,
                    float th_non1hop, int32_t mode_cn, int32_t *type_ptr, void *entries, int64_t ent_cap,
                    int32_t grid_blocks, void *stream);

/* The same selection in ONE launch with PAIR-MAJOR output (select4.hip) -- the form the hot
 * path runs: compute_node_mask + get_ppr_vals + get_non_1hop_ppr (link_transformer.py:214-319,434-481) for the model's
 * own typing adjacency, same index arrays, same per-candidate arithmetic and therefore the same index sets and PPR
 * values bit for bit as lpf_select3_*.  A workgroup owns a block of LPF_SELECT4_BLOCK consecutive pairs: it plans their
 * walks itself (no plan launch, no descriptors in memory), types the block's candidate slots, and compacts the kept
 * entries in slot order -- a pair's entries are contiguous -- with ballot ranks and one scan inside the workgroup (no
 * chained scan over the batch).  Where a block lands in `entries` is one atomic add; consumers address entries through
 * pair_tab only, so results do not depend on it.
 *   entries   16-byte records {pair | type << 29 | from_N(b) << 31, node, pa, pb}, type 1 = common neighbour,
 *             2 = one-hop, 3 = >1-hop; a pair's entries in candidate-slot order (neighbours of a, then of b, then
 *             the >1-hop nodes), types mixed
 *   pair_tab  int32[bs][4] = {first entry, n_cn, n_1hop, n_non1hop} per pair
 *   blk_cnt   int32[ceil(bs / LPF_SELECT4_BLOCK)][2]: {selected entries, pairs with selected entries} per block of pairs;
 *             8-byte aligned (written and read as 8-byte pairs)
 *   ctl       int64[LPF_SELECT4_CTL_WORDS], zero-initialised once by the caller, then owned by the library (one control
 *             block per stream): [0] entries the last batch needed room for (the buffer is cut into 8 regions, workgroup
 *             g allocates its candidate slots -- rounded up to 8 -- in region g % 8: 8 x the fullest region -- an upper
 *             bound)  [1] the candidate slots of the last batch (every workgroup's rounded up to 8: the true sum over the
 *             regions)  [3] STICKY error bits as above  [10] completion counter, [16..23] allocation counters (zero
 *             between launches)
 * A block that does not fit below ent_cap leaves empty pairs and raises LPF_SELECT_ERR_ENTRY_CAP (consumers write NaN
 * rows while the bit is set).  threads: launch shape, workgroup size (256, 512 or 1024) + 4096 * (blocks of 64 pairs a workgroup
 * takes together - 1); 0 = the default (1024 threads; 2 blocks while that gives every CU a workgroup, else 1). */
#define LPF_SELECT4_BLOCK 64
#define LPF_SELECT4_CTL_WORDS 32
int lpf_select4(int64_t bs, const int64_t *batch, int64_t batch_ld, int64_t n_nodes, const void *node_rec,
                const void *adj_cv, const void *a1_cv, const void *px_cv, const void *t0_cv, const void *u_cv,
                const void *mini, int32_t mode_cn, int32_t use_px, float th_cn, float th_1hop, float th_non1hop,
                int64_t *ctl, void *pair_tab, int32_t *blk_cnt, void *blk_types, void *entries, int64_t ent_cap,
                int32_t threads, void *stream);

/* lpf_select4's result in the TYPE-MAJOR form of lpf_select3_run (what the matrix-core attention, the record-merging tail
 * and lpf_select_export read): type_ptr int32[3][bs + 1] per-type segment pointers and three regions of `ent_cap` 16-byte
 * records {pair | from_N(b) << 31, node, pa, pb} ordered by (pair, candidate slot) -- same sets, same values, same order
 * inside a pair's segment as lpf_select3_* leave.  For hub-heavy batches (ogbl-ppa-like: a pair's walks run to thousands
 * of candidate slots, a handful is kept) lpf_select4 + this launch cost the chip about half the CU-time of
 * lpf_select3_plan + _run: a block of 64 pairs keeps one workgroup busy for as long as ITS walks take instead of every
 * CU for as long as the batch's do.
 *   blk_types  int32[ceil(bs / LPF_SELECT4_BLOCK)][4] {common neighbours, one-hop, >1-hop, 0} kept per block: pass the
 *              same buffer to lpf_select4 (its optional blk_types argument; NULL there: not written)
 *   ctl        lpf_select4's control block; receives [4..6] the totals per type and the sticky LPF_SELECT_ERR_ENTRY_CAP
 *              when a region is too small (entries past ent_cap are dropped, type_ptr still counts them)
 * One wavefront per block: the per-pair pointers are an in-wave scan on top of the sum of the blocks in front, the
 * block's entries -- one contiguous run of the pair-major buffer -- move with ballot ranks. */
int lpf_select4_regions(int64_t bs, const void *pair_tab, const void *blk_types, const void *entries4, int64_t ent_cap4,
                        int32_t *type_ptr, void *regions, int64_t ent_cap, int64_t *ctl, void *stream);

/* The reference's layout from the regions above: all CN entries sorted by (pair, node), then all 1-hop (the two runs
 * merged by node id), then all >1-hop (link_transformer.py:161-162); type_ptr64 int64[3*(bs+1)] relative per type,
 * counts_f (optional) the float count features: n_counts = 4 -> get_structure_cnts (:340-356) n_cn, n_1hop,
 * n_non1hop, n_cn + n_1hop; 3 -> without n_non1hop (mask mode "1-hop"); 1 -> n_cn alone (mask mode "cn", :154-155). */
int lpf_select_export(int64_t bs, const int32_t *type_ptr, const void *entries, int64_t ent_cap, int64_t *type_ptr64,
                      float *counts_f, int64_t ldc, int32_t n_counts, int32_t *sel_pair, int32_t *sel_node,
                      float *sel_pa, float *sel_pb, void *stream);

/* Attention scores for every selected entry (layers.py:206-218 with get_pos_encodings
 * link_transformer.py:182-211 folded in; algebra in DESIGN.md):
 *   h_e = ReLU(LN_t(W1_t [pa,pb] + b1_t)) + ReLU(LN_t(W1_t [pb,pa] + b1_t))
 *   k_e = Z[node_e] + Wfold_t h_e + bfold_t ;  score_e = sum_c att_c * leaky_relu(k_e,c * q[pair_e],c, 0.2)
 * Host-prepared, parameter-only tables (lpformer_amd/fold.py builds them in float64, stores fp32):
 *   pe_tab   float[3][D][4]  per type t and hidden unit k: (g_k (w0_k - mean w0), g_k (w1_k - mean w1),
 *                            g_k (b_k - mean b), beta_k) -- first PE Linear with the LayerNorm centring folded in
 *   pe_stat  float[3][8]     centred second moments over k of (w0, w1, b): C00, C11, Cbb, C01, C0b, C1b, 0, 0
 *                            (LayerNorm variance of W1 [x,y] + b1 as a quadratic form in (x, y))
 *   wfold_packed float[3][D/32][D/8][64][4]: element (t, c, sq, lane, u) = Wfold_t[32c + (lane&31)]
 *                            [(lane>>5)*(D/2) + 4 sq + u]  with Wfold_t = W_r[:, D:] W2_t  (MFMA A-operand order)
 *   bfold    float[3][D] = W_r[:, D:] (2 b2_t) ;  att float[D]
 * The number of entries is read from type_ptr on the device; max_entries (host-side capacity) only sizes the grid.
 * D in {32, 64, 128, 256}.  fp32 MFMA. */
int lpf_pair_scores_f32(int32_t D, const int64_t *type_ptr, int64_t bs, const int32_t *sel_pair,
                        const int32_t *sel_node, const float *sel_pa, const float *sel_pb,
                        const float *Z, int64_t ldz, const float *q, int64_t ldq,
                        const float *pe_tab, const float *pe_stat, const float *wfold_packed, const float *bfold,
                        const float *att, float *score, int64_t max_entries, void *stream);

/* Per-pair segment softmax (PyG softmax: max-shift, denominator + 1e-16; layers.py:220) and the weighted sums
 * that the output GEMM needs (layers.py:224 + scatter-sum):
 *   G[k, 0:D]   = sum_e alpha_e Z[node_e]
 *   G[k, (1+t)D : (2+t)D] = sum_{e of type t} alpha_e h_e          t = 0,1,2
 *   G[k, 4D + t] = sum_{e of type t} alpha_e ;  G[k, 4D+3] = 1
 * alpha_out (optional, NULL to skip): alpha per entry in the sel_* order (return_weights path, layers.py:73-75).
 * heavy_scratch: int32[bs+1] scratch (list of the pairs with many selected nodes, handled by a second kernel). */
int lpf_pair_softmax_gather_f32(int32_t D, int64_t bs, const int64_t *type_ptr, const int32_t *sel_node,
                                const float *sel_pa, const float *sel_pb, const float *score,
                                const float *Z, int64_t ldz, const float *pe_tab, const float *pe_stat,
                                float *G, int64_t ldg, float *alpha_out, int32_t *heavy_scratch, void *stream);

/* Fused dense chain (one launch):  y = L2( act( LN( L1(x) + addend ) ) )  on the fp32 matrix cores, hidden activations
 * never leaving registers.  Replaces Linear -> LayerNorm -> ReLU -> Linear chains of other_models.py:125-138 (MLP),
 * :173-179 (mlp_score, N2 == 1 "dot mode": logit/prob out), the hoisted lin_l of layers.py:212-215 (in_mode 2) and the
 * attention-output projection + post_att_norm (layers.py:78; single layer with addend + LN).
 *   in_mode 0: x = X[m, :K1]; 1: x = X[a_m] * X[b_m]; 2: x = X[a_m] + X[b_m]  (batch: int64 [2, M], row stride batch_ld;
 *              n_rows = rows of X, ids outside [0, n_rows) read row 0 -- see lpf_pair_gather_f32)
 *   w1_packed: layer-1 weights [N1, K1] in MFMA A-operand order, output tiles padded to an even count
 *              ntp1 = 2*ceil(N1/32).  k-group ks (16 input features) is ntp1*64 float4: float4 (c, lane = 16q + i) =
 *              W1[16c + i][16ks + 4q + 0..3].  A stage is one k-group, zero padded to a multiple of 512 float4; the
 *              image is the concatenation of ceil(K1 / 16) stages.
 *              b1, ln_g, ln_b: ntp1*16 floats, zero padded; ln_g NULL = no LayerNorm;
 *              flags & LPF_FLAG_RELU: ReLU after (the LayerNorm of) layer 1; addend [M, N1] optional (N1 % 4 == 0)
 *   w2_packed: NULL = single layer (out [M, N1]); N2 == 1: plain zero-padded vector of ntp1*16 floats, b2[0] the bias,
 *              out = logits [M] and/or prob = sigmoid [M]; otherwise [N2, N1] packed like w1 over the ntp1 hidden
 *              tiles as k-groups (ntp2 = 2*ceil(N2/32) output tiles); b2: ntp2*16 floats
 *   K1 % 4 == 0.  Built tile shapes (nt1, nt2): (2|3|4|5|8|9|16|17|32, 0), (2,2) (4,4) (8,8) (16,16), (3,2) (5,4) (9,8)
 *   (17,16); in_mode 1 for the square pairs and (2|4|8|16, 0), in_mode 2 for (2|4|8|16, 0);
 *   anything else returns LPF_ERR_UNSUPPORTED (callers then use lpf_gemm_f32 + lpf_layernorm_f32).
 *   lpformer_amd/fold.py builds the packed images. */
int lpf_dense_chain_f32(int64_t M, int32_t in_mode, const float *X, int64_t ldx, const int64_t *batch,
                        int64_t batch_ld, int64_t n_rows, int32_t K1, const float *w1_packed, int32_t N1, const float *b1,
                        const float *addend, int64_t ldadd, const float *ln_g, const float *ln_b, uint32_t flags,
                        const float *w2_packed, int32_t N2, const float *b2, float *out, int64_t ldo, float *prob,
                        void *stream);
/* lpf_dense_chain_f32 in a gather mode (in_mode 1 or 2) that ALSO leaves side_out[m, :side_dim] = S[a_m] + S[b_m] for a
 * second per-node table S [n_rows, ld_side_tab] read with the same ids -- lpf_pair_gather_f32 (sum) without a launch of
 * its own: the attention's per-pair query q = lin_l(x_a) + lin_l(x_b) from the table lin_l(X) (layers.py:212-215)
 * beside the elementwise branch of the same batch.  side_dim % 4 == 0, rows 16-byte aligned, a + b in that order. */
int lpf_dense_chain_side_f32(int64_t M, int32_t in_mode, const float *X, int64_t ldx, const int64_t *batch,
                             int64_t batch_ld, int64_t n_rows, int32_t K1, const float *w1_packed, int32_t N1,
                             const float *b1, const float *addend, int64_t ldadd, const float *ln_g, const float *ln_b,
                             uint32_t flags, const float *w2_packed, int32_t N2, const float *b2, float *out,
                             int64_t ldo, float *prob, const float *side_tab, int64_t ld_side_tab, int32_t side_dim,
                             float *side_out, int64_t ld_side_out, void *stream);

/* The dense tail of the scoring path in one launch (what LinkTransformer.score_pairs runs after the softmax-gather):
 *   o   = post_att_norm( G[:, D:] Wcat^T + G[:, :D] )                        (layers.py:78; G from
 *                                                                              lpf_pair_softmax_gather_f32)
 *   r_p = ReLU(LayerNorm( W_p0 [o | counts] + b_p0 ))                         first layer of pairwise_lin
 *   s   = w_dot . ReLU( W_C [r_e | r_p] + b_C ) + b_dot ;  prob = sigmoid(s)   score head, boundary Linears folded
 * D in {32, 64, 128} (else LPF_ERR_UNSUPPORTED: run the three lpf_dense_chain_f32 launches).  n_counts = 4 (3 in
 * "1-hop" mode; the fourth float of `counts` must then be 0).  Weight images in the lpf_dense_chain_f32 layout (one
 * k-group per stage): wA = Wcat [D, 3D+4]; wB = W_p0 [D+n, D+n]; wC = [A_e | A_p] laid out over D + 16*2*ceil((D+n)/32)
 * input columns (r_e first, then the even-tile-padded r_p).  lnA_*: D floats; bB, lnB_*: padded to the even tile
 * count of D+n; bC, w_dot: 2D floats; b_dot: 1 float.  r_e [M, D] = hidden activations of elementwise_lin. */
int lpf_tail_chain_f32(int64_t M, int32_t D, int32_t n_counts, const float *G, int64_t ldg, const float *wA_packed,
                       const float *lnA_g, const float *lnA_b, const float *counts, int64_t ldc,
                       const float *wB_packed, const float *bB, const float *lnB_g, const float *lnB_b,
                       const float *r_e, int64_t ldre, const float *wC_packed, const float *bC, const float *w_dot,
                       const float *b_dot, float *logit, float *prob, void *stream);

/* Attention in ONE pass over the regions written by lpf_select_run (pair_fused.hip): per entry
 *   k_e = Z[node] + Wfold_t h_e + bfold_t,  s_e = att . leaky_relu(k_e * q[pair], 0.2)        (layers.py:206-218)
 * then the segment softmax and the weighted sum (layers.py:220-224) as an online softmax per (pair, type) segment;
 * every Z row is gathered once, k_e and s_e never reach memory.  Output: for every non-empty segment
 *   part[(t*bs + pair)*(D+4) ..] = { sum_e exp(s_e - m) k_e [D], m = max_e s_e, l = sum_e exp(s_e - m), -, - }
 * A segment that crosses 16-entry unit boundaries of its type's region leaves instead one boundary record per unit it
 * touches: bnd[((t*units_cap + U)*2 + slot)*(D+4) ..], slot 1 of the unit it starts in, slot 0 of every following
 * unit.  lpf_tail_chain_merge_f32 combines a pair's records (it finds the boundary records from the segment
 * pointers).  Tables as for lpf_pair_scores_f32.
 * bnd: float[3*units_cap*2*(D+4)], units_cap >= ceil(ent_cap/16).  D in {32, 64, 128}. */
int lpf_pair_attention_fused_f32(int32_t D, int64_t bs, const int32_t *type_ptr, const void *entries,
                                 int64_t ent_cap, const float *Z, int64_t ldz, const float *q, int64_t ldq,
                                 const float *pe_tab, const float *pe_stat, const float *wfold_packed,
                                 const float *bfold, const float *att, float *part, float *bnd,
                                 int64_t units_cap, void *stream);

/* bf16 THROUGHPUT MODE of lpf_pair_attention_fused_f32 (BASELINE.json config 2 names bf16 storage): the node table
 * Z is stored in bf16 (half the gather bytes: rows of 2 D bytes staged by LDS-DMA) and Wfold_t h_e runs on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation; h_e is generated in fp32 and rounded to bf16 as the A operand.
 * Everything else (q, bfold, att, scores, softmax, records) stays fp32, and the selection never touches bf16, so the
 * selected index sets are unchanged.  Logits differ from the fp32 path by the bf16 rounding of Z and Wfold (tests state
 * the tolerance).
 *   Z_bf16            bf16 [N, ldz] (ldz in elements, rows 16-byte aligned)
 *   wfold_packed_bf16 bf16 [3][D/32][D/16][64][8]: element (t, c, s, lane, j) =
 *                     Wfold_t[32 c + (lane & 31)][16 s + 8 (lane >> 5) + j]   (MFMA B-operand order) */
int lpf_pair_attention_fused_bf16(int32_t D, int64_t bs, const int32_t *type_ptr, const void *entries,
                                  int64_t ent_cap, const void *Z_bf16, int64_t ldz, const float *q, int64_t ldq,
                                  const float *pe_tab, const float *pe_stat, const void *wfold_packed_bf16,
                                  const float *bfold, const float *att, float *part, float *bnd,
                                  int64_t units_cap, void *stream);

/* One-pass attention WITHOUT the D x D product per entry (pair_flip.hip): same inputs, records and consumers as
 * lpf_pair_attention_fused_f32, fp32 throughout.  The hidden layer of the PE MLP is affine in (r pa, r pb, r, 1) on
 * every set of active units, so with the four D-vectors of the units active at (0, 0) precomputed
 *   Wfold_t h_e + bfold_t = P0 (r1 pa + r2 pb) + Q0 (r1 pb + r2 pa) + R0 (r1 + r2) + C0 + sum_{k flipped} Wfold_t[:,k] |y_k|
 * (exact for every input; a unit is "flipped" when its ReLU state differs from the one at (0, 0)).  Tables from
 * lpformer_amd/fold.py flip_tables: pe_tab_signed float[3][D][4] = the pe_tab row of a unit times +1 when the unit is
 * active at (0, 0) and -1 when it is not (the kernel then sees a flipped unit as a NEGATIVE pre-activation of
 * magnitude |y_k|), base float[3][4][D] = (P0, Q0, R0, C0), wfold_t float[3][D][D] (wfold_t[t][k][c] = Wfold_t[c][k]).
 * Bound: vector-ALU issue (about ninety instructions per entry plus a dozen per flipped unit); the Z-row gather
 * (4 D + 16 bytes per entry) hides under it. */
int lpf_pair_attention_flip_f32(int32_t D, int64_t bs, const int32_t *type_ptr, const void *entries, int64_t ent_cap,
                                const float *Z, int64_t ldz, const float *q, int64_t ldq, const float *pe_tab_signed,
                                const float *pe_stat, const float *base, const float *wfold_t, const float *att,
                                float *part, float *bnd, int64_t units_cap, void *stream);

/* The same with the node table Z stored in bf16 (Z_bf16: uint16 rows, ldz in elements, a multiple of 8): the attention
 * kernel of the bf16 throughput mode for D >= 128.  Z is widened to fp32 on arrival; q, tables, arithmetic and records
 * are those of lpf_pair_attention_flip_f32. */
int lpf_pair_attention_flip_zbf16(int32_t D, int64_t bs, const int32_t *type_ptr, const void *entries, int64_t ent_cap,
                                  const void *Z_bf16, int64_t ldz, const float *q, int64_t ldq,
                                  const float *pe_tab_signed, const float *pe_stat, const float *base,
                                  const float *wfold_t, const float *att, float *part, float *bnd, int64_t units_cap,
                                  void *stream);

/* The attention OUTPUT of every pair from the records of lpf_pair_attention_fused_f32 (for callers that want
 * features, calc_pairwise link_transformer.py:132-178, rather than scores):
 *   out[p, :D] = post_att_norm( merged record of pair p + att_bias )   (layers.py:78,220; PyG softmax, +1e-16 once)
 *   out[p, D..] = n_cn, n_1hop, [n_non1hop,] n_cn + n_1hop             (get_structure_cnts, link_transformer.py:340-356)
 * ldo >= D + n_counts, ldo % 4 == 0.  Rows are NaN when sel_ctl[3] != 0 (selection workspace overflow). */
int lpf_pair_attention_merge_f32(int64_t bs, int32_t D, int32_t n_counts, const float *part, const float *bnd,
                                 int64_t units_cap, const int32_t *type_ptr, const float *att_bias, const float *ln_g,
                                 const float *ln_b, const int64_t *sel_ctl, float *out, int64_t ldo, void *stream);

/* lpf_tail_chain_f32 with the attention output taken from the records of lpf_pair_attention_fused_f32 instead of a
 * GEMM:  o = post_att_norm( sum_t e^{m_t-M} acc_t / (sum_t e^{m_t-M} l_t + 1e-16) + att_bias ), M = max_t m_t over the
 * pair's non-empty segments (PyG softmax over ALL entries of the pair, layers.py:220; no entry => o = LN(att_bias)),
 * and the count features n_cn, n_1hop, [n_non1hop,] n_cn+n_1hop (link_transformer.py:340-356) from type_ptr.
 * sel_ctl (optional): the selection control block; if its error word is set every score of the batch is NaN. */
int lpf_tail_chain_merge_f32(int64_t M, int32_t D, int32_t n_counts, const float *part, const float *bnd,
                             int64_t units_cap, const int32_t *type_ptr, const float *att_bias, const float *lnA_g, const float *lnA_b, const float *wB_packed,
                             const float *bB, const float *lnB_g, const float *lnB_b, const float *r_e, int64_t ldre,
                             const float *wC_packed, const float *bC, const float *w_dot, const float *b_dot,
                             const int64_t *sel_ctl, float *logit, float *prob, void *stream);
/* bf16 THROUGHPUT MODE of the dense tail (SURVEY 8b lpf_pair_head_bf16): the two GEMMs run on
 * v_mfma_f32_16x16x16_bf16 -- weights wB / wC as bf16 images in the same element order as the fp32 ones (a lane's four
 * consecutive fp32 become its four bf16), activations rounded to bf16 as they enter a GEMM, fp32 accumulate; the
 * record merge, both LayerNorms, the dot product and the sigmoid stay fp32. */
int lpf_tail_chain_merge_bf16(int64_t M, int32_t D, int32_t n_counts, const float *part, const float *bnd,
                             int64_t units_cap, const int32_t *type_ptr, const float *att_bias, const float *lnA_g, const float *lnA_b, const void *wB_packed_bf16,
                             const float *bB, const float *lnB_g, const float *lnB_b, const float *r_e, int64_t ldre,
                             const void *wC_packed_bf16, const float *bC, const float *w_dot, const float *b_dot,
                             const int64_t *sel_ctl, float *logit, float *prob, void *stream);

/* One-pass attention PAIR-MAJOR (pair_rows.hip; replaces lpf_pair_attention_flip_* + the record merge on the hot path):
 * LinkAttention.message + PyG softmax + scatter-sum + LinkAttention.bias + post_att_norm (src/modules/layers.py:66-78,
 * 193-224) and get_structure_cnts (src/models/link_transformer.py:340-356), with the positional encodings
 * (link_transformer.py:182-211) folded in as in lpf_pair_attention_flip_f32 (same tables).  A group of D/4 lanes walks
 * the entries PAIR-MAJOR -- a pair's three segments one after the other, found from type_ptr, 16 consecutive entries
 * per work unit -- and the kernel writes every pair's finished row once:
 *   out[p, :D]  = post_att_norm( sum_e alpha_e k_e + att_bias )          (no entry: post_att_norm(att_bias))
 *   out[p, D..] = n_cn, n_1hop, [n_non1hop,] n_cn + n_1hop               (n_counts of them; n_counts = 0: none written)
 * A pair inside one unit is finished in registers; one that crosses units leaves a partial state per unit in `pieces`,
 * merged in unit order by the workgroup that owns the pair before the launch ends: nothing is left for the consumer.
 * ldo >= D + n_counts, ldo % 4 == 0.  Rows are NaN when sel_ctl[3] != 0 (selection workspace overflow).
 * pieces: scratch, float[units_cap * 2 * lpf_pair_rows_piece_floats(D)], units_cap >= ceil(3 * ent_cap / 16) + 1 -- the
 * partial softmax states of the pairs that cross the 16-entry units of the pair-major order (written and merged inside
 * the launch, by the workgroup that owns the pair; contents are meaningless afterwards).  D in {32, 64, 128, 256}. */
int lpf_pair_attention_rows_f32(int32_t D, int64_t bs, const int32_t *type_ptr, const void *entries, int64_t ent_cap,
                                const float *Z, int64_t ldz, const float *q, int64_t ldq, const float *pe_tab_signed,
                                const float *pe_stat, const float *base, const float *wfold_t, const float *att,
                                const float *att_bias, const float *ln_g, const float *ln_b, int32_t n_counts,
                                const int64_t *sel_ctl, float *pieces, int64_t units_cap, float *out, int64_t ldo,
                                void *stream);
int64_t lpf_pair_rows_piece_floats(int32_t D);
/* The same with the node table Z stored in bf16 (uint16 rows, ldz in elements, a multiple of 8). */
int lpf_pair_attention_rows_zbf16(int32_t D, int64_t bs, const int32_t *type_ptr, const void *entries, int64_t ent_cap,
                                  const void *Z_bf16, int64_t ldz, const float *q, int64_t ldq,
                                  const float *pe_tab_signed, const float *pe_stat, const float *base,
                                  const float *wfold_t, const float *att, const float *att_bias, const float *ln_g,
                                  const float *ln_b, int32_t n_counts, const int64_t *sel_ctl, float *pieces,
                                  int64_t units_cap, float *out, int64_t ldo, void *stream);

/* lpf_tail_chain_f32 behind lpf_pair_attention_rows_*: stage A is a plain read of the pair's finished row
 * rows[p] = [post_att_norm(attention output) (D) | count features, zero padded to 4] (ldrows >= D + 4); what is left is
 * the first layer of pairwise_lin, its LayerNorm + ReLU, the folded score head and the sigmoid
 * (other_models.py:80-179, link_transformer.py:170-177).  sel_ctl as for lpf_tail_chain_merge_f32.  D in {32, 64, 128}. */
int lpf_tail_chain_rows_f32(int64_t M, int32_t D, int32_t n_counts, const float *rows, int64_t ldrows,
                            const float *wB_packed, const float *bB, const float *lnB_g, const float *lnB_b,
                            const float *r_e, int64_t ldre, const float *wC_packed, const float *bC, const float *w_dot,
                            const float *b_dot, const int64_t *sel_ctl, float *logit, float *prob, void *stream);
/* ... with bf16 weights and v_mfma_f32_16x16x16_bf16 for the two GEMMs (as lpf_tail_chain_merge_bf16). */
int lpf_tail_chain_rows_bf16(int64_t M, int32_t D, int32_t n_counts, const float *rows, int64_t ldrows,
                             const void *wB_packed_bf16, const float *bB, const float *lnB_g, const float *lnB_b,
                             const float *r_e, int64_t ldre, const void *wC_packed_bf16, const float *bC,
                             const float *w_dot, const float *b_dot, const int64_t *sel_ctl, float *logit, float *prob,
                             void *stream);

/* The pair of launches with the pairs WITHOUT selected nodes handled apart.  Such a pair's attention branch is a constant
 * -- softmax over nothing, so the row is post_att_norm(att_bias) with zero counts (layers.py:66-78,
 * link_transformer.py:340-356) -- and so is everything pairwise_lin makes of it: its score needs the elementwise half of
 * the folded head only.  lpf_pair_attention_rows_perm_* additionally leaves the ORDER the tail walks the pairs in:
 *   perm int32[bs]: the pairs with selected nodes in ascending order, then -- from the back, descending towards the
 *                   front -- the ones without;  *n_nonempty: how many have selected nodes.  (Deterministic: a chained scan
 *                   over the workgroups, no atomics on a counter.)
 *   perm_lb uint64[LPF_ROWS_PERM_LB_WORDS]: the scan's words; zero before the first launch, then left alone (the
 *                   kernel tags them with a launch number it keeps in the last word).  One buffer per stream.
 * lpf_tail_chain_rows_perm_* walks the pairs in that order; a workgroup whose 64 pairs all lie behind *n_full computes
 *   score = w_dot . ReLU(A_e r_e + bC_empty) + b_dot,    bC_empty [2 D] = bC + A_p r_p0
 * with r_p0 the hidden activation of pairwise_lin for the constant row (lpformer_amd/fold.py empty_pair_head_bias); every
 * other workgroup runs the full tail.  In a MIXED workgroup a pair without selected nodes takes its row from row_empty
 * [D] = post_att_norm(att_bias) (lpformer_amd/fold.py empty_pair_row) and zero counts; row_empty NULL: it reads rows[] like
 * any other pair (lpf_pair_attention_rows_perm_* writes the constant row there; lpf_pair_attention_rows4_* with an order
 * does NOT write rows or counts of such pairs: pass row_empty behind it).  Outputs are indexed by pair as before. */
#define LPF_ROWS_PERM_LB_WORDS 1025
int lpf_pair_attention_rows_perm_f32(int32_t D, int64_t bs, const int32_t *type_ptr, const void *entries, int64_t ent_cap,
                                     const float *Z, int64_t ldz, const float *q, int64_t ldq,
                                     const float *pe_tab_signed, const float *pe_stat, const float *base,
                                     const float *wfold_t, const float *att, const float *att_bias, const float *ln_g,
                                     const float *ln_b, int32_t n_counts, const int64_t *sel_ctl, float *pieces,
                                     int64_t units_cap, float *out, int64_t ldo, int32_t *perm, uint64_t *perm_lb,
                                     int64_t *n_nonempty, void *stream);
int lpf_pair_attention_rows_perm_zbf16(int32_t D, int64_t bs, const int32_t *type_ptr, const void *entries,
                                       int64_t ent_cap, const void *Z_bf16, int64_t ldz, const float *q, int64_t ldq,
                                       const float *pe_tab_signed, const float *pe_stat, const float *base,
                                       const float *wfold_t, const float *att, const float *att_bias,
                                       const float *ln_g, const float *ln_b, int32_t n_counts, const int64_t *sel_ctl,
                                       float *pieces, int64_t units_cap, float *out, int64_t ldo, int32_t *perm,
                                       uint64_t *perm_lb, int64_t *n_nonempty, void *stream);
/* lpf_pair_attention_rows_perm_* behind lpf_select4: the entries are pair-major already (a pair's entries contiguous from
 * pair_tab[p][0], the type in bits 29-30 of the record's pair word), so the kernel reads ONE region and needs no per-type
 * pointers; its workgroups split the batch by blk_cnt (entries per LPF_SELECT4_BLOCK pairs) and the 64 table entries of the
 * block a cut falls into.  units_cap: 16-entry units the `pieces` scratch has room for -- it bounds the SELECTED entries
 * of a batch (16 * (units_cap - 1)), not the entry buffer; a batch with more leaves NaN rows and raises
 * LPF_SELECT_ERR_ENTRY_CAP in sel_ctl[3].  perm / n_nonempty: both, or both NULL (no order for the
 * tail; the order needs no scan words here: the selection counted the pairs with entries per block; WITH an order the
 * rows and count features of pairs without entries are not written -- lpf_tail_chain_rows_perm_* takes row_empty for
 * them).
 * ACTIVATION PATTERNS BY TABLE (link_transformer.py:67-76,182-211 -- the PE MLPs' hidden ReLU -- layers.py:193-224): the
 * type-major calls know the base vectors of ONE activation pattern of the PE hidden layer, the one of (pa, pb) = (0, 0),
 * and pay per entry for finding and correcting the units that left it (`base`, `wfold_t`).  Here the plane of PPR value
 * pairs is cut into grid_n x grid_n cells, cell(v) = clamp((bits(fp32(v + grid_ofs)) >> grid_shift) - grid_base, 0,
 * grid_n - 1) on either axis, and
 *   pat_grid uint8 [3][grid_n][grid_n]: per type, cell (cell(x), cell(y)) -> bits 0-4: id s < LPF_ROWS_PATTERNS of a
 *       tabulated pattern, bit 7 clear: s is the pattern of EVERY point (x, y) of the cell; bit 7 set (a boundary may
 *       cross the cell, or its pattern is not tabulated): s is the tabulated pattern nearest to the pattern of the cell's
 *       centre, and the entry takes the exact path -- the units whose state differs from pattern s are found
 *       (pe_tab_signed, signed for pattern 0, with pat_sign) and corrected (wfold_t) one by one; the last cell of either
 *       axis (values > 1, NaN, negative) must be 0x80;
 *   pat_base float [3][LPF_ROWS_PATTERNS][4][D]: (P_s, Q_s, R_s, B_s + bfold / 2) with X_s = sum over the units k active
 *       in pattern s of Wfold[:, k] * (ta_k, tc_k, td_k, beta_k); id 0 = the pattern of (0, 0);
 *   pat_sign uint32 [3][LPF_ROWS_PATTERNS][D / 32]: bit k of pattern s = unit k is active in exactly one of pattern s and
 *       pattern 0.
 * An entry's key is Z[v] + [P r1 pa + Q r1 pb + R r1 + B](pattern of (pa, pb)) + [P r2 pb + Q r2 pa + R r2 + B](pattern
 * of (pb, pa)): no look at its 2 D units at all.  Built by lpformer_amd/patterns.py (a per-cell proof by convexity;
 * which patterns are tabulated only decides how many entries take the slower path, never a result).  A kernel that
 * cannot hold LPF_ROWS_PATTERNS patterns per type in LDS (D = 256) keeps the first half and treats higher ids as 0x80.
 * Everything else as lpf_pair_attention_rows_f32; the rows agree with that call's up to rounding (the order in which a
 * pair's entries are summed, the association of the base vectors). */
#define LPF_ROWS_PATTERNS 16
int lpf_pair_attention_rows4_f32(int32_t D, int64_t bs, const void *pair_tab, const int32_t *blk_cnt, const void *entries,
                                 int64_t ent_cap, const float *Z, int64_t ldz, const float *q, int64_t ldq,
                                 const float *pe_tab_signed, const float *pe_stat, const float *pat_base,
                                 const void *pat_grid, const void *pat_sign, int32_t grid_n, int32_t grid_shift,
                                 int32_t grid_base, float grid_ofs, const float *wfold_t, const float *att, const float *att_bias,
                                 const float *ln_g, const float *ln_b, int32_t n_counts, const int64_t *sel_ctl,
                                 float *pieces, int64_t units_cap, float *out, int64_t ldo, int32_t *perm,
                                 int64_t *n_nonempty, void *stream);
int lpf_pair_attention_rows4_zbf16(int32_t D, int64_t bs, const void *pair_tab, const int32_t *blk_cnt,
                                   const void *entries, int64_t ent_cap, const void *Z_bf16, int64_t ldz, const float *q,
                                   int64_t ldq, const float *pe_tab_signed, const float *pe_stat, const float *pat_base,
                                   const void *pat_grid, const void *pat_sign, int32_t grid_n, int32_t grid_shift,
                                   int32_t grid_base, float grid_ofs, const float *wfold_t, const float *att, const float *att_bias,
                                   const float *ln_g, const float *ln_b, int32_t n_counts, const int64_t *sel_ctl,
                                   float *pieces, int64_t units_cap, float *out, int64_t ldo, int32_t *perm,
                                   int64_t *n_nonempty, void *stream);

/* The dense tail behind that order (perm / n_full / bC_empty / row_empty: the paragraph in front of
 * lpf_pair_attention_rows_perm_f32; other_models.py:80-179, link_transformer.py:101-105,170-177). */
int lpf_tail_chain_rows_perm_f32(int64_t M, int32_t D, int32_t n_counts, const float *rows, int64_t ldrows,
                                 const float *wB_packed, const float *bB, const float *lnB_g, const float *lnB_b,
                                 const float *r_e, int64_t ldre, const float *wC_packed, const float *bC,
                                 const float *w_dot, const float *b_dot, const int64_t *sel_ctl, const int32_t *perm,
                                 const int64_t *n_full, const float *bC_empty, const float *row_empty, float *logit,
                                 float *prob, void *stream);
int lpf_tail_chain_rows_perm_bf16(int64_t M, int32_t D, int32_t n_counts, const float *rows, int64_t ldrows,
                                  const void *wB_packed_bf16, const float *bB, const float *lnB_g, const float *lnB_b,
                                  const float *r_e, int64_t ldre, const void *wC_packed_bf16, const float *bC,
                                  const float *w_dot, const float *b_dot, const int64_t *sel_ctl, const int32_t *perm,
                                  const int64_t *n_full, const float *bC_empty, const float *row_empty, float *logit,
                                  float *prob, void *stream);

/* logit[i] = dot(A[i,:], w) + b ; prob[i] = sigmoid(logit[i])  (mlp_score last layer, other_models.py:178-179).
 * logit or prob may be NULL. */
int lpf_rowdot_sigmoid_f32(int64_t M, int32_t K, const float *A, int64_t lda, const float *w, float b,
                           float *logit, float *prob, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Device-side PPR producer: the same Andersen push as lpf_ppr_push_cpu below, on the GPU.
 * calc_ppr (src/util/calc_ppr_scores.py:136-192) + create_sparse_ppr_matrix (:221-241): LIFO stack and float64
 * arithmetic per source in the reference's order, so index sets and fp32 values are bit-identical.  One wavefront
 * per source (dynamic assignment), neighbours of a popped node across the lanes; every wavefront owns dense
 * epoch-stamped state over all n nodes (32 n bytes) inside `workspace`.
 * ---------------------------------------------------------------------------------------------- */

/* Bytes of `workspace` for n_waves concurrent sources (n_waves: multiple of 4; 4096-8192 fills an MI355X). */
int64_t lpf_ppr_push_workspace_bytes(int64_t n, int64_t n_waves, double alpha, double eps);

/* rowptr/col: CSR of the coalesced directed edge list (get_ppr_matrix, :111-117), device memory.
 * Row i of the result lands UNSORTED at pool_col/pool_val[row_off[i] .. row_off[i] + row_len[i]) (rows are placed in
 * completion order).  counters int64[4] on return: [1] = total entries (if > pool_capacity nothing was written for
 * the rows that did not fit: call again with a pool of that size), [2] = rows that overflowed their
 * 1/(alpha*eps) list bound (must be 0). */
int lpf_ppr_push_f64(int64_t n, const int64_t *rowptr, const int32_t *col, double alpha, double eps,
                     int64_t n_waves, void *workspace, int64_t workspace_bytes, int32_t *pool_col, float *pool_val,
                     int64_t pool_capacity, int64_t *row_off, int32_t *row_len, int64_t *counters, void *stream);

/* Sort every row by column and pack the CSR (out_rowptr int64[n+1], out_col/out_val [nnz], nnz = counters[1] above). */
int64_t lpf_ppr_pack_workspace_bytes(int64_t n, int64_t nnz);
int lpf_ppr_pack_csr(int64_t n, const int64_t *row_off, const int32_t *row_len, const int32_t *pool_col,
                     const float *pool_val, int64_t nnz, int64_t *out_rowptr, int32_t *out_col, float *out_val,
                     void *workspace, int64_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Per-model indexes over the PPR matrix, built on the device.  Selection results are identical
 * with and without them.
 *   mode 0: the >1-hop candidates  p > 0 and fl32(fl32(p+1)-1) >= theta   (link_transformer.py:464-478)  -> "T0"
 *   mode 1: the one-hop candidates fl32(fl32(p+1)-1) >= theta             (link_transformer.py:241-250,290-317) -> "P1"
 * Two steps around an exclusive scan of out_len by the caller: count per row, then fill (column order preserved).
 * ---------------------------------------------------------------------------------------------- */
int lpf_ppr_filter_count(int64_t n, const int64_t *rowptr, const float *val, int32_t mode, float theta,
                         int64_t *out_len, void *stream);
int lpf_ppr_filter_fill(int64_t n, const int64_t *rowptr, const int32_t *col, const float *val, int32_t mode,
                        float theta, const int64_t *out_rowptr, int32_t *out_col, float *out_val, void *stream);

/* selfp[e] = P[i, j] for every adjacency entry e = (i, j) (0 where the PPR matrix stores nothing): the PPR of a node
 * to its own neighbours, aligned with the adjacency CSR (the adj_selfp argument of lpf_select_run). */
int lpf_self_ppr(int64_t n, const int64_t *adj_rowptr, const int32_t *adj_col, const int64_t *ppr_rowptr,
                 const int32_t *ppr_col, const float *ppr_val, float *selfp, void *stream);

/* out[q] = M[rows[q], cols[q]] of a CSR matrix with sorted int32 columns (0 where nothing is stored or an id lies outside
 * [0, n)).  Used for the raw PPR values of the few selected entries whose TYPE changes when the training loop removes
 * the batch's positive edges from the typing adjacency (src/train/train_model.py:40-46; a common neighbour of (a, b)
 * that loses its edge to a becomes a one-hop node and its values are the other type's round trip of the raw value,
 * src/models/link_transformer.py:229-237,290-291): lpformer_amd.LinkTransformer._patch_removed. */
int lpf_csr_lookup_f32(int64_t nq, int64_t n, const int64_t *rows, const int64_t *cols, const int64_t *rowptr,
                       const int32_t *col, const float *val, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Training step of the pair stage (pair_train.hip): the forward of get_pos_encodings
 * (link_transformer.py:182-211) and LinkAttention.message + PyG softmax + scatter-sum (layers.py:193-224) with the
 * state a backward pass needs, and the gradients torch autograd derives from them (the reference's training step,
 * src/train/train_model.py:59-77).  Entries are sorted by (type, pair); seg int64[3][bs+1] = first entry of pair p's
 * type-t segment (global entry index, seg[t][bs] = end of type t).  D in {32, 64, 128, 256}; rows 16-byte aligned.
 * Sums over entries / pairs go through per-block partials added in block order (deterministic); dZ and
 * lpf_pair_scatter_add_f32 use float atomics.  workspace: lpf_train_partial_blocks(units) * k * D floats (k below;
 * units = rows the call reduces over -- today a constant bound, 512 blocks).
 * ---------------------------------------------------------------------------------------------- */
int64_t lpf_train_partial_blocks(int64_t units);
/* H[e,:] = ReLU(LN(W1 [pa,pb] + b1)) + ReLU(LN(W1 [pb,pa] + b1)) for the entries of ONE type (w1 float[D][2],
 * b1 / gamma / beta float[D]: the first Linear and the LayerNorm of that type's ppr_encoder_* MLP). */
int lpf_pe_hidden_fwd_f32(int64_t n_entries, int32_t D, const float *w1, const float *b1, const float *gamma,
                          const float *beta, const float *pa, const float *pb, float *H, int64_t ldh, void *stream);
/* grads float[5][D] = (dW1[:,0], dW1[:,1], db1, dgamma, dbeta) from dH; workspace k = 5. */
int lpf_pe_hidden_bwd_f32(int64_t n_entries, int32_t D, const float *w1, const float *b1, const float *gamma,
                          const float *beta, const float *pa, const float *pb, const float *dH, int64_t ldh,
                          float *grads, float *workspace, void *stream);
/* out[c] = sum_r x[r, c]  (bias gradients); workspace k = 1. */
int lpf_colsum_f32(int64_t M, int32_t D, const float *x, int64_t ldx, float *out, float *workspace, void *stream);
/* k_e = Z[node_e] + KP[e]; s_e = att . leaky_relu(k_e * q[p], 0.2); out[p] = sum_e softmax_p(s)_e k_e + bias.
 * Saves the raw scores and per pair the segment max and 1 / (sum exp + 1e-16) (0 for a pair without entries). */
int lpf_pair_attention_train_fwd_f32(int64_t bs, int64_t n_entries, int32_t D, const int64_t *seg,
                                     const int32_t *e_node, const float *Z, int64_t ldz, const float *KP, int64_t ldk,
                                     const float *q, int64_t ldq, const float *att, const float *bias, float *out,
                                     int64_t ldo, float *score, float *pmax, float *pinv, void *stream);
/* Gradients of the above from dout: dK[e] (entry-major), dq[p], datt_dbias float[2][D]; workspace k = 2.  dZ: NULL (the
 * caller sums dK by node with lpf_segment_rows_sum_f32: deterministic) or float[N][lddz], zeroed by the caller, to which
 * dk_e is added at row node_e with float atomics (order of the additions not fixed). */
int lpf_pair_attention_train_bwd_f32(int64_t bs, int64_t n_entries, int32_t D, const int64_t *seg,
                                     const int32_t *e_node, const float *Z, int64_t ldz, const float *KP, int64_t ldk,
                                     const float *q, int64_t ldq, const float *att, const float *bias, const float *out,
                                     int64_t ldo, const float *score, const float *pmax, const float *pinv,
                                     const float *dout, int64_t lddo, float *dK, int64_t lddk, float *dZ, int64_t lddz,
                                     float *dq, int64_t lddq, float *datt_dbias, float *workspace, void *stream);
/* dst[key] = sum of src[order[j]] over each run of equal keys in keys_sorted (ascending j: a fixed order) -- the gradient
 * of a row gather without atomics: keys_sorted / order = the node ids of the entries sorted and the permutation that sorts
 * them, src = dK, dst = dZ (rows no key names are left untouched: zero them first).  D in {32, 64, 128, 256}. */
int lpf_segment_rows_sum_f32(int64_t n, int32_t D, const int32_t *keys_sorted, const int64_t *order, const float *src,
                             int64_t lds, float *dst, int64_t ldd, void *stream);
/* Gradient of lpf_pair_gather_f32: dX[a_k] += dsum[k] + dmul[k] * X[b_k], dX[b_k] += dsum[k] + dmul[k] * X[a_k]
 * (dmul or dsum may be NULL; dX is accumulated into). */
int lpf_pair_scatter_add_f32(int64_t bs, int32_t D, const int64_t *batch, int64_t batch_ld, int64_t n_rows,
                             const float *X, int64_t ldx, const float *dmul, int64_t ldm, const float *dsum,
                             int64_t lds, float *dX, int64_t lddx, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Host side (liblpformer_host.so)
 * ---------------------------------------------------------------------------------------------- */

/* Approximate personalised PageRank for every source node: calc_ppr (src/util/calc_ppr_scores.py:136-192)
 * followed by create_sparse_ppr_matrix (:221-241).  Same LIFO push order and float64 arithmetic per source, so
 * the produced index sets and fp32 values are bit-identical; OpenMP over sources (num_threads <= 0: all cores).
 * indptr/indices: CSR of the coalesced directed edge list (get_ppr_matrix, :111-117), host memory.
 * out_rowptr: caller-provided int64[n+1].  *out_col / *out_val: malloc'ed by the library, sized out_rowptr[n],
 * rows sorted by column; release both with lpf_host_free. */
int lpf_ppr_push_cpu(int64_t n, const int64_t *indptr_host, const int32_t *indices_host, double alpha, double eps,
                     int64_t *out_rowptr_host, int32_t **out_col_host, float **out_val_host, int32_t num_threads);

void lpf_host_free(void *p);
int lpf_host_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LPFORMER_HIP_H */
