// fp32 GEMM on the gfx950 matrix cores:  C[M,N] = A[M,K] * W[N,K]^T (+bias) (+addend) (ReLU)
//
// v_mfma_f32_32x32x2_f32 (exact fp32 multiply-accumulate, 64 FLOP/clk/SIMD).  Both operands are K-contiguous
// in memory (nn.Linear layout), so A and W tiles are staged the same way:
//   global --float4--> registers --ds_write_b128--> LDS [rows][BK+4] --ds_read_b128--> MFMA operand registers
// Inside a 32-wide K block lane (row = l&31, half = l>>5) owns k = 16*half + s for MFMA step s = 0..15 (any
// bijection of k onto (step, half) is a valid summation order as long as A and W use the same one), which makes
// every operand fetch a 16-byte LDS read; the +4 float row padding (144-byte stride) makes those reads
// bank-conflict free.  Two LDS buffers, one barrier per K block, next tile's global loads in flight during the
// MFMAs.  Block = 4 waves; wave w owns rows [32w, 32w+32) x all BN columns (BN/32 accumulators of 16 registers).
#include "lpf_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDK = BK + 4;  // padded LDS row stride (floats)

template <int BN, bool OUT_BF16 = false>
__global__ __launch_bounds__(256) void gemm_f32_kernel(int64_t M, int N, int K, const float *__restrict__ A,
                                                       int64_t lda, const float *__restrict__ W, int64_t ldw,
                                                       const float *__restrict__ bias,
                                                       const float *__restrict__ addend, int64_t ldadd,
                                                       float *__restrict__ C, int64_t ldc, uint32_t flags) {
    constexpr int NT = BN / 32;            // accumulator tiles per wave
    constexpr int A_F4 = BM * BK / 4 / 256;  // float4 loads per thread for the A tile (4)
    constexpr int B_F4 = BN * BK / 4 / 256;  // for the W tile (4 or 2)
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDK];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    float4 ra[A_F4], rb[B_F4];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int r = 0; r < A_F4; ++r) {
            int f = tid + 256 * r;
            int row = f >> 3, c4 = f & 7;
            int64_t m = m0 + row;
            if (m >= M) m = M - 1;
            int k = k0 + 4 * c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < K) {
                v = *reinterpret_cast<const float4 *>(A + m * lda + k);
                if (k + 3 >= K) {  // ragged K: zero what lies beyond it
                    if (k + 1 >= K) v.y = 0.f;
                    if (k + 2 >= K) v.z = 0.f;
                    v.w = 0.f;
                }
            }
            ra[r] = v;
        }
#pragma unroll
        for (int r = 0; r < B_F4; ++r) {
            int f = tid + 256 * r;
            int row = f >> 3, c4 = f & 7;
            int n = n0 + row;
            if (n >= N) n = N - 1;
            int k = k0 + 4 * c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < K) {
                v = *reinterpret_cast<const float4 *>(W + (int64_t)n * ldw + k);
                if (k + 3 >= K) {
                    if (k + 1 >= K) v.y = 0.f;
                    if (k + 2 >= K) v.z = 0.f;
                    v.w = 0.f;
                }
            }
            rb[r] = v;
        }
    };
    auto store_tiles = [&](int buf) {
        float *As = lds + buf * (BM + BN) * LDK;
        float *Bs = As + BM * LDK;
#pragma unroll
        for (int r = 0; r < A_F4; ++r) {
            int f = tid + 256 * r;
            *reinterpret_cast<float4 *>(As + (f >> 3) * LDK + 4 * (f & 7)) = ra[r];
        }
#pragma unroll
        for (int r = 0; r < B_F4; ++r) {
            int f = tid + 256 * r;
            *reinterpret_cast<float4 *>(Bs + (f >> 3) * LDK + 4 * (f & 7)) = rb[r];
        }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    const int nkb = (K + BK - 1) / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    const int li = lane & 31, lh = lane >> 5;
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nkb) load_tiles((kb + 1) * BK);
        const float *As = lds + buf * (BM + BN) * LDK + (wave * 32 + li) * LDK + lh * 16;
        const float *Bs = lds + buf * (BM + BN) * LDK + BM * LDK + li * LDK + lh * 16;
        float4 a4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) a4[j] = *reinterpret_cast<const float4 *>(As + 4 * j);
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            float4 b4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b4[j] = *reinterpret_cast<const float4 *>(Bs + c * 32 * LDK + 4 * j);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].x, b4[j].x, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].y, b4[j].y, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].z, b4[j].z, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].w, b4[j].w, acc[c], 0, 0, 0);
            }
        }
        if (kb + 1 < nkb) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // epilogue: accumulator register r of lane (li, lh) is C[row = (r&3) + 8*(r>>2) + 4*lh][col = li]
    const bool relu = flags & LPF_FLAG_RELU;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const int n = n0 + c * 32 + li;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m >= M) continue;
            float v = acc[c][r] + bv;
            if (addend) v += addend[m * ldadd + n];
            if (relu) v = fmaxf(v, 0.f);
            if constexpr (OUT_BF16) reinterpret_cast<uint16_t *>(C)[m * ldc + n] = lpf_f32_to_bf16(v);
            else C[m * ldc + n] = v;
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ tall and skinny
// M in the hundreds of thousands, K <= 128, N <= 256 (every GEMM of the encoder, the per-node projections of the
// attention, most of the training step): the tiled kernel above runs at 0.39 of the matrix pipe there -- 128 x 128 x 32
// tiles, two barriers per K block, four K blocks in all.  Here W^T stays in LDS for the life of a (persistent)
// workgroup, in A-operand order of v_mfma_f32_16x16x4_f32 with the ROWS OF A ON THE MFMA COLUMNS (the layout of
// gcn_fused.hip): a wavefront owns a tile of 16 rows, lane (j = lane % 16, q = lane / 16) loads row j's features
// 16 g + 4 q .. + 3 -- a wavefront reads its 16 rows as one contiguous piece -- and those registers are the B operands
// (k = 16 g + 4 q + u for the u-th MFMA of k-group g); the product leaves the lane with C[j][16 c + 4 q .. + 3]: bias,
// addend, ReLU and the store are float4 operations.  No staging of A, no barrier after the fill; wavefronts in their
// load phase cover the ones that multiply.  A row's result does not depend on M or on the tile it falls into.
namespace {

typedef float f32x4r __attribute__((ext_vector_type(4)));

struct RowsArgs {
    int64_t M;
    int N, K;
    const float *A; int64_t lda;
    const float *W; int64_t ldw;
    const float *bias;
    const float *addend; int64_t ldadd;
    float *C; int64_t ldc;
    uint32_t flags;
};

template <int NTI, int NTO, int NTH>
__global__ __launch_bounds__(NTH, 4) void gemm_rows_kernel(const RowsArgs P) {
    extern __shared__ __attribute__((aligned(16))) f32x4r gr_lds[];   // [NTI][NTO][64]
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    for (int e = threadIdx.x; e < NTI * NTO * 64; e += NTH) {
        const int l = e & 63, c = (e >> 6) % NTO, g = (e >> 6) / NTO;
        const int n = 16 * c + (l & 15), k = 16 * g + 4 * (l >> 4);
        f32x4r v = {0.f, 0.f, 0.f, 0.f};
        if (n < P.N && k < P.K) v = *reinterpret_cast<const f32x4r *>(P.W + (int64_t)n * P.ldw + k);
        gr_lds[e] = v;
    }
    __syncthreads();
    const int64_t n_tiles = (P.M + 15) >> 4;
    const int64_t n_waves = (int64_t)gridDim.x * (NTH / 64);
    const bool relu = P.flags & LPF_FLAG_RELU;
    for (int64_t tile = (int64_t)blockIdx.x * (NTH / 64) + (threadIdx.x >> 6); tile < n_tiles; tile += n_waves) {
        const int64_t row = tile * 16 + j;
        const bool live = row < P.M;
        const float *ar = P.A + (live ? row : P.M - 1) * P.lda + 4 * q;
        f32x4r b[NTI];
#pragma unroll
        for (int g = 0; g < NTI; ++g)
#ifdef GR_NOLOAD
            b[g] = (f32x4r){0.5f, 0.25f, (float)j, (float)g};
#else
            b[g] = (16 * g + 4 * q < P.K) ? *reinterpret_cast<const f32x4r *>(ar + 16 * g) : (f32x4r){0.f, 0.f, 0.f, 0.f};
#endif
        f32x4r y[NTO];
#pragma unroll
        for (int c = 0; c < NTO; ++c) y[c] = (f32x4r){0.f, 0.f, 0.f, 0.f};
#ifdef GR_NOMFMA
#pragma unroll
        for (int c = 0; c < NTO; ++c) y[c] = b[c % NTI];
#else
#pragma unroll
        for (int g = 0; g < NTI; ++g) {
            const f32x4r *lg = gr_lds + g * NTO * 64 + lane;
#pragma unroll
            for (int c0 = 0; c0 < NTO; c0 += 4) {      // four tiles' operands at a time
                f32x4r a[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] = (c0 + c < NTO) ? lg[(c0 + c) * 64] : (f32x4r){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c0 + c < NTO)
                            y[c0 + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][u], b[g][u], y[c0 + c], 0, 0, 0);
                // (keeps the operand reads of later blocks behind these MFMAs: hoisted together they spill)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
#ifdef GR_NOSTORE
        if (live && y[0][0] == 123.456f) {
#else
        if (live) {
#endif
#pragma unroll
            for (int c = 0; c < NTO; ++c) {
                const int n0 = 16 * c + 4 * q;
                if (n0 < P.N) {        // (N is a multiple of 4 on this path)
                    f32x4r v = y[c];
                    if (P.bias) v += *reinterpret_cast<const f32x4r *>(P.bias + n0);
                    if (P.addend) v += *reinterpret_cast<const f32x4r *>(P.addend + row * P.ldadd + n0);
                    if (relu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                    *reinterpret_cast<f32x4r *>(P.C + row * P.ldc + n0) = v;
                }
            }
        }
    }
}

template <int NTI, int NTO>
int gemm_rows_go(const RowsArgs &a, hipStream_t s, int n_cu) {
    constexpr size_t lds = (size_t)NTI * NTO * 64 * sizeof(f32x4r);
    constexpr int NTH = lds > 64 * 1024 ? 1024 : 512;       // two workgroups per CU, or one around a big table
    auto kern = gemm_rows_kernel<NTI, NTO, NTH>;
    LPF_SET_MAX_LDS(kern, lds);
    const int64_t n_tiles = (a.M + 15) >> 4;
    int64_t groups = (n_tiles + NTH / 64 - 1) / (NTH / 64);
    const int64_t cap = (int64_t)n_cu * (NTH == 1024 ? 1 : 2);
    if (groups > cap) groups = cap;
    hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(NTH), lds, s, a);
    return LPF_OK;
}

// returns LPF_ERR_UNSUPPORTED when the shape is not one of this path's
int gemm_rows_launch(const RowsArgs &a, hipStream_t s) {
    if (a.K > 128 || a.N > 256 || (a.K & 3) || (a.N & 3) || a.M < 4096) return LPF_ERR_UNSUPPORTED;
    if ((a.ldc & 3) || !lpf_aligned16(a.C) || (a.bias && !lpf_aligned16(a.bias)) ||
        (a.addend && ((a.ldadd & 3) || !lpf_aligned16(a.addend))))
        return LPF_ERR_UNSUPPORTED;
    const int n_cu = lpf_cu_count();
    if (n_cu == 0) return LPF_ERR_NO_DEVICE;
    const int nti = a.K <= 32 ? 2 : (a.K <= 64 ? 4 : 8), nto = a.N <= 32 ? 2 : (a.N <= 64 ? 4 : (a.N <= 128 ? 8 : 16));
#define LPF_ROWS(I, O) if (nti == I && nto == O) return gemm_rows_go<I, O>(a, s, n_cu)
    LPF_ROWS(2, 2); LPF_ROWS(2, 4); LPF_ROWS(2, 8); LPF_ROWS(2, 16);
    LPF_ROWS(4, 2); LPF_ROWS(4, 4); LPF_ROWS(4, 8); LPF_ROWS(4, 16);
    LPF_ROWS(8, 2); LPF_ROWS(8, 4); LPF_ROWS(8, 8); LPF_ROWS(8, 16);
#undef LPF_ROWS
    return LPF_ERR_UNSUPPORTED;
}

}  // namespace

namespace {
template <bool OUT_BF16>
int gemm_launch(int64_t M, int32_t N, int32_t K, const float *A, int64_t lda, const float *W, int64_t ldw,
                const float *bias, const float *addend, int64_t ldadd, float *C, int64_t ldc, uint32_t flags,
                void *stream) {
    if (M == 0 || N == 0) return LPF_OK;
    LPF_REQUIRE(M > 0 && N > 0 && K > 0 && A && W && C);
    LPF_REQUIRE(lda >= K && ldw >= K && ldc >= N && (lda & 3) == 0 && (ldw & 3) == 0);
    LPF_REQUIRE(lpf_aligned16(A) && lpf_aligned16(W));
    LPF_REQUIRE(!addend || ldadd >= N);
    LPF_REQUIRE((M + BM - 1) / BM < (1ll << 31));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if constexpr (!OUT_BF16) {   // tall and skinny: the row-streaming kernel
        const RowsArgs ra{M, N, K, A, lda, W, ldw, bias, addend, ldadd, C, ldc, flags};
        const int rc = gemm_rows_launch(ra, s);
        if (rc != LPF_ERR_UNSUPPORTED) {
            if (rc != LPF_OK) return rc;
            LPF_CHECK_LAUNCH();
            return LPF_OK;
        }
    }
    // Column tile: the one that wastes fewer padded columns; and for small problems the narrow tile, so that the grid
    // has at least two workgroups per CU.
    const int pad128 = ((N + 127) / 128) * 128 - N, pad64 = ((N + 63) / 64) * 64 - N;
    const int64_t blocks128 = ((M + BM - 1) / BM) * ((N + 127) / 128);
    if (pad64 < pad128 || blocks128 < 2 * 256) {
        dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + 63) / 64));
        hipLaunchKernelGGL((gemm_f32_kernel<64, OUT_BF16>), grid, dim3(256), 0, s, M, N, K, A, lda, W, ldw, bias,
                           addend, ldadd, C, ldc, flags);
    } else {
        dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + 127) / 128));
        hipLaunchKernelGGL((gemm_f32_kernel<128, OUT_BF16>), grid, dim3(256), 0, s, M, N, K, A, lda, W, ldw, bias,
                           addend, ldadd, C, ldc, flags);
    }
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}
}  // namespace

extern "C" int lpf_gemm_f32(int64_t M, int32_t N, int32_t K, const float *A, int64_t lda, const float *W,
                            int64_t ldw, const float *bias, const float *addend, int64_t ldadd, float *C,
                            int64_t ldc, uint32_t flags, void *stream) {
    return gemm_launch<false>(M, N, K, A, lda, W, ldw, bias, addend, ldadd, C, ldc, flags, stream);
}

extern "C" int lpf_gemm_f32_out_bf16(int64_t M, int32_t N, int32_t K, const float *A, int64_t lda, const float *W,
                                     int64_t ldw, const float *bias, const float *addend, int64_t ldadd,
                                     void *C_bf16, int64_t ldc, uint32_t flags, void *stream) {
    return gemm_launch<true>(M, N, K, A, lda, W, ldw, bias, addend, ldadd, static_cast<float *>(C_bf16), ldc, flags,
                             stream);
}

// ------------------------------------------------------------------------------------------------ C = A^T B
// Weight gradient of a Linear layer: dW[N, K] = dY[M, N]^T X[M, K] -- the reduction runs over the M rows (hundreds of
// thousands in the encoder), the output is one or a few 128 x 128 tiles.  lpf_gemm_f32 on transposed copies gave that
// single tile to a single workgroup (1.1 ms per call, 3/4 of a training step).  Here the rows are split into chunks,
// a workgroup computes the partial product of one chunk for one output tile (a wavefront: 32 output rows x up to 128
// columns; v_mfma_f32_32x32x2_f32 with k = row of the chunk: lane (i, half) feeds A[i][m = 2 s + half] =
// dY[m][n0 + i] and B[m][c0 + i] = X[m][c0 + i], both coalesced 128-byte row pieces straight from global memory),
// writes it to a partial buffer, and a second kernel adds the partials in order (deterministic).
//
// Round 6.  (a) A wavefront keeps P = 4 rounds of eight rows in flight (P register sets, the loop unrolled over them):
// a round's 20 loads are issued three rounds of MFMAs (3 x 1,024 matrix-pipe cycles, twice that with the SIMD's other
// wavefront) before they are awaited.  With the loads issued and awaited inside the round (rounds 4-5) a workgroup spent
// most of its life waiting: 125 us per call on the collab-like encoder shape.
// Every load is unconditional -- rows and columns past the end are clamped to the last valid one and zeroed when
// used -- so that the wait before a round's MFMAs counts exactly the younger loads it may leave in flight.
// Measured alone (rocprofv3, collab-like encoder shape M = 235,868, N = K = 128, tools/r06_tn_isolate.sh): 113 us; the
// loads alone (-DLPF_TN_NOMFMA) 52 us, the matrix work alone (-DLPF_TN_NOLOAD) 90 us -- 1,856 MFMAs per SIMD at ~92
// cycles each where the instruction's issue rate is 64: the launch is bound by the fp32 matrix pipe, not by memory.
// (b) Narrow outputs: with N <= 64 (<= 32) the workgroup's four wavefronts form G = 2 (4) groups that take every G-th
// round of the chunk and write partials of their own; K <= 64 instantiates two column tiles instead of four.  (Before:
// wavefronts 2, 3 and column tiles 2, 3 multiplied zeros -- 296 us per call on the ppa-like encoder, D = 64.)
namespace {

__host__ static inline int tn_groups(int N) { return N <= 32 ? 4 : (N <= 64 ? 2 : 1); }

// rows of a chunk: about 512 chunks per output tile column (two workgroups per CU), at least 64 rows each
__host__ static inline int64_t tn_chunk_rows(int64_t M) {
    int64_t rows = (M + 511) / 512;
    if (rows < 64) rows = 64;
    return (rows + 7) & ~7ll;  // whole rounds of eight rows
}
__host__ static inline int64_t tn_chunks(int64_t M) {
    const int64_t rows = tn_chunk_rows(M);
    return (M + rows - 1) / rows;
}

template <int KC, int P>
__global__ __launch_bounds__(256) void gemm_tn_partial_kernel(int64_t M, int N, int K, const float *__restrict__ A,
                                                              int64_t lda, const float *__restrict__ B, int64_t ldb,
                                                              float *__restrict__ part, int64_t rows_per_chunk, int G,
                                                              float *__restrict__ colpart) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int per_group = 4 / G;                           // wavefronts (32-row output tiles) of a group
    const int tile = wave % per_group, grp = wave / per_group;
    const int n0 = blockIdx.y * 128 + tile * 32, k0 = blockIdx.z * (32 * KC);
    if (n0 >= N) return;                                   // (no barrier in this kernel)
    const int64_t m_lo = (int64_t)blockIdx.x * rows_per_chunk;
    int64_t m_hi = m_lo + rows_per_chunk;
    if (m_hi > M) m_hi = M;
    f32x16 acc[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    const bool a_ok = n0 + li < N;
    bool b_ok[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) b_ok[c] = k0 + 32 * c + li < K;
    const float *ap = A + (a_ok ? n0 + li : N - 1);
    const float *bp[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) bp[c] = B + (b_ok[c] ? k0 + 32 * c + li : K - 1);
    auto fetch = [&](int64_t m, float (&a)[4], float (&b)[4][KC]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            int64_t mm = m + 2 * s + lh;
            if (mm > M - 1) mm = M - 1;
#ifdef LPF_TN_NOLOAD      // (timing aid: the matrix work alone)
            a[s] = __int_as_float(0x3f800000 | ((int)mm & 0xffff));
#pragma unroll
            for (int c = 0; c < KC; ++c) b[s][c] = __int_as_float(0x3f800000 | (((int)mm + c) & 0xffff));
#else
            a[s] = ap[mm * lda];
#pragma unroll
            for (int c = 0; c < KC; ++c) b[s][c] = bp[c][mm * ldb];
#endif
        }
    };
    float asum = 0.f;      // column sum of A over this wavefront's rows (the bias gradient beside dW), rows of parity lh
    auto mma = [&](int64_t m, const float (&a)[4], const float (&b)[4][KC]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool ok = m + 2 * s + lh < m_hi;
            const float x = (ok && a_ok) ? a[s] : 0.f;
            asum += x;
#ifdef LPF_TN_NOMFMA      // (timing aid: the loads alone)
#pragma unroll
            for (int c = 0; c < KC; ++c) acc[c][s] += x * ((ok && b_ok[c]) ? b[s][c] : 0.f);
#else
#pragma unroll
            for (int c = 0; c < KC; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, (ok && b_ok[c]) ? b[s][c] : 0.f, acc[c], 0, 0, 0);
#endif
        }
    };
    const int64_t step = 8 * G;                            // rows between two rounds of this group
    float a[P][4], b[P][4][KC];
#pragma unroll
    for (int p = 0; p < P; ++p) fetch(m_lo + 8 * grp + step * p, a[p], b[p]);
    for (int64_t m = m_lo + 8 * grp; m < m_hi; m += step * P) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            mma(m + step * p, a[p], b[p]);                 // (a round past the chunk's end multiplies zeros)
            // (the scheduler otherwise sinks the loads to just above their use, three rounds later: no prefetch left)
            __builtin_amdgcn_sched_barrier(0);
            fetch(m + step * (P + p), a[p], b[p]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // partial tile: part[chunk * G + group][n][k] (n < N, k < K), accumulator register r of lane (li, lh) = row
    // (r & 3) + 8 (r >> 2) + 4 lh
    if (colpart && blockIdx.z == 0) {      // colpart[chunk * G + group][n]
        const float tot = asum + __shfl_xor(asum, 32, 64);
        if (lh == 0 && a_ok) colpart[((int64_t)blockIdx.x * G + grp) * N + n0 + li] = tot;
    }
    float *pp = part + ((int64_t)blockIdx.x * G + grp) * N * K;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const int k = k0 + 32 * c + li;
        if (k >= K) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (n < N) pp[(int64_t)n * K + k] = acc[c][r];
        }
    }
}

// 32 output elements x 8 slices of the chunk list per workgroup: a slice adds its chunks in order, the slices meet in
// LDS and are added in slice order (deterministic)
// (elements elems .. elems + n_col - 1: the column sums of A, from colpart[chunk][n])
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(int64_t elems, int chunks, const float *__restrict__ part,
                                                             float *__restrict__ C, int K, int64_t ldc, int n_col,
                                                             const float *__restrict__ colpart,
                                                             float *__restrict__ colsum) {
    __shared__ float red[8][33];
    const int ex = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int64_t e = (int64_t)blockIdx.x * 32 + ex;
    float s = 0.f;
    if (e < elems) {
        for (int c = slice; c < chunks; c += 8) s += part[(int64_t)c * elems + e];
    } else if (e < elems + n_col) {
        for (int c = slice; c < chunks; c += 8) s += colpart[(int64_t)c * n_col + (e - elems)];
    }
    red[slice][ex] = s;
    __syncthreads();
    if (slice == 0 && e < elems + n_col) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][ex];
        if (e < elems) C[(e / K) * ldc + (e % K)] = t;
        else colsum[e - elems] = t;
    }
}

}  // namespace

extern "C" int64_t lpf_gemm_tn_workspace_floats(int64_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return tn_chunks(M) * tn_groups(N) * ((int64_t)N * K + N);      // (+ N: the column-sum partials of _colsum_f32)
}

namespace {
int tn_launch(int64_t M, int32_t N, int32_t K, const float *A, int64_t lda, const float *B, int64_t ldb, float *C,
              int64_t ldc, float *colsum, float *workspace, void *stream) {
    if (N == 0) return LPF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (K == 0 && !colsum) return LPF_OK;
    LPF_REQUIRE(M >= 0 && N > 0 && K > 0 && C && ldc >= K && (M == 0 || (A && B && workspace && lda >= N && ldb >= K)));
    if (M == 0) {
        for (int n = 0; n < N; ++n)
            if (hipMemsetAsync(C + (int64_t)n * ldc, 0, sizeof(float) * K, s) != hipSuccess) return LPF_ERR_LAUNCH;
        if (colsum && hipMemsetAsync(colsum, 0, sizeof(float) * N, s) != hipSuccess) return LPF_ERR_LAUNCH;
        return LPF_OK;
    }
    const int64_t rows = tn_chunk_rows(M), chunks = tn_chunks(M);
    const int G = tn_groups(N);
    float *colpart = colsum ? workspace + chunks * G * (int64_t)N * K : nullptr;
    // (K > 64 as two-tile launches of twice the workgroups, four wavefronts per SIMD: 5-10 % slower, measured)
    if (K <= 64) {
        dim3 grid((unsigned)chunks, (unsigned)((N + 127) / 128), (unsigned)((K + 63) / 64));
        hipLaunchKernelGGL((gemm_tn_partial_kernel<2, 4>), grid, dim3(256), 0, s, M, N, K, A, lda, B, ldb, workspace, rows, G,
                           colpart);
    } else {
        dim3 grid((unsigned)chunks, (unsigned)((N + 127) / 128), (unsigned)((K + 127) / 128));
        hipLaunchKernelGGL((gemm_tn_partial_kernel<4, 3>), grid, dim3(256), 0, s, M, N, K, A, lda, B, ldb, workspace, rows, G,
                           colpart);
    }
    const int64_t elems = (int64_t)N * K;
    const int n_col = colsum ? N : 0;
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)((elems + n_col + 31) / 32)), dim3(256), 0, s, elems,
                       (int)(chunks * G), workspace, C, K, ldc, n_col, colpart, colsum);
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}
}  // namespace

extern "C" int lpf_gemm_tn_f32(int64_t M, int32_t N, int32_t K, const float *A, int64_t lda, const float *B,
                               int64_t ldb, float *C, int64_t ldc, float *workspace, void *stream) {
    if (N == 0 || K == 0) return LPF_OK;
    return tn_launch(M, N, K, A, lda, B, ldb, C, ldc, nullptr, workspace, stream);
}

extern "C" int lpf_gemm_tn_colsum_f32(int64_t M, int32_t N, int32_t K, const float *A, int64_t lda, const float *B,
                                      int64_t ldb, float *C, int64_t ldc, float *colsum, float *workspace,
                                      void *stream) {
    if (N == 0 || K == 0) return LPF_OK;
    LPF_REQUIRE(colsum != nullptr);
    return tn_launch(M, N, K, A, lda, B, ldb, C, ldc, colsum, workspace, stream);
}
