// Shared by the two selection kernels that run over the per-model WALK INDEXES (lpformer_amd/graph.py build_walk_index):
// select3.hip (flat slot space, type-major output, two launches) and select4.hip (blocks of 64 pairs, pair-major output,
// one launch).  What is shared is the ALGEBRA -- which rows a pair walks, what a walked candidate is to the other
// endpoint, the reference's fp32 round trip and thresholds (src/models/link_transformer.py:214-319, 434-481) --, so the
// two paths cannot drift apart: both are bit-exact against the same fixtures.
#pragma once
#include "select_common.h"

// the reference's fp32 round trip must be evaluated op by op: no fused multiply-add in the files that include this
#pragma clang fp contract(off)

namespace walk {

constexpr uint32_t FROM_B = 0x80000000u;     // pair word, bit 31: a one-hop node that is a neighbour of b
constexpr uint32_t HASH_MUL = 2654435761u;
constexpr int BUCKET = 8;
// mini filter of a node's union row (lpformer_amd/graph.py mini_filters): 32 words, two bits of one word per key
constexpr uint32_t BLOOM_MUL1 = 0x85EBCA6Bu, BLOOM_MUL2 = 0xC2B2AE35u, MINI_SALT = 0x9E3779B9u;
constexpr int MINI_WORDS = 32;

// walk kinds
constexpr int K_FULL = 0, K_A1 = 1, K_PX = 2, K_T0 = 3;
constexpr int KF_SRC_A = 4;    // the walked row belongs to endpoint a (its value is pa, the looked-up one pb)
constexpr int KF_SIDE_B = 8;   // a one-hop node emitted by this walk is a neighbour of b (flag bit 31 of the pair word)

struct alignas(16) NodeRec {   // lpformer_amd/graph.py WalkIndex.rec: where node i's rows start in the index arrays
    int64_t adj0, a10, px0, t00, u0;   // element offsets into adj_cv / a1_cv / px_cv / t0_cv; entry offset into u_cv
    int32_t deg, n_a1, n_px, n_t0, u_nb, pad;
};
static_assert(sizeof(NodeRec) == 64, "one 64-byte record per node");

struct alignas(16) Walk3 {
    const int2 *src;     // {node, value bits} entries of the walked row
    int64_t u0;          // first entry of the looked-up endpoint's U row
    int32_t unb;         // ... and its bucket count
    int32_t start;       // slot (inside the pair) at which this walk starts
    int32_t kind;        // K_* | KF_*
    int32_t len;
};
struct alignas(16) PairDesc3 {
    Walk3 w[3];
    int32_t total, a, b, pad[5];
};
static_assert(sizeof(Walk3) == 32 && sizeof(PairDesc3) == 128, "descriptor is one 128-byte line");

__device__ __forceinline__ uint32_t bloom_hash(uint32_t v) {
    uint32_t h = v * BLOOM_MUL1;
    h ^= h >> 15;
    h *= BLOOM_MUL2;
    return h ^ (h >> 13);
}

// fl32((fl32(fl32(p*t)+t)-t)/t) for t in {1,2}: p*1, p*2, x/1 and x/2 are exact, only the add and subtract round
__device__ __forceinline__ float rt1(float p) { return __fsub_rn(__fadd_rn(p, 1.0f), 1.0f); }
__device__ __forceinline__ float rt2(float p) { return 0.5f * __fsub_rn(__fadd_rn(p * 2.0f, 2.0f), 2.0f); }

// The three walks of pair (a, b) from the endpoints' node records (the plan of DESIGN.md 5.2): walk 0 / 1 = the nodes
// that are neighbours of a / b, walk 2 = the >1-hop candidates.  `d` must be zero on entry.
__device__ __forceinline__ void build_desc(PairDesc3 &d, int64_t a, int64_t b, const NodeRec (&r)[2], const int2 *adj_cv,
                                           const int2 *a1_cv, const int2 *px_cv, const int2 *t0_cv, int mode_cn,
                                           int use_px) {
    const int s = r[0].deg <= r[1].deg ? 0 : 1;   // the endpoint whose whole adjacency row is walked
    int start = 0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {                 // walk e: the nodes that are neighbours of endpoint e
        const int o = 1 - e;
        Walk3 &w = d.w[e];
        const int side = e == 1 ? KF_SIDE_B : 0;
        if (e == s) {          // common neighbours + this side's one-hop nodes
            w.src = adj_cv + r[e].adj0; w.len = r[e].deg; w.u0 = r[o].u0; w.unb = r[o].u_nb;
            w.kind = K_FULL | (e == 0 ? KF_SRC_A : 0) | side;
        } else if (mode_cn) {
            w.len = 0;
        } else if (use_px && r[o].n_px < r[e].n_a1) {   // walk the other endpoint's strong non-neighbours
            w.src = px_cv + r[o].px0; w.len = r[o].n_px; w.u0 = r[e].u0; w.unb = r[e].u_nb;
            w.kind = K_PX | (o == 0 ? KF_SRC_A : 0) | side;
        } else {                                          // walk this endpoint's strong neighbours
            w.src = a1_cv + r[e].a10; w.len = r[e].n_a1; w.u0 = r[o].u0; w.unb = r[o].u_nb;
            w.kind = K_A1 | (e == 0 ? KF_SRC_A : 0) | side;
        }
        w.start = start;
        start += w.len;
    }
    Walk3 &w = d.w[2];
    w.start = start;
    if (t0_cv) {
        const int e = r[0].n_t0 <= r[1].n_t0 ? 0 : 1, o = 1 - e;
        w.src = t0_cv + r[e].t00; w.len = r[e].n_t0; w.u0 = r[o].u0; w.unb = r[o].u_nb;
        w.kind = K_T0 | (e == 0 ? KF_SRC_A : 0);
        start += w.len;
    }
    d.total = start;
    d.a = (int32_t)a; d.b = (int32_t)b;
}

// does candidate x pass the looked-up endpoint's mini filter (32 words `flt`)?  A candidate that fails is not in the
// union row: its bucket is never read, "not found" is what the bucket would have said.
__device__ __forceinline__ bool mini_pass(const uint32_t *flt, int32_t x) {
    const uint32_t mh = bloom_hash((uint32_t)x ^ MINI_SALT);
    const uint32_t mw = flt[mh >> 27];
    return ((mw >> (mh & 31u)) & (mw >> ((mh >> 5) & 31u)) & 1u) != 0u;
}

__device__ __forceinline__ uint32_t bucket_of(int32_t x, int32_t unb) {
    return __umulhi((uint32_t)x * HASH_MUL, (uint32_t)unb);
}

// What a walked candidate (x, own value ws) of a walk of kind `kindw` is, given the bucket of the looked-up endpoint's
// union row (bv; nodes of -1 when no bucket was fetched): code 0 = not selected, 1 = common neighbour, 2 = one-hop,
// 3 = >1-hop, | 4 = a one-hop node that is a neighbour of b; va / vb = the round-tripped PPR values of a and b.
struct Typed {
    int code;
    float va, vb;
};
__device__ __forceinline__ Typed type_slot(int32_t x, float ws, int kindw, const int4 (&bv)[BUCKET / 2], float th_cn,
                                           float th_1, float th_n, int mode_cn) {
    bool found = false;
    int bitsv = 0;
#pragma unroll
    for (int q = 0; q < BUCKET / 2; ++q) {
        if (bv[q].x == x) { found = true; bitsv = bv[q].y; }
        if (bv[q].z == x) { found = true; bitsv = bv[q].w; }
    }
    const bool adj = found && bitsv < 0;                     // sign bit: x is adjacent to that endpoint
    const float lv = __int_as_float(bitsv & 0x7fffffff);     // its PPR value (0: nothing stored)
    const int kind = kindw & 3;
    const bool cn = kind == K_FULL && adj;
    const bool hop = kind == K_FULL ? !adj : (kind == K_A1 ? !adj : (kind == K_PX ? adj : false));
    const bool far = kind == K_T0 && found && !adj;
    // the reference's round trips (t = 2 for a common neighbour, 1 otherwise; mode "cn": 1)
    const bool two = cn && !mode_cn;
    const float rs = two ? rt2(ws) : rt1(ws);
    const float rl = two ? rt2(lv) : rt1(lv);
    int c = 0;
    if (cn) c = (rs >= th_cn && rl >= th_cn) ? 1 : 0;
    else if (hop) c = (!mode_cn && rs >= th_1 && rl >= th_1) ? 2 : 0;
    else if (far) c = (ws > 0.f && lv > 0.f && rs >= th_n && rl >= th_n) ? 3 : 0;
    const bool src_a = kindw & KF_SRC_A;
    Typed t;
    t.code = c | ((c == 2 && (kindw & KF_SIDE_B)) ? 4 : 0);
    t.va = src_a ? rs : rl;
    t.vb = src_a ? rl : rs;
    return t;
}

}  // namespace walk
