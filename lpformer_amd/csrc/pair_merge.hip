// Attention output of a pair from the one-pass attention kernel's records, for callers that want the FEATURES rather
// than the score (LinkTransformer.calc_pairwise / forward, src/models/link_transformer.py:132-178): the merge that
// tail_chain.hip does in front of its GEMMs, as a kernel of its own --
//   o = post_att_norm( sum_t e^{m_t - M} acc_t / (sum_t e^{m_t - M} l_t + 1e-16) + att_bias )      (layers.py:78,220)
// written beside the count features of get_structure_cnts (link_transformer.py:340-356), taken from the segment
// pointers: out[p] = [o (D) | n_cn, n_1hop, (n_non1hop,) n_cn + n_1hop].  G = D/4 lanes per pair, 16 bytes of a record
// per lane and read; a segment that crossed 16-entry units is a chain of boundary records (pair_fused.hip) whose
// addresses follow from the segment pointers.  Bound: HBM (one to a few (D+4)-float records per pair and type).
#include "lpf_common.h"

namespace {

template <int G>
__global__ __launch_bounds__(256) void pair_merge_kernel(int64_t bs, int D, int n_counts, const float *__restrict__ part,
                                                         const float *__restrict__ bnd, int64_t units_cap,
                                                         const int32_t *__restrict__ type_ptr,
                                                         const float *__restrict__ att_bias,
                                                         const float *__restrict__ ln_g, const float *__restrict__ ln_b,
                                                         const int64_t *__restrict__ sel_ctl, float *__restrict__ out,
                                                         int64_t ldo) {
    constexpr int PPW = 64 / G;
    const int lane = threadIdx.x & 63, grp = lane / G, lig = lane % G;
    const int off = 4 * lig;
    const int64_t rs = D + 4;
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const bool bad = sel_ctl && sel_ctl[3] != 0;  // the batch did not fit the selection workspace: NaN, never wrong
    for (int64_t p0 = wave_id * PPW; p0 < bs; p0 += n_waves * PPW) {
        const int64_t p = p0 + grp;
        const bool live = p < bs;
        const int64_t pp = live ? p : bs - 1;
        float mx = -INFINITY, den = 0.f;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        int cnt[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int32_t *tp = type_ptr + (int64_t)t * (bs + 1) + pp;
            const int lo = tp[0], hi = tp[1];
            cnt[t] = hi - lo;
            if (hi <= lo || ((hi - 1) >> 4) >= units_cap) continue;
            const int u0 = lo >> 4, n_p = ((hi - 1) >> 4) - u0 + 1;
            for (int pi = 0; pi < n_p; ++pi) {
                const float *rec = n_p == 1 ? part + ((int64_t)t * bs + pp) * rs
                                            : bnd + ((((int64_t)t * units_cap + u0 + pi) * 2) + (pi == 0 ? 1 : 0)) * rs;
                const float4 h = *reinterpret_cast<const float4 *>(rec + D);
                const float4 b = *reinterpret_cast<const float4 *>(rec + off);
                const float mn = fmaxf(mx, h.x);
                const float sa = __expf(mx - mn), sb = __expf(h.x - mn);
                den = fmaf(den, sa, h.y * sb);
                v.x = v.x * sa + b.x * sb; v.y = v.y * sa + b.y * sb;
                v.z = v.z * sa + b.z * sb; v.w = v.w * sa + b.w * sb;
                mx = mn;
            }
        }
        const float inv = 1.0f / (den + 1e-16f);
        const float4 ab = *reinterpret_cast<const float4 *>(att_bias + off);
        float4 y = make_float4(v.x * inv + ab.x, v.y * inv + ab.y, v.z * inv + ab.z, v.w * inv + ab.w);
        const float mean = lpf_group_sum<G>(y.x + y.y + y.z + y.w) / (float)D;
        const float4 d = make_float4(y.x - mean, y.y - mean, y.z - mean, y.w - mean);
        const float var = lpf_group_sum<G>(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) / (float)D;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        const float4 gg = *reinterpret_cast<const float4 *>(ln_g + off), bb = *reinterpret_cast<const float4 *>(ln_b + off);
        y = make_float4(d.x * rstd * gg.x + bb.x, d.y * rstd * gg.y + bb.y, d.z * rstd * gg.z + bb.z,
                        d.w * rstd * gg.w + bb.w);
        if (bad) y = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
        if (!live) continue;
        float *o = out + p * ldo;
        *reinterpret_cast<float4 *>(o + off) = y;
        if (lig == 0) {
            const float n0 = (float)cnt[0], n1 = (float)cnt[1], n2 = (float)cnt[2];
            if (n_counts == 4) { o[D] = n0; o[D + 1] = n1; o[D + 2] = n2; o[D + 3] = n0 + n1; }
            else if (n_counts == 3) { o[D] = n0; o[D + 1] = n1; o[D + 2] = n0 + n1; }
            else { o[D] = n0; }  // mask mode "cn": get_count alone (link_transformer.py:154-155)
        }
    }
}

}  // namespace

extern "C" int lpf_pair_attention_merge_f32(int64_t bs, int32_t D, int32_t n_counts, const float *part, const float *bnd,
                                            int64_t units_cap, const int32_t *type_ptr, const float *att_bias,
                                            const float *ln_g, const float *ln_b, const int64_t *sel_ctl, float *out,
                                            int64_t ldo, void *stream) {
    if (bs == 0) return LPF_OK;
    LPF_REQUIRE(bs > 0 && part && bnd && units_cap > 0 && type_ptr && att_bias && ln_g && ln_b && out);
    LPF_REQUIRE((n_counts == 1 || n_counts == 3 || n_counts == 4) && ldo >= D + n_counts && (ldo & 3) == 0);
    LPF_REQUIRE(lpf_aligned16(part) && lpf_aligned16(bnd) && lpf_aligned16(att_bias) && lpf_aligned16(ln_g) &&
                lpf_aligned16(ln_b) && lpf_aligned16(out));
    hipStream_t s = static_cast<hipStream_t>(stream);
    int64_t blocks;
#define LPF_MERGE(GG)                                                                                             \
    do {                                                                                                          \
        blocks = (bs + 4 * (64 / GG) - 1) / (4 * (64 / GG));                                                      \
        if (blocks > 256 * 16) blocks = 256 * 16;                                                                 \
        hipLaunchKernelGGL(pair_merge_kernel<GG>, dim3((unsigned)blocks), dim3(256), 0, s, bs, D, n_counts, part, \
                           bnd, units_cap, type_ptr, att_bias, ln_g, ln_b, sel_ctl, out, ldo);                    \
    } while (0)
    switch (D) {
        case 32: LPF_MERGE(8); break;
        case 64: LPF_MERGE(16); break;
        case 128: LPF_MERGE(32); break;
        case 256: LPF_MERGE(64); break;
        default: return LPF_ERR_UNSUPPORTED;
    }
#undef LPF_MERGE
    LPF_CHECK_LAUNCH();
    return LPF_OK;
}
