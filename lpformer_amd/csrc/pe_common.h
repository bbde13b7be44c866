// First layer of the PPR positional MLP with its LayerNorm in closed form (shared by the attention kernels).
// Reference: get_pos_encodings (src/models/link_transformer.py:182-211) -> MLP (src/models/other_models.py:125-138).
#pragma once
#include "lpf_common.h"

struct PeStat {
    float c00, c11, cbb, c01, c0b, c1b;
};

__device__ __forceinline__ PeStat pe_load_stat(const float *__restrict__ pe_stat, int t) {
    PeStat s;
    s.c00 = pe_stat[8 * t + 0]; s.c11 = pe_stat[8 * t + 1]; s.cbb = pe_stat[8 * t + 2];
    s.c01 = pe_stat[8 * t + 3]; s.c0b = pe_stat[8 * t + 4]; s.c1b = pe_stat[8 * t + 5];
    return s;
}

// LayerNorm statistics of u_k = w0_k*x + w1_k*y + b_k over k, from the centred second moments of (w0, w1, b):
// mean-free by construction (pe_tab already holds centred, gamma-scaled coefficients), so only 1/std is needed.
__device__ __forceinline__ float pe_rstd(const PeStat &s, float x, float y) {
    const float var = s.c00 * x * x + s.c11 * y * y + s.cbb + 2.0f * (s.c01 * x * y + s.c0b * x + s.c1b * y);
    return 1.0f / sqrtf(fmaxf(var, 0.0f) + 1e-5f);
}

// h_k = ReLU(LN(W1 [pa,pb] + b1))_k + ReLU(LN(W1 [pb,pa] + b1))_k ; k = (g (w0 - mean), g (w1 - mean), g (b - mean), beta)
__device__ __forceinline__ float pe_hidden(const float4 k, float pa, float pb, float r_ab, float r_ba) {
    const float u_ab = fmaf(k.x, pa, fmaf(k.y, pb, k.z));
    const float u_ba = fmaf(k.x, pb, fmaf(k.y, pa, k.z));
    return fmaxf(fmaf(r_ab, u_ab, k.w), 0.0f) + fmaxf(fmaf(r_ba, u_ba, k.w), 0.0f);
}
