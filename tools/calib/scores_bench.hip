// Stand-alone timing harness for lpf_pair_scores_f32 on synthetic data (calibration only, not part of the product).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../lpformer_amd/csrc/pair_attn.hip"
void lpf_set_hip_error(hipError_t) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <typename T> T *dev(const std::vector<T> &h) { T *d; CK(hipMalloc(&d, h.size() * sizeof(T))); CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }
int main(int argc, char **argv) {
    const int D = 128, N = 235868; const int64_t bs = 32768, n1 = argc > 1 ? atoll(argv[1]) : 365000, n0 = 4100, n2 = 3;
    const int64_t ntot = n0 + n1 + n2, cap = argc > 2 ? atoll(argv[2]) : 3600000;
    srand(1);
    std::vector<int64_t> tp(3 * (bs + 1), 0); tp[bs] = n0; tp[(bs + 1) + bs] = n1; tp[2 * (bs + 1) + bs] = n2;
    std::vector<int32_t> pr(cap), nd(cap); std::vector<float> pa(cap), pb(cap);
    for (int64_t i = 0; i < ntot; ++i) { pr[i] = (int32_t)((i * bs) / ntot); nd[i] = rand() % N; pa[i] = (rand() % 1000) * 1e-4f; pb[i] = (rand() % 1000) * 1e-4f; }
    std::vector<float> Z((size_t)N * D), q((size_t)bs * D), tab(3 * D * 4), stat(24), wf(3 * D * D), bf(3 * D), att(D);
    for (auto *v : {&Z, &q, &tab, &stat, &wf, &bf, &att}) for (auto &x : *v) x = (rand() % 2001 - 1000) * 1e-3f;
    for (int t = 0; t < 3; ++t) for (int k = 0; k < 3; ++k) stat[8 * t + k] = 0.5f;
    auto *dtp = dev(tp); auto *dpr = dev(pr); auto *dnd = dev(nd); auto *dpa = dev(pa); auto *dpb = dev(pb);
    auto *dZ = dev(Z); auto *dq = dev(q); auto *dtab = dev(tab); auto *dstat = dev(stat); auto *dwf = dev(wf); auto *dbf = dev(bf); auto *datt = dev(att);
    float *score; CK(hipMalloc(&score, cap * 4));
    auto run = [&] { int rc = lpf_pair_scores_f32(D, dtp, bs, dpr, dnd, dpa, dpb, dZ, D, dq, D, dtab, dstat, dwf, dbf, datt, score, cap, 0); if (rc) { printf("rc=%d\n", rc); exit(1);} };
    run(); CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20; hipEventRecord(e0); for (int i = 0; i < reps; ++i) run(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("entries=%lld cap=%lld : %.1f us/launch  %.1f TFLOP/s\n", (long long)ntot, (long long)cap, ms * 1e3, ntot * 2.0 * D * D / ms / 1e9);
    return 0;
}
