// Library-level entry points: ABI version, error strings, device query.
#include <string.h>

#include "lpf_common.h"

static thread_local char g_hip_err[256] = "";

void lpf_set_hip_error(hipError_t e) {
    const char *s = hipGetErrorString(e);
    strncpy(g_hip_err, s ? s : "unknown HIP error", sizeof(g_hip_err) - 1);
    g_hip_err[sizeof(g_hip_err) - 1] = 0;
}

int lpf_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LPF_MAX_DEVICES) return -1;
    return dev;
}

int lpf_cu_count() {
    static LpfPerDevice cus;
    const int dev = lpf_current_device();
    if (dev < 0) return 0;
    int n = cus.v[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        n = prop.multiProcessorCount;
        cus.v[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

int lpf_set_max_lds(LpfPerDevice &once, const void *kern, int bytes) {
    const int dev = lpf_current_device();
    if (dev < 0) return LPF_ERR_NO_DEVICE;
    if (once.v[dev].load(std::memory_order_relaxed)) return LPF_OK;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        lpf_set_hip_error(e);
        return LPF_ERR_LAUNCH;
    }
    once.v[dev].store(1, std::memory_order_relaxed);
    return LPF_OK;
}

int lpf_blocks_per_cu(LpfPerDevice &cache, const void *kern, int threads, size_t lds, int fallback) {
    const int dev = lpf_current_device();
    if (dev < 0) return fallback;
    int occ = cache.v[dev].load(std::memory_order_relaxed);
    if (occ == 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, threads, lds) != hipSuccess || occ < 1) occ = fallback;
        cache.v[dev].store(occ, std::memory_order_relaxed);
    }
    return occ;
}

extern "C" int lpf_abi_version(void) { return LPF_ABI_VERSION; }

extern "C" const char *lpf_last_hip_error(void) { return g_hip_err; }

extern "C" const char *lpf_strerror(int code) {
    switch (code) {
        case LPF_OK: return "ok";
        case LPF_ERR_INVALID: return "invalid argument (null pointer, size, leading dimension or alignment)";
        case LPF_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
        case LPF_ERR_LAUNCH: return "HIP launch/runtime error (see lpf_last_hip_error)";
        case LPF_ERR_NO_DEVICE: return "no HIP device visible";
        default: return "unknown lpformer_hip error code";
    }
}

extern "C" int lpf_device_info(int *cu_count, int *lds_bytes_per_cu, int *wave_size, char *arch_name,
                               int arch_name_len) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return LPF_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) {
        lpf_set_hip_error(e);
        return LPF_ERR_NO_DEVICE;
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (wave_size) *wave_size = prop.warpSize;
    if (arch_name && arch_name_len > 0) {
        strncpy(arch_name, prop.gcnArchName, (size_t)arch_name_len - 1);
        arch_name[arch_name_len - 1] = 0;
    }
    return LPF_OK;
}
