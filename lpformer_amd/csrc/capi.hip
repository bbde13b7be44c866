// Library-level entry points: ABI version, error strings, device query.
#include <string.h>

#include "lpf_common.h"

static thread_local char g_hip_err[256] = "";

void lpf_set_hip_error(hipError_t e) {
    const char *s = hipGetErrorString(e);
    strncpy(g_hip_err, s ? s : "unknown HIP error", sizeof(g_hip_err) - 1);
    g_hip_err[sizeof(g_hip_err) - 1] = 0;
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
