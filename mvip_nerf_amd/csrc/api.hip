// ABI bookkeeping: version, error strings, last HIP error, device facts.
#include "common.h"
#include <string.h>

namespace mvip {
static thread_local hipError_t g_last = hipSuccess;
void set_last_error(hipError_t e) { g_last = e; }
}  // namespace mvip

extern "C" int mvip_abi_version(void) { return MVIP_ABI_VERSION; }

// 1 when the library was compiled with a -DMVIP_EXPERIMENT_* macro (timing experiments: parts of kernels switched off, the
// results are WRONG).  The Python loader refuses such a library unless explicitly allowed.
extern "C" int mvip_build_is_experiment(void) {
#if defined(MVIP_EXPERIMENT_NO_FUSE_TAIL) || defined(MVIP_EXPERIMENT_CONV) || defined(MVIP_EXPERIMENT_GEMM) || \
    defined(MVIP_EXPERIMENT_NO_BARRIER) || defined(MVIP_EXPERIMENT_NO_EPILOGUE) || defined(MVIP_EXPERIMENT_HG_ONE_ATOMIC)
    return 1;
#else
    return 0;
#endif
}

extern "C" const char *mvip_strerror(int code) {
    switch (code) {
        case MVIP_OK: return "ok";
        case MVIP_EINVAL: return "invalid argument (size, null pointer or configuration)";
        case MVIP_ELAUNCH: return "HIP launch/runtime error (see mvip_last_hip_error)";
        case MVIP_EUNSUP: return "shape or mode not supported by the compiled kernels";
        default: return "unknown mvip error code";
    }
}

extern "C" const char *mvip_last_hip_error(void) { return hipGetErrorString(mvip::g_last); }

extern "C" int mvip_device_info(int *n_cu, int *lds_bytes, char *arch_out, int arch_cap) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) { mvip::set_last_error(e); return MVIP_ELAUNCH; }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)prop.sharedMemPerBlock;
    if (arch_out && arch_cap > 0) { strncpy(arch_out, prop.gcnArchName, arch_cap - 1); arch_out[arch_cap - 1] = 0; }
    return MVIP_OK;
}
