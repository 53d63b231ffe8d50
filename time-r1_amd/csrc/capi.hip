// C-ABI plumbing: error string, version, device probe.
#include "tr1_common.h"

static thread_local char g_err[512] = {0};

extern "C" void tr1_set_error_(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* tr1_last_error(void) { return g_err; }
extern "C" int tr1_version(void) { return 1; }

// Returns 0 and fills arch[] (e.g. "gfx950") when a HIP device is usable, else the hipError_t.
extern "C" int tr1_device_info(int device, char* arch, int64_t arch_len, int64_t* n_cu, int64_t* hbm_bytes) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { tr1_set_error_(hipGetErrorString(e)); return (int)e; }
    if (arch && arch_len > 0) { strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1); arch[arch_len - 1] = 0; }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return 0;
}
