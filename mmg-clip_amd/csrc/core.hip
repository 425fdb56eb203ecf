// ABI probe + error string for the mmg-clip gfx950 library (see include/mmgclip_hip.h).
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void mmg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

MMG_API int mmg_abi_version(void) { return 2; }
MMG_API const char* mmg_last_error(void) { return g_err; }
MMG_API const char* mmg_target_arch(void) { return "gfx950"; }

// Number of compute units of the current device (used to size persistent grids); <0 on failure.
MMG_API int mmg_device_cu_count(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return -1;
    return p.multiProcessorCount;
}
