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

// ---- which kernel did the last entry point launch? (diagnostics: bench.py's roofline leg labels its HIP-event timings with the
// instantiation a dispatcher chose, so they can be set beside the rocprofv3 kernel trace name for name) ----
static thread_local char g_kernel[128] = "";
static int g_notes_on = 0;
int mmg_kernel_notes_on(void) { return g_notes_on; }
void mmg_note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}
MMG_API int mmg_set_kernel_notes(int on) { g_notes_on = on != 0; g_kernel[0] = 0; return 0; }
MMG_API const char* mmg_last_kernel(void) { return g_kernel; }

MMG_API int mmg_abi_version(void) { return 5; }
MMG_API const char* mmg_last_error(void) { return g_err; }
MMG_API const char* mmg_target_arch(void) { return "gfx950"; }

// CU count of the CURRENT device, cached per device id (persistent grids are sized by it on every launch; a process that drives
// several devices, or partitions of different sizes, gets each device's own count - ADVICE r3).  256 when the query fails.
int mmg_cu_count_cached(void) {
    static int cus[64];                       // 0 = not asked yet; racing first calls store the same value
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int c = __atomic_load_n(&cus[dev], __ATOMIC_RELAXED);
    if (!c) {
        hipDeviceProp_t pr;
        c = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
        __atomic_store_n(&cus[dev], c, __ATOMIC_RELAXED);
    }
    return c;
}

// Number of compute units of the current device (used to size persistent grids); <0 on failure.
MMG_API int mmg_device_cu_count(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return -1;
    return p.multiProcessorCount;
}
