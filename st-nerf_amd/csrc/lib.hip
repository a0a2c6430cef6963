// Library-level entry points: version, last error, device info.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"

namespace stnerf {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace stnerf

extern "C" const char* stnerf_version(void) { return "stnerf-hip 0.1.0 (gfx950)"; }

extern "C" const char* stnerf_last_error(void) { return stnerf::g_err; }

extern "C" int stnerf_device_info(int* cu_count, int* lds_bytes_per_cu, int* clock_khz, char* arch, int arch_len) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        stnerf::set_error("device_info: no HIP device");
        return STNERF_ELAUNCH;
    }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) {
        stnerf::set_error("device_info: hipGetDeviceProperties failed");
        return STNERF_ELAUNCH;
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
    if (clock_khz) *clock_khz = p.clockRate;
    if (arch && arch_len > 0) {
        strncpy(arch, p.gcnArchName, (size_t)arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        stnerf::set_error("device_info: device is %s, this library is built for gfx950 only", p.gcnArchName);
        return STNERF_EARCH;
    }
    return STNERF_OK;
}
