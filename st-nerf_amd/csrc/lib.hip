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

// ------------------------------------------------------------------------------------------ LDS opt-in
#include <map>
#include <mutex>
#include <utility>
#include <vector>
namespace stnerf {
int reserve_dynamic_lds(const void* kernel, int bytes, const char* what) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, int> reserved;  // (device, kernel) -> bytes opted in so far
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        set_error("%s: no current HIP device", what);
        return STNERF_ELAUNCH;
    }
    std::lock_guard<std::mutex> lock(mu);
    int& have = reserved[std::make_pair(dev, kernel)];
    if (bytes <= have) return STNERF_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        (void)hipGetLastError();
        set_error("%s: cannot reserve %d B of LDS on device %d", what, bytes, dev);
        return STNERF_ELAUNCH;
    }
    have = bytes;
    return STNERF_OK;
}
}  // namespace stnerf

// ------------------------------------------------------------------------------------------ launch profiler
namespace stnerf {
namespace {
struct ProfRec {
    int kernel, kind, ns, tag;
    int64_t n_rays, bytes_per_ray;
    hipEvent_t e0, e1;
};
bool g_prof_on = false;
std::vector<ProfRec>* g_recs = nullptr;
thread_local int g_tag = -1;
}  // namespace

bool profiling_enabled() { return g_prof_on; }
void set_launch_tag(int tag) { g_tag = tag; }

LaunchTimer::LaunchTimer(int kernel, int kind, int64_t n_rays, int ns, int64_t bytes_per_ray, hipStream_t stream)
    : rec_(nullptr), stream_(stream) {
    if (!g_prof_on || !g_recs) return;
    ProfRec r{kernel, kind, ns, g_tag, n_rays, bytes_per_ray, nullptr, nullptr};
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    (void)hipEventRecord(r.e0, stream);
    g_recs->push_back(r);
    rec_ = reinterpret_cast<void*>(g_recs->size());  // index + 1
}

LaunchTimer::~LaunchTimer() {
    if (!rec_ || !g_recs) return;
    ProfRec& r = (*g_recs)[reinterpret_cast<size_t>(rec_) - 1];
    (void)hipEventRecord(r.e1, stream_);
}
}  // namespace stnerf

extern "C" int stnerf_profile_begin(void) {
    using namespace stnerf;
    if (!g_recs) g_recs = new std::vector<ProfRec>();
    for (auto& r : *g_recs) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_recs->clear();
    g_prof_on = true;
    return STNERF_OK;
}

extern "C" int stnerf_profile_end(stnerf_profile_record* out, int max_records, int* n_records) {
    using namespace stnerf;
    g_prof_on = false;
    if (!n_records) {
        set_error("profile_end: null pointer");
        return STNERF_EINVAL;
    }
    const int total = g_recs ? (int)g_recs->size() : 0;
    *n_records = total;
    for (int i = 0; i < total && i < max_records && out; ++i) {
        ProfRec& r = (*g_recs)[i];
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) {
            set_error("profile_end: event query failed for record %d", i);
            return STNERF_ELAUNCH;
        }
        out[i].kernel = r.kernel;
        out[i].kind = r.kind;
        out[i].ns = r.ns;
        out[i].tag = r.tag;
        out[i].n_rays = r.n_rays;
        out[i].bytes_per_ray = r.bytes_per_ray;
        out[i].ms = ms;
    }
    return STNERF_OK;
}

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
