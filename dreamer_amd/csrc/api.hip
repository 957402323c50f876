// Library-level entry points: version, error text, architecture check.
#include "common.hpp"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace drm {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }

int sync_check_mode() {
    static const int mode = [] { const char* e = std::getenv("DRM_SYNC_CHECK"); return e != nullptr ? atoi(e) : 0; }();
    return mode;
}

int launch_check(const char* func, const char* file, int line) {
    const int mode = sync_check_mode();
    const bool sync = mode != 0;
    hipError_t e = hipGetLastError();
    // mode 2: the launch site goes to stderr before the wait, so that a memory fault (which ends the process inside the
    // wait) leaves the name of its kernel's launch as the last line
    if (mode >= 2) { fprintf(stderr, "[drm] launched %s at %s:%d\n", func, file, line); fflush(stderr); }
    if (e == hipSuccess && sync) e = hipDeviceSynchronize();
    if (e == hipSuccess) return DRM_OK;
    set_error(std::string(func) + ": launch at " + file + ":" + std::to_string(line) + " -> " + hipGetErrorString(e));
    if (sync) fprintf(stderr, "[drm] %s: launch at %s:%d failed: %s\n", func, file, line, hipGetErrorString(e));
    return DRM_EHIP;
}
}  // namespace drm

namespace drm {
// One event per launch, recorded right behind it; a launch's time runs from the previous launch's event
// (same back-to-back run) to its own, i.e. what a kernel trace shows for kernels queued back to back.
struct ProfRecord { int cls; hipEvent_t a, b; double flops, bytes; };
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static bool g_prof_fresh = true;
static std::vector<ProfRecord> g_prof;
bool prof_on() { return g_prof_on; }
void prof_fresh() { std::lock_guard<std::mutex> lk(g_prof_mu); g_prof_fresh = true; }
void prof_begin(int cls, hipStream_t s, double flops, double bytes) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRecord r{cls, nullptr, nullptr, flops, bytes};
    if (hipEventCreate(&r.b) != hipSuccess) return;
    if (g_prof_fresh || g_prof.empty()) {
        if (hipEventCreate(&r.a) != hipSuccess) return;
        (void)hipEventRecord(r.a, s);
        g_prof_fresh = false;
    }
    g_prof.push_back(r);
}
void prof_end(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof.empty()) (void)hipEventRecord(g_prof.back().b, s);
}
static void prof_clear() {
    for (auto& r : g_prof) { if (r.a) (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    g_prof_fresh = true;
}
}  // namespace drm

extern "C" int drm_prof_enable(int enable) {
    std::lock_guard<std::mutex> lk(drm::g_prof_mu);
    drm::prof_clear();
    drm::g_prof_on = enable != 0;
    return DRM_OK;
}

extern "C" const char* drm_prof_class_name(int cls) {
    static const char* names[DRM_PROF_CLASSES] = {"linear", "gru", "linear+sample",
                                                  "actor_head", "k_scalar_heads", "k_critic_value", "k_lambda_return",
                                                  "k_seq_gather", "sample", "pack", "k_linear_big", "elementwise"};
    return (cls >= 0 && cls < DRM_PROF_CLASSES) ? names[cls] : "";
}

extern "C" int drm_prof_summary(int cls, double* total_ms, int64_t* launches, double* flops, double* bytes) {
    DRM_REQUIRE(total_ms && launches && flops && bytes, DRM_EINVAL, "null pointer");
    std::lock_guard<std::mutex> lk(drm::g_prof_mu);
    *total_ms = 0; *launches = 0; *flops = 0; *bytes = 0;
    for (size_t i = 0; i < drm::g_prof.size(); ++i) {
        auto& r = drm::g_prof[i];
        if (r.cls != cls) continue;
        DRM_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        DRM_HIP(hipEventElapsedTime(&ms, r.a ? r.a : drm::g_prof[i - 1].b, r.b));
        *total_ms += ms; *launches += 1; *flops += r.flops; *bytes += r.bytes;
    }
    return DRM_OK;
}

extern "C" int drm_version(void) { return 100; }

extern "C" const char* drm_last_error(void) { return drm::g_last_error.c_str(); }

extern "C" int drm_arch_ok(void) {
    int dev = 0;
    DRM_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    DRM_HIP(hipGetDeviceProperties(&prop, dev));
    DRM_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, DRM_EARCH, "device %d is %s, this library is built for gfx950",
                dev, prop.gcnArchName);
    return DRM_OK;
}

