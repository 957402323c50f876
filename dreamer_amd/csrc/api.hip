// Library-level entry points: version, error text, architecture check.
#include "common.hpp"

#include <cstring>

namespace drm {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
}  // namespace drm

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
