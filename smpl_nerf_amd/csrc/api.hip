// Version / error plumbing of the C-ABI (include/smplnerf.h).
#include "snerf_common.h"

namespace snerf {
char *err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}
int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace snerf

extern "C" int snerf_version(void) { return SNERF_VERSION; }
extern "C" const char *snerf_last_error_string(void) { return snerf::err_buf(); }
extern "C" int snerf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
