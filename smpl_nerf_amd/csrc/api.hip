// Version / error plumbing of the C-ABI (include/smplnerf.h).
#include <stdlib.h>

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

__global__ __launch_bounds__(256) void debug_lds_poison_kernel(int floats) {
    extern __shared__ float lds_all[];
    for (int i = threadIdx.x; i < floats; i += 256) lds_all[i] = __int_as_float(0x7fc00000 | (i & 0xffff));
    __syncthreads();
    // keep the workgroup resident for a moment so that the launch spreads over every CU (one 160 KB workgroup per CU at a time)
    if (lds_all[(threadIdx.x * 7) % floats] == 0.f) __builtin_amdgcn_s_sleep(1);
    for (int k = 0; k < 64; ++k) __builtin_amdgcn_s_sleep(127);
}
void debug_poison_lds() {
    static const int on = [] { const char *e = getenv("SNERF_DEBUG_POISON_LDS"); return e ? atoi(e) : 0; }();
    if (!on) return;
    constexpr int BYTES = 160 * 1024;
    static std::atomic<int> raised{0};
    if (!raised.load() &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(debug_lds_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BYTES) == hipSuccess)
        raised.store(1);
    hipLaunchKernelGGL(debug_lds_poison_kernel, dim3(2048), dim3(256), BYTES, 0, BYTES / 4);
    (void)hipGetLastError();
}

int device_cu_count(const char *what) {
    static std::atomic<int> cus[MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
        (void)hipGetLastError();
        return fail(SNERF_E_LAUNCH, "%s: cannot query the current device", what);
    }
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        return fail(SNERF_E_LAUNCH, "%s: cannot query the CU count of device %d", what, dev);
    }
    cus[dev].store(n, std::memory_order_relaxed);
    return n;
}

int raise_dynamic_lds(const void *kernel, int bytes, LdsRaised &state, const char *what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
        (void)hipGetLastError();
        return fail(SNERF_E_LAUNCH, "%s: cannot query the current device", what);
    }
    if (state.bytes[dev].load(std::memory_order_relaxed) >= bytes) return SNERF_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return fail(SNERF_E_LAUNCH, "%s: cannot raise the dynamic LDS limit to %d bytes on device %d", what, bytes, dev);
    }
    state.bytes[dev].store(bytes, std::memory_order_relaxed);
    return SNERF_OK;
}

const Tuning &tuning() {
    static const Tuning t = [] {   // C++11 magic static: initialised once, thread-safe
        auto flag = [](const char *name, bool dflt) {
            const char *e = getenv(name);
            return e ? atoi(e) != 0 : dflt;
        };
        Tuning k;
        k.warp_fold = flag("SNERF_WARP_FOLD", true);
        k.mlp_fold = flag("SNERF_MLP_FOLD", true);
        return k;
    }();
    return t;
}
}  // namespace snerf

extern "C" int snerf_version(void) { return SNERF_VERSION; }
extern "C" const char *snerf_last_error_string(void) { return snerf::err_buf(); }
namespace snerf {
void destroy_fork_join_events();   // train_step.hip
}
extern "C" int snerf_shutdown(void) {
    snerf::destroy_fork_join_events();
    return SNERF_OK;
}
extern "C" int snerf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
