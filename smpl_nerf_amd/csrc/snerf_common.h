// Shared host/device helpers for libsmplnerf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include <atomic>

#include "../../include/smplnerf.h"

namespace snerf {

// thread-local error text behind snerf_last_error_string()
char *err_buf();
int fail(int code, const char *fmt, ...);

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SNERF_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SNERF_OK;
}

// ---- process-wide state of the library (all of it; include/smplnerf.h "State") ------------------------------------
// The CU count and hipFuncAttributeMaxDynamicSharedMemorySize are PER DEVICE: both are cached per HIP device ordinal of
// the calling thread's current device, so one process may drive several GPUs through the library.
constexpr int MAX_DEVICES = 64;
// CU count of the current device (cached per device); returns < 1 and sets the error text on failure
int device_cu_count(const char *what);
// "dynamic LDS limit of this kernel raised to N bytes on device d" - one per kernel instantiation (function-local static)
struct LdsRaised {
    std::atomic<int> bytes[MAX_DEVICES];
};
// raises the limit once per (kernel, device); idempotent, a race only repeats the driver call.  Returns 0 or SNERF_E_LAUNCH.
int raise_dynamic_lds(const void *kernel, int bytes, LdsRaised &state, const char *what);

// Tuning knobs: environment variables read ONCE, at the first call that consults them (include/smplnerf.h lists them).
// They select between equivalent kernels / launch shapes for A/B measurements and never change results.
struct Tuning {
    bool fwd_persistent;            // SNERF_FWD_PERSISTENT            (default 1)
    int fwd_waves;                  // SNERF_FWD_WAVES                 (8; 4 = two 4-wave workgroups per CU)
    bool bf16_persistent;           // SNERF_BF16_PERSISTENT           (1)
    bool warp_resident;             // SNERF_WARP_RESIDENT             (1)
    bool warp_bwd_ring;             // SNERF_WARP_BWD_RING             (0)
    bool wgrad_bf16;                // SNERF_WGRAD_BF16                (1)
    bool wgrad_f16;                 // SNERF_WGRAD_F16                 (1)
    bool wgrad_narrow_f16;          // SNERF_WGRAD_NARROW_F16          (1)
    bool wgrad_f16_split_per_wave;  // SNERF_WGRAD_F16_SPLIT_PER_WAVE  (0)
    int wgrad_fold;                 // SNERF_WGRAD_FOLD                (1)
};
const Tuning &tuning();

__host__ __device__ inline bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

constexpr int WAVE = 64;

// ---- wave-level primitives (64-wide wavefronts) ------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// inclusive prefix product / sum over the 64 lanes in fp64
__device__ __forceinline__ double wave_scan_mul(double v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        double t = __shfl_up(v, off, 64);
        if (lane >= off) v *= t;
    }
    return v;
}
__device__ __forceinline__ double wave_scan_add(double v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        double t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

}  // namespace snerf
