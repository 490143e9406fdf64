// Shared host/device helpers for libsmplnerf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

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
