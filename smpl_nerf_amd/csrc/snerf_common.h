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

// SNERF_DEBUG_POISON_LDS=1 (debug aid, tests/test_gpu_round4.py): after every checked launch a kernel on the NULL stream
// fills the LDS of every CU with NaNs, so that a kernel that reads LDS it has not written meets NaNs instead of whatever
// the previous kernel left there (the first process on a freshly booted GPU meets arbitrary bits).
void debug_poison_lds();
inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SNERF_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    debug_poison_lds();
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

// Tuning knobs: environment variables read ONCE, at the first call that consults them (INTEGRATION.md lists every knob of the
// library and of the Python host).  r06: the A/B switches of rounds 2-5 whose experiments are decided are gone (their patches
// and scripts under tools/ab/ are the record); what is left switches the per-ray folds of inference off.
struct Tuning {
    bool warp_fold;   // SNERF_WARP_FOLD=0   warp inference: pose k-blocks per sample instead of the per-ray fold
    bool mlp_fold;    // SNERF_MLP_FOLD=0    fp32 inference: per-ray additional inputs as k-blocks per sample
};
const Tuning &tuning();

__host__ __device__ inline bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

constexpr int WAVE = 64;

// ---- wave-level primitives (64-wide wavefronts) ------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Wavefront scans and sums on the DPP network (row_shr within rows of 16 lanes, row_bcast:15 / :31 across rows - the GFX9
// sequence): one VALU move per 32 bits and step instead of a ds_bpermute with its index arithmetic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float identity, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v), CTRL,
                                                                 ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double identity, double v) {
    const long long o = __builtin_bit_cast(long long, identity), x = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)x, CTRL, ROW_MASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(x >> 32), CTRL, ROW_MASK, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// inclusive prefix sum / product over the 64 lanes (lanes without a source take the identity)
template <class T>
__device__ __forceinline__ T wave_scan_add(T v, int /*lane*/ = 0) {
    v += dpp_take<0x111, 0xf>(T(0), v);   // row_shr:1
    v += dpp_take<0x112, 0xf>(T(0), v);   // row_shr:2
    v += dpp_take<0x114, 0xf>(T(0), v);   // row_shr:4
    v += dpp_take<0x118, 0xf>(T(0), v);   // row_shr:8
    v += dpp_take<0x142, 0xa>(T(0), v);   // row_bcast:15 -> rows 1, 3
    v += dpp_take<0x143, 0xc>(T(0), v);   // row_bcast:31 -> rows 2, 3
    return v;
}
template <class T>
__device__ __forceinline__ T wave_scan_mul(T v, int /*lane*/ = 0) {
    v *= dpp_take<0x111, 0xf>(T(1), v);
    v *= dpp_take<0x112, 0xf>(T(1), v);
    v *= dpp_take<0x114, 0xf>(T(1), v);
    v *= dpp_take<0x118, 0xf>(T(1), v);
    v *= dpp_take<0x142, 0xa>(T(1), v);
    v *= dpp_take<0x143, 0xc>(T(1), v);
    return v;
}
__device__ __forceinline__ float wave_last(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63)); }
__device__ __forceinline__ double wave_last(double v) {
    const long long x = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), 63);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// lane i <- lane i-1 (wave_shr:1); lane 0 takes `first`
template <class T>
__device__ __forceinline__ T wave_shift_up1(T v, T first) { return dpp_take<0x138, 0xf>(first, v); }
// the sum over the 64 lanes, in every lane (a scan whose last lane is broadcast through an SGPR)
__device__ __forceinline__ float wave_sum(float v) { return wave_last(wave_scan_add(v)); }
__device__ __forceinline__ double wave_sum(double v) { return wave_last(wave_scan_add(v)); }

}  // namespace snerf
