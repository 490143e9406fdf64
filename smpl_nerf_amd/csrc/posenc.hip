// snerf_posenc_f32 - stand-alone positional encoding (a1, utils.py:114-131).
//
// out[p, :] = [x_p] (if identity) ++ for k in 0..L-1: [sin(2^k x_p) (c values), cos(2^k x_p) (c values)].
// The product pipeline never materialises this tensor (the encoding is fused into the MLP kernel's
// first-layer operand, mlp.hip); this entry exists for the PositionalEncoder.encode() drop-in.
// HBM-bound on the store side: 4*c*(id+2L) B written per 4*c B read.  One workgroup encodes a tile
// of PE_TILE points into LDS (one sincosf per (point, channel, frequency)) and streams the tile
// out with 16-byte stores, so the store pattern is fully coalesced regardless of c and L.
#include "snerf_common.h"

namespace snerf {

constexpr int PE_THREADS = 256;
constexpr int PE_TILE = 64;

__global__ __launch_bounds__(PE_THREADS) void posenc_kernel(const float *__restrict__ x, int64_t n, int c, int L,
                                                           int identity, float *__restrict__ out, int tile) {
    extern __shared__ __attribute__((aligned(16))) float s_out[];  // [tile][outdim]
    const int outdim = c * (identity + 2 * L);
    const int64_t p0 = (int64_t)blockIdx.x * tile;
    const int np = (int)min((int64_t)tile, n - p0);
    const int per_point = c * (L + identity);  // work items: (channel, k) pairs, k == L means identity
    for (int i = threadIdx.x; i < np * per_point; i += PE_THREADS) {
        const int p = i / per_point, r = i - p * per_point;
        const int k = r / c, ch = r - k * c;
        const float xv = x[(p0 + p) * c + ch];
        float *row = s_out + p * outdim;
        if (k == L) {
            row[ch] = xv;  // only reached when identity
        } else {
            // x * freq with freq = 2^k is exact in fp32, so the argument is bit-identical to the
            // reference's `x * freq` (utils.py:127); full-range sincosf (arguments reach ~2e3 rad).
            float sv, cv;
            sincosf(ldexpf(xv, k), &sv, &cv);
            float *blk = row + identity * c + k * 2 * c;
            blk[ch] = sv;
            blk[c + ch] = cv;
        }
    }
    __syncthreads();
    const int tile_floats = np * outdim;
    float *dst = out + p0 * outdim;
    if ((tile_floats & 3) == 0 && aligned(dst, 16)) {
        const float4 *s4 = reinterpret_cast<const float4 *>(s_out);
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        for (int i = threadIdx.x; i < tile_floats / 4; i += PE_THREADS) d4[i] = s4[i];
    } else {
        for (int i = threadIdx.x; i < tile_floats; i += PE_THREADS) dst[i] = s_out[i];
    }
}

// backward of the encoding (autograd through utils.py:123-131): one thread per input value,
//   d x = [identity] g_id + sum_k 2^k (g_sin_k cos(2^k x) - g_cos_k sin(2^k x))
__global__ __launch_bounds__(PE_THREADS) void posenc_bwd_kernel(const float *__restrict__ x, const float *__restrict__ d_out,
                                                               int64_t total, int c, int L, int identity,
                                                               float *__restrict__ d_x) {
    const int64_t i = (int64_t)blockIdx.x * PE_THREADS + threadIdx.x;
    if (i >= total) return;
    const int64_t p = i / c;
    const int ch = (int)(i - p * c);
    const int outdim = c * (identity + 2 * L);
    const float *g = d_out + p * outdim;
    const float xv = x[i];
    float acc = 0.f;
    for (int k = L - 1; k >= 0; --k) {   // autograd visits the last-used frequency first
        float sv, cv;
        sincosf(ldexpf(xv, k), &sv, &cv);
        const float *blk = g + identity * c + k * 2 * c;
        const float f = ldexpf(1.0f, k);
        acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(-blk[c + ch], sv), f));   // cos branch: grad * -sin(x f) * f
        acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(blk[ch], cv), f));       // sin branch: grad * cos(x f) * f
    }
    if (identity) acc = __fadd_rn(acc, g[ch]);
    d_x[i] = acc;
}

}  // namespace snerf

extern "C" int snerf_posenc_bwd_f32(const float *x, const float *d_out, int64_t n, int c, int L, int identity, float *d_x,
                                    snerf_stream_t stream) {
    using namespace snerf;
    if (n < 0 || c <= 0 || L < 0 || L > 30) return fail(SNERF_E_BADARG, "posenc_bwd: bad n/c/L");
    identity = identity ? 1 : 0;
    if (n == 0) return SNERF_OK;
    if (!x || !d_x) return fail(SNERF_E_BADARG, "posenc_bwd: null pointer");
    if (!d_out && (identity + 2 * L) > 0) return fail(SNERF_E_BADARG, "posenc_bwd: d_out is null");
    const int64_t total = n * c;
    const int64_t grid = (total + PE_THREADS - 1) / PE_THREADS;
    if (grid > 0x7fffffffLL) return fail(SNERF_E_BADARG, "posenc_bwd: n too large");
    hipLaunchKernelGGL(posenc_bwd_kernel, dim3((unsigned)grid), dim3(PE_THREADS), 0, (hipStream_t)stream, x, d_out, total, c,
                       L, identity, d_x);
    return check_launch("posenc_bwd");
}

extern "C" int snerf_posenc_f32(const float *x, int64_t n, int c, int L, int identity, float *out,
                                snerf_stream_t stream) {
    using namespace snerf;
    if (n < 0 || c <= 0 || L < 0 || L > 30) return fail(SNERF_E_BADARG, "posenc: bad n/c/L");
    identity = identity ? 1 : 0;
    const int outdim = c * (identity + 2 * L);
    if (n == 0 || outdim == 0) return SNERF_OK;
    if (!x || !out) return fail(SNERF_E_BADARG, "posenc: null pointer");
    // points per workgroup: as many as fit 64 KiB of LDS, at most PE_TILE (wide rows, e.g. a 69-channel
    // pose with L = 10 -> 1380 floats per point, get fewer points per group)
    if (outdim > 16384) return fail(SNERF_E_BADARG, "posenc: c*(identity+2L) too large (%d)", outdim);
    int tile = 16384 / outdim;
    if (tile > PE_TILE) tile = PE_TILE;
    const size_t lds = (size_t)tile * outdim * sizeof(float);
    const int64_t grid = (n + tile - 1) / tile;
    if (grid > 0x7fffffffLL) return fail(SNERF_E_BADARG, "posenc: n too large");
    hipLaunchKernelGGL(posenc_kernel, dim3((unsigned)grid), dim3(PE_THREADS), lds, (hipStream_t)stream, x, n, c, L,
                       identity, out, tile);
    return check_launch("posenc");
}
