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

}  // namespace snerf

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
