// snerf_raygen_f64 - on-device ray generation + stratified coarse sampling.
//
// Replaces, per ray, what the reference does in Python inside Dataset.__getitem__ (82.7 us/ray, 1.2e4 rays/s):
//   get_rays            utils.py:50-54     dirs = [(i - W/2)/f, -(j - H/2)/f, -1]; d = dirs @ R^T; o = t   (fp64)
//   CoarseSampling      datasets/transforms.py:80-89   z = lower + (upper - lower) * rand(); x = o + d z  (fp64)
//   ToTensor            datasets/transforms.py:13-21   cast to fp32
// All arithmetic is fp64 with the reference's operation order (products and sums rounded separately, no FMA),
// then rounded to fp32 once - bit-identical to the numpy path.  The per-launch tables lower[k], span[k] =
// upper[k] - lower[k] (functions of near, far, Nc only) are computed by the host with the reference's own numpy
// expression and passed in.  One wavefront per ray, lane = sample; HBM: 16 B in (index + jitter), 16*Nc + 24 B out.
#include "snerf_common.h"

namespace snerf {

constexpr int RG_THREADS = 256;

__global__ __launch_bounds__(RG_THREADS) void raygen_kernel(const double *__restrict__ poses, int64_t P, int H, int W,
                                                           double focal, const double *__restrict__ lower,
                                                           const double *__restrict__ span, int Nc,
                                                           const int64_t *__restrict__ ray_index,
                                                           const double *__restrict__ jitter, int64_t B,
                                                           float *__restrict__ samples, float *__restrict__ o_out,
                                                           float *__restrict__ d_out, float *__restrict__ z_out) {
    const int lane = lane_id();
    const int64_t ray = (int64_t)blockIdx.x * (RG_THREADS / WAVE) + (threadIdx.x >> 6);
    if (ray >= B) return;
    const int64_t idx = ray_index[ray];
    const int64_t hw = (int64_t)H * W;
    if (idx < 0 || idx >= P * hw) {  // out-of-range ray index: never read poses out of bounds; the ray comes back as NaN
        const float qnan = __builtin_nanf("");
        if (lane < 3) {
            o_out[ray * 3 + lane] = qnan;
            d_out[ray * 3 + lane] = qnan;
        }
        for (int k = lane; k < Nc; k += WAVE) {
            z_out[ray * Nc + k] = qnan;
            float *p = samples + (ray * Nc + k) * 3;
            p[0] = p[1] = p[2] = qnan;
        }
        return;
    }
    const int64_t frame = idx / hw;
    const int pix = (int)(idx - frame * hw);
    const int j = pix / W, i = pix - j * W;
    const double *M = poses + frame * 16;  // row-major 4x4
    // utils.py:50-51: i, j are float32 grids; (i - W*.5) stays float32 (python scalar), "/ focal" promotes to fp64
    const double cx = __ddiv_rn((double)__fsub_rn((float)i, (float)(W * 0.5)), focal);
    const double cy = __ddiv_rn((double)(-__fsub_rn((float)j, (float)(H * 0.5))), focal);
    const double cz = -1.0;
    double d[3], o[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {  // utils.py:52: sum over the last axis of dirs[c] * R[r][c], left to right
        d[r] = __dadd_rn(__dadd_rn(__dmul_rn(cx, M[r * 4 + 0]), __dmul_rn(cy, M[r * 4 + 1])), __dmul_rn(cz, M[r * 4 + 2]));
        o[r] = M[r * 4 + 3];       // utils.py:53
    }
    if (lane < 3) {
        o_out[ray * 3 + lane] = (float)o[lane];
        d_out[ray * 3 + lane] = (float)d[lane];
    }
    const double jit = jitter[ray];
    for (int k = lane; k < Nc; k += WAVE) {
        const double z = __dadd_rn(lower[k], __dmul_rn(span[k], jit));                  // transforms.py:87
        z_out[ray * Nc + k] = (float)z;
        float *p = samples + (ray * Nc + k) * 3;
#pragma unroll
        for (int r = 0; r < 3; ++r) p[r] = (float)__dadd_rn(o[r], __dmul_rn(d[r], z));  // transforms.py:88
    }
}

}  // namespace snerf

extern "C" int snerf_raygen_f64(const double *poses, int64_t n_frames, int H, int W, double focal, const double *lower,
                                const double *span, int Nc, const int64_t *ray_index, const double *jitter, int64_t B,
                                float *samples, float *o, float *d, float *z, snerf_stream_t stream) {
    using namespace snerf;
    if (B < 0 || n_frames < 1 || H < 1 || W < 1 || Nc < 1) return fail(SNERF_E_BADARG, "raygen: bad sizes");
    if (!(focal > 0.0)) return fail(SNERF_E_BADARG, "raygen: focal must be positive");
    if (B == 0) return SNERF_OK;
    if (!poses || !lower || !span || !ray_index || !jitter || !samples || !o || !d || !z)
        return fail(SNERF_E_BADARG, "raygen: null pointer");
    const int per = RG_THREADS / WAVE;
    const int64_t grid = (B + per - 1) / per;
    if (grid > 0x7fffffffLL) return fail(SNERF_E_BADARG, "raygen: B too large");
    hipLaunchKernelGGL(raygen_kernel, dim3((unsigned)grid), dim3(RG_THREADS), 0, (hipStream_t)stream, poses, n_frames, H, W,
                       focal, lower, span, Nc, ray_index, jitter, B, samples, o, d, z);
    return check_launch("raygen");
}
