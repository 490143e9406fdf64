// snerf_composite_fwd_f32 - alpha compositing of one ray per wavefront (a4, utils.py:134-191).
//
//   dist_i  = (z_{i+1} - z_i) * ||dir||,  dist_{N-1} = 1e10 * ||dir||          (utils.py:161-165)
//   alpha_i = 1 - exp(-relu(sigma_i + noise_i) * dist_i)                        (utils.py:159,173)
//   T_i     = prod_{j<i} (1 - alpha_j + 1e-10)      exclusive cumprod           (utils.py:174-179)
//   w_i     = alpha_i * T_i ;  rgb = sum_i w_i * sigmoid(raw_i.rgb) (+ 1 - sum w if white bg)
//
// HBM-bound: 20 B read (raw 16 + z 4) and 8 B written (weights, alpha) per sample, +12 B per ray.
// One 64-lane wavefront owns a ray; samples are walked in chunks of 64 so every load/store is a
// fully coalesced 256 B / 1 KiB wave access; the transmittance is a wavefront prefix product with
// a scalar carry between chunks.  The prefix product runs in fp64 and is rounded once per sample:
// that is what torch's CPU cumprod does (accumulate in double, store float), so the result does
// not depend on the scan order and matches the reference to the last bit of T in practice.
#include "snerf_common.h"

namespace snerf {

constexpr int CP_THREADS = 256;  // 4 rays per workgroup

__device__ __forceinline__ float sigmoidf_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(CP_THREADS) void composite_fwd_kernel(
    const float4 *__restrict__ raw, const float *__restrict__ z, const float *__restrict__ dirs, int dirs_per_sample,
    const float *__restrict__ noise, int64_t B, int N, int white_bg, float *__restrict__ rgb_out,
    float *__restrict__ w_out, float *__restrict__ alpha_out) {
    const int lane = lane_id();
    const int64_t ray = (int64_t)blockIdx.x * (CP_THREADS / WAVE) + (threadIdx.x >> 6);
    if (ray >= B) return;  // wave-uniform
    const int64_t base = ray * N;

    if (N == 1) {  // utils.py:168-169: sigmoid(rgb), weights = alpha = ones
        if (lane == 0) {
            const float4 r = raw[base];
            if (rgb_out) {
                rgb_out[ray * 3 + 0] = sigmoidf_ref(r.x);
                rgb_out[ray * 3 + 1] = sigmoidf_ref(r.y);
                rgb_out[ray * 3 + 2] = sigmoidf_ref(r.z);
            }
            if (w_out) w_out[ray] = 1.0f;
            if (alpha_out) alpha_out[ray] = 1.0f;
        }
        return;
    }

    float ray_norm = 0.f;
    if (!dirs_per_sample) {
        const float dx = dirs[ray * 3 + 0], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
        ray_norm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    }

    double carry = 1.0;  // product of (1 - alpha + 1e-10) over all previous chunks
    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_w = 0.f;
    for (int c0 = 0; c0 < N; c0 += WAVE) {
        const int i = c0 + lane;
        const bool ok = i < N;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        float zi = 0.f, zn = 0.f, nrm = ray_norm, nz = 0.f;
        if (ok) {
            r = raw[base + i];
            zi = z[base + i];
            zn = (i + 1 < N) ? z[base + i + 1] : 0.f;
            if (noise) nz = noise[base + i];
            if (dirs_per_sample) {
                const float *dp = dirs + (base + i) * 3;
                nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dp[0], dp[0]), __fmul_rn(dp[1], dp[1])), __fmul_rn(dp[2], dp[2])));
            }
        }
        float dist = (i + 1 < N) ? __fsub_rn(zn, zi) : 1e10f;
        dist = __fmul_rn(dist, nrm);
        const float sig = noise ? __fadd_rn(r.w, nz) : r.w;
        const float a = ok ? __fsub_rn(1.0f, expf(__fmul_rn(-fmaxf(sig, 0.f), dist))) : 0.f;
        const float om = ok ? __fadd_rn(__fsub_rn(1.0f, a), 1e-10f) : 1.0f;
        // inclusive fp64 prefix product over the chunk, shifted to exclusive
        const double incl = wave_scan_mul((double)om, lane);
        double excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0;
        const float T = (float)(carry * excl);
        carry *= __shfl(incl, 63, 64);
        const float w = __fmul_rn(a, T);
        if (ok) {
            if (w_out) w_out[base + i] = w;
            if (alpha_out) alpha_out[base + i] = a;
            acc_r = __fadd_rn(acc_r, __fmul_rn(w, sigmoidf_ref(r.x)));
            acc_g = __fadd_rn(acc_g, __fmul_rn(w, sigmoidf_ref(r.y)));
            acc_b = __fadd_rn(acc_b, __fmul_rn(w, sigmoidf_ref(r.z)));
            acc_w = __fadd_rn(acc_w, w);
        }
    }
    acc_r = wave_sum(acc_r);
    acc_g = wave_sum(acc_g);
    acc_b = wave_sum(acc_b);
    acc_w = wave_sum(acc_w);
    if (lane == 0 && rgb_out) {
        const float bg = white_bg ? __fsub_rn(1.0f, acc_w) : 0.f;
        rgb_out[ray * 3 + 0] = white_bg ? __fadd_rn(acc_r, bg) : acc_r;
        rgb_out[ray * 3 + 1] = white_bg ? __fadd_rn(acc_g, bg) : acc_g;
        rgb_out[ray * 3 + 2] = white_bg ? __fadd_rn(acc_b, bg) : acc_b;
    }
}

// ---- backward: d rgb [B,3] -> d raw [B,N,4] --------------------------------------------------------
// With c = sigmoid(raw.rgb), a = alpha, om = 1-a+1e-10, T = exclusive cumprod(om), w = a*T:
//   d c_i  = w_i * d rgb                      d raw.rgb_i = d c_i * c_i (1 - c_i)
//   d w_i  = <d rgb, c_i> - [white bg] sum(d rgb)
//   d a_j  = d w_j T_j - (sum_{i>j} d w_i w_i) / om_j        (T_i depends on om_j for every i > j)
//   d sigma_j = d a_j * dist_j * exp(-relu(sigma_j) dist_j) * [sigma_j + noise_j > 0]
// One wave per ray: a forward sweep records the transmittance carried into every 64-sample chunk, a
// reverse sweep recomputes each chunk and runs the suffix sum as a reverse wavefront scan (fp64).
// HBM: raw and z are read twice (40 B/sample), d raw written once (16 B/sample).
constexpr int CP_MAX_CHUNKS = 64;  // N <= 4096 (forward and backward accept the same range)

__global__ __launch_bounds__(CP_THREADS) void composite_bwd_kernel(
    const float4 *__restrict__ raw, const float *__restrict__ z, const float *__restrict__ dirs, int dirs_per_sample,
    const float *__restrict__ noise, int64_t B, int N, int white_bg, const float *__restrict__ d_rgb,
    float4 *__restrict__ d_raw, float *__restrict__ d_dirs) {
    __shared__ double s_carry[CP_THREADS / WAVE][CP_MAX_CHUNKS];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * (CP_THREADS / WAVE) + wave;
    if (ray >= B) return;
    const int64_t base = ray * N;
    const float gr = d_rgb[ray * 3 + 0], gg = d_rgb[ray * 3 + 1], gb = d_rgb[ray * 3 + 2];
    if (N == 1) {  // rgb = sigmoid(raw.rgb); weights/alpha are constants (utils.py:168-169)
        if (lane == 0) {
            const float4 r = raw[base];
            const float cr = sigmoidf_ref(r.x), cg = sigmoidf_ref(r.y), cb = sigmoidf_ref(r.z);
            d_raw[base] = make_float4(gr * cr * (1.f - cr), gg * cg * (1.f - cg), gb * cb * (1.f - cb), 0.f);
        }
        if (d_dirs && lane < 3) d_dirs[base * 3 + lane] = 0.f;
        return;
    }
    float ray_norm = 0.f;
    if (!dirs_per_sample) {
        const float dx = dirs[ray * 3 + 0], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
        ray_norm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    }
    float delta = 0.f, nrm = 0.f;  // of the sample last evaluated by this lane
    auto sample = [&](int i, float4 &r, float &a, float &om, float &dist, float &sig, float &ex) {
        r = raw[base + i];
        const float zi = z[base + i];
        nrm = ray_norm;
        if (dirs_per_sample) {
            const float *dp = dirs + (base + i) * 3;
            nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dp[0], dp[0]), __fmul_rn(dp[1], dp[1])), __fmul_rn(dp[2], dp[2])));
        }
        delta = (i + 1 < N) ? __fsub_rn(z[base + i + 1], zi) : 1e10f;
        dist = __fmul_rn(delta, nrm);
        sig = noise ? __fadd_rn(r.w, noise[base + i]) : r.w;
        ex = expf(__fmul_rn(-fmaxf(sig, 0.f), dist));
        a = __fsub_rn(1.0f, ex);
        om = __fadd_rn(__fsub_rn(1.0f, a), 1e-10f);
    };
    // forward sweep: transmittance carried into each chunk
    const int nchunk = (N + WAVE - 1) / WAVE;
    double carry = 1.0;
    for (int c = 0; c < nchunk; ++c) {
        if (lane == 0) s_carry[wave][c] = carry;
        const int i = c * WAVE + lane;
        float om = 1.0f;
        if (i < N) {
            float4 r;
            float a, dist, sig, ex;
            sample(i, r, a, om, dist, sig, ex);
        }
        const double incl = wave_scan_mul((double)om, lane);
        carry *= __shfl(incl, 63, 64);
    }
    const float gsum = white_bg ? (gr + gg + gb) : 0.f;
    // reverse sweep
    double suffix = 0.0;  // sum_{i > last sample of this chunk} d w_i * w_i
    for (int c = nchunk - 1; c >= 0; --c) {
        const int i = c * WAVE + lane;
        const bool ok = i < N;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        float a = 0.f, om = 1.0f, dist = 0.f, sig = 0.f, ex = 1.0f;
        if (ok) sample(i, r, a, om, dist, sig, ex);
        const double incl = wave_scan_mul((double)om, lane);
        double excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0;
        const float T = (float)(s_carry[wave][c] * excl);
        const float w = a * T;
        const float cr = sigmoidf_ref(r.x), cg = sigmoidf_ref(r.y), cb = sigmoidf_ref(r.z);
        const float dw = ok ? (gr * cr + gg * cg + gb * cb - gsum) : 0.f;
        const double q = (double)dw * (double)w;
        // exclusive reverse scan of q over the lanes + what came from the later chunks
        double incl_rev = q;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double t = __shfl_down(incl_rev, off, 64);
            if (lane + off < 64) incl_rev += t;
        }
        const double after = incl_rev - q + suffix;
        suffix += __shfl(incl_rev, 0, 64);
        if (ok) {
            const float da = dw * T - (float)(after / (double)om);
            const float dsig = sig > 0.f ? da * dist * ex : 0.f;  // ex = exp(-relu(sigma) dist)
            d_raw[base + i] = make_float4(w * gr * cr * (1.f - cr), w * gg * cg * (1.f - cg), w * gb * cb * (1.f - cb), dsig);
            if (d_dirs) {  // per-sample directions: dist = delta * |dir|  =>  d dir = d dist * delta * dir / |dir|
                const float ddist = sig > 0.f ? da * sig * ex : 0.f;
                const float s = nrm > 0.f ? ddist * delta / nrm : 0.f;  // torch.norm's subgradient at 0 is 0
                const float *dp = dirs + (base + i) * 3;
                float *q = d_dirs + (base + i) * 3;
                q[0] = s * dp[0];
                q[1] = s * dp[1];
                q[2] = s * dp[2];
            }
        }
    }
}

}  // namespace snerf

extern "C" int snerf_composite_fwd_f32(const float *raw, const float *z, const float *dirs, int dirs_per_sample,
                                       const float *noise, int64_t B, int N, int white_background, float *rgb,
                                       float *weights, float *alpha, snerf_stream_t stream) {
    using namespace snerf;
    if (B < 0 || N < 1 || N > WAVE * CP_MAX_CHUNKS) return fail(SNERF_E_BADARG, "composite: need B >= 0 and 1 <= N <= 4096");
    if (B == 0) return SNERF_OK;
    if (!raw || !z) return fail(SNERF_E_BADARG, "composite: raw/z is null");
    if (N > 1 && !dirs) return fail(SNERF_E_BADARG, "composite: dirs is null");
    if (!aligned(raw, 16)) return fail(SNERF_E_ALIGN, "composite: raw must be 16-byte aligned");
    const int rays_per_block = CP_THREADS / WAVE;
    const int64_t grid = (B + rays_per_block - 1) / rays_per_block;
    if (grid > 0x7fffffffLL) return fail(SNERF_E_BADARG, "composite: B too large");
    hipLaunchKernelGGL(composite_fwd_kernel, dim3((unsigned)grid), dim3(CP_THREADS), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(raw), z, dirs, dirs_per_sample ? 1 : 0, noise, B, N,
                       white_background ? 1 : 0, rgb, weights, alpha);
    return check_launch("composite_fwd");
}

extern "C" int snerf_composite_bwd_f32(const float *raw, const float *z, const float *dirs, int dirs_per_sample,
                                       const float *noise, int64_t B, int N, int white_background, const float *d_rgb,
                                       float *d_raw, float *d_dirs, snerf_stream_t stream) {
    using namespace snerf;
    if (B < 0 || N < 1 || N > WAVE * CP_MAX_CHUNKS) return fail(SNERF_E_BADARG, "composite_bwd: need 1 <= N <= 4096");
    if (B == 0) return SNERF_OK;
    if (!raw || !z || !d_rgb || !d_raw) return fail(SNERF_E_BADARG, "composite_bwd: null pointer");
    if (N > 1 && !dirs) return fail(SNERF_E_BADARG, "composite_bwd: dirs is null");
    if (d_dirs && !dirs_per_sample) return fail(SNERF_E_BADARG, "composite_bwd: d_dirs needs per-sample directions");
    if (!aligned(raw, 16) || !aligned(d_raw, 16)) return fail(SNERF_E_ALIGN, "composite_bwd: raw/d_raw must be 16-byte aligned");
    const int rays_per_block = CP_THREADS / WAVE;
    const int64_t grid = (B + rays_per_block - 1) / rays_per_block;
    if (grid > 0x7fffffffLL) return fail(SNERF_E_BADARG, "composite_bwd: B too large");
    hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)grid), dim3(CP_THREADS), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(raw), z, dirs, dirs_per_sample ? 1 : 0, noise, B, N,
                       white_background ? 1 : 0, d_rgb, reinterpret_cast<float4 *>(d_raw), d_dirs);
    return check_launch("composite_bwd");
}
