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
        const double incl = wave_scan_mul((double)om);
        const double excl = wave_shift_up1(incl, 1.0);
        const float T = (float)(carry * excl);
        carry *= wave_last(incl);
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

// ---- backward: (d rgb [B,3], d weights [B,N], d alpha [B,N]) -> d raw [B,N,4] (, d dirs, d z) -------------------------
// With c = sigmoid(raw.rgb), a = alpha, om = 1-a+1e-10, T = exclusive cumprod(om), w = a*T:
//   d c_i  = w_i * d rgb                      d raw.rgb_i = d c_i * c_i (1 - c_i)
//   G_i    = <d rgb, c_i> - [white bg] sum(d rgb) + d weights_i                  (everything that arrives at w_i)
//   d a_j  = G_j T_j - (sum_{i>j} G_i w_i) / om_j + d alpha_j                    (T_i depends on om_j for every i > j)
//   d sigma_j = d a_j * dist_j * exp(-relu(sigma_j) dist_j) * [sigma_j + noise_j > 0]
//   d dist_j  = d a_j * relu(sigma_j) * exp(..);  dist_j = (z_{j+1} - z_j) |dir_j|  =>  d z, d dir
// One wave per ray: a forward sweep records the transmittance carried into every 64-sample chunk, a
// reverse sweep recomputes each chunk and runs the suffix sum as a reverse wavefront scan (fp64).
// HBM: raw and z are read twice (40 B/sample), d raw written once (16 B/sample).
constexpr int CP_MAX_CHUNKS = 64;  // backward: N <= 4096

__global__ __launch_bounds__(CP_THREADS) void composite_bwd_kernel(
    const float4 *__restrict__ raw, const float *__restrict__ z, const float *__restrict__ dirs, int dirs_per_sample,
    const float *__restrict__ noise, int64_t B, int N, int white_bg, const float *__restrict__ d_rgb,
    const float *__restrict__ d_w, const float *__restrict__ d_a, float4 *__restrict__ d_raw, float *__restrict__ d_dirs,
    float *__restrict__ d_z) {
    __shared__ double s_carry[CP_THREADS / WAVE][CP_MAX_CHUNKS];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * (CP_THREADS / WAVE) + wave;
    if (ray >= B) return;
    const int64_t base = ray * N;
    float gr = 0.f, gg = 0.f, gb = 0.f;
    if (d_rgb) {
        gr = d_rgb[ray * 3 + 0];
        gg = d_rgb[ray * 3 + 1];
        gb = d_rgb[ray * 3 + 2];
    }
    if (N == 1) {  // rgb = sigmoid(raw.rgb); weights/alpha are constants (utils.py:168-169)
        if (lane == 0) {
            const float4 r = raw[base];
            const float cr = sigmoidf_ref(r.x), cg = sigmoidf_ref(r.y), cb = sigmoidf_ref(r.z);
            d_raw[base] = make_float4(gr * cr * (1.f - cr), gg * cg * (1.f - cg), gb * cb * (1.f - cb), 0.f);
            if (d_z) d_z[base] = 0.f;
        }
        if (d_dirs && lane < 3) d_dirs[(dirs_per_sample ? base : ray) * 3 + lane] = 0.f;
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
        const double incl = wave_scan_mul((double)om);
        carry *= wave_last(incl);
    }
    const float gsum = white_bg ? (gr + gg + gb) : 0.f;
    // reverse sweep
    double suffix = 0.0;   // sum_{i > last sample of this chunk} G_i * w_i
    float ray_dd = 0.f;    // sum_i d dist_i * delta_i (ray-direction gradient)
    float z_pend = 0.f;    // lane 0: -e_i of the first sample of the chunk processed last, waiting for e_{i-1}
    for (int c = nchunk - 1; c >= 0; --c) {
        const int i = c * WAVE + lane;
        const bool ok = i < N;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        float a = 0.f, om = 1.0f, dist = 0.f, sig = 0.f, ex = 1.0f;
        if (ok) sample(i, r, a, om, dist, sig, ex);
        const double incl = wave_scan_mul((double)om);
        const double excl = wave_shift_up1(incl, 1.0);
        const float T = (float)(s_carry[wave][c] * excl);
        const float w = a * T;
        const float cr = sigmoidf_ref(r.x), cg = sigmoidf_ref(r.y), cb = sigmoidf_ref(r.z);
        float dw = ok ? (gr * cr + gg * cg + gb * cb - gsum) : 0.f;
        if (ok && d_w) dw += d_w[base + i];
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
        float e = 0.f;   // d loss / d delta_i = d dist_i * |dir_i|
        if (ok) {
            float da = dw * T - (float)(after / (double)om);
            if (d_a) da += d_a[base + i];
            const float dsig = sig > 0.f ? da * dist * ex : 0.f;  // ex = exp(-relu(sigma) dist)
            d_raw[base + i] = make_float4(w * gr * cr * (1.f - cr), w * gg * cg * (1.f - cg), w * gb * cb * (1.f - cb), dsig);
            const float ddist = sig > 0.f ? da * sig * ex : 0.f;
            e = (i + 1 < N) ? ddist * nrm : 0.f;   // the last interval is the constant 1e10 (utils.py:164)
            if (d_dirs) {  // dist = delta * |dir|  =>  d dir = d dist * delta * dir / |dir|
                if (dirs_per_sample) {
                    const float s = nrm > 0.f ? ddist * delta / nrm : 0.f;  // torch.norm's subgradient at 0 is 0
                    const float *dp = dirs + (base + i) * 3;
                    float *q3 = d_dirs + (base + i) * 3;
                    q3[0] = s * dp[0];
                    q3[1] = s * dp[1];
                    q3[2] = s * dp[2];
                } else {
                    ray_dd += ddist * delta;
                }
            }
        }
        if (d_z) {   // d z_i = e_{i-1} - e_i
            const float e_last = wave_last(e);
            if (lane == 0 && c + 1 < nchunk) d_z[base + (c + 1) * WAVE] = z_pend + e_last;
            const float prev = wave_shift_up1(e, 0.f);
            if (lane == 0) {
                z_pend = -e;
                if (c == 0) d_z[base] = -e;
            } else if (ok) {
                d_z[base + i] = prev - e;
            }
        }
    }
    if (d_dirs && !dirs_per_sample) {
        ray_dd = wave_sum(ray_dd);
        if (lane < 3) d_dirs[ray * 3 + lane] = ray_norm > 0.f ? ray_dd / ray_norm * dirs[ray * 3 + lane] : 0.f;
    }
}

}  // namespace snerf

extern "C" int snerf_composite_fwd_f32(const float *raw, const float *z, const float *dirs, int dirs_per_sample,
                                       const float *noise, int64_t B, int N, int white_background, float *rgb,
                                       float *weights, float *alpha, snerf_stream_t stream) {
    using namespace snerf;
    if (B < 0 || N < 1) return fail(SNERF_E_BADARG, "composite: need B >= 0 and N >= 1");
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

namespace snerf {
static int launch_composite_bwd(const float *raw, const float *z, const float *dirs, int dirs_per_sample, const float *noise,
                                int64_t B, int N, int white_background, const float *d_rgb, const float *d_weights,
                                const float *d_alpha, float *d_raw, float *d_dirs, float *d_z, snerf_stream_t stream) {
    if (B < 0 || N < 1 || N > WAVE * CP_MAX_CHUNKS) return fail(SNERF_E_BADARG, "composite_bwd: need 1 <= N <= 4096");
    if (B == 0) return SNERF_OK;
    if (!raw || !z || !d_raw) return fail(SNERF_E_BADARG, "composite_bwd: null pointer");
    if (N > 1 && !dirs) return fail(SNERF_E_BADARG, "composite_bwd: dirs is null");
    if (!aligned(raw, 16) || !aligned(d_raw, 16)) return fail(SNERF_E_ALIGN, "composite_bwd: raw/d_raw must be 16-byte aligned");
    const int rays_per_block = CP_THREADS / WAVE;
    const int64_t grid = (B + rays_per_block - 1) / rays_per_block;
    if (grid > 0x7fffffffLL) return fail(SNERF_E_BADARG, "composite_bwd: B too large");
    hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)grid), dim3(CP_THREADS), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(raw), z, dirs, dirs_per_sample ? 1 : 0, noise, B, N,
                       white_background ? 1 : 0, d_rgb, d_weights, d_alpha, reinterpret_cast<float4 *>(d_raw), d_dirs, d_z);
    return check_launch("composite_bwd");
}
}  // namespace snerf

extern "C" int snerf_composite_bwd_f32(const float *raw, const float *z, const float *dirs, int dirs_per_sample,
                                       const float *noise, int64_t B, int N, int white_background, const float *d_rgb,
                                       float *d_raw, float *d_dirs, snerf_stream_t stream) {
    using namespace snerf;
    if (!d_rgb) return fail(SNERF_E_BADARG, "composite_bwd: d_rgb is null");
    if (d_dirs && !dirs_per_sample) return fail(SNERF_E_BADARG, "composite_bwd: d_dirs needs per-sample directions");
    return launch_composite_bwd(raw, z, dirs, dirs_per_sample, noise, B, N, white_background, d_rgb, nullptr, nullptr, d_raw,
                                d_dirs, nullptr, stream);
}

extern "C" int snerf_composite_bwd_all_f32(const float *raw, const float *z, const float *dirs, int dirs_per_sample,
                                           const float *noise, int64_t B, int N, int white_background, const float *d_rgb,
                                           const float *d_weights, const float *d_alpha, float *d_raw, float *d_dirs,
                                           float *d_z, snerf_stream_t stream) {
    return snerf::launch_composite_bwd(raw, z, dirs, dirs_per_sample, noise, B, N, white_background, d_rgb, d_weights, d_alpha,
                                       d_raw, d_dirs, d_z, stream);
}
