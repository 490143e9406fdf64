// snerf_sample_pdf_f32 - inverse-CDF hierarchical sampling, merge and point generation (a5).
//
// One 64-lane wavefront per ray does what the reference spreads over ~20 ATen ops, a full
// torch.sort and the torchsearchsorted extension (utils.py:194-264):
//   bins = .5*(z[1:]+z[:-1])                                   utils.py:258
//   w'   = weights[1:-1] + 1e-5 ; pdf = w'/sum(w') ; cdf = [0, cumsum(pdf)]      :200-203
//   inds = searchsorted(cdf, u, 'right'); below/above clamp; denom<1e-5 -> 1; lerp   :212-226
//   z_fine = sort(cat(z, samples)) ; pts = o + d*z_fine                              :261-263
// All per-ray state (z, bins, cdf, samples, merged z) lives in LDS (2 KiB per ray at 64+128), the
// binary searches run against LDS, and HBM sees only the algorithmic traffic: 2*Nc*4 + 24 B in,
// (Nc+Nf)*16 B out (+Nf*8 B if indices are requested, +Nf*4 B for z_samples).
//
// Numerics contract (matches oracle/nerf_oracle.py bit for bit): the normalising sum and the cumsum
// are evaluated in fp64 and rounded to fp32 once per element - identical to torch's CPU cumsum and
// independent of the reduction order, which is what makes a wavefront scan legal here; every other
// operation is a single correctly-rounded fp32 op in the reference's order (no FMA contraction).
// The *_strict entry points take the normalising sums as an input (`tot` [B], what torch.sum returned on the
// reference's host): the only step of the reference whose bits depend on the host.
//
// The merge exploits that both lists are ascending (rank = own index + cross-rank by bisection);
// a wave-uniform check detects an out-of-order input and falls back to an O(n^2) stable rank sort,
// so the output is always exactly sorted(cat(z, samples)).
#include "snerf_common.h"

namespace snerf {

constexpr int SP_THREADS = 256;
constexpr int SP_WAVES = SP_THREADS / WAVE;

__device__ __forceinline__ int count_le(const float *__restrict__ row, int n, float v) {  // #{k: row[k] <= v}
    int lo = 0, len = n;
    while (len > 0) {
        int half = len >> 1;
        bool go = row[lo + half] <= v;
        lo = go ? lo + half + 1 : lo;
        len = go ? len - half - 1 : half;
    }
    return lo;
}
__device__ __forceinline__ int count_lt(const float *__restrict__ row, int n, float v) {  // #{k: row[k] < v}
    int lo = 0, len = n;
    while (len > 0) {
        int half = len >> 1;
        bool go = row[lo + half] < v;
        lo = go ? lo + half + 1 : lo;
        len = go ? len - half - 1 : half;
    }
    return lo;
}

// DIRECT = the literal sample_pdf(bins, weights) calling convention (utils.py:194): `z` holds the
// bins [B, Nc-1] and `weights` the interior weights [B, Nc-2]; no merge, no points.
template <bool DIRECT>
__global__ __launch_bounds__(SP_THREADS) void sample_pdf_kernel(
    const float *__restrict__ z, const float *__restrict__ weights, const float *__restrict__ u,
    const float *__restrict__ o, const float *__restrict__ d, const float *__restrict__ tot_in, int64_t B, int Nc, int Nf,
    int64_t *__restrict__ inds_out, float *__restrict__ zs_out, float *__restrict__ zf_out,
    float *__restrict__ pts_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * SP_WAVES + wave;
    if (ray >= B) return;  // wave-uniform; no workgroup barriers below
    const int Nb = Nc - 1;           // number of bins == len(cdf)
    const int M = Nc - 2;            // number of interior weights
    const int Nt = Nc + Nf;
    const int per_wave = Nc + 2 * Nb + Nf + Nt;
    float *s_z = smem + wave * per_wave;  // [Nc]
    float *s_bins = s_z + Nc;             // [Nb]
    float *s_cdf = s_bins + Nb;           // [Nb]
    float *s_zs = s_cdf + Nb;             // [Nf]
    float *s_out = s_zs + Nf;             // [Nt]

    const float *zr = z + ray * (DIRECT ? Nb : Nc);
    const float *wr = DIRECT ? weights + ray * M - 1 : weights + ray * Nc;  // wr[k+1] = k-th interior weight

    // ---- load z, build bins, sum of (w + 1e-5) in fp64 -----------------------------------------
    double part = 0.0;
    bool sorted_in = true;
    if (DIRECT) {
        for (int i = lane; i < Nb; i += WAVE) s_bins[i] = zr[i];
        for (int i = lane + 1; i <= M; i += WAVE) part += (double)__fadd_rn(wr[i], 1e-5f);
    } else {
        for (int i = lane; i < Nc; i += WAVE) {
            const float zi = zr[i];
            s_z[i] = zi;
            if (i + 1 < Nc) {
                const float zn = zr[i + 1];
                s_bins[i] = __fmul_rn(0.5f, __fadd_rn(zn, zi));
                sorted_in = sorted_in && (zi <= zn);
            }
            if (i >= 1 && i <= M) part += (double)__fadd_rn(wr[i], 1e-5f);
        }
    }
    // strict mode: the caller supplies the normalising sum torch.sum produced on the reference's host (utils.py:201; a
    // vectorised fp32 cascade whose bits depend on that host's SIMD width) - every later step is order-independent, so
    // cdf, indices and samples then equal the reference's bit for bit
    const float tot = tot_in ? tot_in[ray] : (float)wave_sum(part);

    // ---- cdf = [0, cumsum(pdf)] : fp64 wavefront scan with a carry between 64-element chunks ----
    if (lane == 0) s_cdf[0] = 0.f;
    double carry = 0.0;
    for (int c0 = 0; c0 < M; c0 += WAVE) {
        const int k = c0 + lane;  // interior weight index 0..M-1  <->  weights[k+1]
        double p = 0.0;
        if (k < M) p = (double)__fdiv_rn(__fadd_rn(wr[k + 1], 1e-5f), tot);
        const double incl = wave_scan_add(p, lane) + carry;
        if (k < M) s_cdf[k + 1] = (float)incl;
        carry = __shfl(incl, 63, 64);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    // ---- invert the cdf for every u ----------------------------------------------------------------
    for (int f = lane; f < Nf; f += WAVE) {
        const float uf = u[f];
        const int ind = count_le(s_cdf, Nb, uf);                    // searchsorted(cdf, u, 'right')
        const int below = max(0, ind - 1);
        const int above = min(Nb - 1, ind);
        const float c0v = s_cdf[below], c1v = s_cdf[above];
        const float b0 = s_bins[below], b1 = s_bins[above];
        float denom = __fsub_rn(c1v, c0v);
        denom = denom < 1e-5f ? 1.0f : denom;
        const float t = __fdiv_rn(__fsub_rn(uf, c0v), denom);
        const float smp = __fadd_rn(b0, __fmul_rn(t, __fsub_rn(b1, b0)));
        s_zs[f] = smp;
        if (inds_out) inds_out[ray * Nf + f] = ind;
        if (zs_out) zs_out[ray * Nf + f] = smp;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (DIRECT || (!zf_out && !pts_out)) return;

    // ---- merge (stable: coarse samples first among equals) -------------------------------------------
    bool sorted_s = true;
    for (int f = lane; f + 1 < Nf; f += WAVE) sorted_s = sorted_s && (s_zs[f] <= s_zs[f + 1]);
    const bool fast = __all(sorted_in && sorted_s);
    if (fast) {
        for (int i = lane; i < Nc; i += WAVE) {
            const float v = s_z[i];
            s_out[i + count_lt(s_zs, Nf, v)] = v;
        }
        for (int f = lane; f < Nf; f += WAVE) {
            const float v = s_zs[f];
            s_out[f + count_le(s_z, Nc, v)] = v;
        }
    } else {
        // exact stable rank sort of cat(z, samples); NaNs (never produced by the pipeline) sort last
        for (int e = lane; e < Nt; e += WAVE) {
            const float v = e < Nc ? s_z[e] : s_zs[e - Nc];
            int rank = 0;
            for (int q = 0; q < Nt; ++q) {
                const float w = q < Nc ? s_z[q] : s_zs[q - Nc];
                const bool before = (w < v) || (w == v && q < e) || (v != v && (w == w || q < e));
                rank += before ? 1 : 0;
            }
            s_out[rank] = v;
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    // ---- stores: z_fine and pts = o + d * z (mul then add, as the reference's eager ops) ------------------
    if (zf_out)
        for (int e = lane; e < Nt; e += WAVE) zf_out[ray * Nt + e] = s_out[e];
    if (pts_out) {
        const float ox = o[ray * 3 + 0], oy = o[ray * 3 + 1], oz = o[ray * 3 + 2];
        const float dx = d[ray * 3 + 0], dy = d[ray * 3 + 1], dz = d[ray * 3 + 2];
        float *pr = pts_out + ray * Nt * 3;
        for (int e = lane; e < 3 * Nt; e += WAVE) {  // coalesced over the flattened [Nt,3] row
            const int s = e / 3, ch = e - 3 * s;
            const float zv = s_out[s];
            const float ov = ch == 0 ? ox : (ch == 1 ? oy : oz);
            const float dv = ch == 0 ? dx : (ch == 1 ? dy : dz);
            pr[e] = __fadd_rn(ov, __fmul_rn(dv, zv));
        }
    }
}

}  // namespace snerf

namespace snerf {
static int launch_sample_pdf(bool direct, const float *z, const float *weights, const float *u, const float *o,
                             const float *d, const float *tot, int64_t B, int Nc, int Nf, int64_t *inds, float *z_samples,
                             float *z_fine, float *pts, snerf_stream_t stream);
}

extern "C" int snerf_sample_pdf_f32(const float *z, const float *weights, const float *u, const float *o,
                                    const float *d, int64_t B, int Nc, int Nf, int64_t *inds, float *z_samples,
                                    float *z_fine, float *pts, snerf_stream_t stream) {
    return snerf::launch_sample_pdf(false, z, weights, u, o, d, nullptr, B, Nc, Nf, inds, z_samples, z_fine, pts, stream);
}

extern "C" int snerf_sample_pdf_strict_f32(const float *z, const float *weights, const float *u, const float *o,
                                           const float *d, const float *tot, int64_t B, int Nc, int Nf, int64_t *inds,
                                           float *z_samples, float *z_fine, float *pts, snerf_stream_t stream) {
    if (!tot) return snerf::fail(SNERF_E_BADARG, "sample_pdf_strict: tot is null");
    return snerf::launch_sample_pdf(false, z, weights, u, o, d, tot, B, Nc, Nf, inds, z_samples, z_fine, pts, stream);
}

extern "C" int snerf_sample_pdf_bins_strict_f32(const float *bins, const float *weights, const float *u, const float *tot,
                                                int64_t B, int Nb, int Nf, int64_t *inds, float *z_samples,
                                                snerf_stream_t stream) {
    if (Nb < 2) return snerf::fail(SNERF_E_BADARG, "sample_pdf_bins_strict: need Nb >= 2");
    if (!tot) return snerf::fail(SNERF_E_BADARG, "sample_pdf_bins_strict: tot is null");
    return snerf::launch_sample_pdf(true, bins, weights, u, nullptr, nullptr, tot, B, Nb + 1, Nf, inds, z_samples, nullptr,
                                    nullptr, stream);
}

extern "C" int snerf_sample_pdf_bins_f32(const float *bins, const float *weights, const float *u, int64_t B, int Nb,
                                         int Nf, int64_t *inds, float *z_samples, snerf_stream_t stream) {
    if (Nb < 2) return snerf::fail(SNERF_E_BADARG, "sample_pdf_bins: need Nb >= 2");
    return snerf::launch_sample_pdf(true, bins, weights, u, nullptr, nullptr, nullptr, B, Nb + 1, Nf, inds, z_samples,
                                    nullptr, nullptr, stream);
}

static int snerf::launch_sample_pdf(bool direct, const float *z, const float *weights, const float *u, const float *o,
                                    const float *d, const float *tot, int64_t B, int Nc, int Nf, int64_t *inds,
                                    float *z_samples, float *z_fine, float *pts, snerf_stream_t stream) {
    if (B < 0) return fail(SNERF_E_BADARG, "sample_pdf: negative B");
    if (Nc < 3 || Nc > 1024 || Nf < 1 || Nf > 1024)
        return fail(SNERF_E_BADARG, "sample_pdf: need 3 <= Nc <= 1024 and 1 <= Nf <= 1024 (got %d, %d)", Nc, Nf);
    if (B == 0) return SNERF_OK;
    if (!z || !weights || !u) return fail(SNERF_E_BADARG, "sample_pdf: z/weights/u is null");
    if (pts && (!o || !d)) return fail(SNERF_E_BADARG, "sample_pdf: pts requested but o/d is null");
    const int per_wave = Nc + 2 * (Nc - 1) + Nf + (Nc + Nf);
    const size_t lds = (size_t)SP_WAVES * per_wave * sizeof(float);  // <= 4 * 6142 * 4 = 96 KiB
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(sample_pdf_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(sample_pdf_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
            return fail(SNERF_E_LAUNCH, "sample_pdf: cannot raise dynamic LDS limit");
    }
    const int64_t grid = (B + SP_WAVES - 1) / SP_WAVES;
    if (grid > 0x7fffffffLL) return fail(SNERF_E_BADARG, "sample_pdf: B too large");
    if (direct)
        hipLaunchKernelGGL(sample_pdf_kernel<true>, dim3((unsigned)grid), dim3(SP_THREADS), lds, (hipStream_t)stream, z,
                           weights, u, o, d, tot, B, Nc, Nf, inds, z_samples, z_fine, pts);
    else
        hipLaunchKernelGGL(sample_pdf_kernel<false>, dim3((unsigned)grid), dim3(SP_THREADS), lds, (hipStream_t)stream, z,
                           weights, u, o, d, tot, B, Nc, Nf, inds, z_samples, z_fine, pts);
    return check_launch("sample_pdf");
}
