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

// #{k: row[k] <= v} (LE) / #{k: row[k] < v} of an ascending row of n: branch-free bisection over power-of-two steps, the same
// trip count in every lane.  N > 0: n is the compile-time constant N - a row of 2^k - 1 entries needs no range check at all, a
// row of 2^k entries one extra comparison with its last entry, and every read gets an immediate offset.
__device__ __forceinline__ int pow2_floor(int n) { return 1 << (31 - __builtin_clz(n)); }
template <int N, bool LE>
__device__ __forceinline__ int count_below(const float *__restrict__ row, int n, float v) {
    if constexpr (N > 0 && (N & (N + 1)) == 0) {            // N = 2^k - 1
        int pos = 0;
#pragma unroll
        for (int step = (N + 1) >> 1; step > 0; step >>= 1) {
            const float r = row[pos + step - 1];
            pos += (LE ? r <= v : r < v) ? step : 0;
        }
        return pos;
    } else if constexpr (N > 1 && (N & (N - 1)) == 0) {     // N = 2^k
        const float last = row[N - 1];
        const int head = count_below<N - 1, LE>(row, N - 1, v);
        return (LE ? last <= v : last < v) ? N : head;
    } else {
        int pos = 0;
        for (int step = pow2_floor(n); step > 0; step >>= 1) {
            const int p = pos + step;
            const float r = row[min(p, n) - 1];
            pos = (p <= n && (LE ? r <= v : r < v)) ? p : pos;
        }
        return pos;
    }
}
__host__ __device__ inline int sp_round4(int n) { return (n + 3) & ~3; }

// DIRECT = the literal sample_pdf(bins, weights) calling convention (utils.py:194): `z` holds the
// bins [B, Nc-1] and `weights` the interior weights [B, Nc-2]; no merge, no points.
// NC, NF > 0: the sample counts as compile-time constants (the launcher picks the 64 + 128 instance for the pipeline's shape).
template <bool DIRECT, int NC, int NF>
__global__ __launch_bounds__(SP_THREADS) void sample_pdf_kernel(
    const float *__restrict__ z, const float *__restrict__ weights, const float *__restrict__ u,
    const float *__restrict__ o, const float *__restrict__ d, const float *__restrict__ tot_in, int64_t B, int Nc_arg, int Nf_arg,
    int64_t *__restrict__ inds_out, float *__restrict__ zs_out, float *__restrict__ zf_out,
    float *__restrict__ pts_out) {
    const int Nc = NC > 0 ? NC : Nc_arg, Nf = NF > 0 ? NF : Nf_arg;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * SP_WAVES + wave;
    if (ray >= B) return;  // wave-uniform; no workgroup barriers below
    const int Nb = Nc - 1;           // number of bins == len(cdf)
    const int M = Nc - 2;            // number of interior weights
    const int Nt = Nc + Nf;
    const int n_in = sp_round4(Nc + 2 * Nb + Nf);      // the arrays below s_out; s_out and every wave's base are 16-byte aligned
    const int per_wave = n_in + sp_round4(Nt);
    float *s_z = smem + wave * per_wave;  // [Nc]
    float *s_bins = s_z + Nc;             // [Nb]
    float *s_cdf = s_bins + Nb;           // [Nb]
    float *s_zs = s_cdf + Nb;             // [Nf]
    float *s_out = s_z + n_in;            // [Nt]

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
        const double incl = wave_scan_add(p) + carry;
        if (k < M) s_cdf[k + 1] = (float)incl;
        carry = wave_last(incl);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    // ---- invert the cdf for every u ----------------------------------------------------------------
    for (int f = lane; f < Nf; f += WAVE) {
        const float uf = u[f];
        const int ind = count_below<(NC > 0 ? NC - 1 : 0), true>(s_cdf, Nb, uf);                    // searchsorted(cdf, u, 'right')
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
            s_out[i + count_below<NF, false>(s_zs, Nf, v)] = v;
        }
        for (int f = lane; f < Nf; f += WAVE) {
            const float v = s_zs[f];
            s_out[f + count_below<NC, true>(s_z, Nc, v)] = v;
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
    // Rows of a multiple of four samples go out as 16-byte vectors: z_fine straight from s_out, the points through a
    // 64-sample staging area laid over the arrays that are dead by now (one sample per lane -> 3 floats at stride 3,
    // conflict-free; then 48 lanes copy 768 contiguous bytes).  Other shapes take the dword path.
    const bool vec = (Nt & 3) == 0 && n_in >= 3 * WAVE && ((reinterpret_cast<uintptr_t>(zf_out) | reinterpret_cast<uintptr_t>(pts_out)) & 15) == 0;
    if (zf_out) {
        if (vec) {
            for (int q = lane; q < (Nt >> 2); q += WAVE)
                reinterpret_cast<float4 *>(zf_out + ray * Nt)[q] = reinterpret_cast<const float4 *>(s_out)[q];
        } else {
            for (int e = lane; e < Nt; e += WAVE) zf_out[ray * Nt + e] = s_out[e];
        }
    }
    if (pts_out) {
        const float ox = o[ray * 3 + 0], oy = o[ray * 3 + 1], oz = o[ray * 3 + 2];
        const float dx = d[ray * 3 + 0], dy = d[ray * 3 + 1], dz = d[ray * 3 + 2];
        float *pr = pts_out + ray * Nt * 3;
        if (vec) {
            float *stage = s_z;
            for (int c0 = 0; c0 < Nt; c0 += WAVE) {
                if (c0 + lane < Nt) {
                    const float zv = s_out[c0 + lane];
                    stage[3 * lane + 0] = __fadd_rn(ox, __fmul_rn(dx, zv));
                    stage[3 * lane + 1] = __fadd_rn(oy, __fmul_rn(dy, zv));
                    stage[3 * lane + 2] = __fadd_rn(oz, __fmul_rn(dz, zv));
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                const int nvec = (3 * min(WAVE, Nt - c0)) >> 2;
                if (lane < nvec) reinterpret_cast<float4 *>(pr + 3 * c0)[lane] = reinterpret_cast<const float4 *>(stage)[lane];
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
        } else {
            for (int e = lane; e < 3 * Nt; e += WAVE) {  // coalesced over the flattened [Nt,3] row
                const int s = e / 3, ch = e - 3 * s;
                const float zv = s_out[s];
                const float ov = ch == 0 ? ox : (ch == 1 ? oy : oz);
                const float dv = ch == 0 ? dx : (ch == 1 ? dy : dz);
                pr[e] = __fadd_rn(ov, __fmul_rn(dv, zv));
            }
        }
    }
}


// ---- backward of sample_pdf(bins, weights) (utils.py:194-228 under autograd) -------------------------------------------------
// The reference's fine_sampling detaches the samples (utils.py:260), but sample_pdf itself is differentiable w.r.t. bins and
// weights; the searchsorted indices carry no gradient, so with `inds` held fixed the map is piecewise linear:
//     t = (u - c0) / denom,  denom = c1 - c0 (1 where that is < 1e-5),  sample = b0 + t (b1 - b0)
//     d b0 = g (1 - t)   d b1 = g t   d t = g (b1 - b0)   d c0 = -d t / denom - d denom   d c1 = d denom = -d t t / denom
//     d pdf_i = sum_{k > i} d cdf_k        d w_m = d pdf_m / tot - (sum_i d pdf_i (w_i + 1e-5)) / tot^2
// One wave per ray; every lane owns bins k = lane, lane + 64, ... and walks the Nf samples for the ones that gather it
// (deterministic: no atomics); sums in fp64.
__global__ __launch_bounds__(SP_THREADS) void sample_pdf_bwd_kernel(const float *__restrict__ bins, const float *__restrict__ weights,
                                                                    const float *__restrict__ u, const int64_t *__restrict__ inds,
                                                                    const float *__restrict__ tot_in, const float *__restrict__ d_zs,
                                                                    int64_t B, int Nb, int Nf, float *__restrict__ d_bins,
                                                                    float *__restrict__ d_weights) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * SP_WAVES + wave;
    if (ray >= B) return;
    const int M = Nb - 1;
    const int per_wave = sp_round4(2 * Nb) + 4 * sp_round4(Nf);
    float *s_cdf = smem + wave * per_wave;       // [Nb]
    float *s_dc = s_cdf + Nb;                    // [Nb]  d cdf
    float *s_c0 = s_cdf + sp_round4(2 * Nb);     // [Nf] per sample: d c0, d c1, d b0, d b1
    float *s_c1 = s_c0 + sp_round4(Nf), *s_b0 = s_c1 + sp_round4(Nf), *s_b1 = s_b0 + sp_round4(Nf);
    const float *br = bins + ray * Nb, *wr = weights + ray * M;
    double part = 0.0;
    for (int i = lane; i < M; i += WAVE) part += (double)__fadd_rn(wr[i], 1e-5f);
    const float tot = tot_in ? tot_in[ray] : (float)wave_sum(part);   // (strict mode: the forward's normalising sum -> the forward's cdf)
    if (lane == 0) s_cdf[0] = 0.f;
    double carry = 0.0;
    for (int c0 = 0; c0 < M; c0 += WAVE) {
        const int k = c0 + lane;
        double p = 0.0;
        if (k < M) p = (double)__fdiv_rn(__fadd_rn(wr[k], 1e-5f), tot);
        const double incl = wave_scan_add(p) + carry;
        if (k < M) s_cdf[k + 1] = (float)incl;
        carry = wave_last(incl);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    for (int f = lane; f < Nf; f += WAVE) {
        const int ind = (int)inds[ray * Nf + f];
        const int below = max(0, ind - 1), above = min(Nb - 1, ind);
        const float c0 = s_cdf[below], c1 = s_cdf[above], b0 = br[below], b1 = br[above];
        const float raw = c1 - c0;
        const bool active = !(raw < 1e-5f);
        const float denom = active ? raw : 1.f;
        const float t = (u[f] - c0) / denom;
        const float g = d_zs[ray * Nf + f];
        const float dt = g * (b1 - b0);
        const float dden = active ? -dt * t / denom : 0.f;
        s_c0[f] = -dt / denom - dden;
        s_c1[f] = dden;
        s_b0[f] = g * (1.f - t);
        s_b1[f] = g * t;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    for (int k = lane; k < Nb; k += WAVE) {
        double dc = 0.0, db = 0.0;
        for (int f = 0; f < Nf; ++f) {
            const int ind = (int)inds[ray * Nf + f];
            const int below = max(0, ind - 1), above = min(Nb - 1, ind);
            if (below == k) dc += (double)s_c0[f], db += (double)s_b0[f];
            if (above == k) dc += (double)s_c1[f], db += (double)s_b1[f];
        }
        s_dc[k] = (float)dc;
        d_bins[ray * Nb + k] = (float)db;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    // d pdf_i = sum_{k = i + 1}^{Nb - 1} d cdf_k (cdf_0 = 0 is a constant); S = sum_i d pdf_i (w_i + 1e-5)
    double s_part = 0.0;
    for (int i = lane; i < M; i += WAVE) {
        double dp = 0.0;
        for (int k = i + 1; k < Nb; ++k) dp += (double)s_dc[k];
        s_part += dp * (double)__fadd_rn(wr[i], 1e-5f);
        d_weights[ray * M + i] = (float)dp;   // d pdf for now; finished below
    }
    const double S = wave_sum(s_part);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const double tt = (double)tot;
    for (int i = lane; i < M; i += WAVE) d_weights[ray * M + i] = (float)((double)d_weights[ray * M + i] / tt - S / (tt * tt));
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
        return fail(SNERF_E_BADARG, "sample_pdf: need 3 <= Nc <= 1024 and 1 <= Nf <= 1024 (got %d, %d; below three coarse samples the "
                                    "reference's own sample_pdf raises: empty cdf)", Nc, Nf);
    if (B == 0) return SNERF_OK;
    if (!z || !weights || !u) return fail(SNERF_E_BADARG, "sample_pdf: z/weights/u is null");
    if (pts && (!o || !d)) return fail(SNERF_E_BADARG, "sample_pdf: pts requested but o/d is null");
    const int per_wave = sp_round4(Nc + 2 * (Nc - 1) + Nf) + sp_round4(Nc + Nf);
    const size_t lds = (size_t)SP_WAVES * per_wave * sizeof(float);  // <= 4 * 6144 * 4 = 96 KiB
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(sample_pdf_kernel<false, 0, 0>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(sample_pdf_kernel<true, 0, 0>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
            return fail(SNERF_E_LAUNCH, "sample_pdf: cannot raise dynamic LDS limit");
    }
    const int64_t grid = (B + SP_WAVES - 1) / SP_WAVES;
    if (grid > 0x7fffffffLL) return fail(SNERF_E_BADARG, "sample_pdf: B too large");
    if (direct)
        hipLaunchKernelGGL((sample_pdf_kernel<true, 0, 0>), dim3((unsigned)grid), dim3(SP_THREADS), lds, (hipStream_t)stream, z,
                           weights, u, o, d, tot, B, Nc, Nf, inds, z_samples, z_fine, pts);
    else if (Nc == 64 && Nf == 128)
        hipLaunchKernelGGL((sample_pdf_kernel<false, 64, 128>), dim3((unsigned)grid), dim3(SP_THREADS), lds, (hipStream_t)stream, z,
                           weights, u, o, d, tot, B, Nc, Nf, inds, z_samples, z_fine, pts);
    else
        hipLaunchKernelGGL((sample_pdf_kernel<false, 0, 0>), dim3((unsigned)grid), dim3(SP_THREADS), lds, (hipStream_t)stream, z,
                           weights, u, o, d, tot, B, Nc, Nf, inds, z_samples, z_fine, pts);
    return check_launch("sample_pdf");
}

extern "C" int snerf_sample_pdf_bins_bwd_f32(const float *bins, const float *weights, const float *u, const int64_t *inds,
                                             const float *tot, const float *d_z_samples, int64_t B, int Nb, int Nf, float *d_bins,
                                             float *d_weights, snerf_stream_t stream) {
    using namespace snerf;
    if (B < 0) return fail(SNERF_E_BADARG, "sample_pdf_bins_bwd: negative B");
    if (Nb < 2 || Nb > 1023 || Nf < 1 || Nf > 1024) return fail(SNERF_E_BADARG, "sample_pdf_bins_bwd: need 2 <= Nb <= 1023 and 1 <= Nf <= 1024");
    if (B == 0) return SNERF_OK;
    if (!bins || !weights || !u || !inds || !d_z_samples || !d_bins || !d_weights) return fail(SNERF_E_BADARG, "sample_pdf_bins_bwd: null pointer");
    const size_t lds = (size_t)SP_WAVES * (sp_round4(2 * Nb) + 4 * sp_round4(Nf)) * sizeof(float);     // <= 4 * (2048 + 4096) * 4 = 96 KiB
    static LdsRaised raised;
    if (lds > 64 * 1024)
        if (int rc = raise_dynamic_lds(reinterpret_cast<const void *>(sample_pdf_bwd_kernel), 128 * 1024, raised, "sample_pdf_bins_bwd")) return rc;
    const int64_t grid = (B + SP_WAVES - 1) / SP_WAVES;
    if (grid > 0x7fffffffLL) return fail(SNERF_E_BADARG, "sample_pdf_bins_bwd: B too large");
    hipLaunchKernelGGL(sample_pdf_bwd_kernel, dim3((unsigned)grid), dim3(SP_THREADS), lds, (hipStream_t)stream, bins, weights, u, inds,
                       tot, d_z_samples, B, Nb, Nf, d_bins, d_weights);
    return check_launch("sample_pdf_bins_bwd");
}
