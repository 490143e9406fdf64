// WarpFieldNet forward fused with its surroundings in SmplNerfPipeline (a7):
//   warp   = linear2(relu(linear1([PE(x) | PE(pose)])))            models/warp_field_net.py:17-21
//   x'     = x + warp                                               models/smpl_nerf_pipeline.py:48-49 / :77-79
//   sdir   = x' - o   (per-sample view direction)                   models/smpl_nerf_pipeline.py:52-53 / :82-83
// Same machinery as the RenderRayNet kernel (mlp_device.h): one wave owns 16 samples, the 100 -> 256
// layer runs on v_mfma_f32_16x16x4_f32 with the encoding evaluated in registers (the pose encoding is a
// per-ray constant read as "additional input"), the 256 -> 3 head is one padded tile.  52 736 FLOP per
// sample (4 % of a RenderRayNet evaluation); HBM: 12 B in, 36 B out per sample.
#include <stdlib.h>

#include "warp_plan.h"

namespace snerf {

// activation / gradient buffers of the 2-layer warp net in the generic training layout (mlp_plan.h)
inline void warp_train_layout(const Plan &P, TrainLayout &L) {
    const int T = P.width / 16;
    L = TrainLayout{};
    L.T = T;
    L.pe = 0;
    L.add = P.pos_nkb;
    L.x[1] = P.pos_nkb + P.add_nkb;  // hidden activation h
    L.act_rows = L.x[1] + T;
    L.dy[0] = 0;
    L.dy[1] = T;
    L.dy_rows = T + 1;
    int g = 0;
    for (int l = 0; l < 2; ++l) {
        L.gp[l] = g;
        g += P.layer[l].t_out * P.layer[l].nkb * 256 + P.layer[l].t_out * 16;
    }
    L.gp_floats = g;
    for (int sgi = 0; sgi < P.layer[0].nseg; ++sgi)
        L.xrow[0][sgi] = (short)(P.layer[0].seg[sgi].type == SEG_PE ? L.pe : L.add);
    L.xrow[1][0] = (short)L.x[1];
}

template <int WIDTH, int NWAVES, bool TRAIN>
__global__ __launch_bounds__(NWAVES * 64) void warp_fwd_kernel(WarpArgs A) {
    constexpr int NT = NWAVES * 64;
    constexpr int T = WIDTH / 16;
    extern __shared__ __attribute__((aligned(16))) float ring[];  // RING_BYTES (SNERF_LAUNCH_RING)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t sample = ((int64_t)blockIdx.x * NWAVES + wave) * 16 + (lane & 15);
    const bool valid = sample < A.n;
    const int64_t sc = valid ? sample : A.n - 1;
    const int64_t ray = sc / A.spr;
    SampleCtx c;
    c.g = lane >> 4;
    c.enc = nullptr;
    c.px = c.py = c.pz = c.dx = c.dy = c.dz = 0.f;
    if (A.x) {
        c.px = A.x[sc * 3 + 0];
        c.py = A.x[sc * 3 + 1];
        c.pz = A.x[sc * 3 + 2];
    }
    c.add = A.add_dim ? A.add + ray * A.add_dim : nullptr;

    SlabPipe<NT> pipe;
    pipe.prologue(A.packed, ring, tid);
    f4 in[T], acc[T];
    {
        LayerRun<T, NT> run(pipe, lane);
        run.init(acc);
        for (int kb = 0; kb < A.pos_nkb; ++kb) {
            const f4 b = pe_operand<false>(c, false, A.pos_L, A.pos_id, kb, 0);
            if (TRAIN && valid) store_tile(A.act, kb, A.n, sample, c.g, b);
            run.step(b, acc);
        }
        for (int kb = 0; kb < A.add_nkb; ++kb) {
            const f4 b = add_operand(c, A.add_dim, kb);
            if (TRAIN && valid) store_tile(A.act, A.pos_nkb + kb, A.n, sample, c.g, b);
            run.step(b, acc);
        }
        run.finish();
        relu_into(in, acc);
        if (TRAIN && valid) store_tiles(A.act, A.pos_nkb + A.add_nkb, A.n, sample, c.g, in);
    }
    f4 w[1];
    {
        LayerRun<1, NT> run(pipe, lane);
        run.init(w);
#pragma unroll
        for (int kb = 0; kb < T; ++kb) run.step(in[kb], w);
        run.finish();
    }
    if (valid && c.g == 0) {
        float *wp = A.warp + sample * 3;
        wp[0] = w[0][0];
        wp[1] = w[0][1];
        wp[2] = w[0][2];
        if (A.warped) {
            const float wx = __fadd_rn(c.px, w[0][0]), wy = __fadd_rn(c.py, w[0][1]), wz = __fadd_rn(c.pz, w[0][2]);
            float *q = A.warped + sample * 3;
            q[0] = wx;
            q[1] = wy;
            q[2] = wz;
            if (A.sdirs) {
                const float *op = A.o + ray * 3;
                float *s = A.sdirs + sample * 3;
                s[0] = __fsub_rn(wx, op[0]);
                s[1] = __fsub_rn(wy, op[1]);
                s[2] = __fsub_rn(wz, op[2]);
            }
        }
    }
}

// The same forward with the WHOLE net resident in LDS.  The streaming kernel above pays 5 slabs (165 KB of L2 -> LDS
// traffic, five workgroup barriers) per 64 samples for 128 MFMAs per wave, one wave per SIMD: 3.3 ms per 128x128 frame,
// a quarter of the matrix peak.  The warp net is small - linear1: T x nkb tiles of 1 KiB, linear2: T tiles, two bias
// blocks: 130 KiB at width 256 with the default encoders - so a persistent workgroup copies it into LDS once and then
// walks its sample tiles with no barrier and no weight traffic at all; the A-operand prefetch of kblock() runs on across
// layers and tiles.  Used whenever the net fits (warp_resident_bytes <= 160 KiB - what the hardware has).
__host__ __device__ inline int warp_resident_bytes(int T, int nkb0) { return (T * nkb0 + T + 2) * 1024; }

// The pose encoding is a per-RAY constant (SmplNerfPipeline expands goal_pose over the samples, models/smpl_nerf_pipeline.py:
// 40-45): its columns of linear1 contribute the same 256-vector to all 64 / 192 samples of a ray.  For inference that
// vector is evaluated once per ray - ray_bias[ray][o] = linear1.bias[o] + sum_c linear1[o, pos_dim + c] * pose_enc[ray][c],
// read from the packed stream (A tile (kb, to), lane m + 16 g, value r  <->  W[16 to + m][16 kb + 4 g + r]) - and the
// resident kernel starts its accumulators from it instead of running the pose k-blocks per sample (3 of the 7 k-blocks
// of the default net: 37 % of the kernel's MFMAs).  One thread per (ray, output feature).
__global__ __launch_bounds__(256) void warp_ray_bias_kernel(const float *__restrict__ packed, const float *__restrict__ add,
                                                            int64_t n_rays, int add_dim, int pos_nkb, int add_nkb, int T,
                                                            float *__restrict__ out) {
    const int width = T * 16;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_rays * width) return;
    const int64_t ray = e / width;
    const int o = (int)(e - ray * width), to = o >> 4, m = o & 15;
    const float *a = add + ray * add_dim;
    float sum = packed[SLAB_A_FLOATS + o];   // linear1.bias (aux block of the first slab)
    for (int kb = 0; kb < add_nkb; ++kb) {
        const int t = (pos_nkb + kb) * T + to;
        const float *tile = packed + (int64_t)(t / SLAB_TILES) * SLAB_FLOATS + (t % SLAB_TILES) * 256;
        for (int g = 0; g < 4; ++g) {
            const f4 w = *reinterpret_cast<const f4 *>(tile + (m + 16 * g) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * kb + 4 * g + r;
                if (c < add_dim) sum += w[r] * a[c];
            }
        }
    }
    out[e] = sum;
}

template <int WIDTH, int NWAVES, bool TRAIN>
__global__ __launch_bounds__(NWAVES * 64) void warp_fwd_resident_kernel(WarpArgs A) {
    constexpr int NT = NWAVES * 64;
    constexpr int T = WIDTH / 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nkb0 = A.pos_nkb + A.add_nkb;
    float *w1 = lds;                       // [nkb0][T][64 lanes][4]
    float *w2 = w1 + nkb0 * T * 256;       // [T k-blocks][1 tile]
    float *b1 = w2 + T * 256;              // 256 floats (the aux block of linear1's first slab)
    float *b2 = b1 + 256;
    {   // slab stream -> dense LDS image (tile t of a layer sits in slab t / SLAB_TILES at index t % SLAB_TILES)
        const f4 *src = reinterpret_cast<const f4 *>(A.packed);
        f4 *dst = reinterpret_cast<f4 *>(lds);
        const int n1 = nkb0 * T * 64, n2 = T * 64;   // f4 counts
        for (int e = tid; e < n1; e += NT) {
            const int t = e >> 6;
            dst[e] = src[(int64_t)(t / SLAB_TILES) * (SLAB_FLOATS / 4) + (t % SLAB_TILES) * 64 + (e & 63)];
        }
        for (int e = tid; e < n2; e += NT) dst[n1 + e] = src[(int64_t)A.l1_slab * (SLAB_FLOATS / 4) + e];
        for (int e = tid; e < 64; e += NT) {
            dst[n1 + n2 + e] = src[SLAB_A_FLOATS / 4 + e];
            dst[n1 + n2 + 64 + e] = src[(int64_t)A.l1_slab * (SLAB_FLOATS / 4) + SLAB_A_FLOATS / 4 + e];
        }
    }
    __syncthreads();
    const f4 *first = reinterpret_cast<const f4 *>(w1) + lane;
    f4 pa0 = first[0], pa1 = first[64];   // first A pair of the first k-block; kblock() keeps the prefetch rolling
    for (int64_t tile = blockIdx.x; tile < A.n_tiles; tile += gridDim.x) {
        const int64_t sample = (tile * NWAVES + wave) * 16 + (lane & 15);
        const bool valid = sample < A.n;
        const int64_t sc = valid ? sample : A.n - 1;
        const int64_t ray = sc / A.spr;
        SampleCtx c;
        c.g = lane >> 4;
        c.enc = nullptr;
        c.px = c.py = c.pz = c.dx = c.dy = c.dz = 0.f;
        if (A.x) {
            c.px = A.x[sc * 3 + 0];
            c.py = A.x[sc * 3 + 1];
            c.pz = A.x[sc * 3 + 2];
        }
        c.add = A.add_dim ? A.add + ray * A.add_dim : nullptr;
        f4 in[T], acc[T];
        const bool folded = !TRAIN && A.ray_bias;   // the pose columns arrive as this ray's accumulator start
        if (folded) {
            const f4 *rb = reinterpret_cast<const f4 *>(A.ray_bias + ray * WIDTH) + (lane >> 4);
#pragma unroll
            for (int to = 0; to < T; ++to) acc[to] = rb[to * 4];
        } else {
            const f4 *aux = reinterpret_cast<const f4 *>(b1) + (lane >> 4);
#pragma unroll
            for (int to = 0; to < T; ++to) acc[to] = aux[to * 4];
        }
        const int nkb_run = folded ? A.pos_nkb : nkb0;
        for (int kb = 0; kb < nkb_run; ++kb) {
            f4 b;
            if (kb < A.pos_nkb) b = pe_operand<false>(c, false, A.pos_L, A.pos_id, kb, 0);
            else b = add_operand(c, A.add_dim, kb - A.pos_nkb);
            if (TRAIN && valid) store_tile(A.act, kb, A.n, sample, c.g, b);
            const float *cur = w1 + kb * (T * 256);
            kblock<T>(cur, kb + 1 < nkb_run ? cur + T * 256 : w2, b, acc, pa0, pa1, lane);
        }
        relu_into(in, acc);
        if (TRAIN && valid) store_tiles(A.act, nkb0, A.n, sample, c.g, in);
        f4 w[1];
        w[0] = reinterpret_cast<const f4 *>(b2)[lane >> 4];
#pragma unroll
        for (int kb = 0; kb < T; ++kb) kblock<1>(w2 + kb * 256, kb + 1 < T ? w2 + (kb + 1) * 256 : w1, in[kb], w, pa0, pa1, lane);
        if (valid && c.g == 0) {
            float *wp = A.warp + sample * 3;
            wp[0] = w[0][0];
            wp[1] = w[0][1];
            wp[2] = w[0][2];
            if (A.warped) {
                const float wx = __fadd_rn(c.px, w[0][0]), wy = __fadd_rn(c.py, w[0][1]), wz = __fadd_rn(c.pz, w[0][2]);
                float *q = A.warped + sample * 3;
                q[0] = wx;
                q[1] = wy;
                q[2] = wz;
                if (A.sdirs) {
                    const float *op = A.o + ray * 3;
                    float *s = A.sdirs + sample * 3;
                    s[0] = __fsub_rn(wx, op[0]);
                    s[1] = __fsub_rn(wy, op[1]);
                    s[2] = __fsub_rn(wz, op[2]);
                }
            }
        }
    }
}

// backward: d warp [n,3] -> d h = linear2^T d warp masked by h > 0 (stored as the layer-0 dY tile-rows) and the
// head's dY tile-row; the weight gradients then come from the generic split-K wgrad (mlp_train.hip).
struct WarpBwdArgs {
    const float *packed_t, *act, *d_warp;
    float *dy;
    int64_t n;
    int h_row;
};

// No slab ring: the transposed head is ONE k-block of T tiles (T KiB), so a workgroup keeps it in a small static LDS block and
// walks its sample tiles (the slab-ring form of r02 loaded three 33 KiB slabs per 64 samples and, with its 99 KiB of LDS, ran one
// 4-wave workgroup per CU - for a kernel that only moves 2 KB per sample).
template <int WIDTH, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void warp_bwd_light_kernel(WarpBwdArgs A, int64_t n_tiles) {
    constexpr int NT = NWAVES * 64;
    constexpr int T = WIDTH / 16;
    __shared__ __attribute__((aligned(16))) float wt[T * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    for (int e = tid; e < T * 64; e += NT) reinterpret_cast<f4 *>(wt)[e] = reinterpret_cast<const f4 *>(A.packed_t)[e];
    __syncthreads();
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t sample = (tile * NWAVES + wave) * 16 + (lane & 15);
        const bool valid = sample < A.n;
        const int64_t sc = valid ? sample : A.n - 1;
        const float *dp = A.d_warp + sc * 3;
        const f4 dw = g == 0 ? f4{dp[0], dp[1], dp[2], 0.f} : f4{0.f, 0.f, 0.f, 0.f};
        if (valid) store_tile(A.dy, T, A.n, sample, g, dw);
        f4 m[T];
#pragma unroll
        for (int t = 0; t < T; ++t) m[t] = load_tile(A.act, A.h_row + t, A.n, sc, g);   // in flight behind the MFMAs
        f4 dh[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const f4 a = reinterpret_cast<const f4 *>(wt)[t * 64 + lane];
            f4 acc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], dw[r], acc, 0, 0, 0);
            dh[t] = acc;
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            dh[t][0] = m[t][0] > 0.f ? dh[t][0] : 0.f;
            dh[t][1] = m[t][1] > 0.f ? dh[t][1] : 0.f;
            dh[t][2] = m[t][2] > 0.f ? dh[t][2] : 0.f;
            dh[t][3] = m[t][3] > 0.f ? dh[t][3] : 0.f;
        }
        if (valid) store_tiles(A.dy, 0, A.n, sample, g, dh);
    }
}

// in mlp_train.hip
int launch_pack_t(const Plan &P, const BwdPlan &B, const float *params_flat, float *packed_t, hipStream_t s, const char *what);

}  // namespace snerf

extern "C" int snerf_warp_train_sizes(const snerf_warp_desc *desc, int64_t n, int64_t *act_floats, int64_t *dy_floats,
                                      int64_t *packed_t_floats, int64_t *gpart_floats) {
    using namespace snerf;
    Plan P;
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "warp_train_sizes: desc is null");
    if (make_warp_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "warp_train_sizes: %s", why);
    if (n < 0) return fail(SNERF_E_BADARG, "warp_train_sizes: negative n");
    TrainLayout L;
    warp_train_layout(P, L);
    if (act_floats) *act_floats = (int64_t)L.act_rows * n * 16;
    if (dy_floats) *dy_floats = (int64_t)L.dy_rows * n * 16;
    if (packed_t_floats) *packed_t_floats = (int64_t)(1 + SLAB_PAD) * SLAB_FLOATS;
    if (gpart_floats) *gpart_floats = (int64_t)wgrad_chunks(n) * L.gp_floats;
    return SNERF_OK;
}

extern "C" int snerf_warp_pack_t_f32(const snerf_warp_desc *desc, const float *params_flat, float *packed_t,
                                     snerf_stream_t stream) {
    using namespace snerf;
    Plan P;
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "warp_pack_t: desc is null");
    if (make_warp_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "warp_pack_t: %s", why);
    if (!params_flat || !packed_t) return fail(SNERF_E_BADARG, "warp_pack_t: null pointer");
    if (!aligned(packed_t, 16)) return fail(SNERF_E_ALIGN, "warp_pack_t: packed_t must be 16-byte aligned");
    BwdPlan B;
    B.nl = 1;
    B.total_slabs = 1;
    B.layer[0] = BwdLayer{1, 0, P.width / 16, 1, -1, 0, 1};  // linear2^T: 3 output rows -> width hidden columns
    return launch_pack_t(P, B, params_flat, packed_t, (hipStream_t)stream, "warp_pack_t");
}

extern "C" int snerf_warp_bwd_f32(const snerf_warp_desc *desc, const float *packed_t, const float *act,
                                  const float *d_warp, int64_t n, float *dy, float *gpart, float *flat_grad,
                                  snerf_stream_t stream) {
    return snerf::launch_warp_bwd(desc, packed_t, act, d_warp, n, dy, gpart, flat_grad, stream, false);
}

// accumulate: flat_grad += (the warp net is evaluated twice per step - coarse and fine stage - and once more per ray chunk)
int snerf::launch_warp_bwd(const snerf_warp_desc *desc, const float *packed_t, const float *act, const float *d_warp, int64_t n,
                           float *dy, float *gpart, float *flat_grad, snerf_stream_t stream, bool accumulate) {
    Plan P;
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "warp_bwd: desc is null");
    if (make_warp_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "warp_bwd: %s", why);
    if (n < 0) return fail(SNERF_E_BADARG, "warp_bwd: negative n");
    if (n == 0) return SNERF_OK;
    if (!packed_t || !act || !d_warp || !dy || !gpart || !flat_grad) return fail(SNERF_E_BADARG, "warp_bwd: null pointer");
    if (!aligned(packed_t, 16) || !aligned(act, 16) || !aligned(dy, 16) || !aligned(gpart, 16))
        return fail(SNERF_E_ALIGN, "warp_bwd: buffers must be 16-byte aligned");
    TrainLayout L;
    warp_train_layout(P, L);
    WarpBwdArgs A{packed_t, act, d_warp, dy, n, L.x[1]};
    hipStream_t s = (hipStream_t)stream;
    {   // the ring-free dgrad (the slab-ring form of r02 lost its A/B in r03 and is gone)
        constexpr int LW = 8;
        const int64_t n_tiles = (n + LW * 16 - 1) / (LW * 16);
        const int64_t g = n_tiles < 2048 ? n_tiles : 2048;   // grid-stride over the sample tiles
        if (P.width == 256) hipLaunchKernelGGL((warp_bwd_light_kernel<256, LW>), dim3((unsigned)g), dim3(LW * 64), 0, s, A, n_tiles);
        else hipLaunchKernelGGL((warp_bwd_light_kernel<128, LW>), dim3((unsigned)g), dim3(LW * 64), 0, s, A, n_tiles);
    }
    int rc = check_launch("warp_bwd");
    if (rc) return rc;
    return launch_wgrad(P, L, act, dy, n, gpart, flat_grad, s, 0, accumulate);
}

extern "C" int64_t snerf_warp_param_floats(const snerf_warp_desc *desc) {
    using namespace snerf;
    Plan P;
    const char *why;
    if (!desc || make_warp_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "warp: bad descriptor");
    return P.param_floats;
}

extern "C" int64_t snerf_warp_packed_floats(const snerf_warp_desc *desc) {
    using namespace snerf;
    Plan P;
    const char *why;
    if (!desc || make_warp_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "warp: bad descriptor");
    return (int64_t)(P.total_slabs + SLAB_PAD) * SLAB_FLOATS;
}

extern "C" int snerf_warp_pack_f32(const snerf_warp_desc *desc, const float *params_flat, float *packed,
                                   snerf_stream_t stream) {
    using namespace snerf;
    Plan P;
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "warp_pack: desc is null");
    if (make_warp_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "warp_pack: %s", why);
    if (!params_flat || !packed) return fail(SNERF_E_BADARG, "warp_pack: null pointer");
    if (!aligned(packed, 16)) return fail(SNERF_E_ALIGN, "warp_pack: packed must be 16-byte aligned");
    return launch_pack(P, params_flat, packed, (hipStream_t)stream, "warp_pack");
}

namespace snerf {
static int launch_warp_fwd(const snerf_warp_desc *desc, const float *packed, const float *x, const float *pose_enc,
                           const float *o, int64_t n, int samples_per_ray, float *warp, float *warped, float *sdirs,
                           float *act, snerf_stream_t stream, void *workspace = nullptr, int64_t workspace_bytes = 0);
// bytes of the per-ray pose-fold table of an inference call, 0 when the fold does not apply (warp_ray_bias_kernel)
static int64_t warp_fold_bytes(const Plan &P, int64_t n, int spr) {
    const int T = P.width / 16, nkb0 = P.pos_nkb + P.add_nkb;
    if (warp_resident_bytes(T, nkb0) > 160 * 1024) return 0;
    if (!(P.add_dim > 0 && P.pos_nkb > 0 && tuning().warp_fold && spr >= 8 && n > 0 && n % spr == 0)) return 0;
    const int64_t floats = (n / spr) * P.width;
    if ((floats + 255) / 256 > 0x7fffffffLL) return 0;   // (more rays than a grid holds: the per-sample form)
    return floats * (int64_t)sizeof(float);
}
}
extern "C" int64_t snerf_warp_fold_workspace_bytes(const snerf_warp_desc *desc, int64_t n, int samples_per_ray) {
    using namespace snerf;
    Plan P;
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "warp_fold_workspace_bytes: desc is null");
    if (make_warp_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "warp_fold_workspace_bytes: %s", why);
    if (n < 0 || samples_per_ray < 1) return fail(SNERF_E_BADARG, "warp_fold_workspace_bytes: bad n/samples_per_ray");
    return warp_fold_bytes(P, n, samples_per_ray);
}
extern "C" int snerf_warp_fwd_ws_f32(const snerf_warp_desc *desc, const float *packed, const float *x,
                                     const float *pose_enc, const float *o, int64_t n, int samples_per_ray,
                                     float *warp, float *warped, float *sdirs, void *workspace, int64_t workspace_bytes,
                                     snerf_stream_t stream) {
    if (workspace_bytes < 0) return snerf::fail(SNERF_E_BADARG, "warp_fwd: negative workspace_bytes");
    return snerf::launch_warp_fwd(desc, packed, x, pose_enc, o, n, samples_per_ray, warp, warped, sdirs, nullptr, stream,
                                  workspace, workspace ? workspace_bytes : 0);
}
extern "C" int snerf_warp_fwd_f32(const snerf_warp_desc *desc, const float *packed, const float *x,
                                  const float *pose_enc, const float *o, int64_t n, int samples_per_ray,
                                  float *warp, float *warped, float *sdirs, snerf_stream_t stream) {
    return snerf::launch_warp_fwd(desc, packed, x, pose_enc, o, n, samples_per_ray, warp, warped, sdirs, nullptr, stream);
}
extern "C" int snerf_warp_fwd_train_f32(const snerf_warp_desc *desc, const float *packed, const float *x,
                                        const float *pose_enc, const float *o, int64_t n, int samples_per_ray,
                                        float *warp, float *warped, float *sdirs, float *act, snerf_stream_t stream) {
    if (!act) return snerf::fail(SNERF_E_BADARG, "warp_fwd_train: act is null");
    return snerf::launch_warp_fwd(desc, packed, x, pose_enc, o, n, samples_per_ray, warp, warped, sdirs, act, stream);
}
static int snerf::launch_warp_fwd(const snerf_warp_desc *desc, const float *packed, const float *x, const float *pose_enc,
                                  const float *o, int64_t n, int samples_per_ray, float *warp, float *warped,
                                  float *sdirs, float *act, snerf_stream_t stream, void *workspace, int64_t workspace_bytes) {
    Plan P;
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "warp_fwd: desc is null");
    if (make_warp_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "warp_fwd: %s", why);
    if (n < 0 || samples_per_ray < 1) return fail(SNERF_E_BADARG, "warp_fwd: bad n/samples_per_ray");
    if (n == 0) return SNERF_OK;
    if (!packed || !warp) return fail(SNERF_E_BADARG, "warp_fwd: null pointer");
    if (P.pos_dim && !x) return fail(SNERF_E_BADARG, "warp_fwd: x is null");
    if (P.add_dim && !pose_enc) return fail(SNERF_E_BADARG, "warp_fwd: pose_enc is null");
    if (warped && !x) return fail(SNERF_E_BADARG, "warp_fwd: warped requested without x");
    if (sdirs && (!warped || !o)) return fail(SNERF_E_BADARG, "warp_fwd: sdirs needs warped and o");
    if (!aligned(packed, 16)) return fail(SNERF_E_ALIGN, "warp_fwd: packed must be 16-byte aligned");
    WarpArgs A{};
    A.packed = packed;
    A.x = P.pos_dim ? x : (warped ? x : nullptr);
    A.add = pose_enc;
    A.o = o;
    A.warp = warp;
    A.warped = warped;
    A.sdirs = sdirs;
    A.n = n;
    A.spr = samples_per_ray;
    A.pos_L = desc->pos_freqs;
    A.pos_id = desc->pos_identity ? 1 : 0;
    A.pos_nkb = P.pos_nkb;
    A.add_dim = P.add_dim;
    A.add_nkb = P.add_nkb;
    A.act = act;
    hipStream_t s = (hipStream_t)stream;
    {   // the LDS-resident persistent kernel whenever the net fits the 160 KiB of a CU (else the slab-streaming kernel)
        const bool resident = true;
        const int T = P.width / 16, nkb0 = P.pos_nkb + P.add_nkb;
        const int bytes = warp_resident_bytes(T, nkb0);
        if (resident && bytes <= 160 * 1024) {
            // inference: 16 waves (121 registers: 4 waves per SIMD cover each other's sincos phases); training forward: 8
            const int RW = act ? 8 : 16;
            A.l1_slab = P.layer[1].first_slab;
            A.n_tiles = (n + RW * 16 - 1) / (RW * 16);
            const int n_cu = device_cu_count("warp_fwd");
            if (n_cu < 1) return n_cu;
            const int64_t g = A.n_tiles < n_cu ? A.n_tiles : n_cu;
            // inference with pose columns, at least one position k-block and a workspace: the per-ray fold (warp_ray_bias_kernel),
            // its table in the CALLER's workspace (snerf_warp_fold_workspace_bytes); no workspace: the per-sample form
            if (!act && workspace) {
                if (const int64_t need = warp_fold_bytes(P, n, samples_per_ray)) {
                    if (workspace_bytes < need)
                        return fail(SNERF_E_BADARG, "warp_fwd: workspace of %lld bytes, the per-ray fold needs %lld "
                                    "(snerf_warp_fold_workspace_bytes)", (long long)workspace_bytes, (long long)need);
                    if (!aligned(workspace, 16)) return fail(SNERF_E_ALIGN, "warp_fwd: workspace must be 16-byte aligned");
                    float *ray_bias = reinterpret_cast<float *>(workspace);
                    const int64_t floats = need / (int64_t)sizeof(float);
                    hipLaunchKernelGGL(warp_ray_bias_kernel, dim3((unsigned)((floats + 255) / 256)), dim3(256), 0, s, packed, pose_enc,
                                       n / samples_per_ray, P.add_dim, P.pos_nkb, P.add_nkb, T, ray_bias);
                    A.ray_bias = ray_bias;
                }
            }
#define SNERF_WARP_RES(W_, RW_, TR_)                                                                                      \
    do {                                                                                                                  \
        static LdsRaised raised; /* per device */                                                                         \
        if (int rc_ = raise_dynamic_lds(reinterpret_cast<const void *>(warp_fwd_resident_kernel<W_, RW_, TR_>), 160 * 1024, \
                                        raised, "warp_fwd"))                                                              \
            return rc_;                                                                                                   \
        hipLaunchKernelGGL((warp_fwd_resident_kernel<W_, RW_, TR_>), dim3((unsigned)g), dim3(RW_ * 64), bytes, s, A);      \
    } while (0)
            if (P.width == 256) {
                if (act) SNERF_WARP_RES(256, 8, true);
                else SNERF_WARP_RES(256, 16, false);
            } else {
                if (act) SNERF_WARP_RES(128, 8, true);
                else SNERF_WARP_RES(128, 16, false);
            }
#undef SNERF_WARP_RES
            return check_launch("warp_fwd(resident)");
        }
    }
    constexpr int NW = 4;
    const int64_t grid = (n + NW * 16 - 1) / (NW * 16);
    if (grid > 0x7fffffffLL) return fail(SNERF_E_BADARG, "warp_fwd: n too large");
    if (P.width == 256) {
        if (act) SNERF_LAUNCH_RING((warp_fwd_kernel<256, NW, true>), dim3((unsigned)grid), dim3(NW * 64), s, A);
        else SNERF_LAUNCH_RING((warp_fwd_kernel<256, NW, false>), dim3((unsigned)grid), dim3(NW * 64), s, A);
    } else {
        if (act) SNERF_LAUNCH_RING((warp_fwd_kernel<128, NW, true>), dim3((unsigned)grid), dim3(NW * 64), s, A);
        else SNERF_LAUNCH_RING((warp_fwd_kernel<128, NW, false>), dim3((unsigned)grid), dim3(NW * 64), s, A);
    }
    return check_launch("warp_fwd");
}
