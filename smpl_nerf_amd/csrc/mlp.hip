// Fused positional-encoding + RenderRayNet forward on the CDNA4 matrix cores (a1+a2).
//
// Replaces RenderRayNet.forward (models/render_ray_net.py:42-61) together with the encoder calls and
// the torch.cat that feed it in NerfPipeline.forward (models/nerf_pipeline.py:29-39, :49-57).
//
// Design (see mlp_plan.h for the operand algebra):
//   * one wavefront owns 16 samples for the whole network; its activations live in registers in
//     MFMA accumulator layout from the first layer to the [rgb|sigma] store - nothing but the 12 B
//     position, the direction and the 16 B result of a sample ever touches HBM;
//   * v_mfma_f32_16x16x4_f32: exact fp32 products and accumulation (bitwise an fmaf chain), so the
//     result sits at the fp32 round-off floor of the reference's MKL sgemm path (parity 1e-4 on RGB
//     needs this: plain bf16 misses it by 100x).  Peak 157.3 TFLOP/s on MI355X;
//   * the weights of all layers form one contiguous stream of 17 KiB slabs in consumption order
//     (packed once per weight update by mlp_pack_kernel).  The NWAVES waves of a workgroup stream
//     it from L2 through a 3-slot LDS ring: slab p is consumed from LDS while slab p+1 already
//     sits in LDS and slab p+2 is in flight in registers - one workgroup barrier per slab
//     (64 MFMAs per wave);
//   * several independent workgroups per CU (2 waves per SIMD) keep the matrix pipe busy while
//     one of them is at its barrier or evaluating sin/cos;
//   * positional encodings are evaluated in registers, straight into B-operand layout, with
//     full-range sincosf (arguments reach 2^9*|x|); they are recomputed at the skip layer instead of
//     being kept live.
#include "snerf_common.h"
#include "mlp_plan.h"

namespace snerf {

typedef float f4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// weight packing: params_flat (state_dict order) -> slab stream
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mlp_pack_kernel(Plan P, const float *__restrict__ params,
                                                       float *__restrict__ packed) {
    const int slab = blockIdx.x;
    float *dst = packed + (int64_t)slab * SLAB_FLOATS;
    if (slab >= P.total_slabs) {  // zero padding behind the stream
        for (int e = threadIdx.x; e < SLAB_FLOATS; e += 256) dst[e] = 0.f;
        return;
    }
    int li = 0;
    while (li + 1 < P.nlayers && slab >= P.layer[li + 1].first_slab) ++li;
    const Layer &Ly = P.layer[li];
    const int sl = slab - Ly.first_slab;
    const int kps = 16 / Ly.t_out;
    const float *Wm = params + Ly.w_off;
    const float *bias = params + Ly.b_off;
    for (int e = threadIdx.x; e < SLAB_FLOATS; e += 256) {
        float val = 0.f;
        if (e < SLAB_A_FLOATS) {
            const int per_kb = Ly.t_out * 256;
            const int kbl = e / per_kb;
            int rem = e - kbl * per_kb;
            const int to = rem >> 8;
            rem &= 255;
            const int l = rem >> 2, r = rem & 3;
            const int i = l & 15, g = l >> 4;
            const int row = 16 * to + i;
            int kb = sl * kps + kbl;
            if (kbl < kps && kb < Ly.nkb && row < Ly.n_out) {
                int col = -1;
                for (int s = 0; s < Ly.nseg; ++s) {
                    const Seg &sg = Ly.seg[s];
                    if (kb < sg.nkb) {
                        int c;
                        if (sg.type == SEG_PE) {
                            c = pe_slot_col(sg.L, sg.ident, kb, g, r);
                        } else {
                            c = 16 * kb + 4 * g + r;
                            if (c >= sg.ncols) c = -1;
                        }
                        col = c < 0 ? -1 : sg.col_off + c;
                        break;
                    }
                    kb -= sg.nkb;
                }
                if (col >= 0) val = Wm[(int64_t)row * Ly.n_in + col];
            }
        } else if (sl == 0) {
            const int jj = e - SLAB_A_FLOATS;
            if (jj < Ly.n_out) val = bias[jj];
        }
        dst[e] = val;
    }
}

// ------------------------------------------------------------------------------------------------
// forward kernel
// ------------------------------------------------------------------------------------------------
struct FwdArgs {
    const float *packed;
    const float *x;      // [n,3] positions, or x_enc [n, enc_stride] when ENCODED
    const float *dirs;   // [n/spr,3] or [n,3]
    const float *add;    // [n/spr, add_dim] or null
    float *raw;          // [n,4]
    int64_t n;
    int spr;             // samples per ray
    int dirs_per_sample;
    int n_hidden;        // positional_net layers
    unsigned skip_mask;
    int pos_L, pos_id, pos_nkb, pos_dim;
    int dir_L, dir_id, dir_nkb, dir_dim;
    int add_dim, add_nkb;
    int use_dir;
    int enc_stride;
};

// Streams the slab sequence global -> registers -> LDS ring (3 slots).
template <int NT>
struct SlabPipe {
    static constexpr int NA = SLAB_A_FLOATS / 4 / NT;  // f4 per thread in the A region (NT=256: 4, 512: 2)
    const f4 *g;   // this thread's read cursor in the packed stream
    float *ring;
    f4 st[NA], st_aux;
    int tid, rd, wr;

    __device__ __forceinline__ void load() {
#pragma unroll
        for (int i = 0; i < NA; ++i) st[i] = g[i * NT];
        if (tid < 64) st_aux = g[SLAB_A_FLOATS / 4];
        g += SLAB_FLOATS / 4;
    }
    __device__ __forceinline__ void store(int slot) {
        f4 *d = reinterpret_cast<f4 *>(ring + slot * SLAB_FLOATS) + tid;
#pragma unroll
        for (int i = 0; i < NA; ++i) d[i * NT] = st[i];
        if (tid < 64) d[SLAB_A_FLOATS / 4] = st_aux;
    }
    __device__ __forceinline__ void prologue(const float *packed, float *ring_, int tid_) {
        ring = ring_;
        tid = tid_;
        g = reinterpret_cast<const f4 *>(packed) + tid;
        load(); store(0);
        load(); store(1);
        load();
        rd = 0;
        wr = 2;
        __syncthreads();
    }
    __device__ __forceinline__ const float *acquire() const { return ring + rd * SLAB_FLOATS; }
    __device__ __forceinline__ void release() {
        store(wr);
        load();
        __syncthreads();
        rd = rd == 2 ? 0 : rd + 1;
        wr = wr == 2 ? 0 : wr + 1;
    }
};

// per-lane view of the sample this lane works for
struct SampleCtx {
    float px, py, pz;  // position
    float dx, dy, dz;  // normalised direction
    const float *enc;  // row of x_enc (ENCODED) or null
    const float *add;  // row of add or null
    int g;
};

__device__ __forceinline__ void pe_unit(float x, float y, float z, int L, int ident, int p, float &a, float &b) {
    const int nid = ident ? 3 : 0;
    a = 0.f;
    b = 0.f;
    if (p < nid) {
        a = p == 0 ? x : (p == 1 ? y : z);
        return;
    }
    const int pp = p - nid;
    if (pp >= 3 * L) return;
    const int k = pp / 3, c = pp - 3 * k;
    const float v = c == 0 ? x : (c == 1 ? y : z);
    sincosf(ldexpf(v, k), &a, &b);  // 2^k * v is exact: same argument bits as utils.py:127
}

// B operand (4 k-steps) of PE k-block kb for this lane
template <bool ENCODED>
__device__ __forceinline__ f4 pe_operand(const SampleCtx &c, bool is_dir, int L, int ident, int kb, int enc_off) {
    f4 b;
    if (ENCODED) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = pe_slot_col(L, ident, kb, c.g, r);
            b[r] = col >= 0 ? c.enc[enc_off + col] : 0.f;
        }
    } else {
        const float x = is_dir ? c.dx : c.px, y = is_dir ? c.dy : c.py, z = is_dir ? c.dz : c.pz;
        float s0, c0, s1, c1;
        pe_unit(x, y, z, L, ident, 4 * (2 * kb) + c.g, s0, c0);
        pe_unit(x, y, z, L, ident, 4 * (2 * kb + 1) + c.g, s1, c1);
        b = f4{s0, c0, s1, c1};
    }
    return b;
}

__device__ __forceinline__ f4 add_operand(const SampleCtx &c, int add_dim, int kb) {
    f4 b;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int col = 16 * kb + 4 * c.g + r;
        b[r] = col < add_dim ? c.add[col] : 0.f;
    }
    return b;
}

// One k-block: T_OUT x (ds_read_b128 + 4 MFMA).  Tiles are walked in pairs so that consecutive MFMAs
// never share an accumulator (dependent latency of 16x16x4 is 40 cycles vs 32 issue).
template <int T_OUT>
__device__ __forceinline__ void kblock(const float *a_kb, f4 b, f4 (&acc)[T_OUT], int lane) {
    const f4 *ap = reinterpret_cast<const f4 *>(a_kb) + lane;
    if constexpr (T_OUT == 1) {
        const f4 a = ap[0];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc[0], 0, 0, 0);
    } else {
#pragma unroll
        for (int to = 0; to < T_OUT; to += 2) {
            const f4 a0 = ap[to * 64], a1 = ap[(to + 1) * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], b[r], acc[to], 0, 0, 0);
                acc[to + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[r], b[r], acc[to + 1], 0, 0, 0);
            }
        }
    }
}

// Walks the k-blocks of one layer through the slab pipe.
template <int T_OUT, int NT>
struct LayerRun {
    static constexpr int KPS = 16 / T_OUT;
    SlabPipe<NT> &pipe;
    const float *slab;
    int kbl;  // k-block index inside the current slab
    int lane;

    __device__ __forceinline__ LayerRun(SlabPipe<NT> &p, int lane_) : pipe(p), slab(p.acquire()), kbl(0), lane(lane_) {}
    // bias -> accumulator init (aux block of the layer's first slab: bias[16*to + 4*g + r])
    __device__ __forceinline__ void init(f4 (&acc)[T_OUT]) {
        const f4 *aux = reinterpret_cast<const f4 *>(slab + SLAB_A_FLOATS) + (lane >> 4);
#pragma unroll
        for (int to = 0; to < T_OUT; ++to) acc[to] = aux[to * 4];
    }
    __device__ __forceinline__ void step(f4 b, f4 (&acc)[T_OUT]) {
        if (kbl == KPS) {
            pipe.release();
            slab = pipe.acquire();
            kbl = 0;
        }
        kblock<T_OUT>(slab + kbl * (T_OUT * 256), b, acc, lane);
        ++kbl;
    }
    __device__ __forceinline__ void finish() { pipe.release(); }
};

template <int N>
__device__ __forceinline__ void relu_into(f4 (&dst)[N], const f4 (&src)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        dst[i][0] = fmaxf(src[i][0], 0.f);
        dst[i][1] = fmaxf(src[i][1], 0.f);
        dst[i][2] = fmaxf(src[i][2], 0.f);
        dst[i][3] = fmaxf(src[i][3], 0.f);
    }
}
template <int N>
__device__ __forceinline__ void copy_into(f4 (&dst)[N], const f4 (&src)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) dst[i] = src[i];
}

template <int WIDTH, int NWAVES, bool ENCODED>
__global__ __launch_bounds__(NWAVES * 64) void mlp_fwd_kernel(FwdArgs A) {
    constexpr int NT = NWAVES * 64;
    constexpr int T = WIDTH / 16;   // tiles of the trunk
    constexpr int TD = WIDTH / 32;  // tiles of the directional branch
    __shared__ __attribute__((aligned(16))) float ring[3 * SLAB_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t sample = ((int64_t)blockIdx.x * NWAVES + wave) * 16 + (lane & 15);
    const bool valid = sample < A.n;
    const int64_t sc = valid ? sample : A.n - 1;

    SampleCtx c;
    c.g = lane >> 4;
    c.enc = nullptr;
    c.add = nullptr;
    c.px = c.py = c.pz = c.dx = c.dy = c.dz = 0.f;
    if (ENCODED) {
        c.enc = A.x + sc * A.enc_stride;
    } else {
        c.px = A.x[sc * 3 + 0];
        c.py = A.x[sc * 3 + 1];
        c.pz = A.x[sc * 3 + 2];
        const int64_t ray = sc / A.spr;
        if (A.use_dir) {
            const float *dp = A.dirs + (A.dirs_per_sample ? sc : ray) * 3;
            const float ux = dp[0], uy = dp[1], uz = dp[2];
            const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(ux, ux), __fmul_rn(uy, uy)), __fmul_rn(uz, uz)));
            c.dx = __fdiv_rn(ux, nrm);  // models/nerf_pipeline.py:33-34
            c.dy = __fdiv_rn(uy, nrm);
            c.dz = __fdiv_rn(uz, nrm);
        }
        if (A.add_dim) c.add = A.add + ray * A.add_dim;
    }
    const int enc_add_off = A.pos_dim, enc_dir_off = A.enc_stride - A.dir_dim;  // directions = x[..., -dir_dim:] (:43)

    SlabPipe<NT> pipe;
    pipe.prologue(A.packed, ring, tid);

    f4 in[T], acc[T];

    // extra input segments [PE(x) | add] of layer 0 and of the skip layers
    auto pos_segments = [&](LayerRun<T, NT> &run) {
        for (int kb = 0; kb < A.pos_nkb; ++kb) run.step(pe_operand<ENCODED>(c, false, A.pos_L, A.pos_id, kb, 0), acc);
        for (int kb = 0; kb < A.add_nkb; ++kb) {
            f4 b;
            if (ENCODED) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = 16 * kb + 4 * c.g + r;
                    b[r] = col < A.add_dim ? c.enc[enc_add_off + col] : 0.f;
                }
            } else {
                b = add_operand(c, A.add_dim, kb);
            }
            run.step(b, acc);
        }
    };

    {  // positions_pose_input + relu (models/render_ray_net.py:45)
        LayerRun<T, NT> run(pipe, lane);
        run.init(acc);
        pos_segments(run);
        run.finish();
        relu_into(in, acc);
    }
    for (int i = 0; i < A.n_hidden; ++i) {  // positional_net[i] + relu (:46-50)
        LayerRun<T, NT> run(pipe, lane);
        run.init(acc);
#pragma unroll
        for (int kb = 0; kb < T; ++kb) run.step(in[kb], acc);
        if ((A.skip_mask >> i) & 1u) pos_segments(run);
        run.finish();
        relu_into(in, acc);
    }
    {  // additional_linear_layer, no activation (:51)
        LayerRun<T, NT> run(pipe, lane);
        run.init(acc);
#pragma unroll
        for (int kb = 0; kb < T; ++kb) run.step(in[kb], acc);
        run.finish();
        copy_into(in, acc);
    }
    f4 sig[1];
    {  // sigma_out_layer (:52): one padded tile, row 0 is sigma
        LayerRun<1, NT> run(pipe, lane);
        run.init(sig);
#pragma unroll
        for (int kb = 0; kb < T; ++kb) run.step(in[kb], sig);
        run.finish();
    }
    f4 ind[TD], accd[TD];
    {  // directional_input, no activation (:54-57)
        LayerRun<TD, NT> run(pipe, lane);
        run.init(accd);
#pragma unroll
        for (int kb = 0; kb < T; ++kb) run.step(in[kb], accd);
        for (int kb = 0; kb < A.dir_nkb; ++kb)
            run.step(pe_operand<ENCODED>(c, true, A.dir_L, A.dir_id, kb, enc_dir_off), accd);
        run.finish();
        copy_into(ind, accd);
    }
    {  // directional_net[0] + relu (:58-59)
        LayerRun<TD, NT> run(pipe, lane);
        run.init(accd);
#pragma unroll
        for (int kb = 0; kb < TD; ++kb) run.step(ind[kb], accd);
        run.finish();
        relu_into(ind, accd);
    }
    f4 rgb[1];
    {  // rgb_out_layer (:60): rows 0..2
        LayerRun<1, NT> run(pipe, lane);
        run.init(rgb);
#pragma unroll
        for (int kb = 0; kb < TD; ++kb) run.step(ind[kb], rgb);
        run.finish();
    }
    if (valid && c.g == 0) {  // [rgb | sigma] (:61): one 16 B store per sample, 256 B contiguous per wave
        f4 o = f4{rgb[0][0], rgb[0][1], rgb[0][2], sig[0][0]};
        reinterpret_cast<f4 *>(A.raw)[sample] = o;
    }
}

static int fill_args(const snerf_mlp_desc *desc, Plan &P, FwdArgs &A) {
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "mlp: desc is null");
    if (make_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "mlp: %s", why);
    A.n_hidden = P.n_hidden;
    A.skip_mask = desc->skip_mask;
    A.pos_L = desc->pos_freqs;
    A.pos_id = desc->pos_identity ? 1 : 0;
    A.pos_nkb = P.pos_nkb;
    A.pos_dim = P.pos_dim;
    A.dir_L = desc->dir_freqs;
    A.dir_id = desc->dir_identity ? 1 : 0;
    A.dir_nkb = P.dir_nkb;
    A.dir_dim = P.dir_dim;
    A.add_dim = P.add_dim;
    A.add_nkb = P.add_nkb;
    A.use_dir = desc->use_dir ? 1 : 0;
    A.enc_stride = P.pos_dim + P.add_dim + (desc->use_dir ? P.dir_dim : 0);
    return SNERF_OK;
}

constexpr int FWD_WAVES = 4;  // 64 samples per workgroup; several workgroups share a CU

template <bool ENCODED>
static int launch_fwd(const Plan &P, const FwdArgs &A, hipStream_t s) {
    const int64_t tile = FWD_WAVES * 16;
    const int64_t grid = (A.n + tile - 1) / tile;
    if (grid > 0x7fffffffLL) return fail(SNERF_E_BADARG, "mlp_fwd: n too large");
    if (P.width == 256)
        hipLaunchKernelGGL((mlp_fwd_kernel<256, FWD_WAVES, ENCODED>), dim3((unsigned)grid), dim3(FWD_WAVES * 64), 0, s, A);
    else
        hipLaunchKernelGGL((mlp_fwd_kernel<128, FWD_WAVES, ENCODED>), dim3((unsigned)grid), dim3(FWD_WAVES * 64), 0, s, A);
    return check_launch("mlp_fwd");
}

}  // namespace snerf

extern "C" int64_t snerf_mlp_param_floats(const snerf_mlp_desc *desc) {
    using namespace snerf;
    Plan P;
    const char *why;
    if (!desc || make_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "mlp: bad descriptor");
    return P.param_floats;
}

extern "C" int64_t snerf_mlp_packed_floats(const snerf_mlp_desc *desc) {
    using namespace snerf;
    Plan P;
    const char *why;
    if (!desc || make_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "mlp: bad descriptor");
    return (int64_t)(P.total_slabs + SLAB_PAD) * SLAB_FLOATS;
}

extern "C" int snerf_mlp_pack_f32(const snerf_mlp_desc *desc, const float *params_flat, float *packed,
                                  snerf_stream_t stream) {
    using namespace snerf;
    Plan P;
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "mlp_pack: desc is null");
    if (make_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "mlp_pack: %s", why);
    if (!params_flat || !packed) return fail(SNERF_E_BADARG, "mlp_pack: null pointer");
    if (!aligned(packed, 16)) return fail(SNERF_E_ALIGN, "mlp_pack: packed must be 16-byte aligned");
    hipLaunchKernelGGL(mlp_pack_kernel, dim3(P.total_slabs + SLAB_PAD), dim3(256), 0, (hipStream_t)stream, P,
                       params_flat, packed);
    return check_launch("mlp_pack");
}

extern "C" int snerf_mlp_fwd_f32(const snerf_mlp_desc *desc, const float *packed, const float *x, const float *dirs,
                                 int dirs_per_sample, const float *add, int64_t n, int samples_per_ray, float *raw,
                                 snerf_stream_t stream) {
    using namespace snerf;
    Plan P;
    FwdArgs A{};
    int rc = fill_args(desc, P, A);
    if (rc) return rc;
    if (n < 0 || samples_per_ray < 1) return fail(SNERF_E_BADARG, "mlp_fwd: bad n/samples_per_ray");
    if (n == 0) return SNERF_OK;
    if (!packed || !x || !raw) return fail(SNERF_E_BADARG, "mlp_fwd: null pointer");
    if (A.use_dir && !dirs) return fail(SNERF_E_BADARG, "mlp_fwd: dirs is null");
    if (A.add_dim && !add) return fail(SNERF_E_BADARG, "mlp_fwd: add is null");
    if (!aligned(packed, 16) || !aligned(raw, 16)) return fail(SNERF_E_ALIGN, "mlp_fwd: packed/raw must be 16-byte aligned");
    A.packed = packed;
    A.x = x;
    A.dirs = dirs;
    A.add = add;
    A.raw = raw;
    A.n = n;
    A.spr = samples_per_ray;
    A.dirs_per_sample = dirs_per_sample ? 1 : 0;
    return launch_fwd<false>(P, A, (hipStream_t)stream);
}

extern "C" int snerf_mlp_fwd_encoded_f32(const snerf_mlp_desc *desc, const float *packed, const float *x_enc,
                                         int64_t n, int64_t row_floats, float *raw, snerf_stream_t stream) {
    using namespace snerf;
    Plan P;
    FwdArgs A{};
    int rc = fill_args(desc, P, A);
    if (rc) return rc;
    if (n < 0) return fail(SNERF_E_BADARG, "mlp_fwd_encoded: bad n");
    if (n == 0) return SNERF_OK;
    if (!packed || !x_enc || !raw) return fail(SNERF_E_BADARG, "mlp_fwd_encoded: null pointer");
    if (row_floats < A.pos_dim + A.add_dim || row_floats < A.dir_dim || row_floats > 0x7fffffff)
        return fail(SNERF_E_BADARG, "mlp_fwd_encoded: row of %lld floats is too short for this network", (long long)row_floats);
    A.enc_stride = (int)row_floats;
    if (!aligned(packed, 16) || !aligned(raw, 16)) return fail(SNERF_E_ALIGN, "mlp_fwd_encoded: packed/raw must be 16-byte aligned");
    A.packed = packed;
    A.x = x_enc;
    A.raw = raw;
    A.n = n;
    A.spr = 1;
    return launch_fwd<true>(P, A, (hipStream_t)stream);
}
