// Split-bf16 variant of the fused positional-encoding + RenderRayNet forward (inference path).
//
// fp32-input MFMA runs at 1/16 of the bf16 MFMA rate on CDNA4.  This kernel keeps fp32-class accuracy on the
// bf16 matrix cores by splitting every fp32 operand into NS bf16 parts (x = p0 + p1 [+ p2], p_k = bf16(x -
// p_0 - .. - p_{k-1}), round-to-nearest-even) and summing the cross terms in fp32 accumulators:
//     NS = 3 ("bf16x6"): p0q0 + p0q1 + p1q0 + p1q1 + p0q2 + p2q0   -> relative error ~2^-24 per product, the
//                        rendered RGB sits at the fp32 round-off floor (4e-6 on the bench frame): parity mode
//     NS = 2 ("bf16x3"): p0q0 + p0q1 + p1q0                         -> ~2^-16 per product, RGB within 8e-5
// Products of bf16 values are exact in the fp32 accumulator.  6 (3) v_mfma_f32_16x16x32_bf16 replace the 8
// v_mfma_f32_16x16x4_f32 of a 16x16x32 block: 2.7x (5.3x) less matrix-pipe time.
//
// Everything else is the fp32 kernel's design (mlp.hip, mlp_plan.h) with 32-wide k-blocks: one wave owns 16
// samples, the accumulator layout of a layer is the B-operand layout of the next (k-block b = accumulators
// of tiles 2b, 2b+1), activations stay in registers as NS packed-bf16 B operands (96 VGPRs for 256 features
// at NS = 3), weights are pre-split and stream L2 -> registers -> 3-slot LDS ring (one 48 KiB slab = one
// k-block x 16 output tiles x 3 parts), positional encodings are evaluated in registers.
#include <stdlib.h>

#include "mlp_device.h"

namespace snerf {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__host__ __device__ constexpr int slab16_bytes(int ns) { return ns * 16384 + 1024; }

// ------------------------------------------------------------------------------------------------
// weight packing: params_flat -> split-bf16 slab stream
// slab = [k-block in slab][output tile][part][lane][8 bf16] then 256 fp32 of bias
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mlp_pack_bf16_kernel(Plan P, int NS, const float *__restrict__ params,
                                                            unsigned char *__restrict__ packed) {
    const int slab = blockIdx.x;
    const int SB = slab16_bytes(NS);
    unsigned char *dst = packed + (int64_t)slab * SB;
    if (slab >= P.total_slabs) {
        for (int e = threadIdx.x; e < SB / 4; e += 256) reinterpret_cast<float *>(dst)[e] = 0.f;
        return;
    }
    int li = 0;
    while (li + 1 < P.nlayers && slab >= P.layer[li + 1].first_slab) ++li;
    const Layer &Ly = P.layer[li];
    const int sl = slab - Ly.first_slab;
    const int kps = 16 / Ly.t_out;
    const float *Wm = params + Ly.w_off;
    __bf16 *a = reinterpret_cast<__bf16 *>(dst);
    const int per_kb = Ly.t_out * NS * 512;
    for (int q = threadIdx.x; q < NS * 8192; q += 256) {
        const int kbl = q / per_kb;
        int rem = q - kbl * per_kb;
        const int to = rem / (NS * 512);
        rem -= to * NS * 512;
        const int s = rem >> 9;
        rem &= 511;
        const int lane = rem >> 3, e = rem & 7;
        const int row = 16 * to + (lane & 15), g = lane >> 4;
        const int kb = sl * kps + kbl;
        float w = 0.f;
        if (kb < Ly.nkb && row < Ly.n_out) {
            const int col = slot_to_col32(Ly, kb, g, e);
            if (col >= 0) w = Wm[(int64_t)row * Ly.n_in + col];
        }
        __bf16 h = (__bf16)w;
        for (int t = 0; t < s; ++t) {
            w = w - (float)h;
            h = (__bf16)w;
        }
        a[q] = h;
    }
    float *aux = reinterpret_cast<float *>(dst + NS * 16384);
    for (int j = threadIdx.x; j < 256; j += 256) aux[j] = (sl == 0 && j < Ly.n_out) ? params[Ly.b_off + j] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// slab pipe with a run-time slab size (dynamic LDS)
// ------------------------------------------------------------------------------------------------
template <int NT, int NS>
struct SlabPipe16 {
    static constexpr int SB = NS * 16384 + 1024;
    static constexpr int NA = NS * 16384 / 16 / NT;
    const f4 *g;
    char *ring;
    f4 st[NA], st_aux;
    int tid, rd, wr;
    __device__ __forceinline__ void load() {
#pragma unroll
        for (int i = 0; i < NA; ++i) st[i] = g[i * NT];
        if (tid < 64) st_aux = g[NS * 1024];
        g += SB / 16;
    }
    __device__ __forceinline__ void store(int slot) {
        f4 *d = reinterpret_cast<f4 *>(ring + slot * SB) + tid;
#pragma unroll
        for (int i = 0; i < NA; ++i) d[i * NT] = st[i];
        if (tid < 64) d[NS * 1024] = st_aux;
    }
    __device__ __forceinline__ void prologue(const void *packed, char *ring_, int tid_) {
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
    __device__ __forceinline__ const char *acquire() const { return ring + rd * SB; }
    __device__ __forceinline__ void release() {
        store(wr);
        load();
        __syncthreads();
        rd = rd == 2 ? 0 : rd + 1;
        wr = wr == 2 ? 0 : wr + 1;
    }
};

// cross terms (A part, B part), smallest first
template <int NS> struct Terms;
template <> struct Terms<2> {
    static constexpr int N = 3;
    static constexpr int A[3] = {1, 0, 0};
    static constexpr int B[3] = {0, 1, 0};
};
template <> struct Terms<3> {
    static constexpr int N = 6;
    static constexpr int A[6] = {2, 0, 1, 1, 0, 0};
    static constexpr int B[6] = {0, 2, 1, 0, 1, 0};
};

// One 32-wide k-block: per output tile NS ds_read_b128 + Terms<NS>::N MFMAs; tiles in pairs so consecutive
// MFMAs alternate accumulators; the A parts of the next pair are read while this pair's MFMAs issue.
template <int T_OUT, int NS>
__device__ __forceinline__ void kblock16(const char *a_kb, const bf8 (&b)[NS], f4 (&acc)[T_OUT], int lane) {
    const bf8 *ap = reinterpret_cast<const bf8 *>(a_kb) + lane;  // [(to*NS + s)*64]
    using Tm = Terms<NS>;
    if constexpr (T_OUT == 1) {
        bf8 a[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) a[s] = ap[s * 64];
#pragma unroll
        for (int t = 0; t < Tm::N; ++t) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[Tm::A[t]], b[Tm::B[t]], acc[0], 0, 0, 0);
    } else {
        bf8 a0[NS], a1[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            a0[s] = ap[s * 64];
            a1[s] = ap[(NS + s) * 64];
        }
#pragma unroll
        for (int to = 0; to < T_OUT; to += 2) {
            bf8 n0[NS], n1[NS];
            if (to + 2 < T_OUT) {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    n0[s] = ap[((to + 2) * NS + s) * 64];
                    n1[s] = ap[((to + 3) * NS + s) * 64];
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * NS, 0);
            }
#pragma unroll
            for (int t = 0; t < Tm::N; ++t) {
                acc[to] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[Tm::A[t]], b[Tm::B[t]], acc[to], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                acc[to + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[Tm::A[t]], b[Tm::B[t]], acc[to + 1], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (to + 2 < T_OUT) {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    a0[s] = n0[s];
                    a1[s] = n1[s];
                }
            }
        }
    }
}

template <int T_OUT, int NT, int NS>
struct LayerRun16 {
    static constexpr int KPS = 16 / T_OUT;
    SlabPipe16<NT, NS> &pipe;
    const char *slab;
    int kbl, lane;
    __device__ __forceinline__ LayerRun16(SlabPipe16<NT, NS> &p, int lane_) : pipe(p), slab(p.acquire()), kbl(0), lane(lane_) {}
    __device__ __forceinline__ void init(f4 (&acc)[T_OUT]) {
        const f4 *aux = reinterpret_cast<const f4 *>(slab + NS * 16384) + (lane >> 4);
#pragma unroll
        for (int to = 0; to < T_OUT; ++to) acc[to] = aux[to * 4];
    }
    __device__ __forceinline__ void step(const bf8 (&b)[NS], f4 (&acc)[T_OUT]) {
        if (kbl == KPS) {
            pipe.release();
            slab = pipe.acquire();
            kbl = 0;
        }
        kblock16<T_OUT, NS>(slab + kbl * (T_OUT * NS * 1024), b, acc, lane);
        ++kbl;
    }
    __device__ __forceinline__ void finish() { pipe.release(); }
};

// fp32 -> NS bf16 parts, element e of the packed B operand
template <int NS>
__device__ __forceinline__ void split_into(float v, bf8 (&dst)[NS], int e) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const __bf16 h = (__bf16)v;
        dst[s][e] = h;
        v = v - (float)h;
    }
}

// accumulators of N tiles (optionally through ReLU) -> B operands of N/2 k-blocks
template <int N, int NS, bool RELU>
__device__ __forceinline__ void pack_acts(const f4 (&acc)[N], bf8 (&bin)[N / 2][NS]) {
#pragma unroll
    for (int t = 0; t < N; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) split_into<NS>(RELU ? fmaxf(acc[t][r], 0.f) : acc[t][r], bin[t >> 1], 4 * (t & 1) + r);
}

// B operand of encoder k-block kb: 4 units (sin, cos pairs) per lane
template <int NS>
__device__ __forceinline__ void pe_operand16(const SampleCtx &c, bool is_dir, int L, int ident, int kb, bf8 (&b)[NS]) {
    const float x = is_dir ? c.dx : c.px, y = is_dir ? c.dy : c.py, z = is_dir ? c.dz : c.pz;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float s0, c0;
        pe_unit(x, y, z, L, ident, 4 * (4 * kb + u) + c.g, s0, c0);
        split_into<NS>(s0, b, 2 * u);
        split_into<NS>(c0, b, 2 * u + 1);
    }
}
template <int NS>
__device__ __forceinline__ void add_operand16(const SampleCtx &c, int add_dim, int kb, bf8 (&b)[NS]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int col = 32 * kb + 16 * (e >> 2) + 4 * c.g + (e & 3);
        split_into<NS>(col < add_dim ? c.add[col] : 0.f, b, e);
    }
}

template <int WIDTH, int NWAVES, int NS>
__global__ __launch_bounds__(NWAVES * 64) void mlp_fwd_bf16_kernel(FwdArgs A) {
    constexpr int NT = NWAVES * 64;
    constexpr int T = WIDTH / 16, TD = WIDTH / 32;
    constexpr int KB = T / 2, KBD = TD / 2;
    extern __shared__ __attribute__((aligned(16))) char ring[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t sample = ((int64_t)blockIdx.x * NWAVES + wave) * 16 + (lane & 15);
    const bool valid = sample < A.n;
    const int64_t sc = valid ? sample : A.n - 1;
    SampleCtx c;
    c.g = lane >> 4;
    c.enc = nullptr;
    c.add = nullptr;
    c.px = A.x[sc * 3 + 0];
    c.py = A.x[sc * 3 + 1];
    c.pz = A.x[sc * 3 + 2];
    c.dx = c.dy = c.dz = 0.f;
    const int64_t ray = sc / A.spr;
    if (A.use_dir) {
        const float *dp = A.dirs + (A.dirs_per_sample ? sc : ray) * 3;
        const float ux = dp[0], uy = dp[1], uz = dp[2];
        const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(ux, ux), __fmul_rn(uy, uy)), __fmul_rn(uz, uz)));
        c.dx = __fdiv_rn(ux, nrm);
        c.dy = __fdiv_rn(uy, nrm);
        c.dz = __fdiv_rn(uz, nrm);
    }
    if (A.add_dim) c.add = A.add + ray * A.add_dim;

    SlabPipe16<NT, NS> pipe;
    pipe.prologue(A.packed, ring, tid);

    bf8 bin[KB][NS];
    f4 acc[T];
    auto pe_segment = [&](LayerRun16<T, NT, NS> &run) {
        for (int kb = 0; kb < A.pos_nkb; ++kb) {
            bf8 b[NS];
            pe_operand16<NS>(c, false, A.pos_L, A.pos_id, kb, b);
            run.step(b, acc);
        }
    };
    auto add_segment = [&](LayerRun16<T, NT, NS> &run) {
        for (int kb = 0; kb < A.add_nkb; ++kb) {
            bf8 b[NS];
            add_operand16<NS>(c, A.add_dim, kb, b);
            run.step(b, acc);
        }
    };
    auto pos_segments = [&](LayerRun16<T, NT, NS> &run) {
        if (A.add_first) add_segment(run);
        pe_segment(run);
        if (!A.add_first) add_segment(run);
    };
    {  // positions_pose_input + relu
        LayerRun16<T, NT, NS> run(pipe, lane);
        run.init(acc);
        pos_segments(run);
        run.finish();
        pack_acts<T, NS, true>(acc, bin);
    }
    for (int i = 0; i < A.n_hidden; ++i) {  // positional_net[i] + relu
        LayerRun16<T, NT, NS> run(pipe, lane);
        run.init(acc);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) run.step(bin[kb], acc);
        if ((A.skip_mask >> i) & 1u) pos_segments(run);
        run.finish();
        pack_acts<T, NS, true>(acc, bin);
    }
    {  // additional_linear_layer (no activation)
        LayerRun16<T, NT, NS> run(pipe, lane);
        run.init(acc);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) run.step(bin[kb], acc);
        run.finish();
        pack_acts<T, NS, false>(acc, bin);
    }
    f4 sig[1];
    {
        LayerRun16<1, NT, NS> run(pipe, lane);
        run.init(sig);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) run.step(bin[kb], sig);
        run.finish();
    }
    bf8 bind[KBD][NS];
    f4 accd[TD];
    {  // directional_input (no activation)
        LayerRun16<TD, NT, NS> run(pipe, lane);
        run.init(accd);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) run.step(bin[kb], accd);
        for (int kb = 0; kb < A.dir_nkb; ++kb) {
            bf8 b[NS];
            pe_operand16<NS>(c, true, A.dir_L, A.dir_id, kb, b);
            run.step(b, accd);
        }
        run.finish();
        pack_acts<TD, NS, false>(accd, bind);
    }
    {  // directional_net[0] + relu
        LayerRun16<TD, NT, NS> run(pipe, lane);
        run.init(accd);
#pragma unroll
        for (int kb = 0; kb < KBD; ++kb) run.step(bind[kb], accd);
        run.finish();
        pack_acts<TD, NS, true>(accd, bind);
    }
    f4 rgb[1];
    {
        LayerRun16<1, NT, NS> run(pipe, lane);
        run.init(rgb);
#pragma unroll
        for (int kb = 0; kb < KBD; ++kb) run.step(bind[kb], rgb);
        run.finish();
    }
    if (valid && c.g == 0) reinterpret_cast<f4 *>(A.raw)[sample] = f4{rgb[0][0], rgb[0][1], rgb[0][2], sig[0][0]};
}

static int plan16(const snerf_mlp_desc *desc, Plan &P, const char *what) {
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "%s: desc is null", what);
    if (make_plan(*desc, P, why, 32) != 0) return fail(SNERF_E_BADARG, "%s: %s", what, why);
    if (P.width != 256) return fail(SNERF_E_BADARG, "%s: the split-bf16 path supports width 256 only", what);
    return SNERF_OK;
}

template <int NS>
static int launch_bf16(const FwdArgs &A, hipStream_t s) {
    constexpr int NW = 8;
    const int lds = 3 * slab16_bytes(NS);
    static bool attr = false;  // idempotent; a race only repeats the call
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_fwd_bf16_kernel<256, NW, NS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return fail(SNERF_E_LAUNCH, "mlp_fwd_bf16: cannot raise the dynamic LDS limit to %d bytes", lds);
        attr = true;
    }
    const int64_t grid = (A.n + NW * 16 - 1) / (NW * 16);
    if (grid > 0x7fffffffLL) return fail(SNERF_E_BADARG, "mlp_fwd_bf16: n too large");
    hipLaunchKernelGGL((mlp_fwd_bf16_kernel<256, NW, NS>), dim3((unsigned)grid), dim3(NW * 64), lds, s, A);
    return check_launch("mlp_fwd_bf16");
}

}  // namespace snerf

extern "C" int64_t snerf_mlp_packed_bf16_bytes(const snerf_mlp_desc *desc, int nsplit) {
    using namespace snerf;
    Plan P;
    if (nsplit != 2 && nsplit != 3) return fail(SNERF_E_BADARG, "mlp_packed_bf16_bytes: nsplit must be 2 or 3");
    int rc = plan16(desc, P, "mlp_packed_bf16_bytes");
    if (rc) return rc;
    return (int64_t)(P.total_slabs + SLAB_PAD) * slab16_bytes(nsplit);
}

extern "C" int snerf_mlp_pack_bf16(const snerf_mlp_desc *desc, const float *params_flat, void *packed, int nsplit,
                                   snerf_stream_t stream) {
    using namespace snerf;
    Plan P;
    if (nsplit != 2 && nsplit != 3) return fail(SNERF_E_BADARG, "mlp_pack_bf16: nsplit must be 2 or 3");
    int rc = plan16(desc, P, "mlp_pack_bf16");
    if (rc) return rc;
    if (!params_flat || !packed) return fail(SNERF_E_BADARG, "mlp_pack_bf16: null pointer");
    if (!aligned(packed, 16)) return fail(SNERF_E_ALIGN, "mlp_pack_bf16: packed must be 16-byte aligned");
    hipLaunchKernelGGL(mlp_pack_bf16_kernel, dim3(P.total_slabs + SLAB_PAD), dim3(256), 0, (hipStream_t)stream, P, nsplit,
                       params_flat, reinterpret_cast<unsigned char *>(packed));
    return check_launch("mlp_pack_bf16");
}

extern "C" int snerf_mlp_fwd_bf16_f32(const snerf_mlp_desc *desc, const void *packed, int nsplit, const float *x,
                                      const float *dirs, int dirs_per_sample, const float *add, int64_t n,
                                      int samples_per_ray, float *raw, snerf_stream_t stream) {
    using namespace snerf;
    Plan P;
    if (nsplit != 2 && nsplit != 3) return fail(SNERF_E_BADARG, "mlp_fwd_bf16: nsplit must be 2 or 3");
    int rc = plan16(desc, P, "mlp_fwd_bf16");
    if (rc) return rc;
    if (n < 0 || samples_per_ray < 1) return fail(SNERF_E_BADARG, "mlp_fwd_bf16: bad n/samples_per_ray");
    if (n == 0) return SNERF_OK;
    if (!packed || !x || !raw) return fail(SNERF_E_BADARG, "mlp_fwd_bf16: null pointer");
    if (desc->use_dir && !dirs) return fail(SNERF_E_BADARG, "mlp_fwd_bf16: dirs is null");
    if (P.add_dim && !add) return fail(SNERF_E_BADARG, "mlp_fwd_bf16: add is null");
    if (!aligned(packed, 16) || !aligned(raw, 16)) return fail(SNERF_E_ALIGN, "mlp_fwd_bf16: packed/raw must be 16-byte aligned");
    FwdArgs A{};
    A.packed = reinterpret_cast<const float *>(packed);
    A.x = x;
    A.dirs = dirs;
    A.add = add;
    A.raw = raw;
    A.n = n;
    A.spr = samples_per_ray;
    A.dirs_per_sample = dirs_per_sample ? 1 : 0;
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
    A.add_first = (P.add_dim && desc->add_first) ? 1 : 0;
    A.use_dir = desc->use_dir ? 1 : 0;
    if (nsplit == 3) return launch_bf16<3>(A, (hipStream_t)stream);
    return launch_bf16<2>(A, (hipStream_t)stream);
}
