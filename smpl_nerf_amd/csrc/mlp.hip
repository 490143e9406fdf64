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
#include <stdlib.h>

#include "mlp_device.h"

namespace snerf {

// ------------------------------------------------------------------------------------------------
// weight packing: params_flat (state_dict order) -> slab stream
// ------------------------------------------------------------------------------------------------
// PACK_PARTS workgroups per slab, one element per thread (r06: with one workgroup per slab every thread walked 33 elements of
// index arithmetic - 29 us for the warp net's 8 slabs, on the critical path of every smpl_nerf training step, whose warp
// streams are re-packed behind Adam)
constexpr int PACK_PARTS = (SLAB_FLOATS + 255) / 256;
__global__ __launch_bounds__(256) void mlp_pack_kernel(Plan P, const float *__restrict__ params,
                                                       float *__restrict__ packed) {
    const int slab = blockIdx.x / PACK_PARTS;
    const int e = (blockIdx.x - slab * PACK_PARTS) * 256 + threadIdx.x;
    if (e >= SLAB_FLOATS) return;
    float *dst = packed + (int64_t)slab * SLAB_FLOATS;
    if (slab >= P.total_slabs) {  // zero padding behind the stream
        dst[e] = 0.f;
        return;
    }
    int li = 0;
    while (li + 1 < P.nlayers && slab >= P.layer[li + 1].first_slab) ++li;
    const Layer &Ly = P.layer[li];
    const int64_t src = fwd_slab_src(Ly, slab - Ly.first_slab, e);
    dst[e] = src >= 0 ? params[src] : 0.f;
}

// TRAIN additionally stores every layer input (post-activation) for the backward kernels.
// FOLD (inference, per-ray additional inputs): the additional columns of layer 0 and of the skip layers are per-RAY
// constants (the pipelines expand one row per ray over its samples: models/append_smpl_params_pipeline.py,
// append_to_nerf_pipeline.py, append_vertices_pipeline.py), so W_add . add is evaluated once per ray by
// mlp_add_fold_kernel and added to the accumulators here; their k-blocks of the stream are skipped, not multiplied.
template <int WIDTH, int NWAVES, bool ENCODED, bool TRAIN, bool FOLD>
__device__ __forceinline__ void mlp_fwd_body(const FwdArgs &A) {
    constexpr int NT = NWAVES * 64;
    constexpr int T = WIDTH / 16;   // tiles of the trunk
    constexpr int TD = WIDTH / 32;  // tiles of the directional branch
    extern __shared__ __attribute__((aligned(16))) float ring[];  // RING4_BYTES (DMA pipe: SNERF_LAUNCH_RING4) or RING_BYTES (SNERF_LAUNCH_RING)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int enc_pos_off = A.add_first ? A.add_dim : 0, enc_add_off = A.add_first ? 0 : A.pos_dim;
    const int enc_dir_off = A.enc_stride - A.dir_dim;  // directions = x[..., -dir_dim:] (:43)

    // Persistent workgroups: one per CU (the 99 KiB ring allows no more), each walking the sample tiles blockIdx.x,
    // blockIdx.x + gridDim.x, ...  The weight ring keeps rolling from one tile into the next (the stream wraps around),
    // so only the first tile of a workgroup pays the pipeline fill and no CU idles between two workgroups.
    using Pipe = std::conditional_t<(!TRAIN && WIDTH == 256 && NWAVES == 8), SlabPipeDma<NT>, PipeFor<WIDTH, NT>>;
    Pipe pipe;
    // raw inputs of a tile (positions, direction), fetched while the previous tile's last layers run so that a tile
    // never starts by waiting on HBM (inference variant; the training variant sits at the register limit)
    // The training variant (at the register limit: the persistent loop would spill) runs one workgroup per tile.
    constexpr bool PERSIST = !TRAIN;
    constexpr bool PREFETCH = !ENCODED && PERSIST && !TRAIN;
    float raw_in[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto load_raw = [&](int64_t t) __attribute__((always_inline)) {
        const int64_t s0 = (t * NWAVES + wave) * 16 + (lane & 15);
        const int64_t s1 = s0 < A.n ? s0 : A.n - 1;
        raw_in[0] = A.x[s1 * 3 + 0];
        raw_in[1] = A.x[s1 * 3 + 1];
        raw_in[2] = A.x[s1 * 3 + 2];
        if (A.use_dir) {
            const float *dp = A.dirs + (A.dirs_per_sample ? s1 : s1 / A.spr) * 3;
            raw_in[3] = dp[0], raw_in[4] = dp[1], raw_in[5] = dp[2];
        }
    };
    if (PREFETCH && blockIdx.x < A.n_tiles) load_raw(blockIdx.x);
    int64_t tile = blockIdx.x;
    do {
    const int64_t sample = (tile * NWAVES + wave) * 16 + (lane & 15);
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
        if (!PREFETCH) load_raw(tile);
        c.px = raw_in[0];
        c.py = raw_in[1];
        c.pz = raw_in[2];
        const int64_t ray = sc / A.spr;
        if (A.use_dir) {
            const float ux = raw_in[3], uy = raw_in[4], uz = raw_in[5];
            const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(ux, ux), __fmul_rn(uy, uy)), __fmul_rn(uz, uz)));
            c.dx = __fdiv_rn(ux, nrm);  // models/nerf_pipeline.py:33-34
            c.dy = __fdiv_rn(uy, nrm);
            c.dz = __fdiv_rn(uz, nrm);
        }
        if (A.add_dim) c.add = A.add + ray * A.add_dim;
    }
    if (tile == blockIdx.x) pipe.prologue(A.packed, ring, tid, PERSIST ? A.total_slabs : 0x7fffffff);

    f4 in[T], acc[T];

    // extra input segments [PE(x) | add] of layer 0 and of the skip layers
    auto pe_segment = [&](LayerRun<T, NT, Pipe> &run, bool first) {
        for (int kb = 0; kb < A.pos_nkb; ++kb) {
            const f4 b = pe_operand<ENCODED>(c, false, A.pos_L, A.pos_id, kb, enc_pos_off);
            if (TRAIN && first && valid) store_tile(A.act, A.act_pe + kb, A.n, sample, c.g, b);
            run.step(b, acc);
        }
    };
    int fold_slot = 0;
    auto add_segment = [&](LayerRun<T, NT, Pipe> &run, bool first) {
        if (FOLD) {
            for (int kb = 0; kb < A.add_nkb; ++kb) run.skip();
            const f4 *row = reinterpret_cast<const f4 *>(A.fold + ((sc / A.spr) * A.fold_slots + fold_slot) * WIDTH) + c.g;
#pragma unroll
            for (int to = 0; to < T; ++to) {
                const f4 v = row[to * 4];
                acc[to][0] += v[0], acc[to][1] += v[1], acc[to][2] += v[2], acc[to][3] += v[3];
            }
            ++fold_slot;
            return;
        }
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
            if (TRAIN && first && valid) store_tile(A.act, A.act_add + kb, A.n, sample, c.g, b);
            run.step(b, acc);
        }
    };
    // extra input segments of layer 0 and of the skip layers, in the column order of the weight matrix
    auto pos_segments = [&](LayerRun<T, NT, Pipe> &run, bool first) {
        if (A.add_first) add_segment(run, first);
        pe_segment(run, first);
        if (!A.add_first) add_segment(run, first);
    };

    {  // positions_pose_input + relu (models/render_ray_net.py:45)
        LayerRun<T, NT, Pipe> run(pipe, lane);
        run.init(acc);
        pos_segments(run, true);
        run.finish();
        relu_into(in, acc);
        if (TRAIN && valid) {
            store_tiles(A.act, A.act_x1, A.n, sample, c.g, in);
            store_mask(A.act, A.act_mask, 0, A.n, sample, c.g, in);
        }
    }
    for (int i = 0; i < A.n_hidden; ++i) {  // positional_net[i] + relu (:46-50)
        LayerRun<T, NT, Pipe> run(pipe, lane);
        run.init(acc);
#pragma unroll
        for (int kb = 0; kb < T; ++kb) run.step(in[kb], acc);
        if ((A.skip_mask >> i) & 1u) pos_segments(run, false);
        run.finish();
        relu_into(in, acc);
        if (TRAIN && valid) {
            store_tiles(A.act, A.act_x1 + (i + 1) * T, A.n, sample, c.g, in);
            store_mask(A.act, A.act_mask, i + 1, A.n, sample, c.g, in);
        }
    }
    {  // additional_linear_layer, no activation (:51)
        LayerRun<T, NT, Pipe> run(pipe, lane);
        run.init(acc);
#pragma unroll
        for (int kb = 0; kb < T; ++kb) run.step(in[kb], acc);
        run.finish();
        copy_into(in, acc);
        if (TRAIN && valid) store_tiles(A.act, A.act_o, A.n, sample, c.g, in);
    }
    f4 sig[1];
    {  // sigma_out_layer (:52): one padded tile, row 0 is sigma
        LayerRun<1, NT, Pipe> run(pipe, lane);
        run.init(sig);
#pragma unroll
        for (int kb = 0; kb < T; ++kb) run.step(in[kb], sig);
        run.finish();
    }
    f4 ind[TD], accd[TD];
    {  // directional_input, no activation (:54-57)
        LayerRun<TD, NT, Pipe> run(pipe, lane);
        run.init(accd);
#pragma unroll
        for (int kb = 0; kb < T; ++kb) run.step(in[kb], accd);
        for (int kb = 0; kb < A.dir_nkb; ++kb) {
            const f4 b = pe_operand<ENCODED>(c, true, A.dir_L, A.dir_id, kb, enc_dir_off);
            if (TRAIN && valid) store_tile(A.act, A.act_dpe + kb, A.n, sample, c.g, b);
            run.step(b, accd);
        }
        run.finish();
        copy_into(ind, accd);
        if (TRAIN && valid) store_tiles(A.act, A.act_h1, A.n, sample, c.g, ind);
        if (PREFETCH && tile + gridDim.x < A.n_tiles) load_raw(tile + gridDim.x);   // lands behind the last two layers
    }
    {  // directional_net[0] + relu (:58-59)
        LayerRun<TD, NT, Pipe> run(pipe, lane);
        run.init(accd);
#pragma unroll
        for (int kb = 0; kb < TD; ++kb) run.step(ind[kb], accd);
        run.finish();
        relu_into(ind, accd);
        if (TRAIN && valid) {
            store_tiles(A.act, A.act_h2, A.n, sample, c.g, ind);
            store_mask<TD, (T > 16)>(A.act, A.act_mask, A.n_hidden + 1, A.n, sample, c.g, ind);
        }
    }
    f4 rgb[1];
    {  // rgb_out_layer (:60): rows 0..2
        LayerRun<1, NT, Pipe> run(pipe, lane);
        run.init(rgb);
#pragma unroll
        for (int kb = 0; kb < TD; ++kb) run.step(ind[kb], rgb);
        run.finish();
    }
    if (valid && c.g == 0) {  // [rgb | sigma] (:61): one 16 B store per sample, 256 B contiguous per wave
        f4 o = f4{rgb[0][0], rgb[0][1], rgb[0][2], sig[0][0]};
        reinterpret_cast<f4 *>(A.raw)[sample] = o;
    }
    } while (PERSIST && (tile += gridDim.x) < A.n_tiles);
    pipe.drain();
}

// the kernels: the body above with and without the per-ray fold (two kernel names: the profiles of the rounds key on the
// first one's)
template <int WIDTH, int NWAVES, bool ENCODED, bool TRAIN>
__global__ __launch_bounds__(NWAVES * 64) void mlp_fwd_kernel(FwdArgs A) {
    mlp_fwd_body<WIDTH, NWAVES, ENCODED, TRAIN, false>(A);
}
template <int WIDTH, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void mlp_fwd_fold_kernel(FwdArgs A) {
    mlp_fwd_body<WIDTH, NWAVES, false, false, true>(A);
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
    A.add_first = (P.add_dim && desc->add_first) ? 1 : 0;
    A.use_dir = desc->use_dir ? 1 : 0;
    A.enc_stride = P.pos_dim + P.add_dim + (desc->use_dir ? P.dir_dim : 0);
    return SNERF_OK;
}

int launch_pack(const Plan &P, const float *params_flat, float *packed, hipStream_t s, const char *what) {
    hipLaunchKernelGGL(mlp_pack_kernel, dim3((P.total_slabs + SLAB_PAD) * PACK_PARTS), dim3(256), 0, s, P, params_flat, packed);
    return check_launch(what);
}

constexpr int FWD_WAVES = 8;  // 128 samples per workgroup, one workgroup per CU (2 waves per SIMD)

// FOLD pre-pass: out[(ray * slots + slot) * W + o] = sum_c W_l[o, additional column c] * add[ray][c], l = layer 0 and the
// skip layers in order, read from the packed stream (A tile (kb, to) of a layer: lane m + 16 g, value r  <->
// W[16 to + m][column of slot (kb, g, r)]).  One thread per (ray, slot, output feature); W = 16 * t_out of the trunk.
__global__ __launch_bounds__(256) void mlp_add_fold_kernel(Plan P, const float *__restrict__ packed, const float *__restrict__ add,
                                                           int64_t n_rays, int slots, float *__restrict__ out) {
    const int W = P.layer[0].t_out * 16;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_rays * slots * W) return;
    const int o = (int)(e % W), slot = (int)((e / W) % slots);
    const int64_t ray = e / ((int64_t)W * slots);
    int l = 0;   // the slot-th layer with an additional-input segment
    for (int seen = 0; l < P.nlayers; ++l) {
        bool has = false;
        for (int sg = 0; sg < P.layer[l].nseg; ++sg) has = has || P.layer[l].seg[sg].type == SEG_ADD;
        if (has && seen++ == slot) break;
    }
    const Layer &Ly = P.layer[l];
    int kb0 = 0, sg = 0;
    while (Ly.seg[sg].type != SEG_ADD) kb0 += Ly.seg[sg++].nkb;
    const int to = o >> 4, m = o & 15, T = Ly.t_out;
    const float *a = add + ray * P.add_dim;
    float sum = 0.f;
    for (int kb = 0; kb < Ly.seg[sg].nkb; ++kb) {
        const int t = (kb0 + kb) * T + to;
        const float *tile = packed + (int64_t)(Ly.first_slab + t / SLAB_TILES) * SLAB_FLOATS + (t % SLAB_TILES) * 256;
        for (int g = 0; g < 4; ++g) {
            const f4 w = *reinterpret_cast<const f4 *>(tile + (m + 16 * g) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * kb + 4 * g + r;
                if (c < P.add_dim) sum += w[r] * a[c];
            }
        }
    }
    out[e] = sum;
}

#define SNERF_LAUNCH_WIDE SNERF_LAUNCH_RING4
template <int NW, bool ENCODED, bool TRAIN, bool FOLD = false>
static int launch_fwd_nw(const Plan &P, const FwdArgs &A, hipStream_t s, int64_t n_limit = -1) {
    const int64_t tile = NW * 16;
    FwdArgs B = A;
    B.n_tiles = ((n_limit >= 0 ? n_limit : A.n) + tile - 1) / tile;   // (n_limit: only the first n_limit samples - the others are another launch's)
    B.total_slabs = P.total_slabs;
    if (B.n_tiles > 0x7fffffffLL) return fail(SNERF_E_BADARG, "mlp_fwd: n too large");
    const int n_cu = device_cu_count("mlp_fwd");  // one persistent workgroup per CU
    if (n_cu < 1) return n_cu;
    const int64_t grid = (!TRAIN && B.n_tiles > n_cu) ? n_cu : B.n_tiles;
    if constexpr (FOLD) {
        if (P.width == 256) {
            if constexpr (NW == 8) SNERF_LAUNCH_RING4((mlp_fwd_fold_kernel<256, NW>), dim3((unsigned)grid), dim3(NW * 64), s, B);
            else SNERF_LAUNCH_RING((mlp_fwd_fold_kernel<256, NW>), dim3((unsigned)grid), dim3(NW * 64), s, B);
        }
        else if (P.width == 128) SNERF_LAUNCH_RING((mlp_fwd_fold_kernel<128, NW>), dim3((unsigned)grid), dim3(NW * 64), s, B);
        else SNERF_LAUNCH_RING((mlp_fwd_fold_kernel<64, NW>), dim3((unsigned)grid), dim3(NW * 64), s, B);
    } else {
        if (P.width > 256) {   // one wave per SIMD (20 .. 32-tile chains need the whole register file): 4-wave workgroups only
            if constexpr (NW == 4 && !FOLD) {
                if (P.width == 320) SNERF_LAUNCH_WIDE((mlp_fwd_kernel<320, 4, ENCODED, TRAIN>), dim3((unsigned)grid), dim3(256), s, B);
                else if (P.width == 384) SNERF_LAUNCH_WIDE((mlp_fwd_kernel<384, 4, ENCODED, TRAIN>), dim3((unsigned)grid), dim3(256), s, B);
                else if (P.width == 448) SNERF_LAUNCH_WIDE((mlp_fwd_kernel<448, 4, ENCODED, TRAIN>), dim3((unsigned)grid), dim3(256), s, B);
                else SNERF_LAUNCH_WIDE((mlp_fwd_kernel<512, 4, ENCODED, TRAIN>), dim3((unsigned)grid), dim3(256), s, B);
            } else {
                return fail(SNERF_E_BADARG, "mlp_fwd: widths above 256 run 4-wave workgroups");
            }
        } else if (P.width == 256) {
            if constexpr (!TRAIN && NW == 8) SNERF_LAUNCH_RING4((mlp_fwd_kernel<256, NW, ENCODED, TRAIN>), dim3((unsigned)grid), dim3(NW * 64), s, B);
            else SNERF_LAUNCH_RING((mlp_fwd_kernel<256, NW, ENCODED, TRAIN>), dim3((unsigned)grid), dim3(NW * 64), s, B);
        }
        else if (P.width == 128) SNERF_LAUNCH_RING((mlp_fwd_kernel<128, NW, ENCODED, TRAIN>), dim3((unsigned)grid), dim3(NW * 64), s, B);
        else SNERF_LAUNCH_RING((mlp_fwd_kernel<64, NW, ENCODED, TRAIN>), dim3((unsigned)grid), dim3(NW * 64), s, B);
    }
    return check_launch("mlp_fwd");
}

// slots (layers with an additional-input segment) and bytes of the per-ray fold table of a call, 0 when the fold does not apply
// (a fold pays when a ray's vector is reused: with fewer than 8 samples per ray the table costs more than it saves)
static int64_t fold_table_bytes(const Plan &P, int64_t n, int spr, int *slots_out = nullptr) {
    if (!P.add_dim || P.width > 256 || !tuning().mlp_fold || spr < 8 || n <= 0 || n % spr != 0) return 0;
    int slots = 0;
    for (int l = 0; l < P.nlayers; ++l)
        for (int sg = 0; sg < P.layer[l].nseg; ++sg) slots += P.layer[l].seg[sg].type == SEG_ADD ? 1 : 0;
    const int64_t floats = (n / spr) * slots * P.width;
    if (!slots || (floats + 255) / 256 > 0x7fffffffLL) return 0;
    if (slots_out) *slots_out = slots;
    return floats * (int64_t)sizeof(float);
}

// inference with per-ray additional inputs: pre-pass + FOLD kernel; the table lives in the CALLER's workspace
// (snerf_mlp_fold_workspace_bytes).  Returns 1 when the fold does not apply or no workspace was given (the caller then runs
// the per-sample form); a workspace that is too small is an error, never a silent switch.
template <int NW>
static int launch_fwd_folded(const Plan &P, const FwdArgs &A, hipStream_t s) {
    if (A.no_fold || !A.fold_ws) return 1;
    int slots = 0;
    const int64_t bytes = fold_table_bytes(P, A.n, A.spr, &slots);
    if (!bytes) return 1;
    if (A.fold_ws_bytes < bytes)
        return fail(SNERF_E_BADARG, "mlp_fwd: workspace of %lld bytes, the per-ray fold needs %lld (snerf_mlp_fold_workspace_bytes)",
                    (long long)A.fold_ws_bytes, (long long)bytes);
    if (!aligned(A.fold_ws, 16)) return fail(SNERF_E_ALIGN, "mlp_fwd: workspace must be 16-byte aligned");
    const int64_t n_rays = A.n / A.spr, floats = bytes / (int64_t)sizeof(float);
    hipLaunchKernelGGL(mlp_add_fold_kernel, dim3((unsigned)((floats + 255) / 256)), dim3(256), 0, s, P, A.packed, A.add, n_rays, slots,
                       A.fold_ws);
    FwdArgs B = A;
    B.fold = A.fold_ws;
    B.fold_slots = slots;
    return launch_fwd_nw<NW, false, false, true>(P, B, s);
}

template <bool ENCODED, bool TRAIN>
static int launch_fwd(const Plan &P, const FwdArgs &A, hipStream_t s) {
    // 8 waves (128 samples) per workgroup = one workgroup per CU, 2 waves per SIMD (two independent 4-wave workgroups per CU
    // measured 83.1 % against 85.7 % of the fp32 MFMA peak on the 128 x 128 frame, at twice the L2 -> LDS weight traffic: r02)
    if (P.width > 256) return launch_fwd_nw<4, ENCODED, TRAIN>(P, A, s);   // (--netwidth above 256: mlp_plan.h make_plan)
    if (!ENCODED && !TRAIN) {   // per-ray additional inputs + a workspace: the folded form (8-wave tiles)
        const int rc = launch_fwd_folded<FWD_WAVES>(P, A, s);
        if (rc != 1) return rc;
    }
    // Calls of a few 16-sample tiles per CU (the README's 64-ray batches, inference.py's 800 rays): the latency-class kernels
    // (mlp_lat.hip: a tile's output features split over the waves of a workgroup; bit-identical results)
    if constexpr (!ENCODED) {
        const LatChoice c = lat_choose_fwd<TRAIN>(P, A.n);
        if (c.mode == 1) return launch_fwd_lat<TRAIN>(P, A, s, 0);
        if (c.mode == 2) {      // whole rounds of 128-sample tiles on the throughput kernel, the rest on the latency kernels
            if (int rc = launch_fwd_nw<FWD_WAVES, ENCODED, TRAIN>(P, A, s, c.n_main)) return rc;
            return launch_fwd_lat<TRAIN>(P, A, s, c.n_main);
        }
    }
    // Small calls (the README's 64-ray batches: 4096 + 12 288 samples): while 64-sample tiles still fit one round of the chip,
    // the 4-wave form finishes in half the time of a 128-sample tile's pass through the weight stream - a call of up to
    // 64 x CUs samples is one tile's latency, not throughput (same per-sample arithmetic: bit-identical results).
    const int n_cu = device_cu_count("mlp_fwd");
    if (n_cu < 1) return n_cu;
    if (A.n <= (int64_t)64 * n_cu) return launch_fwd_nw<4, ENCODED, TRAIN>(P, A, s);
    return launch_fwd_nw<FWD_WAVES, ENCODED, TRAIN>(P, A, s);
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
    return launch_pack(P, params_flat, packed, (hipStream_t)stream, "mlp_pack");
}

extern "C" int64_t snerf_mlp_fold_workspace_bytes(const snerf_mlp_desc *desc, int64_t n, int samples_per_ray) {
    using namespace snerf;
    Plan P;
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "mlp_fold_workspace_bytes: desc is null");
    if (make_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "mlp_fold_workspace_bytes: %s", why);
    if (n < 0 || samples_per_ray < 1) return fail(SNERF_E_BADARG, "mlp_fold_workspace_bytes: bad n/samples_per_ray");
    return fold_table_bytes(P, n, samples_per_ray);
}

extern "C" int snerf_mlp_fwd_ws_f32(const snerf_mlp_desc *desc, const float *packed, const float *x, const float *dirs,
                                    int dirs_per_sample, const float *add, int64_t n, int samples_per_ray, float *raw,
                                    void *workspace, int64_t workspace_bytes, snerf_stream_t stream) {
    using namespace snerf;
    Plan P;
    FwdArgs A{};
    int rc = fill_args(desc, P, A);
    if (rc) return rc;
    if (n < 0 || samples_per_ray < 1) return fail(SNERF_E_BADARG, "mlp_fwd: bad n/samples_per_ray");
    if (dirs_per_sample & ~(SNERF_FWD_DIRS_PER_SAMPLE | SNERF_FWD_NO_RAY_FOLD))
        return fail(SNERF_E_BADARG, "mlp_fwd: dirs_per_sample is a bit set of SNERF_FWD_DIRS_PER_SAMPLE | SNERF_FWD_NO_RAY_FOLD (got %d)",
                    dirs_per_sample);
    if (n == 0) return SNERF_OK;
    if (!packed || !x || !raw) return fail(SNERF_E_BADARG, "mlp_fwd: null pointer");
    if (A.use_dir && !dirs) return fail(SNERF_E_BADARG, "mlp_fwd: dirs is null");
    if (A.add_dim && !add) return fail(SNERF_E_BADARG, "mlp_fwd: add is null");
    if (!aligned(packed, 16) || !aligned(raw, 16)) return fail(SNERF_E_ALIGN, "mlp_fwd: packed/raw must be 16-byte aligned");
    if (workspace_bytes < 0) return fail(SNERF_E_BADARG, "mlp_fwd: negative workspace_bytes");
    A.packed = packed;
    A.x = x;
    A.dirs = dirs;
    A.add = add;
    A.raw = raw;
    A.n = n;
    A.spr = samples_per_ray;
    A.dirs_per_sample = (dirs_per_sample & SNERF_FWD_DIRS_PER_SAMPLE) ? 1 : 0;
    A.no_fold = (dirs_per_sample & SNERF_FWD_NO_RAY_FOLD) ? 1 : 0;
    A.fold_ws = reinterpret_cast<float *>(workspace);
    A.fold_ws_bytes = workspace ? workspace_bytes : 0;
    return launch_fwd<false, false>(P, A, (hipStream_t)stream);
}

extern "C" int snerf_mlp_fwd_f32(const snerf_mlp_desc *desc, const float *packed, const float *x, const float *dirs,
                                 int dirs_per_sample, const float *add, int64_t n, int samples_per_ray, float *raw,
                                 snerf_stream_t stream) {
    return snerf_mlp_fwd_ws_f32(desc, packed, x, dirs, dirs_per_sample, add, n, samples_per_ray, raw, nullptr, 0, stream);
}

extern "C" int snerf_mlp_train_sizes(const snerf_mlp_desc *desc, int64_t n, int64_t *act_floats, int64_t *dy_floats,
                                     int64_t *packed_t_floats, int64_t *gpart_floats, int32_t *gpart_count) {
    using namespace snerf;
    Plan P;
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "mlp_train_sizes: desc is null");
    if (make_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "mlp_train_sizes: %s", why);
    if (n < 0) return fail(SNERF_E_BADARG, "mlp_train_sizes: negative n");
    TrainLayout L;
    make_train_layout(P, L);
    // (+ STAT_INTS ints behind the rows of each: per-layer exponents of the largest |X| / |dY|, left by the f16x3 forward /
    // dgrad for the f16x3 wgrad)
    if (act_floats) *act_floats = (int64_t)L.act_rows * n * 16 + STAT_INTS;
    if (dy_floats) *dy_floats = (int64_t)L.dy_rows * n * 16 + STAT_INTS;
    if (packed_t_floats) *packed_t_floats = (int64_t)(bwd_total_slabs(P, true) + SLAB_PAD) * SLAB_FLOATS;  // covers both streams
    const int G = wgrad_chunks(n);
    if (gpart_count) *gpart_count = G;
    if (gpart_floats) *gpart_floats = (int64_t)G * L.gp_floats;
    return SNERF_OK;
}

extern "C" int snerf_mlp_dy_layout(const snerf_mlp_desc *desc, int32_t *n_layers, int32_t *first_row, int32_t *n_out,
                                   int32_t *n_in) {
    using namespace snerf;
    Plan P;
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "mlp_dy_layout: desc is null");
    if (make_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "mlp_dy_layout: %s", why);
    TrainLayout L;
    make_train_layout(P, L);
    if (n_layers) *n_layers = P.nlayers;
    for (int l = 0; l < P.nlayers; ++l) {
        if (first_row) first_row[l] = L.dy[l];
        if (n_out) n_out[l] = P.layer[l].n_out;
        if (n_in) n_in[l] = P.layer[l].n_in;
    }
    return SNERF_OK;
}

extern "C" int snerf_mlp_fwd_train_f32(const snerf_mlp_desc *desc, const float *packed, const float *x,
                                       const float *dirs, int dirs_per_sample, const float *add, int64_t n,
                                       int samples_per_ray, float *raw, float *act, snerf_stream_t stream) {
    using namespace snerf;
    Plan P;
    FwdArgs A{};
    int rc = fill_args(desc, P, A);
    if (rc) return rc;
    if (n < 0 || samples_per_ray < 1) return fail(SNERF_E_BADARG, "mlp_fwd_train: bad n/samples_per_ray");
    if (n == 0) return SNERF_OK;
    if (!packed || !x || !raw || !act) return fail(SNERF_E_BADARG, "mlp_fwd_train: null pointer");
    if (A.use_dir && !dirs) return fail(SNERF_E_BADARG, "mlp_fwd_train: dirs is null");
    if (A.add_dim && !add) return fail(SNERF_E_BADARG, "mlp_fwd_train: add is null");
    if (!aligned(packed, 16) || !aligned(raw, 16) || !aligned(act, 16))
        return fail(SNERF_E_ALIGN, "mlp_fwd_train: packed/raw/act must be 16-byte aligned");
    TrainLayout L;
    make_train_layout(P, L);
    A.packed = packed;
    A.x = x;
    A.dirs = dirs;
    A.add = add;
    A.raw = raw;
    A.n = n;
    A.spr = samples_per_ray;
    A.dirs_per_sample = dirs_per_sample ? 1 : 0;
    A.act = act;
    A.act_pe = L.pe;
    A.act_add = L.add;
    A.act_dpe = L.dpe;
    A.act_x1 = L.x[1];
    A.act_o = L.o;
    A.act_h1 = L.h1;
    A.act_h2 = L.h2;
    A.act_mask = L.mask;
    return launch_fwd<false, true>(P, A, (hipStream_t)stream);
}

extern "C" int snerf_mlp_fwd_encoded_train_f32(const snerf_mlp_desc *desc, const float *packed, const float *x_enc,
                                               int64_t n, int64_t row_floats, float *raw, float *act,
                                               snerf_stream_t stream) {
    using namespace snerf;
    Plan P;
    FwdArgs A{};
    int rc = fill_args(desc, P, A);
    if (rc) return rc;
    if (n < 0) return fail(SNERF_E_BADARG, "mlp_fwd_encoded_train: bad n");
    if (n == 0) return SNERF_OK;
    if (!packed || !x_enc || !raw || !act) return fail(SNERF_E_BADARG, "mlp_fwd_encoded_train: null pointer");
    if (!aligned(packed, 16) || !aligned(raw, 16) || !aligned(act, 16))
        return fail(SNERF_E_ALIGN, "mlp_fwd_encoded_train: packed/raw/act must be 16-byte aligned");
    if (row_floats < A.pos_dim + A.add_dim || row_floats < A.dir_dim || row_floats > 0x7fffffff)
        return fail(SNERF_E_BADARG, "mlp_fwd_encoded_train: row of %lld floats is too short for this network", (long long)row_floats);
    TrainLayout L;
    make_train_layout(P, L);
    A.enc_stride = (int)row_floats;
    A.packed = packed;
    A.x = x_enc;
    A.raw = raw;
    A.n = n;
    A.spr = 1;
    A.act = act;
    A.act_pe = L.pe;
    A.act_add = L.add;
    A.act_dpe = L.dpe;
    A.act_x1 = L.x[1];
    A.act_o = L.o;
    A.act_h1 = L.h1;
    A.act_h2 = L.h2;
    A.act_mask = L.mask;
    return launch_fwd<true, true>(P, A, (hipStream_t)stream);
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
    return launch_fwd<true, false>(P, A, (hipStream_t)stream);
}
