// Split-bf16 variant of the fused positional-encoding + RenderRayNet forward (inference, and the training forward that
// also saves the layer inputs).
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
// at NS = 3), weights are pre-split and stream L2 -> LDS by DMA through a 3-slot ring (one 48 KiB slab = one
// k-block x 16 output tiles x 3 parts), positional encodings are evaluated in registers.  One persistent workgroup
// per CU walks the 128-sample tiles with the ring rolling on from tile to tile.  The device machinery (operand
// split, ring, hand-laid k-block stream, layer runner) is in mlp_bf16_device.h.
#include <stdlib.h>

#include "mlp_bf16_device.h"

namespace snerf {

// ------------------------------------------------------------------------------------------------
// weight packing: params_flat -> split-bf16 slab stream
// slab = [k-block in slab][output tile][part][lane][8 bf16] then 256 fp32 of bias
// ------------------------------------------------------------------------------------------------
// f16x3 (fmt = FMT_F16): exponent that brings the largest |weight| of layer li to [2^14, 2^15) - block-wide (1024 threads)
__device__ int layer_weight_exp(const Plan &P, int li, const float *__restrict__ params) {
    __shared__ float red[1024];
    const Layer &Ly = P.layer[li];
    const float *w = params + Ly.w_off;
    const int64_t count = (int64_t)Ly.n_out * Ly.n_in;
    float m = 0.f;
    for (int64_t e = threadIdx.x; e < count; e += 1024) m = fmaxf(m, fabsf(w[e]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    m = red[0];
    const int e = ((__float_as_int(m) >> 23) & 0xff) - 127;
    return m > 0.f ? min(14 - e, 50) : 0;
}

// f16x3: the table of weight exponents (one block per layer), left in the first pad slab of the stream - where the
// forward kernel reads it - before the pack kernel runs
__global__ __launch_bounds__(1024) void mlp_wexp_kernel(Plan P, int NS, const float *__restrict__ params,
                                                       unsigned char *__restrict__ packed, int table_slab) {
    const int we = layer_weight_exp(P, blockIdx.x, params);
    if (threadIdx.x == 0) reinterpret_cast<int *>(packed + (int64_t)table_slab * slab16_bytes(NS))[blockIdx.x] = we;
}
int launch_wexp(const Plan &P, int ns, const float *params_flat, void *packed, int table_slab, hipStream_t s, const char *what) {
    hipLaunchKernelGGL(mlp_wexp_kernel, dim3(P.nlayers), dim3(1024), 0, s, P, ns, params_flat, reinterpret_cast<unsigned char *>(packed),
                       table_slab);
    return check_launch(what);
}

__global__ __launch_bounds__(256) void mlp_pack_bf16_kernel(Plan P, int NS, const float *__restrict__ params,
                                                            unsigned char *__restrict__ packed, int fmt) {
    const int slab = blockIdx.x;
    const int SB = slab16_bytes(NS);
    unsigned char *dst = packed + (int64_t)slab * SB;
    const int *wexp_tab = reinterpret_cast<const int *>(packed + (int64_t)P.total_slabs * SB);
    if (slab >= P.total_slabs) {
        const int keep = (fmt == FMT_F16 && slab == P.total_slabs) ? MAX_LAYERS : 0;   // (the table of weight exponents)
        for (int e = threadIdx.x; e < SB / 4; e += 256)
            if (e >= keep) reinterpret_cast<float *>(dst)[e] = 0.f;
        return;
    }
    int li = 0;
    while (li + 1 < P.nlayers && slab >= P.layer[li + 1].first_slab) ++li;
    const Layer &Ly = P.layer[li];
    const int we = fmt == FMT_F16 ? wexp_tab[li] : 0;
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
        if (fmt == FMT_F16) {   // fp16 parts (RNE) of the scaled weight
            w = ldexpf(w, we);
            _Float16 h = (_Float16)w;
            if (s == 1) h = (_Float16)(w - (float)h);
            reinterpret_cast<_Float16 *>(a)[q] = h;
            continue;
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

// post-activation tiles -> the tile-row-major activation buffer
// `unscale` (f16x3): the accumulators hold (value) x 2^unscale
template <bool RELU, int N>
__device__ __forceinline__ void store_act(float *buf, int row0, int64_t n, int64_t sample, int g, const f4 (&t)[N],
                                          int unscale = 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        f4 v = t[i];
        if (unscale != 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = __builtin_ldexpf(v[r], -unscale);
        }
        if (RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        store_tile(buf, row0 + i, n, sample, g, v);
    }
}

// TRAIN additionally stores every layer input (post-activation, fp32, the fp32 kernel's layout) for the backward
// kernels of mlp_train.hip.
//
// FMT_F16 ("f16x3"): two fp16 parts per operand and three products per MAC on
// v_mfma_f32_16x16x32_f16.  fp16 keeps 11 mantissa bits per part (2^-22 relative for the pair, against 2^-16 for two
// bf16 parts) but only 5 exponent bits, so every operand is scaled by a power of two first - exactly, and undone
// exactly: the weights of layer l by 2^wexp[l] (largest |w| of the layer at 2^14; the pack kernel leaves the table in
// the first pad slab of the stream), the B operands of a layer per SAMPLE by 2^kx, kx from the largest activation of
// the sample that enters the layer (all of a sample's values sit in the four lanes that share its column), capped at
// 14 when encoder columns (|sin|, |cos| <= 1) enter too (lower if identity columns or additional inputs exceed 1).  An accumulator column then carries 2^(wexp + kx): the bias is
// loaded with that scale, the next layer's split rescales by the difference of the two exponents (one v_ldexp per
// value), the heads are scaled back before the store.  Nothing can overflow: scaled operands are < 2^15, a sum of 320
// products < 2^39.
template <int WIDTH, int NWAVES, int NS, bool TRAIN, int FMT = FMT_BF16>
__global__ __launch_bounds__(NWAVES * 64) void mlp_fwd_bf16_kernel(FwdArgs A) {
    constexpr int NT = NWAVES * 64;
    constexpr int T = WIDTH / 16, TD = WIDTH / 32;
    constexpr bool F16 = FMT == FMT_F16;
    static_assert(!(F16 && NS != 2), "f16x3: two parts");
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int *wexp_tab = reinterpret_cast<const int *>(reinterpret_cast<const char *>(A.packed) + (int64_t)A.total_slabs * slab16_bytes(NS));
    auto wexp = [&](int l) __attribute__((always_inline)) -> int {
        if constexpr (F16) return wexp_tab[l];
        else return 0;
    };
    // operand scale when encoder columns take part / upper limit (activations below 2^-50: with a weight exponent <= 50 a
    // bias of up to 2^13 still fits the scaled accumulator)
    constexpr int KX_PE = 14, KX_MAX = 64;
    auto sample_exp = [&](const auto &src, bool relu) __attribute__((always_inline)) -> int { return sample_exp16(src, relu); };
    // f16x3 training: exponent of the largest |X| that enters forward layer l, for the f16x3 wgrad (behind the rows of act)
    int *xstat = nullptr;
    if constexpr (F16 && TRAIN) xstat = reinterpret_cast<int *>(A.act + (int64_t)A.act_rows * A.n * 16);
    auto note_x = [&](int l, int e_true) __attribute__((always_inline)) {
        if constexpr (F16 && TRAIN) stat_max16(xstat, l, e_true);
    };
    auto operand_scale = [&](int e_src, int es_, int cap) __attribute__((always_inline)) -> int { return operand_scale16(e_src, es_, cap); };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Persistent workgroups: one per CU (the LDS ring allows no more), each walking the sample tiles
    // blockIdx.x, blockIdx.x + gridDim.x, ...  The weight ring keeps rolling from one tile into the next (the stream
    // wraps around), so only the first tile of a workgroup pays the pipeline fill.
    SlabPipe16<NT, NS> pipe;
    // the raw inputs of a tile are fetched while the previous tile's last two layers run (load_raw at the end of the
    // directional_input layer), so that a tile never starts by waiting on HBM
    float raw[6];
    auto load_raw = [&](int64_t t, float(&r)[6]) __attribute__((always_inline)) {
        const int64_t s0 = (t * NWAVES + wave) * 16 + (lane & 15);
        const int64_t s1 = s0 < A.n ? s0 : A.n - 1;
        // read-once / write-once streams bypass L2 retention: the 2.4 - 3.6 MB weight stream that every workgroup
        // re-reads is what each XCD's 4 MB L2 should keep
        r[0] = __builtin_nontemporal_load(A.x + s1 * 3 + 0);
        r[1] = __builtin_nontemporal_load(A.x + s1 * 3 + 1);
        r[2] = __builtin_nontemporal_load(A.x + s1 * 3 + 2);
        r[3] = r[4] = r[5] = 0.f;
        if (A.use_dir) {
            const float *dp = A.dirs + (A.dirs_per_sample ? s1 : s1 / A.spr) * 3;
            r[3] = dp[0], r[4] = dp[1], r[5] = dp[2];
        }
    };
    // (not in the TRAIN variants: they sit at the register limit)
    if (!TRAIN && blockIdx.x < A.n_tiles) load_raw(blockIdx.x, raw);
    for (int64_t tile = blockIdx.x; tile < A.n_tiles; tile += gridDim.x) {
    if (TRAIN) load_raw(tile, raw);
    const int64_t sample = (tile * NWAVES + wave) * 16 + (lane & 15);
    const bool valid = sample < A.n;
    const int64_t sc = valid ? sample : A.n - 1;
    SampleCtx c;
    c.g = lane >> 4;
    c.enc = nullptr;
    c.add = nullptr;
    c.px = raw[0], c.py = raw[1], c.pz = raw[2];
    c.dx = c.dy = c.dz = 0.f;
    const int64_t ray = sc / A.spr;
    if (A.use_dir) {
        const float ux = raw[3], uy = raw[4], uz = raw[5];
        const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(ux, ux), __fmul_rn(uy, uy)), __fmul_rn(uz, uz)));
        c.dx = __fdiv_rn(ux, nrm);
        c.dy = __fdiv_rn(uy, nrm);
        c.dz = __fdiv_rn(uz, nrm);
    }
    if (A.add_dim) c.add = A.add + ray * A.add_dim;
    // make the compiler retire the input loads HERE: a later first use would put its s_waitcnt vmcnt(0) inside
    // the slab loop and drain the in-flight weight DMA every time
    asm volatile("" ::"v"(c.px), "v"(c.py), "v"(c.pz), "v"(c.dx), "v"(c.dy), "v"(c.dz));

    if (tile == blockIdx.x) pipe.prologue(A.packed, ring, tid, A.total_slabs);

    // f16x3: operand scale limit of the layers that take position-encoder / additional columns: sin and cos are <= 1,
    // identity columns and additional inputs are whatever the caller passes - the sample's largest one decides
    int kx_pos = KX_PE;
    if constexpr (F16) {
        unsigned m = __float_as_uint(1.0f);
        if (A.pos_id) m = max(m, max(__float_as_uint(fabsf(c.px)), max(__float_as_uint(fabsf(c.py)), __float_as_uint(fabsf(c.pz)))));
        if (A.add_dim) {
            unsigned ma = 0u;
            for (int kb = 0; kb < A.add_nkb; ++kb)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int col = 32 * kb + 16 * (e >> 2) + 4 * c.g + (e & 3);
                    if (col < A.add_dim) ma = max(ma, __float_as_uint(fabsf(c.add[col])));
                }
            typedef unsigned u2v __attribute__((ext_vector_type(2)));
            u2v w = __builtin_amdgcn_permlane16_swap(ma, ma, false, false);
            ma = max(w[0], w[1]);
            w = __builtin_amdgcn_permlane32_swap(ma, ma, false, false);
            m = max(m, max(w[0], w[1]));
        }
        kx_pos = 14 - ((int)((m >> 23) & 0xffu) - 127);
        if constexpr (TRAIN) stat_max16(reinterpret_cast<int *>(A.act + (int64_t)A.act_rows * A.n * 16), STAT_INTS - 1, 14 - kx_pos);
    }

    // two accumulator sets ping-pong between consecutive layers: the finished set feeds the next layer's B
    // operands (split just in time, k-block by k-block) while the other set accumulates
    f4 accA[T], accB[T];
    auto pos_segments = [&](LayerRun16<T, NT, NS, FMT> &run, f4(&acc)[T], bool first, int kx) __attribute__((always_inline)) {
        auto pe_segment = [&]() __attribute__((always_inline)) {
            for (int kb = 0; kb < A.pos_nkb; ++kb)
                run.step_make([&](bf8(&b)[NS]) __attribute__((always_inline)) {
                    f4 half[2];
                    pe_operand16<NS, FMT>(c, false, A.pos_L, A.pos_id, kb, b, half, kx);
                    if (TRAIN && first && valid) {
                        store_tile(A.act, A.act_pe + 2 * kb, A.n, sample, c.g, half[0]);
                        if (2 * kb + 1 < A.pos_nkb16) store_tile(A.act, A.act_pe + 2 * kb + 1, A.n, sample, c.g, half[1]);
                    }
                }, acc);
        };
        auto add_segment = [&]() __attribute__((always_inline)) {
            for (int kb = 0; kb < A.add_nkb; ++kb)
                run.step_make([&](bf8(&b)[NS]) __attribute__((always_inline)) {
                    f4 half[2];
                    add_operand16<NS, FMT>(c, A.add_dim, kb, b, half, kx);
                    if (TRAIN && first && valid) {
                        store_tile(A.act, A.act_add + 2 * kb, A.n, sample, c.g, half[0]);
                        if (2 * kb + 1 < A.add_nkb16) store_tile(A.act, A.act_add + 2 * kb + 1, A.n, sample, c.g, half[1]);
                    }
                }, acc);
        };
        if (A.add_first) add_segment();
        pe_segment();
        if (!A.add_first) add_segment();
    };
    // positional_net[i] + relu: src (pre-activation of the previous layer) -> dst
    int es = 0;   // f16x3: the accumulators of the layer just finished hold (true value) x 2^es
    auto hidden = [&](int i, const f4(&src)[T], f4(&dst)[T]) __attribute__((always_inline)) {
        const bool skip = (A.skip_mask >> i) & 1u;
        int kx = 0;
        if constexpr (F16) {
            const int e = sample_exp(src, true);
            note_x(i + 1, e - es);
            kx = operand_scale(e, es, skip ? kx_pos : KX_MAX);
        }
        LayerRun16<T, NT, NS, FMT> run(pipe, lane);
        run.init(dst, wexp(i + 1) + kx);
        run.template run_hidden<true>(src, dst, kx - es);
        if (skip) pos_segments(run, dst, false, kx);
        run.finish();
        if constexpr (F16) es = wexp(i + 1) + kx;
        if (TRAIN && valid) {
            store_mask(A.act, A.act_mask, i + 1, A.n, sample, c.g, dst);
            asm volatile("" ::: "memory");  // mask first, then the tiles: interleaved, the two run out of registers
            store_act<true>(A.act, A.act_x1 + (i + 1) * T, A.n, sample, c.g, dst, es);
        }
    };
    {  // positions_pose_input (its relu is applied when the next layer splits accA)
        LayerRun16<T, NT, NS, FMT> run(pipe, lane);
        if constexpr (F16) es = wexp(0) + kx_pos;
        run.init(accA, es);
        pos_segments(run, accA, true, kx_pos);
        run.finish();
        if (TRAIN && valid) {
            store_mask(A.act, A.act_mask, 0, A.n, sample, c.g, accA);
            asm volatile("" ::: "memory");  // mask first, then the tiles: interleaved, the two run out of registers
            store_act<true>(A.act, A.act_x1, A.n, sample, c.g, accA, es);
        }
    }
    for (int i = 0; i < A.n_hidden; i += 2) {
        hidden(i, accA, accB);
        if (i + 1 < A.n_hidden) {
            hidden(i + 1, accB, accA);
        } else {
#pragma unroll
            for (int t = 0; t < T; ++t) accA[t] = accB[t];
        }
    }
    const int nh = A.n_hidden;
    {  // additional_linear_layer: relu(accA) -> accB (no activation on its output)
        int kx = 0;
        if constexpr (F16) {
            const int e = sample_exp(accA, true);
            note_x(nh + 1, e - es);
            kx = operand_scale(e, es, KX_MAX);
        }
        LayerRun16<T, NT, NS, FMT> run(pipe, lane);
        run.init(accB, wexp(nh + 1) + kx);
        run.template run_hidden<true>(accA, accB, kx - es);
        run.finish();
        if constexpr (F16) es = wexp(nh + 1) + kx;
        if (TRAIN && valid) store_act<false>(A.act, A.act_o, A.n, sample, c.g, accB, es);
    }
    int e_o = 0;   // f16x3: exponent of the largest |additional output| of the sample (feeds two layers)
    if constexpr (F16) {
        e_o = sample_exp(accB, false);
        note_x(nh + 3, e_o - es);
    }
    f4 sig[1];
    {
        const int kx = F16 ? operand_scale(e_o, es, KX_MAX) : 0;
        LayerRun16<1, NT, NS, FMT> run(pipe, lane);
        run.init(sig, wexp(nh + 2) + kx);
        run.template run_hidden<false>(accB, sig, kx - es);
        run.finish();
        if constexpr (F16) sig[0][0] = __builtin_ldexpf(sig[0][0], -(wexp(nh + 2) + kx));
    }
    f4 accd[TD], acce[TD];
    {  // directional_input (no activation)
        const int kx = F16 ? operand_scale(e_o, es, A.dir_nkb > 0 ? KX_PE : KX_MAX) : 0;
        LayerRun16<TD, NT, NS, FMT> run(pipe, lane);
        run.init(accd, wexp(nh + 3) + kx);
        run.template run_hidden<false>(accB, accd, kx - es);
        for (int kb = 0; kb < A.dir_nkb; ++kb)
            run.step_make([&](bf8(&b)[NS]) __attribute__((always_inline)) {
                f4 half[2];
                pe_operand16<NS, FMT>(c, true, A.dir_L, A.dir_id, kb, b, half, kx);
                if (TRAIN && valid) {
                    store_tile(A.act, A.act_dpe + 2 * kb, A.n, sample, c.g, half[0]);
                    if (2 * kb + 1 < A.dir_nkb16) store_tile(A.act, A.act_dpe + 2 * kb + 1, A.n, sample, c.g, half[1]);
                }
            }, accd);
        run.finish();
        if constexpr (F16) es = wexp(nh + 3) + kx;
        if (TRAIN && valid) store_act<false>(A.act, A.act_h1, A.n, sample, c.g, accd, es);
        if (!TRAIN && tile + gridDim.x < A.n_tiles) load_raw(tile + gridDim.x, raw);
    }
    {  // directional_net[0] (its relu is applied when the rgb head splits acce)
        int kx = 0;
        if constexpr (F16) {
            const int e = sample_exp(accd, false);
            note_x(nh + 4, e - es);
            kx = operand_scale(e, es, KX_MAX);
        }
        LayerRun16<TD, NT, NS, FMT> run(pipe, lane);
        run.init(acce, wexp(nh + 4) + kx);
        run.template run_hidden<false>(accd, acce, kx - es);
        run.finish();
        if constexpr (F16) es = wexp(nh + 4) + kx;
        if (TRAIN && valid) {
            store_mask(A.act, A.act_mask, A.n_hidden + 1, A.n, sample, c.g, acce);
            asm volatile("" ::: "memory");  // mask first, then the tiles: interleaved, the two run out of registers
            store_act<true>(A.act, A.act_h2, A.n, sample, c.g, acce, es);
        }
    }
    f4 rgb[1];
    {
        int kx = 0;
        if constexpr (F16) {
            const int e = sample_exp(acce, true);
            note_x(nh + 5, e - es);
            kx = operand_scale(e, es, KX_MAX);
        }
        LayerRun16<1, NT, NS, FMT> run(pipe, lane);
        run.init(rgb, wexp(nh + 5) + kx);
        run.template run_hidden<true>(acce, rgb, kx - es);
        run.finish();
        if constexpr (F16) {
#pragma unroll
            for (int r = 0; r < 3; ++r) rgb[0][r] = __builtin_ldexpf(rgb[0][r], -(wexp(nh + 5) + kx));
        }
    }
    if (valid && c.g == 0)
        __builtin_nontemporal_store(f4{rgb[0][0], rgb[0][1], rgb[0][2], sig[0][0]}, reinterpret_cast<f4 *>(A.raw) + sample);
    }  // tiles
    // the last k-block prefetched the first tile pair of the wrapped-around stream: retire those loads before their
    // registers can be reused
    if (blockIdx.x < A.n_tiles) wait_pair<NS, 0>(pipe.fa0, pipe.fa1);
}

// any (kw = 32) Plan -> split-bf16 slab stream; shared with warp_bf16.hip
int launch_pack_bf16(const Plan &P, int ns, const float *params_flat, void *packed, hipStream_t s, const char *what, int fmt) {
    if (fmt == FMT_F16)
        if (int rc = launch_wexp(P, ns, params_flat, packed, P.total_slabs, s, what)) return rc;
    hipLaunchKernelGGL(mlp_pack_bf16_kernel, dim3(P.total_slabs + SLAB_PAD), dim3(256), 0, s, P, ns, params_flat,
                       reinterpret_cast<unsigned char *>(packed), fmt);
    return check_launch(what);
}

static int plan16(const snerf_mlp_desc *desc, Plan &P, const char *what) {
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "%s: desc is null", what);
    if (make_plan(*desc, P, why, 32) != 0) return fail(SNERF_E_BADARG, "%s: %s", what, why);
    if (desc->width != 256) return fail(SNERF_E_BADARG, "%s: the split-bf16 path supports width 256 only", what);
    return SNERF_OK;
}

template <int NS, bool TRAIN, int FMT = FMT_BF16>
static int launch_bf16(const FwdArgs &A, hipStream_t s) {
    constexpr int NW = 8;
    const int lds = 3 * slab16_bytes(NS);
    static LdsRaised raised;   // per device
    if (int rc = raise_dynamic_lds(reinterpret_cast<const void *>(mlp_fwd_bf16_kernel<256, NW, NS, TRAIN, FMT>), lds, raised,
                                   "mlp_fwd_bf16"))
        return rc;
    const int n_cu = device_cu_count("mlp_fwd_bf16");  // one persistent workgroup per CU
    if (n_cu < 1) return n_cu;
    const int64_t grid = A.n_tiles > n_cu ? n_cu : A.n_tiles;
    if (grid > 0x7fffffffLL) return fail(SNERF_E_BADARG, "mlp_fwd_bf16: n too large");
    hipLaunchKernelGGL((mlp_fwd_bf16_kernel<256, NW, NS, TRAIN, FMT>), dim3((unsigned)grid), dim3(NW * 64), lds, s, A);
    return check_launch("mlp_fwd_bf16");
}

// argument checks + FwdArgs shared by the inference and the training forward; act == nullptr: inference
static int fwd_bf16(const snerf_mlp_desc *desc, const void *packed, int nsplit, const float *x, const float *dirs,
                    int dirs_per_sample, const float *add, int64_t n, int samples_per_ray, float *raw, float *act,
                    bool train, snerf_stream_t stream, const char *what) {
    Plan P;
    if (nsplit != 2 && nsplit != 3 && nsplit != SNERF_SPLIT_F16X3)
        return fail(SNERF_E_BADARG, "%s: nsplit must be 2, 3 or %d (f16x3)", what, SNERF_SPLIT_F16X3);
    int rc = plan16(desc, P, what);
    if (rc) return rc;
    if (n < 0 || samples_per_ray < 1) return fail(SNERF_E_BADARG, "%s: bad n/samples_per_ray", what);
    if (n == 0) return SNERF_OK;
    if (!packed || !x || !raw || (train && !act)) return fail(SNERF_E_BADARG, "%s: null pointer", what);
    if (desc->use_dir && !dirs) return fail(SNERF_E_BADARG, "%s: dirs is null", what);
    if (P.add_dim && !add) return fail(SNERF_E_BADARG, "%s: add is null", what);
    if (!aligned(packed, 16) || !aligned(raw, 16) || (train && !aligned(act, 16)))
        return fail(SNERF_E_ALIGN, "%s: packed/raw/act must be 16-byte aligned", what);
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
    A.total_slabs = P.total_slabs;
    A.n_tiles = (n + 8 * 16 - 1) / (8 * 16);
    if (train) {
        // the activation buffer has the layout of the 16-wide plan the backward kernels are built on
        Plan Q;
        const char *why;
        if (make_plan(*desc, Q, why) != 0) return fail(SNERF_E_BADARG, "%s: %s", what, why);
        TrainLayout L;
        make_train_layout(Q, L);
        A.act = act;
        A.act_pe = L.pe;
        A.act_add = L.add;
        A.act_dpe = L.dpe;
        A.act_x1 = L.x[1];
        A.act_o = L.o;
        A.act_h1 = L.h1;
        A.act_h2 = L.h2;
        A.act_mask = L.mask;
        A.act_rows = L.act_rows;
        if (nsplit == SNERF_SPLIT_F16X3 &&
            hipMemsetAsync(act + (int64_t)L.act_rows * n * 16, 0x80, STAT_INTS * sizeof(int), (hipStream_t)stream) != hipSuccess)
            return fail(SNERF_E_LAUNCH, "%s: cannot reset the layer statistics", what);
        A.pos_nkb16 = Q.pos_nkb;
        A.add_nkb16 = Q.add_nkb;
        A.dir_nkb16 = Q.dir_nkb;
        if (nsplit == SNERF_SPLIT_F16X3) return launch_bf16<2, true, FMT_F16>(A, (hipStream_t)stream);
        if (nsplit == 3) return launch_bf16<3, true>(A, (hipStream_t)stream);
        return launch_bf16<2, true>(A, (hipStream_t)stream);
    }
    if (nsplit == SNERF_SPLIT_F16X3) return launch_bf16<2, false, FMT_F16>(A, (hipStream_t)stream);
    if (nsplit == 3) return launch_bf16<3, false>(A, (hipStream_t)stream);
    return launch_bf16<2, false>(A, (hipStream_t)stream);
}

}  // namespace snerf

extern "C" int64_t snerf_mlp_packed_bf16_bytes(const snerf_mlp_desc *desc, int nsplit) {
    using namespace snerf;
    Plan P;
    if (nsplit != 2 && nsplit != 3 && nsplit != SNERF_SPLIT_F16X3)
        return fail(SNERF_E_BADARG, "mlp_packed_bf16_bytes: nsplit must be 2, 3 or %d (f16x3)", SNERF_SPLIT_F16X3);
    int rc = plan16(desc, P, "mlp_packed_bf16_bytes");
    if (rc) return rc;
    return (int64_t)(P.total_slabs + SLAB_PAD) * slab16_bytes(nsplit == SNERF_SPLIT_F16X3 ? 2 : nsplit);
}

extern "C" int snerf_mlp_pack_bf16(const snerf_mlp_desc *desc, const float *params_flat, void *packed, int nsplit,
                                   snerf_stream_t stream) {
    using namespace snerf;
    Plan P;
    if (nsplit != 2 && nsplit != 3 && nsplit != SNERF_SPLIT_F16X3)
        return fail(SNERF_E_BADARG, "mlp_pack_bf16: nsplit must be 2, 3 or %d (f16x3)", SNERF_SPLIT_F16X3);
    int rc = plan16(desc, P, "mlp_pack_bf16");
    if (rc) return rc;
    if (!params_flat || !packed) return fail(SNERF_E_BADARG, "mlp_pack_bf16: null pointer");
    if (!aligned(packed, 16)) return fail(SNERF_E_ALIGN, "mlp_pack_bf16: packed must be 16-byte aligned");
    if (nsplit == SNERF_SPLIT_F16X3) return launch_pack_bf16(P, 2, params_flat, packed, (hipStream_t)stream, "mlp_pack_bf16", FMT_F16);
    return launch_pack_bf16(P, nsplit, params_flat, packed, (hipStream_t)stream, "mlp_pack_bf16");
}

extern "C" int snerf_mlp_fwd_bf16_f32(const snerf_mlp_desc *desc, const void *packed, int nsplit, const float *x,
                                      const float *dirs, int dirs_per_sample, const float *add, int64_t n,
                                      int samples_per_ray, float *raw, snerf_stream_t stream) {
    return snerf::fwd_bf16(desc, packed, nsplit, x, dirs, dirs_per_sample, add, n, samples_per_ray, raw, nullptr, false,
                           stream, "mlp_fwd_bf16");
}

extern "C" int snerf_mlp_fwd_train_bf16_f32(const snerf_mlp_desc *desc, const void *packed, int nsplit, const float *x,
                                            const float *dirs, int dirs_per_sample, const float *add, int64_t n,
                                            int samples_per_ray, float *raw, float *act, snerf_stream_t stream) {
    return snerf::fwd_bf16(desc, packed, nsplit, x, dirs, dirs_per_sample, add, n, samples_per_ray, raw, act, true, stream,
                           "mlp_fwd_train_bf16");
}
