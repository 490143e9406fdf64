// dgrad of RenderRayNet on the bf16 matrix cores (split-bf16, fp32-class accuracy): the transposed network of
// mlp_train.hip's mlp_bwd_kernel run with the machinery of mlp_bf16.hip.
//
//   * same program: rgb head^T -> mask(h2) -> directional_net[0]^T -> directional_input^T + sigma head row ->
//     additional^T -> positional_net[i]^T with the ReLU masks taken from the saved activations; every d Y_l is
//     stored (fp32, the tile-row layout) for the wgrad kernel, which stays fp32;
//   * the transposed weights are pre-split into NS bf16 parts (snerf_mlp_pack_t_bf16): k-blocks of 32 forward
//     output rows x 16-wide output tiles, slabs of NS x 16 KiB + 1 KiB (the fp32 sigma-head row rides in the 1 KiB
//     block and initialises the accumulators exactly like the fp32 kernel);
//   * gradients are split just in time into NS parts (6 or 3 products per fp32 MAC, fp32 accumulate);
//   * the ReLU mask is applied in place on the accumulators (4 tiles at a time: the mask loads must not cost 64
//     more registers), then the masked tiles are stored and feed the next transposed layer.
#include <stdlib.h>

#include "mlp_bf16_device.h"
#include "mlp_train_device.h"

namespace snerf {

// ------------------------------------------------------------------------------------------------
// transposed weight stream, split-bf16:  slab = [k-block in slab][output tile][part][lane][8 bf16] then 256 fp32 aux
//   A[(kb, to, s, lane (i,g), e)] = part_s(W_fwd[32*kb + 16*(e>>2) + 4*g + (e&3)][col(to, i)])
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mlp_pack_t_bf16_kernel(Plan P, BwdPlan B, int NS, int fmt, const float *__restrict__ params,
                                                              unsigned char *__restrict__ packed) {
    const int slab = blockIdx.x;
    const int SB = slab16_bytes(NS);
    unsigned char *dst = packed + (int64_t)slab * SB;
    // f16x3: the table of weight exponents (launch_wexp, indexed by forward layer) sits in the first pad slab
    const int *wexp_tab = reinterpret_cast<const int *>(packed + (int64_t)B.total_slabs * SB);
    if (slab >= B.total_slabs) {
        const int keep = (fmt == FMT_F16 && slab == B.total_slabs) ? MAX_LAYERS : 0;
        for (int e = threadIdx.x; e < SB / 4; e += 256)
            if (e >= keep) reinterpret_cast<float *>(dst)[e] = 0.f;
        return;
    }
    int bi = 0;
    while (bi + 1 < B.nl && slab >= B.layer[bi + 1].first_slab) ++bi;
    const BwdLayer &Bl = B.layer[bi];
    const Layer &Ly = P.layer[Bl.fwd];
    const Seg &sg = Ly.seg[Bl.seg];
    const int sl = slab - Bl.first_slab;
    const int kps = 16 / Bl.t_out;
    const float *Wm = params + Ly.w_off;
    __bf16 *a = reinterpret_cast<__bf16 *>(dst);
    const int per_kb = Bl.t_out * NS * 512;
    for (int q = threadIdx.x; q < NS * 8192; q += 256) {
        const int kbl = q / per_kb;
        int rem = q - kbl * per_kb;
        const int to = rem / (NS * 512);
        rem -= to * NS * 512;
        const int s = rem >> 9;
        rem &= 511;
        const int lane = rem >> 3, e = rem & 7;
        const int i = lane & 15, g = lane >> 4;
        const int kb = sl * kps + kbl;
        const int row = 32 * kb + 16 * (e >> 2) + 4 * g + (e & 3);  // forward output feature (contraction index)
        int col = -1;                                               // forward input column produced by output row (to, i)
        if (sg.type == SEG_PE) {
            if (to < sg.nkb) {  // slot (i>>2, i&3) of encoder k-block `to` (16-wide plan)
                const int c = pe_slot_col(sg.L, sg.ident, to, i >> 2, i & 3);
                if (c >= 0) col = sg.col_off + c;
            }
        } else if (16 * to + i < sg.ncols) {
            col = sg.col_off + 16 * to + i;
        }
        float w = 0.f;
        if (kb < Bl.nkb && row < Ly.n_out && col >= 0) w = Wm[(int64_t)row * Ly.n_in + col];
        if (fmt == FMT_F16) {   // fp16 parts (RNE) of the scaled weight
            w = ldexpf(w, wexp_tab[Bl.fwd]);
            _Float16 hh = (_Float16)w;
            if (s == 1) hh = (_Float16)(w - (float)hh);
            reinterpret_cast<_Float16 *>(a)[q] = hh;
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
    for (int j = threadIdx.x; j < 256; j += 256) {
        float v = 0.f;
        if (sl == 0 && Bl.aux_fwd >= 0) {
            const Layer &La = P.layer[Bl.aux_fwd];
            if (j < La.seg[0].ncols) v = params[La.w_off + j];  // row 0 of the sigma head
        }
        aux[j] = v;
    }
}

// t[i] = (ReLU output > 0) ? t[i] : 0 from the forward kernel's sign mask (store_mask, mlp_device.h: 8 bytes per lane
// instead of the 256-byte activation row), then store t as d Y tile-rows
// `unscale` (f16x3): t holds (value) x 2^unscale - stays so for the next layer, is stored unscaled
template <int N>
__device__ __forceinline__ void mask_store(f4 (&t)[N], uint2 m, float *dy, int dy_row0, int64_t n, int64_t sample,
                                           bool valid, int g, int unscale = 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const unsigned w = i < 8 ? m.x : m.y;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int keep = static_cast<int>(w << (31 - (((i & 7) << 2) | r))) >> 31;  // 0 or ~0
            t[i][r] = __int_as_float(__float_as_int(t[i][r]) & keep);
        }
        if (valid) {
            f4 v = t[i];
            if (unscale != 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = __builtin_ldexpf(v[r], -unscale);
            }
            store_tile(dy, dy_row0 + i, n, sample, g, v);
        }
    }
}
template <int N>
__device__ __forceinline__ void store_tiles_unscaled(float *buf, int row0, int64_t n, int64_t sample, int g, const f4 (&t)[N],
                                                     int unscale) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        f4 v = t[i];
        if (unscale != 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = __builtin_ldexpf(v[r], -unscale);
        }
        store_tile(buf, row0 + i, n, sample, g, v);
    }
}

// FMT_F16: the f16x3 scheme of mlp_fwd_bf16_kernel on the transposed network - weights scaled per (forward) layer, the
// gradient operands per sample; the masked accumulators keep their scale for the next layer and are stored unscaled.
// No bias here; the fp32 sigma-head term of d o is brought to the scale of its accumulator.
template <int WIDTH, int NWAVES, int NS, bool INPUT_GRAD, int FMT = FMT_BF16>
__global__ __launch_bounds__(NWAVES * 64) void mlp_bwd_bf16_kernel(BwdArgs A) {
    constexpr int NT = NWAVES * 64;
    constexpr int T = WIDTH / 16, TD = WIDTH / 32;
    constexpr int TPP = 4, TPD = 2;
    constexpr bool F16 = FMT == FMT_F16;
    static_assert(!(F16 && NS != 2), "f16x3: two parts");
    constexpr int KX_MAX = 40;   // gradients below 2^-26 keep fewer bits; bounds the fp32 sigma-head term of d o
    const int *wexp_tab = reinterpret_cast<const int *>(reinterpret_cast<const char *>(A.packed_t) + (int64_t)A.total_slabs * slab16_bytes(NS));
    auto wexp = [&](int l) __attribute__((always_inline)) -> int {
        if constexpr (F16) return wexp_tab[l];
        else return 0;
    };
    int es = 0;   // f16x3: scale exponent of the accumulators of the layer just finished
    // f16x3: exponent of the largest |dY| of forward layer l, for the f16x3 wgrad (behind the rows of dy)
    int *ystat = nullptr;
    if constexpr (F16) ystat = reinterpret_cast<int *>(A.dy + (int64_t)A.dy_rows * A.n * 16);
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4;
    // persistent workgroups over the sample tiles, like mlp_fwd_bf16_kernel: the weight ring rolls on from tile to tile.
    // The INPUT_GRAD variants sit at the 256-register limit; carrying the ring state and the prefetched tile pair
    // across tiles makes them spill, so they run one tile per workgroup (the loop folds away).
    constexpr bool PERSIST = !INPUT_GRAD;
    SlabPipe16<NT, NS> pipe;
    int64_t tile = blockIdx.x;
    if (tile >= A.n_tiles) return;
    do {
    const int64_t sample = (tile * NWAVES + wave) * 16 + (lane & 15);
    const bool valid = sample < A.n;
    const int64_t sc = valid ? sample : A.n - 1;
    const int nh = A.n_hidden;

    f4 dr = *reinterpret_cast<const f4 *>(A.d_raw + sc * 4);
    // the sign mask of the next masked layer is fetched one layer ahead (2 registers)
    uint2 mk = *mask_ptr(A.act, A.act_mask, nh + 1, A.n, sc, g);
    asm volatile("" : "+v"(dr), "+v"(mk.x), "+v"(mk.y));  // retire the loads here, before the weight DMA starts (cf. mlp_fwd_bf16_kernel)
    const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
    // head gradients as tile-rows for the wgrad kernel (rows 0..2 = rgb, row 0 = sigma)
    if (valid) {
        store_tile(A.dy, A.dy_rgb, A.n, sample, g, g == 0 ? f4{dr[0], dr[1], dr[2], 0.f} : zero);
        store_tile(A.dy, A.dy_sig, A.n, sample, g, g == 0 ? f4{dr[3], 0.f, 0.f, 0.f} : zero);
    }

    if (!PERSIST || tile == blockIdx.x) pipe.prologue(A.packed_t, ring, tid, A.total_slabs);

    f4 accd[TD], acce[TD];
    {  // rgb_out_layer^T, then the ReLU mask of directional_net[0] (models/render_ray_net.py:58-60)
        LayerRun16<TD, NT, NS, FMT> run(pipe, lane);
        run.init_plain(accd);
        const f4 src[2] = {g == 0 ? f4{dr[0], dr[1], dr[2], 0.f} : zero, zero};
        int kx = 0;
        if constexpr (F16) {
            const int e = sample_exp16(src, false);
            stat_max16(ystat, nh + 5, e);                                            // d rgb
            stat_max16(ystat, nh + 2, (int)((__float_as_uint(dr[3]) >> 23) & 0xffu) - 127);  // d sigma
            kx = operand_scale16(e, 0, KX_MAX);
        }
        run.template run_hidden<false>(src, accd, kx);
        run.finish();
        if constexpr (F16) es = wexp(nh + 5) + kx;
        mask_store(accd, mk, A.dy, A.dy_dn0, A.n, sample, valid, g, es);
        mk = *mask_ptr(A.act, A.act_mask, nh, A.n, sc, g);
    }
    {  // directional_net[0]^T; directional_input has no activation (:54-57)
        LayerRun16<TD, NT, NS, FMT> run(pipe, lane);
        run.init_plain(acce);
        int kx = 0;
        if constexpr (F16) {
            const int e = sample_exp16(accd, false);
            stat_max16(ystat, nh + 4, e - es);
            kx = operand_scale16(e, es, KX_MAX);
        }
        run.template run_hidden<false>(accd, acce, kx - es);
        run.finish();
        if constexpr (F16) es = wexp(nh + 4) + kx;
        if (valid) store_tiles_unscaled(A.dy, A.dy_din, A.n, sample, g, acce, es);
    }
    // d h1 (acce, scale es) feeds the two transposes of directional_input: one operand scale for both
    int kx_din = 0;
    if constexpr (F16) {
        const int e = sample_exp16(acce, false);
        stat_max16(ystat, nh + 3, e - es);
        kx_din = operand_scale16(e, es, KX_MAX);
    }
    if (INPUT_GRAD && A.use_dir && A.dir_nkb > 0) {
        // d (direction encoding) = directional_input[:, W:]^T d h1, then encoder and normalisation backward
        f4 ddpe[TPD];
        LayerRun16<TPD, NT, NS, FMT> run(pipe, lane);
        run.init_plain(ddpe);
        run.template run_hidden<false>(acce, ddpe, kx_din - es);
        run.finish();
        if constexpr (F16) {
#pragma unroll
            for (int q = 0; q < TPD; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) ddpe[q][r] = __builtin_ldexpf(ddpe[q][r], -(wexp(nh + 3) + kx_din));
        }
        const float *dp = A.dirs + (A.dirs_per_sample ? sc : sc / A.spr) * 3;
        const float ux = dp[0], uy = dp[1], uz = dp[2];
        const float nrm = sqrtf(ux * ux + uy * uy + uz * uz);
        const float nx = ux / nrm, ny = uy / nrm, nz = uz / nrm;
        float gx = 0.f, gy = 0.f, gz = 0.f;
        pe_backward<TPD>(ddpe, A.dir_nkb, nx, ny, nz, A.dir_L, A.dir_id, g, gx, gy, gz);
        gx = sum_over_g(gx);
        gy = sum_over_g(gy);
        gz = sum_over_g(gz);
        if (valid && g == 0) {  // d (u/|u|) -> d u = (g - n (n.g)) / |u|   (models/smpl_nerf_pipeline.py:54-55)
            const float dot = nx * gx + ny * gy + nz * gz;
            float *q = A.d_dirs + sample * 3;
            q[0] = (gx - nx * dot) / nrm;
            q[1] = (gy - ny * dot) / nrm;
            q[2] = (gz - nz * dot) / nrm;
        }
    }
    f4 accA[T], accB[T];
    f4 dpe[TPP];
#pragma unroll
    for (int t = 0; t < TPP; ++t) dpe[t] = zero;
    // position-encoding columns of forward layer l (layer 0 or a skip layer), given d Y_l in `cur`
    auto pe_columns = [&](int l, const f4(&cur)[T]) __attribute__((always_inline)) {
        if (!INPUT_GRAD || A.pos_nkb <= 0) return;
        if (!(l == 0 || ((A.skip_mask >> (l - 1)) & 1u))) return;
        f4 t[TPP];
        LayerRun16<TPP, NT, NS, FMT> run(pipe, lane);
        run.init_plain(t);
        int kx = 0;
        if constexpr (F16) kx = operand_scale16(sample_exp16(cur, false), es, KX_MAX);
        run.template run_hidden<false>(cur, t, kx - es);
        run.finish();
#pragma unroll
        for (int q = 0; q < TPP; ++q) {
            if constexpr (F16) {
#pragma unroll
                for (int r = 0; r < 4; ++r) t[q][r] = __builtin_ldexpf(t[q][r], -(wexp(l) + kx));
            }
            dpe[q] += t[q];
        }
    };
    {  // d o = directional_input[:, :W]^T d h1 + sigma_out_layer^T d sigma; additional layer has no activation (:51-52)
        LayerRun16<T, NT, NS, FMT> run(pipe, lane);
        run.init_plain(accA);  // aux block = sigma head weights (fp32)
        const int es_o = wexp(nh + 3) + kx_din;
        const float dsig = F16 ? __builtin_ldexpf(dr[3], es_o) : dr[3];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            accA[t][0] *= dsig;
            accA[t][1] *= dsig;
            accA[t][2] *= dsig;
            accA[t][3] *= dsig;
        }
        run.template run_hidden<false>(acce, accA, kx_din - es);
        run.finish();
        if constexpr (F16) es = es_o;
        if (valid) store_tiles_unscaled(A.dy, (nh + 1) * T, A.n, sample, g, accA, es);
    }
    // additional^T, positional_net[nh-1]^T ... positional_net[0]^T: forward layer l+1 transposed yields
    // d X_{l+1}; masking with X_{l+1} > 0 gives d Y of forward layer l (:46-50).  Two accumulator sets ping-pong.
    auto layer = [&](int l, const f4(&src)[T], f4(&dst)[T]) __attribute__((always_inline)) {
        LayerRun16<T, NT, NS, FMT> run(pipe, lane);
        run.init_plain(dst);
        int kx = 0;
        if constexpr (F16) {
            const int e = sample_exp16(src, false);
            stat_max16(ystat, l + 1, e - es);
            kx = operand_scale16(e, es, KX_MAX);
        }
        run.template run_hidden<false>(src, dst, kx - es);
        run.finish();
        if constexpr (F16) es = wexp(l + 1) + kx;
        mask_store(dst, mk, A.dy, l * T, A.n, sample, valid, g, es);
        if constexpr (F16) {
            if (l == 0) stat_max16(ystat, 0, sample_exp16(dst, false) - es);   // (the other layers' |dY| are noted where they are consumed)
        }
        if (l > 0) mk = *mask_ptr(A.act, A.act_mask, l - 1, A.n, sc, g);
        pe_columns(l, dst);
    };
    for (int l = nh; l >= 0; l -= 2) {
        layer(l, accA, accB);
        if (l - 1 >= 0) {
            layer(l - 1, accB, accA);
        }
    }
    if (INPUT_GRAD) {
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (A.pos_nkb > 0) {
            const float px = A.x[sc * 3 + 0], py = A.x[sc * 3 + 1], pz = A.x[sc * 3 + 2];
            pe_backward<TPP>(dpe, A.pos_nkb, px, py, pz, A.pos_L, A.pos_id, g, gx, gy, gz);
        }
        gx = sum_over_g(gx);
        gy = sum_over_g(gy);
        gz = sum_over_g(gz);
        if (valid && g == 0) {
            float *q = A.d_x + sample * 3;
            q[0] = gx;
            q[1] = gy;
            q[2] = gz;
        }
    }
    tile += gridDim.x;
    } while (PERSIST && tile < A.n_tiles);
    // the last k-block prefetched the first tile pair of the (wrapped-around) stream: retire those loads before their
    // registers can be reused
    wait_pair<NS, 0>(pipe.fa0, pipe.fa1);
}

// ------------------------------------------------------------------------------------------------
// wgrad of the wide jobs on the bf16 matrix cores
// ------------------------------------------------------------------------------------------------
// dW[i][j] = sum_s dY[s][i] X[s][j] with v_mfma_f32_16x16x32_bf16: the contraction index is the sample, so both
// operands are "8 consecutive samples of one feature per lane" - the transpose of the tile-row layout
// ([sample][16 features]).  One workgroup = 8 waves = one wide (layer, segment) job x one chunk of samples; wave
// (bi, bj) owns output tiles 8bi.. x input tiles 4bj.. (8 x 4 accumulator tiles = 128 VGPRs).  A stage is 32 samples
// of every tile-row of the job, brought into LDS by DMA as fp32 (two 1 KiB pieces per tile-row; 2 slots of 64 KiB);
// each wave then gathers its operands with 8 ds_read_b32 per tile (feature i = lane & 15, samples 8 (lane >> 4) + e),
// splits them into NS bf16 parts in registers (the same split as the forward) and runs NS(NS+1)/2 MFMAs per tile pair.
// The DMA places sample s of a 16-sample piece at position s ^ ((s >> 3) & 1): the four lane groups of a gather then
// hit disjoint banks.  Products are exact, accumulation fp32; partials and the reduce are those of the fp32 kernel.
constexpr int WB_WAVES = 8, WB_THREADS = WB_WAVES * 64;
constexpr int WB_STAGE = 32;                            // samples per stage = one MFMA k-block
constexpr int WB_ROW_FLOATS = WB_STAGE * 16;            // one tile-row of a stage: 2 KiB
constexpr int WB_SLOT_FLOATS = 32 * WB_ROW_FLOATS;      // 16 dY rows + 16 X rows: 64 KiB
constexpr int WB_LDS_BYTES = 2 * WB_SLOT_FLOATS * 4;

// FMT_F16 (f16x3 training): two fp16 parts and three products.  The contraction runs over samples, so the operand
// scales must not depend on the sample: X and dY of the job are scaled by 2^(14 - exponent of their largest value over
// ALL samples), which the f16x3 forward and dgrad kernels leave behind the activation / dY rows (stat_max16).  Samples
// with small gradients then lose bits of their own, but not relative to the sum they are added to.
template <int NS, int FMT = FMT_BF16>
__global__ __launch_bounds__(WB_THREADS) void mlp_wgrad_bf16_kernel(Plan P, TrainLayout L, WgradArgs A) {
    constexpr int TI = 8, TJ = 4;
    using Tm = Terms<NS>;
    constexpr bool F16 = FMT == FMT_F16;
    extern __shared__ __attribute__((aligned(16))) float wring[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- decode the job: (wide layer, segment, group of 16 input k-blocks), like mlp_wgrad_kernel ----
    int job = blockIdx.x, l = 0, s = 0, kb0 = 0;
    for (l = 0; l < P.nlayers; ++l) {
        bool found = false;
        kb0 = 0;
        for (s = 0; s < P.layer[l].nseg; ++s) {
            const int cnt = wgrad_wide(P.layer[l], s) ? (P.layer[l].seg[s].nkb + 15) / 16 : 0;
            if (job < cnt) { found = true; break; }
            job -= cnt;
            kb0 += P.layer[l].seg[s].nkb;
        }
        if (found) break;
    }
    const Layer &Ly = P.layer[l];
    const int jb = job;
    const int n_rows_y = Ly.t_out, n_rows_x = min(16, Ly.seg[s].nkb - 16 * jb);
    const int64_t n = A.n;
    int first_seg = 0;
    while (first_seg < Ly.nseg && Ly.seg[first_seg].nkb == 0) ++first_seg;
    const int bi = wave >> 2, bj = wave & 3;
    const int n_ti = max(0, min(TI, n_rows_y - TI * bi)), n_tj = max(0, min(TJ, n_rows_x - TJ * bj));
    const bool active = n_ti > 0 && n_tj > 0;
    const bool want_bias = (s == first_seg && jb == 0 && bj == 0);
    const int i16 = lane & 15, kq = lane >> 4;

    const int64_t begin = (int64_t)blockIdx.y * A.chunk;
    const int64_t end = min(n, begin + A.chunk);
    const int nstages = begin < end ? (int)((end - begin + WB_STAGE - 1) / WB_STAGE) : 0;
    float sx = 1.f, sy = 1.f, unscale = 1.f;   // f16x3: operand scales of the job and the scale of its result
    if constexpr (F16) {
        // X scale: the statistic of THIS segment (xstat_index: the layer's hidden input, or the encoder / additional-input
        // columns).  r04: this read xstat[l] - the hidden input's - for every wide job of the layer; with encoded pose columns
        // (1380 of them: wide jobs of their own) layer 0 has no hidden input, its statistic is unset and the scale was 2^114
        // (found by tools/ab/fuzz_train.py in chunked f16x3 steps)
        const int xi = xstat_index(P, l, s);
        const int ex = 14 - (xi < 0 ? 0 : min(max(A.xstat[xi], -100), 100)), ey = 14 - min(max(A.ystat[l], -100), 100);
        sx = __builtin_ldexpf(1.f, ex);
        sy = __builtin_ldexpf(1.f, ey);
        unscale = __builtin_ldexpf(1.f, -(ex + ey));
    }

    // ---- stage loader: this wave brings LDS rows 4*wave .. 4*wave+3 (rows 0..15 = dY, 16..31 = X), two 16-sample
    // pieces each; rows the job does not have re-load row 0 of dY so that every wave issues exactly 8 pieces per stage
    const float *row_src[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 4 * wave + q;
        int64_t grow = L.dy[l];
        if (r < 16) {
            if (r < n_rows_y) grow = L.dy[l] + r;
            row_src[q] = A.dy + grow * n * 16;
        } else if (r - 16 < n_rows_x) {
            row_src[q] = A.act + (int64_t)(seg_act_row(P, L, l, s) + 16 * jb + (r - 16)) * n * 16;
        } else {
            row_src[q] = A.dy + grow * n * 16;
        }
    }
    auto issue = [&](int stage, int slot) {
        // lane covers 16 B of a piece: position q = lane >> 2 (of 16), feature quad lane & 3; position q holds sample
        // q ^ ((q >> 3) & 1) of the piece (bank swizzle, see above)
        const int q = lane >> 2, sp = q ^ ((q >> 3) & 1);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int64_t smp = min(begin + (int64_t)stage * WB_STAGE + 16 * sub + sp, n - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(row_src[r] + smp * 16 + (lane & 3) * 4),
                    (__attribute__((address_space(3))) void *)(wring + slot * WB_SLOT_FLOATS + (4 * wave + r) * WB_ROW_FLOATS +
                                                               sub * 256),
                    16, 0, 0);
        }
    };

    f4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    float bsum[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) bsum[i] = 0.f;

    // operand of one tile: this lane's 8 samples (8 kq + e) of feature i16 -> NS packed-bf16 parts.  MASK: zero the
    // samples at or past `limit` (ragged end of the chunk); SUM: also accumulate the values (bias gradient)
    auto gather = [&](const float *row, auto mask, auto want_sum, int limit, bf8(&parts)[NS], float &sum, float scale)
                      __attribute__((always_inline)) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int smp = 8 * kq + e;                               // sample within the stage
            const int pos = (smp & 15) ^ (((smp & 15) >> 3) & 1);      // its position within its 16-sample piece
            v[e] = row[(smp >> 4) * 256 + pos * 16 + i16];
            if constexpr (decltype(mask)::value) v[e] = smp < limit ? v[e] : 0.f;
        }
        if constexpr (decltype(want_sum)::value) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += v[e];
        }
        if constexpr (F16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= scale;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) split_pair_into<NS, FMT>(v[2 * e], v[2 * e + 1], parts, e);
    };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;

    if (nstages > 0) {
        issue(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    for (int st = 0; st < nstages; ++st) {
        const int slot = st & 1;
        if (st + 1 < nstages) issue(st + 1, slot ^ 1);  // the other slot was freed by the barrier that ended stage st-1
        if (active) {
            const float *base = wring + slot * WB_SLOT_FLOATS;
            const int64_t s0 = begin + (int64_t)st * WB_STAGE;
            // samples past the chunk end contribute a = 0 (b is finite data); tiles past the edge of the job (the
            // 128-row layer has no second row block... ) are skipped - wave-uniform branches
            const int limit = (int)min((int64_t)WB_STAGE, end - s0);
            const bool tail = limit < WB_STAGE;
            bf8 bpart[TJ][NS];
            float dummy = 0.f;
#pragma unroll
            for (int j = 0; j < TJ; ++j) gather(base + (16 + TJ * bj + j) * WB_ROW_FLOATS, No{}, No{}, 0, bpart[j], dummy, sx);
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                if (i < n_ti) {
                    bf8 apart[NS];
                    const float *row = base + (TI * bi + i) * WB_ROW_FLOATS;
                    if (tail) gather(row, Yes{}, Yes{}, limit, apart, bsum[i], sy);
                    else if (want_bias) gather(row, No{}, Yes{}, 0, apart, bsum[i], sy);
                    else gather(row, No{}, No{}, 0, apart, dummy, sy);
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
#pragma unroll
                        for (int t = 0; t < Tm::N; ++t)
                            acc[i][j] = mfma16<FMT>(apart[Tm::A[t]], bpart[j][Tm::B[t]], acc[i][j]);
                }
            }
        }
        if (st + 1 < nstages) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stage st+1 landed (this wave's pieces)
            __builtin_amdgcn_s_barrier();                      // ... everyone's, and everyone is done reading `slot`
        }
    }
    if (!active) return;
    // ---- write the partial of this (block, chunk): same format as mlp_wgrad_kernel -----------------------
    float *part = A.part + (int64_t)blockIdx.y * L.gp_floats + L.gp[l];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        if (i >= n_ti) continue;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            if (j >= n_tj) continue;
            const int ti = TI * bi + i, tj = kb0 + 16 * jb + TJ * bj + j;
            if constexpr (F16) acc[i][j] *= unscale;
            *reinterpret_cast<f4 *>(part + ((int64_t)(ti * Ly.nkb + tj) * 64 + lane) * 4) = acc[i][j];
        }
        if (want_bias) {   // lane (i16, kq) summed samples 8 kq .. of feature i16
            float v = bsum[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 16) part[(int64_t)Ly.t_out * Ly.nkb * 256 + (TI * bi + i) * 16 + lane] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// the narrow wgrad jobs of an f16x3 step
// ------------------------------------------------------------------------------------------------
// mlp_wgrad_direct_kernel (mlp_train.hip) with two fp16 parts instead of fp32 MFMA: one wave per <= 4x4-tile block of dW
// and one chunk of samples; 32-sample stages of its 8 tile-rows arrive in LDS by DMA (16 KiB per workgroup), are gathered into the operand layout of v_mfma_f32_16x16x32_f16 (feature lane & 15,
// samples 8 (lane >> 4) + e), scaled by the per-layer powers of two of the wide kernel (xstat_index) and split; 3 MFMAs
// of 16 cycles per tile pair and 32 samples where the fp32 kernel spends 8 of 32 cycles.
__global__ __launch_bounds__(64) void mlp_wgrad_direct_f16_kernel(Plan P, TrainLayout L, WgradArgs A) {
    using Tm = Terms<2>;
    const int lane = threadIdx.x;
    // ---- decode the job: (narrow layer, segment, 4x4-tile block), like mlp_wgrad_direct_kernel ----
    int job = blockIdx.x, l = 0, s = 0, kb0 = 0, nbj = 1;
    for (l = 0; l < P.nlayers; ++l) {
        bool found = false;
        kb0 = 0;
        for (s = 0; s < P.layer[l].nseg; ++s) {
            nbj = (P.layer[l].seg[s].nkb + 3) / 4;
            const int cnt = wgrad_wide(P.layer[l], s) ? 0 : ((P.layer[l].t_out + 3) / 4) * nbj;
            if (job < cnt) { found = true; break; }
            job -= cnt;
            kb0 += P.layer[l].seg[s].nkb;
        }
        if (found) break;
    }
    const Layer &Ly = P.layer[l];
    const int bi = job / nbj, bj = job - bi * nbj;
    const int n_ti = min(4, Ly.t_out - 4 * bi), n_tj = min(4, Ly.seg[s].nkb - 4 * bj);
    const int64_t n = A.n;
    const int i16 = lane & 15, kq = lane >> 4;
    int first_seg = 0;  // the bias sums ride with the first non-empty input segment of the layer
    while (first_seg < Ly.nseg && Ly.seg[first_seg].nkb == 0) ++first_seg;
    const bool want_bias = (s == first_seg && bj == 0);
    const int xi = xstat_index(P, l, s);
    const int ex = 14 - (xi < 0 ? 0 : min(max(A.xstat[xi], -100), 100)), ey = 14 - min(max(A.ystat[l], -100), 100);
    const float sx = __builtin_ldexpf(1.f, ex), sy = __builtin_ldexpf(1.f, ey), unscale = __builtin_ldexpf(1.f, -(ex + ey));

    const int64_t begin = (int64_t)blockIdx.y * A.chunk;
    const int64_t end = min(n, begin + A.chunk);

    // ---- stage loader: 32 samples of the block's 4 dY and 4 X tile-rows per stage, by DMA into LDS (two 1 KiB pieces per
    // row, sample s of a piece at position s ^ ((s >> 3) & 1) like the wide kernel; ONE slot of 16 KiB, refilled behind the
    // MFMAs: ten workgroups per CU hide each other's loads better than a second slot would); rows the block does not have
    // re-load its first dY row
    extern __shared__ __attribute__((aligned(16))) float dring[];
    constexpr int ROW = 32 * 16;
    const float *row_src[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int64_t grow = (int64_t)(L.dy[l] + 4 * bi) * n * 16;
        const float *base = A.dy;
        if (r < 4) {
            if (r < n_ti) grow += (int64_t)r * n * 16;
        } else if (r - 4 < n_tj) {
            base = A.act;
            grow = (int64_t)(seg_act_row(P, L, l, s) + 4 * bj + (r - 4)) * n * 16;
        }
        row_src[r] = base + grow;
    }
    const int nstages = begin < end ? (int)((end - begin + 31) / 32) : 0;
    auto issue = [&](int stage) __attribute__((always_inline)) {
        const int q = lane >> 2, sp = q ^ ((q >> 3) & 1);
        float *slot = dring;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int64_t smp = min(begin + (int64_t)stage * 32 + 16 * sub + sp, n - 1);
#pragma unroll
            for (int r = 0; r < 8; ++r)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(row_src[r] + smp * 16 + (lane & 3) * 4),
                                                 (__attribute__((address_space(3))) void *)(slot + r * ROW + sub * 256), 16, 0, 0);
        }
    };

    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};

    if (nstages > 0) issue(0);
    for (int st = 0; st < nstages; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float *slot = dring;
        const int limit = (int)min((int64_t)32, end - (begin + (int64_t)st * 32));   // dY samples past the chunk end count as zero
        bf8 ap[4][2], bp[4][2];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int smp = 8 * kq + e;
                const int pos = (smp & 15) ^ (((smp & 15) >> 3) & 1);
                v[e] = slot[r * ROW + (smp >> 4) * 256 + pos * 16 + i16];
            }
            if (r < 4) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = (r < n_ti && 8 * kq + e < limit) ? v[e] : 0.f;
                    bsum[r] += v[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) split_pair_into<2, FMT_F16>(v[2 * e] * sy, v[2 * e + 1] * sy, ap[r], e);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) split_pair_into<2, FMT_F16>(v[2 * e] * sx, v[2 * e + 1] * sx, bp[r - 4], e);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the stage is in registers: refill the slot behind the MFMAs
        if (st + 1 < nstages) issue(st + 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < n_ti) {
#pragma unroll
                for (int t = 0; t < Tm::N; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < n_tj) acc[i][j] = mfma16<FMT_F16>(ap[i][Tm::A[t]], bp[j][Tm::B[t]], acc[i][j]);
            }
        }
    }
    // ---- write the partial of this (block, chunk): the format of mlp_wgrad_direct_kernel ----------------
    float *part = A.part + (int64_t)blockIdx.y * L.gp_floats + L.gp[l];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i >= n_ti) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j >= n_tj) continue;
            const int ti = 4 * bi + i, tj = kb0 + 4 * bj + j;
            *reinterpret_cast<f4 *>(part + ((int64_t)(ti * Ly.nkb + tj) * 64 + lane) * 4) = acc[i][j] * unscale;
        }
        if (want_bias) {   // lane (i16, kq) summed samples 8 kq .. of feature i16
            float v = bsum[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 16) part[(int64_t)Ly.t_out * Ly.nkb * 256 + (4 * bi + i) * 16 + lane] = v;
        }
    }
}

int launch_wgrad_direct_f16(const Plan &P, const TrainLayout &L, const WgradArgs &W, int jobs, int G, hipStream_t s) {
    hipLaunchKernelGGL(mlp_wgrad_direct_f16_kernel, dim3(jobs, G), dim3(64), 8 * 32 * 16 * 4, s, P, L, W);
    return check_launch("wgrad_direct_f16");
}

// ------------------------------------------------------------------------------------------------
// the wide wgrad jobs of an f16x3 step, operands converted once per workgroup
// ------------------------------------------------------------------------------------------------
// v_mfma_f32_32x32x16_f16, 16-sample stages: wave w brings tile-rows 4w .. 4w+3 of the job into LDS by DMA two stages
// ahead and turns these same rows - two 32-feature operand tiles - into scaled fp16 parts ONCE for the workgroup
// (mlp_wgrad_bf16_kernel splits in every wave: each dY tile four times and each X tile twice, and with three products per
// MAC that VALU work outweighs the MFMAs); every wave then reads its 4 + 2 operand tiles and runs 3 MFMAs per tile pair.
// Landing and operand areas are both double-buffered (2 x 32 KiB + 2 x 32 KiB): one barrier per stage.  The same kernel
// with three bf16 parts needs all 160 KiB and ran exactly as fast as mlp_wgrad_bf16_kernel<3> (energy-bound).
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int WC_THREADS = 8 * 64;
constexpr int WC_STAGE = 16;                            // samples per stage = one MFMA k-block
constexpr int WC_ROW_FLOATS = WC_STAGE * 16;            // one tile-row of a stage: 1 KiB
constexpr int WC_LAND_FLOATS = 32 * WC_ROW_FLOATS;      // 16 dY rows + 16 X rows: 32 KiB
constexpr int WC_LDS_BYTES = 2 * WC_LAND_FLOATS * 4 + 2 * 16 * 2 * 1024;

__global__ __launch_bounds__(WC_THREADS) void mlp_wgrad_f16_kernel(Plan P, TrainLayout L, WgradArgs A) {
    constexpr int NS = 2;
    constexpr int TI = 4, TJ = 2;   // accumulator tiles (32 x 32) per wave
    using Tm = Terms<NS>;
    extern __shared__ __attribute__((aligned(16))) float wring[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- decode the job: (wide layer, segment, group of 16 input k-blocks), like mlp_wgrad_kernel ----
    int job = blockIdx.x, l = 0, s = 0, kb0 = 0;
    for (l = 0; l < P.nlayers; ++l) {
        bool found = false;
        kb0 = 0;
        for (s = 0; s < P.layer[l].nseg; ++s) {
            const int cnt = wgrad_wide(P.layer[l], s) ? (P.layer[l].seg[s].nkb + 15) / 16 : 0;
            if (job < cnt) { found = true; break; }
            job -= cnt;
            kb0 += P.layer[l].seg[s].nkb;
        }
        if (found) break;
    }
    const Layer &Ly = P.layer[l];
    const int jb = job;
    const int n_rows_y = Ly.t_out, n_rows_x = min(16, Ly.seg[s].nkb - 16 * jb);   // 16-feature tile-rows of the job
    const int64_t n = A.n;
    int first_seg = 0;
    while (first_seg < Ly.nseg && Ly.seg[first_seg].nkb == 0) ++first_seg;
    const int bi = wave >> 2, bj = wave & 3;
    const bool active = 2 * TI * bi < n_rows_y && 2 * TJ * bj < n_rows_x;
    const int m32 = lane & 31, kg = lane >> 5;

    const int64_t begin = (int64_t)blockIdx.y * A.chunk;
    const int64_t end = min(n, begin + A.chunk);
    const int nstages = begin < end ? (int)((end - begin + WC_STAGE - 1) / WC_STAGE) : 0;
    // operand scales of the job (per layer, over all samples) and the scale of its result
    const int xi = xstat_index(P, l, s);   // (the statistic of this segment, not of the layer's hidden input - see mlp_wgrad_bf16_kernel)
    const int ex = 14 - (xi < 0 ? 0 : min(max(A.xstat[xi], -100), 100)), ey = 14 - min(max(A.ystat[l], -100), 100);
    const float sx = __builtin_ldexpf(1.f, ex), sy = __builtin_ldexpf(1.f, ey), unscale = __builtin_ldexpf(1.f, -(ex + ey));

    // ---- stage loader: this wave brings (and later converts) tile-rows 4*wave .. 4*wave+3 (rows 0..15 = dY, 16..31 =
    // X); rows the job does not have re-load row 0 of dY (finite filler), so that every wave issues exactly 4 pieces per
    // stage and one counted vmcnt serves all waves
    const float *row_src[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 4 * wave + q;
        int64_t grow = L.dy[l];
        if (r < 16) {
            if (r < n_rows_y) grow = L.dy[l] + r;
            row_src[q] = A.dy + grow * n * 16;
        } else if (r - 16 < n_rows_x) {
            row_src[q] = A.act + (int64_t)(seg_act_row(P, L, l, s) + 16 * jb + (r - 16)) * n * 16;
        } else {
            row_src[q] = A.dy + grow * n * 16;
        }
    }
    auto issue = [&](int stage) {
        // lane covers 16 B of a piece: position q = lane >> 2 (of 16), feature quad lane & 3; position q of row r holds
        // sample q ^ ((((q >> 3) & 1) << 1) | (r & 1)) (bank swizzle, see above; 4*wave + rr has the parity of rr)
        float *slot = wring + (stage & 1) * WC_LAND_FLOATS;
        const int q = lane >> 2;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int sp = q ^ ((((q >> 3) & 1) << 1) | (rr & 1));
            const int64_t smp = min(begin + (int64_t)stage * WC_STAGE + sp, n - 1);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(row_src[rr] + smp * 16 + (lane & 3) * 4),
                (__attribute__((address_space(3))) void *)(slot + (4 * wave + rr) * WC_ROW_FLOATS), 16, 0, 0);
        }
    };

    f16v acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float bsum[2] = {0.f, 0.f};   // bias sums of the two dY operand tiles this wave converts (waves 0-3)
    // operand area: [2 slots][16 operand tiles: 8 of dY, 8 of X][NS parts][64 lanes] x 16 B
    bf8 *const ops = reinterpret_cast<bf8 *>(wring + 2 * WC_LAND_FLOATS);
    constexpr int OPS_SLOT = 16 * NS * 64;

    // operand tiles 2*wave, 2*wave+1 of `stage`: lane (m32, kg) takes samples 8 kg .. 8 kg + 7 of feature m32 of the tile
    auto convert = [&](int stage) __attribute__((always_inline)) {
        const float *slot = wring + (stage & 1) * WC_LAND_FLOATS;
        bf8 *dst = ops + (stage & 1) * OPS_SLOT;
        const int limit = (int)min((int64_t)WC_STAGE, end - (begin + (int64_t)stage * WC_STAGE));  // dY: ragged chunk end
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int r = 4 * wave + 2 * t + (m32 >> 4);   // parity of r = m32 >> 4
            const float *row = slot + r * WC_ROW_FLOATS + (m32 & 15);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int smp = 8 * kg + e;
                v[e] = row[(smp ^ ((kg << 1) | (m32 >> 4))) * 16];
            }
            if (wave < 4) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = 8 * kg + e < limit ? v[e] : 0.f;
                    bsum[t] += v[e];
                }
            }
            const float sc = wave < 4 ? sy : sx;
            bf8 parts[NS];
#pragma unroll
            for (int e = 0; e < 4; ++e) split_pair_into<NS, FMT_F16>(v[2 * e] * sc, v[2 * e + 1] * sc, parts, e);
#pragma unroll
            for (int p = 0; p < NS; ++p) dst[((2 * wave + t) * NS + p) * 64 + lane] = parts[p];
        }
    };
    auto multiply = [&](int stage) __attribute__((always_inline)) {
        const bf8 *src = ops + (stage & 1) * OPS_SLOT;
        bf8 bpart[TJ][NS];
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int p = 0; p < NS; ++p) bpart[j][p] = src[((8 + TJ * bj + j) * NS + p) * 64 + lane];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            bf8 apart[NS];
#pragma unroll
            for (int p = 0; p < NS; ++p) apart[p] = src[((TI * bi + i) * NS + p) * 64 + lane];
#pragma unroll
            for (int t = 0; t < Tm::N; ++t)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8v, apart[Tm::A[t]]), __builtin_bit_cast(h8v, bpart[j][Tm::B[t]]), acc[i][j], 0, 0, 0);
        }
    };

    // (not __syncthreads(): its fence would also wait for the DMA in flight)
    auto ops_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // every wave issues 4 pieces per stage, in stage order: vmcnt(4k) leaves the k newest stages in flight
    if (nstages > 0) {
        issue(0);
        if (nstages > 1) issue(1);
        if (nstages > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        convert(0);
        if (nstages > 2) issue(2);   // into the rows just converted
        ops_barrier();
    }
    for (int st = 0; st < nstages; ++st) {
        if (st + 1 < nstages) {
            // this wave's rows of stage st+1 have landed (stage st+2 may still be in flight)
            if (st + 2 < nstages) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            convert(st + 1);
            if (st + 3 < nstages) issue(st + 3);   // into the rows just converted
        }
        if (active) multiply(st);
        ops_barrier();     // operands of stage st+1 complete; those of stage st free for stage st+2
    }
    // ---- write the partial of this (block, chunk): the [ti][tj][64 lanes][4] format of mlp_wgrad_kernel ----
    float *part = A.part + (int64_t)blockIdx.y * L.gp_floats + L.gp[l];
    if (s == first_seg && jb == 0 && wave < 4) {   // bias sums: lane (m32, kg) summed samples 8 kg .. of feature m32
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v = bsum[t];
            v += __shfl_xor(v, 32, 64);
            const int feat = 32 * (2 * wave + t) + m32;
            if (feat < n_rows_y * 16 && lane < 32) part[(int64_t)Ly.t_out * Ly.nkb * 256 + feat] = v;
        }
    }
    if (!active) return;
    // 32x32 accumulator: register 4 g + r of lane (m32, kg) is element (row 8 g + 4 kg + r, column m32) -> 16x16 tile
    // (row >> 4, column >> 4), lane (column & 15) + 16 ((row & 15) >> 2), register r
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ti = 2 * (TI * bi + i) + (g >> 1), tjl = 2 * (TJ * bj + j) + (m32 >> 4);
                if (ti >= n_rows_y || tjl >= n_rows_x) continue;
                const int tj = kb0 + 16 * jb + tjl;
                const int lane16 = (m32 & 15) + 16 * (2 * (g & 1) + kg);
                *reinterpret_cast<f4 *>(part + ((int64_t)(ti * Ly.nkb + tj) * 64 + lane16) * 4) =
                    f4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]} * unscale;
            }
        }
}

int launch_wgrad_wide_bf16(const Plan &P, const TrainLayout &L, const WgradArgs &W, int jobs, int G, int nsplit, hipStream_t s) {
    static LdsRaised r2, r3, rc16;   // per device
    int rc;
    if (nsplit == SNERF_SPLIT_F16X3) {
        if ((rc = raise_dynamic_lds(reinterpret_cast<const void *>(mlp_wgrad_f16_kernel), WC_LDS_BYTES, rc16, "wgrad_f16"))) return rc;
        hipLaunchKernelGGL(mlp_wgrad_f16_kernel, dim3(jobs, G), dim3(WC_THREADS), WC_LDS_BYTES, s, P, L, W);
    } else if (nsplit == 3) {
        if ((rc = raise_dynamic_lds(reinterpret_cast<const void *>(mlp_wgrad_bf16_kernel<3>), WB_LDS_BYTES, r3, "wgrad_bf16"))) return rc;
        hipLaunchKernelGGL(mlp_wgrad_bf16_kernel<3>, dim3(jobs, G), dim3(WB_THREADS), WB_LDS_BYTES, s, P, L, W);
    } else {
        if ((rc = raise_dynamic_lds(reinterpret_cast<const void *>(mlp_wgrad_bf16_kernel<2>), WB_LDS_BYTES, r2, "wgrad_bf16"))) return rc;
        hipLaunchKernelGGL(mlp_wgrad_bf16_kernel<2>, dim3(jobs, G), dim3(WB_THREADS), WB_LDS_BYTES, s, P, L, W);
    }
    return check_launch("wgrad_bf16");
}

static int plans_t(const snerf_mlp_desc *desc, Plan &P, const char *what) {
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "%s: desc is null", what);
    if (make_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "%s: %s", what, why);
    if (desc->width != 256) return fail(SNERF_E_BADARG, "%s: the split-bf16 path supports width 256 only", what);
    return SNERF_OK;
}

template <int NS, bool INPUT_GRAD, int FMT = FMT_BF16>
static int launch_dgrad_bf16(const BwdArgs &A, hipStream_t s) {
    constexpr int NW = 8;
    const int lds = 3 * slab16_bytes(NS);
    static LdsRaised raised;   // per device
    if (int rc = raise_dynamic_lds(reinterpret_cast<const void *>(mlp_bwd_bf16_kernel<256, NW, NS, INPUT_GRAD, FMT>), lds, raised,
                                   "mlp_bwd_bf16"))
        return rc;
    const int n_cu = device_cu_count("mlp_bwd_bf16");  // one persistent workgroup per CU
    if (n_cu < 1) return n_cu;
    const int64_t grid = (!INPUT_GRAD && A.n_tiles > n_cu) ? n_cu : A.n_tiles;
    if (grid > 0x7fffffffLL) return fail(SNERF_E_BADARG, "mlp_bwd_bf16: n too large");
    hipLaunchKernelGGL((mlp_bwd_bf16_kernel<256, NW, NS, INPUT_GRAD, FMT>), dim3((unsigned)grid), dim3(NW * 64), lds, s, A);
    return check_launch("mlp_bwd_bf16(dgrad)");
}

int launch_bwd_bf16(const snerf_mlp_desc *desc, const void *packed_t, int nsplit, const float *act,
                    const float *d_raw, int64_t n, float *dy, float *gpart, float *flat_grad, const float *x,
                    const float *dirs, int dirs_per_sample, int spr, float *d_x, float *d_dirs,
                    snerf_stream_t stream, bool accumulate) {
    Plan P;
    if (nsplit != 2 && nsplit != 3 && nsplit != SNERF_SPLIT_F16X3)
        return fail(SNERF_E_BADARG, "mlp_bwd_bf16: nsplit must be 2, 3 or %d (f16x3)", SNERF_SPLIT_F16X3);
    int rc = plans_t(desc, P, "mlp_bwd_bf16");
    if (rc) return rc;
    if (n < 0) return fail(SNERF_E_BADARG, "mlp_bwd_bf16: negative n");
    if (n == 0) return SNERF_OK;
    if (!packed_t || !act || !d_raw || !dy || !gpart || !flat_grad) return fail(SNERF_E_BADARG, "mlp_bwd_bf16: null pointer");
    if (!aligned(packed_t, 16) || !aligned(act, 16) || !aligned(d_raw, 16) || !aligned(dy, 16) || !aligned(gpart, 16))
        return fail(SNERF_E_ALIGN, "mlp_bwd_bf16: buffers must be 16-byte aligned");
    const bool input_grad = d_x != nullptr;
    if (input_grad) {
        if (!x || !d_dirs || (desc->use_dir && !dirs) || spr < 1)
            return fail(SNERF_E_BADARG, "mlp_bwd_bf16: input gradients need x, dirs, d_x, d_dirs");
        if (P.pos_nkb > 4 || P.dir_nkb > 2)
            return fail(SNERF_E_BADARG, "mlp_bwd_bf16: input gradients support at most 4 position / 2 direction encoder k-blocks");
    }
    hipStream_t s = (hipStream_t)stream;
    TrainLayout L;
    make_train_layout(P, L);
    const int nh = P.n_hidden;
    BwdArgs A{};
    A.packed_t = reinterpret_cast<const float *>(packed_t);
    A.act = act;
    A.d_raw = d_raw;
    A.dy = dy;
    A.n = n;
    A.n_hidden = nh;
    A.act_x1 = L.x[1];
    A.act_h2 = L.h2;
    A.act_mask = L.mask;
    A.dy_sig = L.dy[nh + 2];
    A.dy_din = L.dy[nh + 3];
    A.dy_dn0 = L.dy[nh + 4];
    A.dy_rgb = L.dy[nh + 5];
    A.x = x;
    A.dirs = dirs;
    A.d_x = d_x;
    A.d_dirs = d_dirs;
    A.dirs_per_sample = dirs_per_sample ? 1 : 0;
    A.spr = spr < 1 ? 1 : spr;
    A.skip_mask = desc->skip_mask;
    A.pos_L = desc->pos_freqs;
    A.pos_id = desc->pos_identity ? 1 : 0;
    A.pos_nkb = P.pos_nkb;
    A.dir_L = desc->dir_freqs;
    A.dir_id = desc->dir_identity ? 1 : 0;
    A.dir_nkb = P.dir_nkb;
    A.use_dir = desc->use_dir ? 1 : 0;
    A.total_slabs = bwd_total_slabs(P, input_grad, 32);
    A.n_tiles = (n + 8 * 16 - 1) / (8 * 16);
    A.dy_rows = L.dy_rows;
    const bool f16 = nsplit == SNERF_SPLIT_F16X3;
    if (f16 && hipMemsetAsync(dy + (int64_t)L.dy_rows * n * 16, 0x80, STAT_INTS * sizeof(int), s) != hipSuccess)
        return fail(SNERF_E_LAUNCH, "mlp_bwd_bf16: cannot reset the layer statistics");
    if (nsplit == SNERF_SPLIT_F16X3) rc = input_grad ? launch_dgrad_bf16<2, true, FMT_F16>(A, s) : launch_dgrad_bf16<2, false, FMT_F16>(A, s);
    else if (nsplit == 3) rc = input_grad ? launch_dgrad_bf16<3, true>(A, s) : launch_dgrad_bf16<3, false>(A, s);
    else rc = input_grad ? launch_dgrad_bf16<2, true>(A, s) : launch_dgrad_bf16<2, false>(A, s);
    if (rc) return rc;
    // wide jobs on the 16-bit matrix cores in the same format (f16x3: with per-layer scales - a per-sample scale cannot be
    // factored out of a contraction over samples), narrow jobs and the reduce in fp32
    return launch_wgrad(P, L, act, dy, n, gpart, flat_grad, s, nsplit, accumulate);
}

}  // namespace snerf

extern "C" int64_t snerf_mlp_packed_t_bf16_bytes(const snerf_mlp_desc *desc, int nsplit, int input_grad) {
    using namespace snerf;
    Plan P;
    if (nsplit != 2 && nsplit != 3 && nsplit != SNERF_SPLIT_F16X3)
        return fail(SNERF_E_BADARG, "mlp_packed_t_bf16_bytes: nsplit must be 2, 3 or %d (f16x3)", SNERF_SPLIT_F16X3);
    int rc = plans_t(desc, P, "mlp_packed_t_bf16_bytes");
    if (rc) return rc;
    return (int64_t)(bwd_total_slabs(P, input_grad != 0, 32) + SLAB_PAD) * slab16_bytes(nsplit == SNERF_SPLIT_F16X3 ? 2 : nsplit);
}

extern "C" int snerf_mlp_pack_t_bf16(const snerf_mlp_desc *desc, const float *params_flat, void *packed_t, int nsplit,
                                     int input_grad, snerf_stream_t stream) {
    using namespace snerf;
    Plan P;
    if (nsplit != 2 && nsplit != 3 && nsplit != SNERF_SPLIT_F16X3)
        return fail(SNERF_E_BADARG, "mlp_pack_t_bf16: nsplit must be 2, 3 or %d (f16x3)", SNERF_SPLIT_F16X3);
    int rc = plans_t(desc, P, "mlp_pack_t_bf16");
    if (rc) return rc;
    if (!params_flat || !packed_t) return fail(SNERF_E_BADARG, "mlp_pack_t_bf16: null pointer");
    if (!aligned(packed_t, 16)) return fail(SNERF_E_ALIGN, "mlp_pack_t_bf16: packed_t must be 16-byte aligned");
    BwdPlan B;
    make_bwd_plan(P, B, input_grad != 0, 32);
    const int fmt = nsplit == SNERF_SPLIT_F16X3 ? FMT_F16 : FMT_BF16, ns = fmt == FMT_F16 ? 2 : nsplit;
    if (fmt == FMT_F16)
        if ((rc = launch_wexp(P, ns, params_flat, packed_t, B.total_slabs, (hipStream_t)stream, "mlp_pack_t_bf16"))) return rc;
    hipLaunchKernelGGL(mlp_pack_t_bf16_kernel, dim3(B.total_slabs + SLAB_PAD), dim3(256), 0, (hipStream_t)stream, P, B, ns, fmt,
                       params_flat, reinterpret_cast<unsigned char *>(packed_t));
    return check_launch("mlp_pack_t_bf16");
}

extern "C" int snerf_mlp_bwd_bf16_f32(const snerf_mlp_desc *desc, const void *packed_t, int nsplit, const float *act,
                                      const float *d_raw, int64_t n, float *dy, float *gpart, float *flat_grad,
                                      snerf_stream_t stream) {
    return snerf::launch_bwd_bf16(desc, packed_t, nsplit, act, d_raw, n, dy, gpart, flat_grad, nullptr, nullptr, 0, 1,
                                  nullptr, nullptr, stream);
}

extern "C" int snerf_mlp_bwd_inputs_bf16_f32(const snerf_mlp_desc *desc, const void *packed_t, int nsplit, const float *act,
                                             const float *d_raw, const float *x, const float *dirs, int dirs_per_sample,
                                             int samples_per_ray, int64_t n, float *dy, float *gpart, float *flat_grad,
                                             float *d_x, float *d_dirs, snerf_stream_t stream) {
    if (!d_x) return snerf::fail(SNERF_E_BADARG, "mlp_bwd_inputs_bf16: d_x is null");
    return snerf::launch_bwd_bf16(desc, packed_t, nsplit, act, d_raw, n, dy, gpart, flat_grad, x, dirs, dirs_per_sample,
                                  samples_per_ray, d_x, d_dirs, stream);
}
