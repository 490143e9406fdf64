// Backward of the fused RenderRayNet (a2 backward; gradients w.r.t. all weights and biases).
//
// The reference gets these from autograd over 13 nn.Linear calls (solver/nerf_solver.py:83-87,
// loss.backward()).  Here a training step of one net is four kernels around HBM-resident, tile-row-major
// activation buffers (mlp_plan.h: TrainLayout):
//
//   mlp_fwd_kernel<.., TRAIN>   forward, additionally stores every layer input X_l           (mlp.hip)
//   mlp_bwd_kernel               dgrad: the forward pass of the TRANSPOSED network on d raw, with the same
//                                register-resident chaining (dX^T = W^T dY^T is again "accumulator layout =
//                                next B operand"); applies the ReLU masks from the stored X_l and stores
//                                every dY_l
//   mlp_wgrad_kernel             dW_l = dY_l X_l^T, db_l = sum_s dY_l: split-K MFMA GEMMs over the samples;
//                                operands are read from HBM/L2 straight into MFMA registers (the tile-row
//                                layout makes 4 samples x 16 features one 256 B wave access), each
//                                workgroup owns a <=64x64 block of one dW and one chunk of samples and
//                                writes a partial
//   mlp_wgrad_reduce_kernel      sums the partials in a fixed order (deterministic) and scatters from slot
//                                order to the reference's [out, in] weight layout
//
// All MFMA work is v_mfma_f32_16x16x4_f32 (exact fp32), like the forward.
#include <algorithm>
#include <type_traits>

#include "mlp_train_device.h"

namespace snerf {

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ------------------------------------------------------------------------------------------------
// transposed weight stream
// ------------------------------------------------------------------------------------------------
constexpr int PACK_T_PARTS = (SLAB_FLOATS + 255) / 256;   // workgroups per slab, one element per thread (mlp.hip: mlp_pack_kernel)
__global__ __launch_bounds__(256) void mlp_pack_t_kernel(Plan P, BwdPlan B, const float *__restrict__ params,
                                                         float *__restrict__ packed) {
    const int slab = blockIdx.x / PACK_T_PARTS;
    const int e = (blockIdx.x - slab * PACK_T_PARTS) * 256 + threadIdx.x;
    if (e >= SLAB_FLOATS) return;
    float *dst = packed + (int64_t)slab * SLAB_FLOATS;
    if (slab >= B.total_slabs) {
        dst[e] = 0.f;
        return;
    }
    int bi = 0;
    while (bi + 1 < B.nl && slab >= B.layer[bi + 1].first_slab) ++bi;
    const BwdLayer &Bl = B.layer[bi];
    const int64_t src = bwd_slab_src(P, Bl, slab - Bl.first_slab, e);
    dst[e] = src >= 0 ? params[src] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// dgrad
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void mask_into(f4 (&dst)[N], const f4 (&src)[N], const float *act, int row0, int64_t n,
                                          int64_t sample, int g) {
    f4 m[N];
#pragma unroll
    for (int t = 0; t < N; ++t) m[t] = load_tile(act, row0 + t, n, sample, g);
#pragma unroll
    for (int t = 0; t < N; ++t) {
        dst[t][0] = m[t][0] > 0.f ? src[t][0] : 0.f;
        dst[t][1] = m[t][1] > 0.f ? src[t][1] : 0.f;
        dst[t][2] = m[t][2] > 0.f ? src[t][2] : 0.f;
        dst[t][3] = m[t][3] > 0.f ? src[t][3] : 0.f;
    }
}

// The same from the 1-bit-per-output sign masks the training forward leaves behind the activation rows (store_mask,
// mlp_device.h): 8 bytes per lane and layer instead of a 1 KiB tile-row per 16 samples and tile - the dgrad then reads
// 0.3 GB instead of 3.6 GB per 786 k-sample launch.  The word is fetched before the layer's MFMA run and used after it.
template <int N>
__device__ __forceinline__ void mask_bits_into(f4 (&dst)[N], const f4 (&src)[N], uint4 w) {
#pragma unroll
    for (int t = 0; t < N; ++t) {
        const unsigned word = (t >> 3) == 0 ? w.x : (t >> 3) == 1 ? w.y : (t >> 3) == 2 ? w.z : w.w;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[t][r] = ((word >> (((t & 7) << 2) | r)) & 1u) ? src[t][r] : 0.f;
    }
}
// the mask words of layer output `idx` (store_mask, mlp_device.h): two per lane for nets of up to 16 tiles, four (a tile-row per
// mask) for the 32-tile nets - whose 16-tile directional branch has two words in that layout
template <int NTILES, bool ROW_PER_MASK>
__device__ __forceinline__ uint4 load_mask(const float *act, int mask_row, int idx, int64_t n, int64_t sample, int g) {
    if constexpr (NTILES > 16) {
        return *mask_ptr4(act, mask_row, idx, n, sample, g);
    } else if constexpr (ROW_PER_MASK) {
        const uint2 w = *reinterpret_cast<const uint2 *>(mask_ptr4(act, mask_row, idx, n, sample, g));
        return uint4{w.x, w.y, 0u, 0u};
    } else {
        const uint2 w = *mask_ptr(act, mask_row, idx, n, sample, g);
        return uint4{w.x, w.y, 0u, 0u};
    }
}

// INPUT_GRAD: additionally back-propagates into the network inputs (SmplNerfPipeline: the warped samples and
// their per-sample view directions are functions of the warp net, models/smpl_nerf_pipeline.py:49-56), for
// encoders of up to TPP position / TPD direction k-blocks (mlp_plan.h: bwd_pe_tiles - 4 / 2 for the default
// encoders, 8 / 8 for identity columns and up to 16 frequencies).
template <int WIDTH, int NWAVES, bool INPUT_GRAD, int TPP = 4, int TPD = 2>
__global__ __launch_bounds__(NWAVES * 64) void mlp_bwd_kernel(BwdArgs A) {
    constexpr int NT = NWAVES * 64;
    constexpr int T = WIDTH / 16;
    constexpr int TD = WIDTH / 32;
    extern __shared__ __attribute__((aligned(16))) float ring[];  // RING_BYTES (SNERF_LAUNCH_RING)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = lane >> 4;
    const int64_t sample = ((int64_t)blockIdx.x * NWAVES + wave) * 16 + (lane & 15);
    const bool valid = sample < A.n;
    const int64_t sc = valid ? sample : A.n - 1;
    const int nh = A.n_hidden;

    const f4 dr = *reinterpret_cast<const f4 *>(A.d_raw + sc * 4);
    const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
    // head gradients as tile-rows for the wgrad kernel (rows 0..2 = rgb, row 0 = sigma)
    if (valid) {
        store_tile(A.dy, A.dy_rgb, A.n, sample, g, g == 0 ? f4{dr[0], dr[1], dr[2], 0.f} : zero);
        store_tile(A.dy, A.dy_sig, A.n, sample, g, g == 0 ? f4{dr[3], 0.f, 0.f, 0.f} : zero);
    }

    using Pipe = PipeFor<WIDTH, NT>;
    Pipe pipe;
    pipe.prologue(A.packed_t, ring, tid);

    f4 ind[TD], accd[TD];
    {  // rgb_out_layer^T, then the ReLU mask of directional_net[0] (models/render_ray_net.py:58-60)
        const uint4 mw = load_mask<TD, (T > 16)>(A.act, A.act_mask, nh + 1, A.n, sc, g);
        LayerRun<TD, NT, Pipe> run(pipe, lane);
        run.init(accd);
        run.step(g == 0 ? f4{dr[0], dr[1], dr[2], 0.f} : zero, accd);
        run.finish();
        mask_bits_into(ind, accd, mw);
        if (valid) store_tiles(A.dy, A.dy_dn0, A.n, sample, g, ind);
    }
    {  // directional_net[0]^T; directional_input has no activation (:54-57)
        LayerRun<TD, NT, Pipe> run(pipe, lane);
        run.init(accd);
#pragma unroll
        for (int kb = 0; kb < TD; ++kb) run.step(ind[kb], accd);
        run.finish();
        copy_into(ind, accd);
        if (valid) store_tiles(A.dy, A.dy_din, A.n, sample, g, ind);
    }
    if (INPUT_GRAD && A.use_dir && A.dir_nkb > 0) {
        // d (direction encoding) = directional_input[:, W:]^T d h1, then encoder and normalisation backward
        f4 ddpe[TPD];
        LayerRun<TPD, NT, Pipe> run(pipe, lane);
        run.init(ddpe);
#pragma unroll
        for (int kb = 0; kb < TD; ++kb) run.step(ind[kb], ddpe);
        run.finish();
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
    f4 in[T], acc[T];
    f4 dpe[TPP];
#pragma unroll
    for (int t = 0; t < TPP; ++t) dpe[t] = f4{0.f, 0.f, 0.f, 0.f};
    // position-encoding columns of forward layer l (layer 0 or a skip layer), given d Y_l in `in`
    auto pe_columns = [&](int l) {
        if (!INPUT_GRAD || A.pos_nkb <= 0) return;
        if (!(l == 0 || ((A.skip_mask >> (l - 1)) & 1u))) return;
        f4 t[TPP];
        LayerRun<TPP, NT, Pipe> run(pipe, lane);
        run.init(t);
#pragma unroll
        for (int kb = 0; kb < T; ++kb) run.step(in[kb], t);
        run.finish();
#pragma unroll
        for (int q = 0; q < TPP; ++q) dpe[q] += t[q];
    };
    {  // d o = directional_input[:, :W]^T d h1 + sigma_out_layer^T d sigma; additional layer has no activation (:51-52)
        LayerRun<T, NT, Pipe> run(pipe, lane);
        run.init(acc);  // aux block = sigma head weights
#pragma unroll
        for (int t = 0; t < T; ++t) {
            acc[t][0] *= dr[3];
            acc[t][1] *= dr[3];
            acc[t][2] *= dr[3];
            acc[t][3] *= dr[3];
        }
#pragma unroll
        for (int kb = 0; kb < TD; ++kb) run.step(ind[kb], acc);
        run.finish();
        copy_into(in, acc);
        if (valid) store_tiles(A.dy, (nh + 1) * T, A.n, sample, g, in);
    }
    (void)0;
    // additional^T, positional_net[nh-1]^T ... positional_net[0]^T: forward layer l+1 transposed yields
    // d X_{l+1}; masking with X_{l+1} > 0 gives d Y of forward layer l (:46-50)
    for (int l = nh; l >= 0; --l) {
        const uint4 mw = load_mask<T, (T > 16)>(A.act, A.act_mask, l, A.n, sc, g);   // lands behind the layer's MFMAs
        LayerRun<T, NT, Pipe> run(pipe, lane);
        run.init(acc);
#pragma unroll
        for (int kb = 0; kb < T; ++kb) run.step(in[kb], acc);
        run.finish();
        mask_bits_into(in, acc, mw);
        if (valid) store_tiles(A.dy, l * T, A.n, sample, g, in);
        pe_columns(l);
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
    pipe.drain();
}

// ------------------------------------------------------------------------------------------------
// wgrad
// ------------------------------------------------------------------------------------------------
// One workgroup = 16 waves = one (layer, input segment, <=16 input k-blocks) job x one chunk of samples: the whole
// <=256x256 block of dW.  Wave (bi, bj) owns the 64x64 sub-block of output tiles 4bi.. x input tiles 4bj.. (4x4
// accumulator tiles, 64 VGPRs).  Operands reach the matrix cores through LDS: a stage is 16 samples of every
// tile-row of the job (<=16 rows of dY, <=16 of X; one tile-row x 16 samples = 1 KiB contiguous in the tile-row
// layout = one global_load_lds per wave), 4-slot ring, one barrier per stage = per 64 MFMAs per wave.  Every byte
// of dY and X is read from HBM once per job instead of once per 64x64 block (4x fewer operand bytes than the
// one-wave-per-block kernel this replaces, which ran at 51 % of the fp32 MFMA peak on L2/HBM operand traffic).
constexpr int WL_WAVES = 16, WL_THREADS = WL_WAVES * 64;
constexpr int WL_STAGE = 16;                      // samples per stage (4 MFMA k-steps)
constexpr int WL_ROW_FLOATS = WL_STAGE * 16;      // one tile-row of a stage: 1 KiB
constexpr int WL_XROW0 = 32, WL_YROW0 = 36;         // LDS rows of the folded operands: <= 4 extra X rows, one extra dY row
constexpr int WL_SLOT_FLOATS = 37 * WL_ROW_FLOATS;  // 16 dY rows + 16 X rows + the folded rows (mlp_train_device.h)
constexpr int WL_SLOTS = 4;                          // ring depth: up to WL_SLOTS - 2 stages in flight behind the one awaited
// (Measured r03: 32-sample stages in a double buffer - half the barriers per MFMA - 4.79 instead of 4.65 ms per launch: the
// ~20 % of a stage that does not overlap with the MFMAs is not the barrier.  The 8-tile job lasts 0.7 of a 16-tile one for
// the same 32 KiB per stage, i.e. a stage cannot stream in faster than ~0.7 of a 16-tile stage's MFMA time: the kernel runs
// close to what the tile-row reads (1 KiB pieces, 2 TB/s aggregate) allow.)
constexpr int WL_LDS_BYTES = WL_SLOTS * WL_SLOT_FLOATS * 4;

// the stage loop and the epilogue for one wave that owns TI x TJ accumulator tiles
// what rides with a wide job (mlp_train_device.h: wgrad_kind == 1)
struct WgradFold {
    int ex;            // extra X tile-rows: the folded segment's k-blocks (0 = none)
    int ex_tj0;        // ... whose partial tiles are columns ex_tj0 .. of the SAME layer
    int ey_layer;      // forward layer whose single dY tile-row rides as extra dY row (the sigma head), or -1
    const float *src;  // this wave's third piece per stage (extra X row `wave` < ex, or the extra dY row for wave 4), or null
};

template <int TI, int TJ, bool FX = false, bool FY = false>
__device__ __forceinline__ void wgrad_wave(const Plan &P, const Layer &Ly, const TrainLayout &L, const WgradArgs &A, float *ring,
                                           int l, int kb0, int jb, int n_rows_y, int n_rows_x, bool bias_job,
                                           const float *const (&row_src)[2], const WgradFold &F, int wave, int lane, int ti0 = 0) {
    const int nbj = (n_rows_x + TJ - 1) / TJ, nbi = (n_rows_y + TI - 1) / TI;
    const int bi = wave / nbj, bj = wave - bi * nbj;
    const bool active = bi < nbi;
    const int n_ti = active ? min(TI, n_rows_y - TI * bi) : 0, n_tj = min(TJ, n_rows_x - TJ * bj);
    const bool want_bias = bias_job && bj == 0;
    const int kslot = lane >> 4;
    const int64_t n = A.n;
    const int64_t begin = (int64_t)blockIdx.y * A.chunk;
    const int64_t end = min(n, begin + A.chunk);
    const int nstages = begin < end ? (int)((end - begin + WL_STAGE - 1) / WL_STAGE) : 0;
    // folded tiles of this wave (compile-time variants: the plain jobs keep their instruction stream):
    // (dY rows TI*bi .., extra X row bj) and (extra dY row, X row TJ*bj + bi)
    const bool has3 = (FX || FY) && F.src != nullptr;
    const bool fx = FX && active && bj < F.ex, fy = FY && active && bi < TJ && TJ * bj + bi < n_rows_x;

    auto issue = [&](int stage, int slot) {
        // lane covers 16 B: features 4*(lane&3).. of sample (lane>>2) of the stage; clamped at the end of the buffer
        const int64_t smp = min(begin + (int64_t)stage * WL_STAGE + (lane >> 2), n - 1);
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(row_src[q] + smp * 16 + (lane & 3) * 4),
                (__attribute__((address_space(3))) void *)(ring + slot * WL_SLOT_FLOATS + (2 * wave + q) * WL_ROW_FLOATS), 16, 0, 0);
        if ((FX || FY) && has3)   // wave-uniform: the folded rows are brought by waves 0 .. 4
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(F.src + smp * 16 + (lane & 3) * 4),
                (__attribute__((address_space(3))) void *)(ring + slot * WL_SLOT_FLOATS + (WL_XROW0 + wave) * WL_ROW_FLOATS), 16, 0, 0);
    };

    f4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    float bsum[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) bsum[i] = 0.f;
    f4 accx[TI], accy = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < TI; ++i) accx[i] = f4{0.f, 0.f, 0.f, 0.f};
    float ysum = 0.f;

    // stage st lives in slot st % WL_SLOTS.  A wave issues exactly two (three with a folded row) pieces per stage, so
    // `vmcnt(2k)` (`vmcnt(3k)`) leaves its k newest stages in flight.
    auto wait_landed = [&](int k) {  // k = stages allowed to stay in flight, 0 .. WL_SLOTS - 2
        if ((FX || FY) && has3) {
            if (k >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (k == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (k >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (k == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    // one stage (16 samples = 4 k-steps) of this wave's tiles.  MASKED: tiles past the edge of the job run with a = 0 and
    // are not stored (their LDS rows hold finite filler data), samples past `end` contribute a = 0.
    // (Measured r04 and rejected: each wave issuing its pieces of stage st + 3 in front of k-step `wave >> 2` instead of at the
    // top of the stage - so that only one of the four waves of a SIMD sits in its DMA issue at a time - 4.63 -> 6.19 ms per
    // launch: the issue inside the pinned LDS-read / MFMA groups costs far more than the collision at the stage top.)
    auto stage_body = [&](auto masked_c, auto bias_c, int slot, int64_t s0) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_c)::value, BIAS = decltype(bias_c)::value;
        const float *ya = ring + slot * WL_SLOT_FLOATS + (TI * bi) * WL_ROW_FLOATS + lane;
        const float *xb = ring + slot * WL_SLOT_FLOATS + (16 + TJ * bj) * WL_ROW_FLOATS + lane;
        if constexpr (!MASKED) {
            // Unmasked stage: the operands of k-step s + 1 are read from LDS BEFORE the MFMAs of k-step s issue.  The four
            // waves of a SIMD leave every barrier in lock-step and the round-robin matrix pipe keeps them there: with
            // read -> wait -> 16 MFMAs per step all four sat in the LDS round trip at the same moment, every k-step.
            float a[2][TI], b[2][TJ];
            auto read = [&](int step, float (&av)[TI], float (&bv)[TJ]) __attribute__((always_inline)) {
#pragma unroll
                for (int t = 0; t < TI; ++t) av[t] = ya[t * WL_ROW_FLOATS + 64 * step];
#pragma unroll
                for (int t = 0; t < TJ; ++t) bv[t] = xb[t * WL_ROW_FLOATS + 64 * step];
            };
            read(0, a[0], b[0]);
#pragma unroll
            for (int step = 0; step < WL_STAGE / 4; ++step) {
                const int cur = step & 1;
                if (step + 1 < WL_STAGE / 4) read(step + 1, a[cur ^ 1], b[cur ^ 1]);
#pragma unroll
                for (int i = 0; i < TI; ++i) {
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
                    if constexpr (BIAS) bsum[i] += a[cur][i];
                }
                if (FX && fx) {   // wave-uniform: this wave's dY rows x folded X row bj
                    const float bx = ring[slot * WL_SLOT_FLOATS + (WL_XROW0 + bj) * WL_ROW_FLOATS + lane + 64 * step];
#pragma unroll
                    for (int i = 0; i < TI; ++i) accx[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][i], bx, accx[i], 0, 0, 0);
                }
                if (FY && fy) {   // folded dY row x this wave's X row bi (read again: cheaper than selecting among b[])
                    const float ay = ring[slot * WL_SLOT_FLOATS + WL_YROW0 * WL_ROW_FLOATS + lane + 64 * step];
                    const float by = xb[bi * WL_ROW_FLOATS + 64 * step];
                    accy = __builtin_amdgcn_mfma_f32_16x16x4f32(ay, by, accy, 0, 0, 0);
                    if constexpr (BIAS) ysum += ay;
                }
                // pin: the LDS reads of the next step first, then this step's MFMAs; nothing crosses the step boundary
                __builtin_amdgcn_sched_group_barrier(0x100, TI + TJ, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, TI * TJ + TI + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
#pragma unroll
        for (int step = 0; step < WL_STAGE / 4; ++step) {
            bool ok = true;
            if constexpr (MASKED) ok = s0 + 4 * step + kslot < end;
            float a[TI], b[TJ];
#pragma unroll
            for (int t = 0; t < TI; ++t) {
                const float va = ya[t * WL_ROW_FLOATS + 64 * step];
                if constexpr (MASKED) a[t] = (t < n_ti && ok) ? va : 0.f;
                else a[t] = va;
            }
#pragma unroll
            for (int t = 0; t < TJ; ++t) b[t] = xb[t * WL_ROW_FLOATS + 64 * step];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
                if constexpr (BIAS) bsum[i] += a[i];
            }
            if (FX && fx) {   // wave-uniform: this wave's dY rows x folded X row bj
                const float bx = ring[slot * WL_SLOT_FLOATS + (WL_XROW0 + bj) * WL_ROW_FLOATS + lane + 64 * step];
#pragma unroll
                for (int i = 0; i < TI; ++i) accx[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bx, accx[i], 0, 0, 0);
            }
            if (FY && fy) {   // folded dY row x this wave's X row bi (read again: cheaper than selecting among b[])
                const float va = ring[slot * WL_SLOT_FLOATS + WL_YROW0 * WL_ROW_FLOATS + lane + 64 * step];
                const float ay = (!MASKED || ok) ? va : 0.f;
                const float by = xb[bi * WL_ROW_FLOATS + 64 * step];
                accy = __builtin_amdgcn_mfma_f32_16x16x4f32(ay, by, accy, 0, 0, 0);
                if constexpr (BIAS) ysum += ay;
            }
            // keep the k-steps apart: without the masks nothing stops the scheduler from hoisting the LDS reads of all four
            // steps to the top (32 more live registers: spills at the 128 the 16-wave workgroup allows)
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    static_assert(WL_SLOTS == 4, "wait_landed covers k <= 2");
    if (nstages > 0) {
#pragma unroll
        for (int q = 0; q < WL_SLOTS - 1; ++q)
            if (q < nstages) issue(q, q);
        wait_landed(min(nstages, WL_SLOTS - 1) - 1);
        __builtin_amdgcn_s_barrier();
    }
    // The matrix pipe does not co-issue with the vector ALU in fp32 (SQ_VALU_MFMA_COEXEC_CYCLES = 0): every vector
    // instruction between two MFMAs is matrix time lost (r03: 0.89 per MFMA here = 11 % of the kernel).  So the common
    // stages - all TI tiles real, all 16 samples inside the chunk - run without the per-value masks, and only the four waves
    // that own the bias sums add them up; the masked variant serves the last stage of the buffer and the edge waves of
    // odd-shaped jobs.  One loop per variant (a loop that chooses per stage makes the register allocator give up the
    // in-place accumulators: 500 spilled registers).
    int slot = 0;
    auto run = [&](auto masked_c, auto bias_c, int st_begin, int st_end) __attribute__((always_inline)) {
        for (int st = st_begin; st < st_end; ++st) {
            // the slot of stage st-1 was freed by the barrier that ended the previous iteration (st = 0: the one slot the
            // prologue left empty)
            if (st + WL_SLOTS - 1 <= nstages - 1) issue(st + WL_SLOTS - 1, slot == 0 ? WL_SLOTS - 1 : slot - 1);
            if (active) stage_body(masked_c, bias_c, slot, begin + (int64_t)st * WL_STAGE);
            if (st + 1 < nstages) {
                // stage st+1 landed (this wave's pieces); the stages issued after it may stay in flight; everyone is done
                // reading `slot`
                wait_landed(min(st + WL_SLOTS - 1, nstages - 1) - (st + 1));
                __builtin_amdgcn_s_barrier();
            }
            slot = slot == WL_SLOTS - 1 ? 0 : slot + 1;
        }
    };
    // stages [0, nfast) lie inside the chunk with all their samples
    // (begin < end: a chunk behind the end of the buffer has no stages at all.  r04: without this the difference went
    // negative and the masked loop ran stages -k .. -1 - a = 0 against whatever the LDS held, which is 0 unless that is a NaN:
    // the first process on a freshly booted GPU got NaN gradients, everybody else the right ones.  tools/ab/nan_hunt.py)
    const int nfast = (begin < end && (!active || n_ti == TI)) ? (int)((end - begin) / WL_STAGE) : 0;
    if (want_bias) run(std::false_type{}, std::true_type{}, 0, nfast);
    else run(std::false_type{}, std::false_type{}, 0, nfast);
    run(std::true_type{}, std::true_type{}, nfast, nstages);
    if (!active) return;
    // ---- write the partial of this (block, chunk) ---------------------------------------------------
    float *part = A.part + (int64_t)blockIdx.y * L.gp_floats + L.gp[l];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        if (i >= n_ti) continue;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            if (j >= n_tj) continue;
            const int ti = ti0 + TI * bi + i, tj = kb0 + 16 * jb + TJ * bj + j;   // (ti0: the job's first output tile)
            *reinterpret_cast<f4 *>(part + ((int64_t)(ti * Ly.nkb + tj) * 64 + lane) * 4) = acc[i][j];
        }
        if (want_bias) {
            float v = bsum[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 16) part[(int64_t)Ly.t_out * Ly.nkb * 256 + (ti0 + TI * bi + i) * 16 + lane] = v;
        }
        if (FX && fx)   // folded segment of the same layer: tile column ex_tj0 + bj
            *reinterpret_cast<f4 *>(part + ((int64_t)((TI * bi + i) * Ly.nkb + F.ex_tj0 + bj) * 64 + lane) * 4) = accx[i];
    }
    if (FY && fy) {     // the folded layer's (one output tile) partial: tile column = this wave's X row
        const Layer &Le = P.layer[F.ey_layer];
        float *pe = A.part + (int64_t)blockIdx.y * L.gp_floats + L.gp[F.ey_layer];
        const int tj = TJ * bj + bi;
        *reinterpret_cast<f4 *>(pe + ((int64_t)tj * 64 + lane) * 4) = accy;
        if (tj == 0) {   // its bias sums
            float v = ysum;
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 16) pe[(int64_t)Le.t_out * Le.nkb * 256 + lane] = v;
        }
    }
}

__global__ __launch_bounds__(WL_THREADS) void mlp_wgrad_kernel(Plan P, TrainLayout L, WgradArgs A) {
    extern __shared__ __attribute__((aligned(16))) float ring[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- decode the job: (layer, segment, group of 16 input k-blocks) --------------------------------
    int job = blockIdx.x, l = 0, s = 0, kb0 = 0;
    for (l = 0; l < P.nlayers; ++l) {
        bool found = false;
        kb0 = 0;
        for (s = 0; s < P.layer[l].nseg; ++s) {
            const int cnt = wgrad_wide_jobs(P.layer[l], s);
            if (job < cnt) { found = true; break; }
            job -= cnt;
            kb0 += P.layer[l].seg[s].nkb;
        }
        if (found) break;
    }
    const Layer &Ly = P.layer[l];
    const int nkg = (Ly.seg[s].nkb + 15) / 16;
    const int ib = job / nkg, jb = job - ib * nkg;         // output tiles 16*ib .., k-blocks 16*jb .. of the segment
    const int ti0 = 16 * ib;
    const int n_rows_y = min(16, Ly.t_out - ti0), n_rows_x = min(16, Ly.seg[s].nkb - 16 * jb);
    const int64_t n = A.n;
    int first_seg = 0;  // the bias sums ride with the first non-empty input segment of the layer
    while (first_seg < Ly.nseg && Ly.seg[first_seg].nkb == 0) ++first_seg;
    const bool bias_job = (s == first_seg && jb == 0);

    // ---- stage loader: this wave brings LDS rows 2*wave, 2*wave+1 (rows 0..15 = dY, 16..31 = X) ----------
    // rows the job does not have re-load row 0 of dY, so that every wave issues exactly two pieces per stage
    // and one counted vmcnt serves all waves
    const float *row_src[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = 2 * wave + q;
        int64_t grow = L.dy[l] + ti0;
        if (r < 16) {
            if (r < n_rows_y) grow = L.dy[l] + ti0 + r;
            row_src[q] = A.dy + grow * n * 16;
        } else if (r - 16 < n_rows_x) {
            row_src[q] = A.act + (int64_t)(seg_act_row(P, L, l, s) + 16 * jb + (r - 16)) * n * 16;
        } else {
            row_src[q] = A.dy + grow * n * 16;
        }
    }
    // ---- what rides with this job (first group of the layer's first wide segment only) ------------------------------
    WgradFold F{0, 0, -1, nullptr};
    if (A.fold && jb == 0 && ib == 0 && s == wgrad_first_wide_seg(Ly)) {
        const int xs = wgrad_fold_xseg(P, l, A.fold);
        if (xs >= 0) {
            F.ex = Ly.seg[xs].nkb;
            F.ex_tj0 = 0;
            for (int q = 0; q < xs; ++q) F.ex_tj0 += Ly.seg[q].nkb;
            if (wave < F.ex) F.src = A.act + (int64_t)(seg_act_row(P, L, l, xs) + wave) * n * 16;
        }
        if (l == P.n_hidden + 3 && wgrad_fold_sigma(P)) {
            F.ey_layer = P.n_hidden + 2;
            if (wave == WL_YROW0 - WL_XROW0) F.src = A.dy + (int64_t)L.dy[F.ey_layer] * n * 16;
        }
    }
    // wave block shape: the job's <=16 x <=16 tiles are cut so that (up to) all 16 waves own a block
    const int ti = n_rows_y > 8 ? 4 : (n_rows_y > 4 ? 2 : 1), tj = n_rows_x > 8 ? 4 : (n_rows_x > 4 ? 2 : 1);
    // the folding variants (mlp_train_device.h guarantees these shapes: 16 or 8 output tiles, 16 input k-blocks)
    if (F.ex > 0 || F.ey_layer >= 0) {
        const bool x = F.ex > 0, y = F.ey_layer >= 0;
        if (ti == 2 && tj == 4 && x && y)
            return wgrad_wave<2, 4, true, true>(P, Ly, L, A, ring, l, kb0, jb, n_rows_y, n_rows_x, bias_job, row_src, F, wave, lane);
        if (ti == 2 && tj == 4 && !x && y)
            return wgrad_wave<2, 4, false, true>(P, Ly, L, A, ring, l, kb0, jb, n_rows_y, n_rows_x, bias_job, row_src, F, wave, lane);
        if (ti == 2 && tj == 4 && x && !y)
            return wgrad_wave<2, 4, true, false>(P, Ly, L, A, ring, l, kb0, jb, n_rows_y, n_rows_x, bias_job, row_src, F, wave, lane);
        __builtin_trap();   // unreachable: wgrad_kind only folds into these shapes
    }
#define SNERF_WG_CASE(TI_, TJ_)                                                                                       \
    if (ti == TI_ && tj == TJ_)                                                                                       \
        return wgrad_wave<TI_, TJ_>(P, Ly, L, A, ring, l, kb0, jb, n_rows_y, n_rows_x, bias_job, row_src, F, wave, lane, ti0);
    SNERF_WG_CASE(4, 4)
    SNERF_WG_CASE(4, 2)
    SNERF_WG_CASE(4, 1)
    SNERF_WG_CASE(2, 4)
    SNERF_WG_CASE(2, 2)
    SNERF_WG_CASE(2, 1)
    SNERF_WG_CASE(1, 4)
    SNERF_WG_CASE(1, 2)
    SNERF_WG_CASE(1, 1)
#undef SNERF_WG_CASE
}

// The narrow (layer, segment) pairs: a workgroup owns a <= 4x4-tile block of dW and one chunk of samples; operands go
// straight from L2/HBM into MFMA registers (4 samples x 16 features = one 256 B access per tile-row), WG_PREFETCH k-steps
// in flight per wave; no LDS in the loop, no barrier.
// r03: the chunk is split over the WG_WAVES waves of the workgroup (quarter chunks, summed through LDS at the end): the
// single-wave version put 1792 unequal waves on 1024 SIMDs - the launch lasted as long as the SIMDs that drew two long
// jobs (42 % matrix-pipe busy); 7168 quarter waves balance.  And the loop is free of per-value vector work (tile / sample
// masks, 64-bit address arithmetic: 3.2 vector instructions per MFMA before, and fp32 MFMAs do not co-issue with the vector
// ALU): full k-steps run unmasked from wave-uniform row bases + one running 32-bit lane offset; masks only in the tail.
constexpr int WG_WAVES = 4, WG_THREADS = WG_WAVES * 64;

// buffer resource over this wave's segment of one tile-row: wave-uniform base in SGPRs, so a load is `buffer_load_dword v,
// voff, s[rsrc] offen offset:imm` - no per-load 64-bit vector address arithmetic (the flat form cost 8 x v_lshl_add_u64 per
// k-step) - and the hardware range check (offset >= num_records returns 0) IS the sample mask: k-steps and lanes past the
// end of the segment read zeros, so the loop needs neither a tail nor a per-value select.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const float *base, int bytes) {
    // the base IS wave-uniform; say so, or the resource lands in vector registers and every load becomes a waterfall loop
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float *>(((uint64_t)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
template <int IMM>
__device__ __forceinline__ float row_load(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff + IMM, 0, 0));
}

// (Measured r03: a third instantiation with the rgb head's 1 x 4 tiles as compile-time counts - its loop then has no
// predicates and no scratch traffic - made the launch SLOWER, 0.565 -> 0.587 ms: the kernel is bound by its operand stream,
// and the two rgb blocks per chunk are not its tail.)
template <int WG_PREFETCH, bool FULL>   // FULL: all 4 x 4 tiles of the block are real
__device__ __forceinline__ void wgrad_direct_wave(const float *const (&ya)[4], const float *const (&xb)[4], int n_ti, int n_tj,
                                                  int64_t wb, int64_t we, int lane, f4 (&acc)[4][4], float (&bsum)[4]) {
    const int bytes = (int)(we - wb) * 64;               // this wave's samples of a tile-row (a quarter chunk: far below 2 GB)
    __amdgpu_buffer_rsrc_t ry[4], rx[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        ry[t] = row_rsrc(ya[t], bytes);
        rx[t] = row_rsrc(xb[t], bytes);
    }
    // lane (i = lane&15, kslot = lane>>4) reads feature i of sample s0 + kslot: byte offset (s0 - wb)*64 + lane*4 from the
    // row base; k-step p of a group sits IMM = 256 p bytes further
    auto load_ab = [&](auto imm_c, unsigned voff, float (&a)[4], float (&b)[4]) __attribute__((always_inline)) {
        constexpr int IMM = decltype(imm_c)::value;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = (FULL || t < n_ti) ? row_load<IMM>(ry[t], voff) : 0.f;
            b[t] = (FULL || t < n_tj) ? row_load<IMM>(rx[t], voff) : 0.f;
        }
    };
    // software pipeline over groups of WG_PREFETCH k-steps (4 samples each), one group ahead; ONE loop (more than one loop
    // that updates `acc` costs the in-place accumulators), the prefetch of the group behind the last one reads zeros
    const int ngroups = ((int)(we - wb) + 4 * WG_PREFETCH - 1) / (4 * WG_PREFETCH);
    unsigned voff = (unsigned)lane * 4u;
    float ra[WG_PREFETCH][4], rb[WG_PREFETCH][4];
    static_for<0, WG_PREFETCH>([&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        load_ab(std::integral_constant<int, 256 * p>{}, voff, ra[p], rb[p]);
    });
    for (int g = 0; g < ngroups; ++g) {
        voff += 256u * WG_PREFETCH;
        static_for<0, WG_PREFETCH>([&](auto pc) __attribute__((always_inline)) {
            constexpr int p = decltype(pc)::value;
            float a0[4], b0[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a0[t] = ra[p][t];
                b0[t] = rb[p][t];
            }
            load_ab(std::integral_constant<int, 256 * p>{}, voff, ra[p], rb[p]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (FULL || i < n_ti) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (FULL || j < n_tj) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i], b0[j], acc[i][j], 0, 0, 0);
                    bsum[i] += a0[i];   // (unconditional: the four adds cost less than a second loop variant)
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // one k-step at a time: hoisting all 32 loads of a group spills
        });
    }
}

// everything behind the job decode, one instantiation per FULL (the accumulators of the two variants never meet)
// gp_off: floats from the start of this chunk's partial to the layer's block; lnkb / lt_out: the layer's k-blocks and output tiles
template <int WG_PREFETCH, bool FULL>
__device__ __forceinline__ void wgrad_direct_block(float *part_chunk, int gp_off, int lnkb, int lt_out, float *s_acc,
                                                   const float *const (&ya)[4], const float *const (&xb)[4], int kb0,
                                                   int bi, int bj, int n_ti, int n_tj, int64_t wb, int64_t we, bool want_bias,
                                                   int wave, int lane) {
    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    wgrad_direct_wave<WG_PREFETCH, FULL>(ya, xb, n_ti, n_tj, wb, we, lane, acc, bsum);

    // ---- sum the waves' quarters in a fixed order (wave 0 + 1 + 2 + 3) ------------------------------------------
    constexpr int PER_WAVE = 16 * 256 + 4 * 64;
    if (wave > 0) {
        float *dst = s_acc + (wave - 1) * PER_WAVE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<f4 *>(dst + ((i * 4 + j) * 64 + lane) * 4) = acc[i][j];
            dst[16 * 256 + i * 64 + lane] = bsum[i];
        }
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < WG_WAVES - 1; ++w) {
        const float *src = s_acc + w * PER_WAVE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] += *reinterpret_cast<const f4 *>(src + ((i * 4 + j) * 64 + lane) * 4);
            bsum[i] += src[16 * 256 + i * 64 + lane];
        }
    }
    // ---- write the partial of this (block, chunk) ---------------------------------------------------
    float *part = part_chunk + gp_off;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i >= n_ti) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j >= n_tj) continue;
            const int ti = 4 * bi + i, tj = kb0 + 4 * bj + j;
            *reinterpret_cast<f4 *>(part + ((int64_t)(ti * lnkb + tj) * 64 + lane) * 4) = acc[i][j];
        }
        if (want_bias) {
            float v = bsum[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 16) part[(int64_t)lt_out * lnkb * 256 + (4 * bi + i) * 16 + lane] = v;
        }
    }
}

// The jobs of a launch as a table the HOST builds from the Plan (r06).  The kernel used to find its job by walking P.layer[l].seg[s]
// with run-time indices: the compiler copies a by-value struct that is indexed dynamically into scratch (500 bytes per lane, 121
// scratch instructions and 84 scalar-load waits in front of the first operand load - a fifth of a small launch).  Now a workgroup
// reads ONE 32-byte descriptor through the kernarg segment pointer (scalar load with a run-time offset, like mlp_lat_device.h).
struct NarrowJob {
    int dy_row;     // first d Y tile-row of the block: L.dy[l] + 4 bi
    int x_row;      // first X tile-row: seg_act_row(l, s) + 4 bj
    int gp;         // L.gp[l]: the layer's block inside a chunk's partial
    int nkb, t_out; // of the layer: the partial's tile grid, and where its bias sums start
    int bi_bj_kb0;  // bi | bj << 8 | kb0 << 16
    int nt_bias;    // n_ti | n_tj << 4 | want_bias << 8
    int pad_;
};
static_assert(sizeof(NarrowJob) == 32, "NarrowJob is read as 8 dwords");
constexpr int MAX_NARROW_JOBS = 108;
struct NarrowTable {
    int n, pad_[7];
    NarrowJob j[MAX_NARROW_JOBS];
};
// host: the jobs of wgrad_direct_jobs(P, fold) in the order the f16 twin of this kernel walks them; false: too many for the table
static bool make_narrow_table(const Plan &P, const TrainLayout &L, int fold, NarrowTable &T) {
    T.n = 0;
    for (int l = 0; l < P.nlayers; ++l) {
        const Layer &Ly = P.layer[l];
        int first_seg = 0;   // the bias sums ride with the first non-empty input segment of the layer
        while (first_seg < Ly.nseg && Ly.seg[first_seg].nkb == 0) ++first_seg;
        int kb0 = 0;
        for (int sg = 0; sg < Ly.nseg; ++sg) {
            if (wgrad_kind(P, l, sg, fold) == 2) {
                const int nbi = (Ly.t_out + 3) / 4, nbj = (Ly.seg[sg].nkb + 3) / 4;
                for (int bi = 0; bi < nbi; ++bi)
                    for (int bj = 0; bj < nbj; ++bj) {
                        if (T.n >= MAX_NARROW_JOBS || bi > 255 || bj > 255 || kb0 > 32767) return false;
                        NarrowJob &J = T.j[T.n++];
                        J.dy_row = L.dy[l] + 4 * bi;
                        J.x_row = seg_act_row(P, L, l, sg) + 4 * bj;
                        J.gp = L.gp[l];
                        J.nkb = Ly.nkb;
                        J.t_out = Ly.t_out;
                        J.bi_bj_kb0 = bi | (bj << 8) | (kb0 << 16);
                        const int n_ti = std::min(4, Ly.t_out - 4 * bi), n_tj = std::min(4, Ly.seg[sg].nkb - 4 * bj);
                        J.nt_bias = n_ti | (n_tj << 4) | ((sg == first_seg && bj == 0) ? 256 : 0);
                        J.pad_ = 0;
                    }
            }
            kb0 += Ly.seg[sg].nkb;
        }
    }
    return true;
}
typedef int nj_i8 __attribute__((ext_vector_type(8)));

template <int WG_PREFETCH>
__global__ __launch_bounds__(WG_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void mlp_wgrad_direct_tab_kernel(NarrowTable tab_in_kernarg, WgradArgs A,
                                                                                                                   int64_t gp_floats) {
    __shared__ __attribute__((aligned(16))) float s_acc[(WG_WAVES - 1) * (16 * 256 + 4 * 64)];   // 3 x (16 tiles + 4 bias sums)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    typedef const __attribute__((address_space(4))) NarrowTable *TabPtr;
    const TabPtr tab = (TabPtr)__builtin_amdgcn_kernarg_segment_ptr();   // = &tab_in_kernarg, read in place
    const nj_i8 v = *reinterpret_cast<const __attribute__((address_space(4))) nj_i8 *>(&tab->j[blockIdx.x]);
    const NarrowJob J = __builtin_bit_cast(NarrowJob, v);
    const int bi = J.bi_bj_kb0 & 255, bj = (J.bi_bj_kb0 >> 8) & 255, kb0 = J.bi_bj_kb0 >> 16;
    const int n_ti = J.nt_bias & 15, n_tj = (J.nt_bias >> 4) & 15;
    const bool want_bias = (J.nt_bias & 256) != 0;
    const int64_t n = A.n;
    // this wave's quarter of the chunk (multiples of 4 samples)
    const int64_t begin = (int64_t)blockIdx.y * A.chunk;
    const int64_t end = min(n, begin + A.chunk);
    const int64_t sub = begin < end ? ((end - begin + 4 * WG_WAVES - 1) / (4 * WG_WAVES)) * 4 : 0;
    const int64_t wb = min(end, begin + wave * sub), we = min(end, wb + sub);
    // wave-uniform row bases at sample wb (rows the block does not have alias row 0: never read)
    const float *ya[4], *xb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        ya[t] = A.dy + ((int64_t)(J.dy_row + (t < n_ti ? t : 0)) * n + wb) * 16;
        xb[t] = A.act + ((int64_t)(J.x_row + (t < n_tj ? t : 0)) * n + wb) * 16;
    }
    float *part_chunk = A.part + (int64_t)blockIdx.y * gp_floats;
    // Every block runs the unpredicated 4 x 4 loop: the tile-rows a block does not have alias its first row (valid addresses, cache
    // hits), their products are computed and not stored.  (Until r06 such blocks - the one-tile heads, the last k-block column of the
    // warp net's 7-k-block layer: 8 of its 12 jobs - ran a second instantiation with a branch around every load and every MFMA row:
    // 137 branches and 32 scratch accesses per group of four k-steps.  The kernel is bound by operand latency, not by the matrix
    // pipe: the wasted MFMAs are free, the branches were not.)
    wgrad_direct_block<WG_PREFETCH, true>(part_chunk, J.gp, J.nkb, J.t_out, s_acc, ya, xb, kb0, bi, bj, n_ti, n_tj, wb, we, want_bias, wave, lane);
}

template <int WG_PREFETCH>
__global__ __launch_bounds__(WG_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void mlp_wgrad_direct_kernel(Plan P, TrainLayout L, WgradArgs A) {
    __shared__ __attribute__((aligned(16))) float s_acc[(WG_WAVES - 1) * (16 * 256 + 4 * 64)];   // 3 x (16 tiles + 4 bias sums)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // ---- decode the job: (narrow layer, segment, 4x4-tile block) -------------------------------------
    int job = blockIdx.x, l = 0, s = 0, kb0 = 0, nbj = 1;
    for (l = 0; l < P.nlayers; ++l) {
        bool found = false;
        kb0 = 0;
        for (s = 0; s < P.layer[l].nseg; ++s) {
            nbj = (P.layer[l].seg[s].nkb + 3) / 4;
            const int cnt = wgrad_kind(P, l, s, A.fold) == 2 ? ((P.layer[l].t_out + 3) / 4) * nbj : 0;
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
    int first_seg = 0;  // the bias sums ride with the first non-empty input segment of the layer
    while (first_seg < Ly.nseg && Ly.seg[first_seg].nkb == 0) ++first_seg;
    const bool want_bias = (s == first_seg && bj == 0);

    // this wave's quarter of the chunk (multiples of 4 samples)
    const int64_t begin = (int64_t)blockIdx.y * A.chunk;
    const int64_t end = min(n, begin + A.chunk);
    const int64_t sub = begin < end ? ((end - begin + 4 * WG_WAVES - 1) / (4 * WG_WAVES)) * 4 : 0;
    const int64_t wb = min(end, begin + wave * sub), we = min(end, wb + sub);
    // wave-uniform row bases at sample wb (rows the block does not have alias row 0: never read)
    const float *ya[4], *xb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        ya[t] = A.dy + ((int64_t)(L.dy[l] + 4 * bi + (t < n_ti ? t : 0)) * n + wb) * 16;
        xb[t] = A.act + ((int64_t)(seg_act_row(P, L, l, s) + 4 * bj + (t < n_tj ? t : 0)) * n + wb) * 16;
    }
    wgrad_direct_block<WG_PREFETCH, true>(A.part + (int64_t)blockIdx.y * L.gp_floats, L.gp[l], Ly.nkb, Ly.t_out, s_acc, ya, xb, kb0, bi, bj, n_ti, n_tj, wb, we, want_bias, wave, lane);
}

// sum over the G partials and scatter slot order -> state_dict order
// G_wide / G_narrow: number of sample chunks (partials) the wide and the narrow jobs were split into
__global__ __launch_bounds__(256) void mlp_wgrad_reduce_kernel(Plan P, TrainLayout L, const float *__restrict__ part,
                                                               int G_wide, int G_narrow, int fold, int accumulate,
                                                               float *__restrict__ flat_grad) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= L.gp_floats) return;
    int l = 0;
    while (l + 1 < P.nlayers && e >= L.gp[l + 1]) ++l;
    const Layer &Ly = P.layer[l];
    int rel = e - L.gp[l];
    int64_t dst = -1;
    const int nw = Ly.t_out * Ly.nkb * 256;
    int seg = 0;   // the input segment whose job wrote this element (the bias sums ride with the first non-empty one)
    while (seg < Ly.nseg && Ly.seg[seg].nkb == 0) ++seg;
    if (rel < nw) {
        const int r = rel & 3, lane = (rel >> 2) & 63, tile = rel >> 8;
        const int ti = tile / Ly.nkb, tj = tile - ti * Ly.nkb;
        int kb0 = 0;
        for (seg = 0; seg + 1 < Ly.nseg && tj >= kb0 + Ly.seg[seg].nkb; ++seg) kb0 += Ly.seg[seg].nkb;
        const int row = 16 * ti + 4 * (lane >> 4) + r;  // MFMA D layout: row = 4*(lane>>4)+r, col = lane&15
        const int jj = lane & 15;
        const int col = slot_to_col(Ly, tj, jj >> 2, jj & 3);
        if (row < Ly.n_out && col >= 0) dst = Ly.w_off + (int64_t)row * Ly.n_in + col;
    } else {
        const int row = rel - nw;
        if (row < Ly.n_out) dst = Ly.b_off + row;
    }
    if (dst < 0) return;
    // the bias sums of a layer ride with its first non-empty segment; a folded pair was written by the wide job's chunks
    // a folded pair was written by its carrier's (wide) workgroups
    const int G = (seg < Ly.nseg && wgrad_kind(P, l, seg, fold) != 2) ? G_wide : G_narrow;
    // accumulate: this launch is one ray chunk of a larger batch (train_step.hip) - the chunks' sums are added in chunk order
    // four running sums (partials c, c + 4, c + 8, ... each): four loads in flight per thread instead of one - the kernel reads
    // G x 2.6 MB and sat at the latency of its one outstanding load (r04: 0.125 -> ~0.07 ms at G = 113); the order is fixed
    const float *p = part + e;
    const int64_t gp = L.gp_floats;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = 0;
    for (; c + 4 <= G; c += 4) {
        s0 += p[(int64_t)c * gp];
        s1 += p[(int64_t)(c + 1) * gp];
        s2 += p[(int64_t)(c + 2) * gp];
        s3 += p[(int64_t)(c + 3) * gp];
    }
    for (; c < G; ++c) s0 += p[(int64_t)c * gp];
    const float sum = (s0 + s1) + (s2 + s3);
    flat_grad[dst] = accumulate ? flat_grad[dst] + sum : sum;
}

int launch_pack_t(const Plan &P, const BwdPlan &B, const float *params_flat, float *packed_t, hipStream_t s, const char *what) {
    hipLaunchKernelGGL(mlp_pack_t_kernel, dim3((B.total_slabs + SLAB_PAD) * PACK_T_PARTS), dim3(256), 0, s, P, B, params_flat, packed_t);
    return check_launch(what);
}

// split-K wgrad + reduce for any (Plan, TrainLayout)
int launch_wgrad(const Plan &P, const TrainLayout &L, const float *act, const float *dy, int64_t n, float *gpart,
                 float *flat_grad, hipStream_t s, int wide_nsplit, bool accumulate, int64_t n_beside) {
    // chunks: at most the wgrad_chunks(n) the partial buffer is sized for; with more wide workgroups than CUs, as many as
    // fill whole rounds of the chip (9 wide jobs x 128 chunks on 256 CUs = 4.5 rounds, the last one half empty: 113
    // chunks = 3.97 rounds of 13 % longer workgroups)
    int G_narrow = wgrad_chunks_1k(n);
    int G = G_narrow;
    const int fold = wide_nsplit ? 0 : 1;   // the split-precision wide kernels do not carry folded tiles
    {
        const int n_cu = device_cu_count("wgrad");
        if (n_cu < 1) return n_cu;
        // this launch's share of the chip when another net's weight gradient runs beside it (r06).  Two small launches that each
        // split their jobs over ALL CUs are two rounds of workgroups with half-length chunks and twice the partials: the 64-ray
        // step's wide launches lasted 132 + 177 us side by side for 95 + 32 us of matrix work.  Split for the share instead and
        // both nets' workgroups are resident at once, with equal chunk lengths.
        const int cu_share = n_beside > 0 ? std::max(1, (int)((double)n_cu * (double)n / (double)(n + n_beside) + 0.5)) : n_cu;
        // narrow jobs: 4-wave workgroups, three per CU (mlp_wgrad_direct_kernel: waves_per_eu(3, 3), 52 KB of LDS) - the same
        // whole-rounds rule: 14 jobs x 128 chunks = 1792 workgroups are 2.33 rounds of 768 slots, 109 chunks are 1.99
        if (const int jobs_d = wgrad_direct_jobs(P, fold)) {
            const int slots = 3 * n_cu;
            if ((int64_t)jobs_d * G_narrow > slots) G_narrow = min(G_narrow, max(1, (int)((int64_t)jobs_d * G_narrow / slots) * slots / jobs_d));
            // small calls (the README's 64-ray batches): 1024-sample chunks would fill a fifth of the slots, each wave walking 256
            // samples at the latency of its 4 k-steps in flight - shorter chunks (>= 64 samples), one round of the slots
            else G_narrow = max(G_narrow, min(min(wgrad_chunks(n), (int)((n + 63) / 64)), max(1, 3 * cu_share / jobs_d)));
        }
        const int jobs = wgrad_jobs(P);
        if (jobs > 0 && (int64_t)jobs * G > n_cu) {
            // (fewer whole rounds = fewer partials for the reduce: measured r03, 4 / 3 / 2 / 1 rounds -> the same step time)
            const int rounds = (int)((int64_t)jobs * G / n_cu);
            G = min(G, max(1, rounds * n_cu / jobs));
        } else if (jobs > 0) {
            // small calls (the README's 64-ray batches: 4096 + 12 288 samples): 1024-sample chunks would put 9 x 4 workgroups
            // on 256 CUs - shorter chunks, one round of the chip (r04: 267 -> ~70 us per launch at 4096 samples; the reduce
            // reads 28 partials instead of 4)
            G = max(G, min(wgrad_chunks(n), max(1, cu_share / jobs)));
        }
    }
    WgradArgs W{};
    W.act = act;
    W.dy = dy;
    W.part = gpart;
    W.n = n;
    W.xstat = reinterpret_cast<const int *>(act + (int64_t)L.act_rows * n * 16);   // (f16x3 wide jobs only)
    W.ystat = reinterpret_cast<const int *>(dy + (int64_t)L.dy_rows * n * 16);
    W.chunk = (((n + G - 1) / G) + 15) / 16 * 16;
    G = (int)((n + W.chunk - 1) / W.chunk);   // rounding the chunk up to whole k-steps may leave trailing chunks empty: not launched
    W.fold = fold;
    static LdsRaised raised;   // per device
    int rc;
    if ((rc = raise_dynamic_lds(reinterpret_cast<const void *>(mlp_wgrad_kernel), WL_LDS_BYTES, raised, "wgrad"))) return rc;
    if (const int jobs = wgrad_jobs(P)) {
        if (wide_nsplit) {   // the wide jobs on the bf16 matrix cores (mlp_train_bf16.hip)
            if ((rc = launch_wgrad_wide_bf16(P, L, W, jobs, G, wide_nsplit, s))) return rc;
        } else {
            hipLaunchKernelGGL(mlp_wgrad_kernel, dim3(jobs, G), dim3(WL_THREADS), WL_LDS_BYTES, s, P, L, W);
            if ((rc = check_launch("wgrad"))) return rc;
        }
    }
    if (const int jobs = wgrad_direct_jobs(P, W.fold)) {
        W.chunk = (((n + G_narrow - 1) / G_narrow) + 15) / 16 * 16;
        G_narrow = (int)((n + W.chunk - 1) / W.chunk);
        // f16x3 step: the narrow jobs with two fp16 parts as well
        if (wide_nsplit == SNERF_SPLIT_F16X3) {
            if ((rc = launch_wgrad_direct_f16(P, L, W, jobs, G_narrow, s))) return rc;
        } else {
            // 4 k-steps of operands in flight per wave (measured r03: 2 / 3 / 4 / 6 -> 0.61 / 0.58 / 0.56 / 0.57 ms per launch)
            NarrowTable T;
            if (make_narrow_table(P, L, W.fold, T) && T.n == jobs)
                hipLaunchKernelGGL(mlp_wgrad_direct_tab_kernel<4>, dim3(jobs, G_narrow), dim3(WG_THREADS), 0, s, T, W, (int64_t)L.gp_floats);
            else   // (more narrow jobs than the table holds - nets with a hundred additional-input k-blocks: the job found in the kernel)
                hipLaunchKernelGGL(mlp_wgrad_direct_kernel<4>, dim3(jobs, G_narrow), dim3(WG_THREADS), 0, s, P, L, W);
            if ((rc = check_launch("wgrad_direct"))) return rc;
        }
    }
    hipLaunchKernelGGL(mlp_wgrad_reduce_kernel, dim3((L.gp_floats + 255) / 256), dim3(256), 0, s, P, L, gpart, G, G_narrow, W.fold, accumulate ? 1 : 0, flat_grad);
    return check_launch("wgrad_reduce");
}

}  // namespace snerf

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
extern "C" int snerf_mlp_pack_t_f32(const snerf_mlp_desc *desc, const float *params_flat, float *packed_t,
                                    int input_grad, snerf_stream_t stream) {
    using namespace snerf;
    Plan P;
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "mlp_pack_t: desc is null");
    if (make_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "mlp_pack_t: %s", why);
    if (!params_flat || !packed_t) return fail(SNERF_E_BADARG, "mlp_pack_t: null pointer");
    if (!aligned(packed_t, 16)) return fail(SNERF_E_ALIGN, "mlp_pack_t: packed_t must be 16-byte aligned");
    BwdPlan B;
    make_bwd_plan(P, B, input_grad != 0);
    return launch_pack_t(P, B, params_flat, packed_t, (hipStream_t)stream, "mlp_pack_t");
}

namespace snerf {
// accumulate: flat_grad += instead of = (one ray chunk of a larger batch, train_step.hip)
int launch_bwd(const snerf_mlp_desc *desc, const float *packed_t, const float *act, const float *d_raw, int64_t n,
               float *dy, float *gpart, float *flat_grad, const float *x, const float *dirs, int dirs_per_sample,
               int spr, float *d_x, float *d_dirs, snerf_stream_t stream, bool accumulate, bool beside_another_net, int64_t n_beside) {
    Plan P;
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "mlp_bwd: desc is null");
    if (make_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "mlp_bwd: %s", why);
    if (n < 0) return fail(SNERF_E_BADARG, "mlp_bwd: negative n");
    if (n == 0) return SNERF_OK;
    if (!packed_t || !act || !d_raw || !dy || !gpart || !flat_grad) return fail(SNERF_E_BADARG, "mlp_bwd: null pointer");
    if (!aligned(packed_t, 16) || !aligned(act, 16) || !aligned(d_raw, 16) || !aligned(dy, 16) || !aligned(gpart, 16))
        return fail(SNERF_E_ALIGN, "mlp_bwd: buffers must be 16-byte aligned");
    const bool input_grad = d_x != nullptr;
    if (input_grad) {
        if (!x || !d_dirs || (desc->use_dir && !dirs) || spr < 1) return fail(SNERF_E_BADARG, "mlp_bwd: input gradients need x, dirs, d_x, d_dirs");
        if (P.pos_nkb > 8 || P.dir_nkb > 8)
            return fail(SNERF_E_BADARG, "mlp_bwd: input gradients support at most 8 position / 8 direction encoder k-blocks");
    }
    hipStream_t s = (hipStream_t)stream;
    TrainLayout L;
    make_train_layout(P, L);
    const int nh = P.n_hidden;
    BwdArgs A{};
    A.packed_t = packed_t;
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
    // 8 waves = 128 samples per workgroup, like the forward; calls of <= 64 x CUs samples (the README's 64-ray batches) run
    // 4-wave workgroups on 64-sample tiles like the forward does (mlp.hip: launch_fwd): below one tile per CU the launch is
    // the latency of one tile's pass through the weight stream, and a wave that has its SIMD's matrix pipe to itself
    // passes in two thirds of the time (r04: 290 -> 190 us at 4096 samples)
    const int n_cu = device_cu_count("mlp_bwd");
    if (n_cu < 1) return n_cu;
    const bool small = P.width > 256 || n <= (int64_t)64 * n_cu;
    const bool wide_pe = input_grad && bwd_pe_tiles(P).pos == 8;
    // calls of a few 16-sample tiles per CU: the latency-class dgrad (mlp_lat.hip; bit-identical d Y); a call of a few rounds and a
    // fraction: whole rounds here, the fraction there
    const LatChoice lc = lat_choose_bwd(P, n, input_grad, beside_another_net);
    if (lc.mode == 1) {
        if (int lrc = launch_bwd_lat(P, A, s, 0)) return lrc;
        return launch_wgrad(P, L, act, dy, n, gpart, flat_grad, s, 0, accumulate, n_beside);
    }
    const int64_t n_dgrad = lc.mode == 2 ? lc.n_main : n;   // samples of the throughput kernel below
    auto launch = [&](auto bw_c) -> int {
        constexpr int BW = decltype(bw_c)::value;
        const int64_t grid = (n_dgrad + BW * 16 - 1) / (BW * 16);
        if (grid > 0x7fffffffLL) return fail(SNERF_E_BADARG, "mlp_bwd: n too large");
        if (P.width > 256) {   // one wave per SIMD, like the forward
            if constexpr (BW == 4) {
#define SNERF_LAUNCH_WIDE SNERF_LAUNCH_RING4
#define SNERF_BWD_WIDE(W_)                                                                                                          \
    do {                                                                                                                            \
        if (wide_pe) SNERF_LAUNCH_WIDE((mlp_bwd_kernel<W_, 4, true, 8, 8>), dim3((unsigned)grid), dim3(256), s, A);                 \
        else if (input_grad) SNERF_LAUNCH_WIDE((mlp_bwd_kernel<W_, 4, true>), dim3((unsigned)grid), dim3(256), s, A);               \
        else SNERF_LAUNCH_WIDE((mlp_bwd_kernel<W_, 4, false>), dim3((unsigned)grid), dim3(256), s, A);                              \
    } while (0)
                if (P.width == 320) SNERF_BWD_WIDE(320);
                else if (P.width == 384) SNERF_BWD_WIDE(384);
                else if (P.width == 448) SNERF_BWD_WIDE(448);
                else SNERF_BWD_WIDE(512);
#undef SNERF_BWD_WIDE
            } else {
                return fail(SNERF_E_BADARG, "mlp_bwd: widths above 256 run 4-wave workgroups");
            }
        } else if (P.width == 256) {
            if (wide_pe) SNERF_LAUNCH_RING((mlp_bwd_kernel<256, BW, true, 8, 8>), dim3((unsigned)grid), dim3(BW * 64), s, A);
            else if (input_grad) SNERF_LAUNCH_RING((mlp_bwd_kernel<256, BW, true>), dim3((unsigned)grid), dim3(BW * 64), s, A);
            else SNERF_LAUNCH_RING((mlp_bwd_kernel<256, BW, false>), dim3((unsigned)grid), dim3(BW * 64), s, A);
        } else if (P.width == 128) {
            if (wide_pe) SNERF_LAUNCH_RING((mlp_bwd_kernel<128, BW, true, 8, 8>), dim3((unsigned)grid), dim3(BW * 64), s, A);
            else if (input_grad) SNERF_LAUNCH_RING((mlp_bwd_kernel<128, BW, true>), dim3((unsigned)grid), dim3(BW * 64), s, A);
            else SNERF_LAUNCH_RING((mlp_bwd_kernel<128, BW, false>), dim3((unsigned)grid), dim3(BW * 64), s, A);
        } else {
            if (wide_pe) SNERF_LAUNCH_RING((mlp_bwd_kernel<64, BW, true, 8, 8>), dim3((unsigned)grid), dim3(BW * 64), s, A);
            else if (input_grad) SNERF_LAUNCH_RING((mlp_bwd_kernel<64, BW, true>), dim3((unsigned)grid), dim3(BW * 64), s, A);
            else SNERF_LAUNCH_RING((mlp_bwd_kernel<64, BW, false>), dim3((unsigned)grid), dim3(BW * 64), s, A);
        }
        return SNERF_OK;
    };
    if (int lrc = small ? launch(std::integral_constant<int, 4>{}) : launch(std::integral_constant<int, 8>{})) return lrc;
    int rc = check_launch("mlp_bwd(dgrad)");
    if (rc) return rc;
    if (lc.mode == 2 && (rc = launch_bwd_lat(P, A, s, lc.n_main))) return rc;
    return launch_wgrad(P, L, act, dy, n, gpart, flat_grad, s, 0, accumulate, n_beside);
}
}  // namespace snerf

extern "C" int snerf_mlp_bwd_f32(const snerf_mlp_desc *desc, const float *packed_t, const float *act,
                                 const float *d_raw, int64_t n, float *dy, float *gpart, float *flat_grad,
                                 snerf_stream_t stream) {
    return snerf::launch_bwd(desc, packed_t, act, d_raw, n, dy, gpart, flat_grad, nullptr, nullptr, 0, 1, nullptr,
                             nullptr, stream);
}

extern "C" int snerf_mlp_bwd_inputs_f32(const snerf_mlp_desc *desc, const float *packed_t, const float *act,
                                        const float *d_raw, const float *x, const float *dirs, int dirs_per_sample,
                                        int samples_per_ray, int64_t n, float *dy, float *gpart, float *flat_grad,
                                        float *d_x, float *d_dirs, snerf_stream_t stream) {
    if (!d_x) return snerf::fail(SNERF_E_BADARG, "mlp_bwd_inputs: d_x is null");
    return snerf::launch_bwd(desc, packed_t, act, d_raw, n, dy, gpart, flat_grad, x, dirs, dirs_per_sample,
                             samples_per_ray, d_x, d_dirs, stream);
}
