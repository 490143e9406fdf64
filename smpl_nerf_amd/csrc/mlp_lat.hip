// Latency-class forward and dgrad of the fused RenderRayNet for small calls (mlp_lat_device.h has the design): the reference's own
// operating points - README.md:23 trains at --batchsize=64 (4096 coarse + 12 288 fine samples per step), inference.py:231 renders 800
// rays per call - are below one 128-sample tile per CU, where the throughput kernels of mlp.hip / mlp_train.hip are one wave's serial
// pass through the weight stream.  Same C-ABI entry points, same packed streams, same activation / dY / mask buffers, bit-identical
// results: launch_fwd (mlp.hip) and launch_bwd (mlp_train.hip) route here by size.
//
// Replaces, like they do: RenderRayNet.forward (models/render_ray_net.py:42-61) + the encoders feeding it
// (models/nerf_pipeline.py:29-39, :49-57), and loss.backward() through it (solver/nerf_solver.py:83-87).
#include <stdlib.h>

#include <algorithm>

#include "mlp_lat_device.h"

namespace snerf {

struct LatLds {   // byte offsets into the dynamic LDS of a workgroup
    int act[2];   // two activation buffers, S x 16 KiB each
    // forward: the input k-blocks of a sample tile that are not a layer's output, pe_stride bytes per tile: [position-encoding and
    // per-ray additional-input k-blocks in the column order of the weight matrix][zero k-blocks up to a multiple of LAT_PF]
    // [direction-encoding k-blocks][zero k-blocks: 16 + dir_nkb up to a multiple of LAT_PF]; dgrad: unused
    int pe, pe_stride, pe_dir;   // pe_dir: offset of the direction k-blocks inside a tile's region
    int pe_pos, pe_add;          // offsets of the position-encoding / additional-input k-blocks inside the first block
    int aux, aux_stride;         // dgrad: the ReLU sign-mask words of the pass, S x n_mask x 512 B (forward: unused)
    int total;
};
static inline int lat_pad(int nkb) { return (nkb + LAT_PF - 1) / LAT_PF * LAT_PF; }
static inline LatLds lat_lds(int S, int pos_nkb, int dir_nkb, int aux_bytes, int add_nkb = 0, int add_first = 0) {
    LatLds o;
    o.act[0] = 0;
    o.act[1] = S * LAT_ACT_BYTES;
    o.pe = 2 * S * LAT_ACT_BYTES;
    o.pe_pos = add_first ? add_nkb * 1024 : 0;
    o.pe_add = add_first ? 0 : pos_nkb * 1024;
    o.pe_dir = lat_pad(pos_nkb + add_nkb) * 1024;
    o.pe_stride = o.pe_dir + (lat_pad(16 + dir_nkb) - 16) * 1024;
    o.aux = o.pe + S * o.pe_stride;
    o.aux_stride = aux_bytes;
    o.total = o.aux + S * aux_bytes;
    return o;
}
// forward: sigma of a sample travels from the sigma head's wave to the rgb head's through the last KiB of the sample tile's
// directional-branch buffer (a 128-wide layer fills half of the 16 KiB)
constexpr int LAT_SIG_OFF = 15 * 1024;

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// wg: this workgroup's index among those that share G (blockIdx.x, or its index inside its half of a mixed launch)
template <int S, bool TRAIN>
__device__ __forceinline__ void mlp_fwd_lat_body(const LatTabPtr tab, const FwdArgs &A, const LatGeom &G, const LatLds &Lo, const int wg, char *lds) {
    const int tid = threadIdx.x;
    LatWave W;
    W.start(tab, A.packed, tid, G.passes);
    const int lane = W.lane, wave = W.wave, g = lane >> 4;
    const int in_nkb = A.pos_nkb + A.add_nkb, enc_nkb = in_nkb + A.dir_nkb;
    const unsigned n32 = (unsigned)A.n;
    const __amdgpu_buffer_rsrc_t act_rs = lat_rsrc(TRAIN ? A.act : A.raw, LAT_STORE_RANGE);   // (TRAIN only)
    const __amdgpu_buffer_rsrc_t raw_rs = lat_rsrc(A.raw, LAT_STORE_RANGE);
    const int n_layers = tab->n;

#pragma clang loop unroll(disable)
    for (int pass = 0; pass < G.passes; ++pass) {
        const int64_t tile0 = G.tile_off + ((int64_t)wg * G.passes + pass) * S;
        // sample of this lane in sample tile s (clamped for loads; `okay` gates every store)
        auto sample_of = [&](int s) { return (tile0 + s) * 16 + (lane & 15); };
        auto okay = [&](int s) { return tile0 + s < G.tile_end && sample_of(s) < A.n; };
        // TRAIN: byte offset of this lane's tile in row `row` of the activation buffer, out of range for a masked sample
        auto act_off = [&](int s, int row) { return okay(s) ? lat_tile_off(row, n32, (unsigned)sample_of(s), g) : LAT_OOB; };

        // ---- phase 0: the encoder k-blocks of the pass, one (tile, k-block) unit per wave at a time -> LDS ---------------
        if (pass == 0) {   // the zero k-blocks behind them (padding steps of layers whose k-block count is not a multiple of LAT_PF)
            const int zp = Lo.pe_dir / 1024 - in_nkb, zd = (Lo.pe_stride - Lo.pe_dir) / 1024 - A.dir_nkb;
            for (int u = wave; u < S * (zp + zd); u += LAT_NW) {
                const int s = u / (zp + zd), k = u - s * (zp + zd);
                const int off = k < zp ? (in_nkb + k) * 1024 : Lo.pe_dir + (A.dir_nkb + k - zp) * 1024;
                *reinterpret_cast<f4 *>(lds + Lo.pe + s * Lo.pe_stride + off + W.voff) = f4{0.f, 0.f, 0.f, 0.f};
            }
        }
        for (int u = wave; u < S * enc_nkb; u += LAT_NW) {
            const int s = u / enc_nkb, k = u - s * enc_nkb;
            const bool is_dir = k >= in_nkb, is_add = !is_dir && k >= A.pos_nkb;
            const int kb = is_dir ? k - in_nkb : is_add ? k - A.pos_nkb : k;
            const int64_t smp = sample_of(s), sc = min(smp, A.n - 1);
            SampleCtx c;
            c.g = g;
            c.enc = nullptr;
            c.add = nullptr;
            c.px = c.py = c.pz = c.dx = c.dy = c.dz = 0.f;
            f4 b;
            if (is_add) {      // per-ray additional inputs (the pose rows of models/append_smpl_params_pipeline.py:29-52): this sample's ray's row
                c.add = A.add + (sc / A.spr) * A.add_dim;
                b = add_operand(c, A.add_dim, kb);
            } else {
                if (is_dir) {
                    const float *dp = A.dirs + (A.dirs_per_sample ? sc : sc / A.spr) * 3;
                    const float ux = dp[0], uy = dp[1], uz = dp[2];
                    const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(ux, ux), __fmul_rn(uy, uy)), __fmul_rn(uz, uz)));
                    c.dx = __fdiv_rn(ux, nrm);  // models/nerf_pipeline.py:33-34
                    c.dy = __fdiv_rn(uy, nrm);
                    c.dz = __fdiv_rn(uz, nrm);
                } else {
                    c.px = A.x[sc * 3 + 0];
                    c.py = A.x[sc * 3 + 1];
                    c.pz = A.x[sc * 3 + 2];
                }
                b = pe_operand<false>(c, is_dir, is_dir ? A.dir_L : A.pos_L, is_dir ? A.dir_id : A.pos_id, kb, 0);
            }
            const int off = is_dir ? Lo.pe_dir + kb * 1024 : (is_add ? Lo.pe_add : Lo.pe_pos) + kb * 1024;
            *reinterpret_cast<f4 *>(lds + Lo.pe + s * Lo.pe_stride + off + W.voff) = b;
            if (TRAIN) lat_store_f4(act_rs, act_off(s, (is_dir ? A.act_dpe : is_add ? A.act_add : A.act_pe) + kb), b);
        }
        __syncthreads();

        // ---- the layers of the table (host: lat_table_fwd), one site for all of them ---------------------------------------
#pragma clang loop unroll(disable)
        for (int l = 0; l < n_layers; ++l) {
            const LatLayer Ly = lat_layer_at(tab, l);
            const int w0 = Ly.wave0, t_out = Ly.t_out, op = Ly.op;
            if (wave >= w0 && wave < w0 + ((t_out + 1) >> 1)) {
                const int tile = 2 * (wave - w0);
                f4 acc[S][2];
#pragma unroll
                for (int s = 0; s < S; ++s) acc[s][0] = W.aux[0], acc[s][1] = W.aux[1];
                lat_run_layer<S>(W, tab, lds, Ly, acc);
                // (a head: out_base is the directional-branch buffer, whose last KiB per sample tile carries sigma)
                float *sig = reinterpret_cast<float *>(lds + Ly.out_base + LAT_SIG_OFF);
                if (op & LAT_HEAD_SIGMA) {          // sigma_out_layer (:52): row 0 of the padded tile (lanes 0..15 hold it)
#pragma unroll
                    for (int s = 0; s < S; ++s) sig[s * (LAT_ACT_BYTES / 4) + lane] = acc[s][0][0];
                } else if (op & LAT_HEAD_RGB) {     // rgb_out_layer (:60): rows 0..2, and the [rgb | sigma] store (:61)
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const f4 o = f4{acc[s][0][0], acc[s][0][1], acc[s][0][2], sig[s * (LAT_ACT_BYTES / 4) + (lane & 15)]};
                        lat_store_f4(raw_rs, (lane < 16 && okay(s)) ? (unsigned)sample_of(s) * 16u : LAT_OOB, o);
                    }
                } else {
                    // a wide layer: activation -> LDS in B-operand layout; TRAIN: the tiles into rows store_row + tile .. of the
                    // activation buffer and this wave's byte of the sign-mask word (store_mask, mlp_device.h: tiles 2w, 2w+1 are
                    // byte w of the 64-bit word)
                    const bool relu = op & LAT_RELU;
                    const int out_base = Ly.out_base, store_row = Ly.store_row, mask_idx = Ly.mask_idx;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        f4 v[2];
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[t][r] = relu ? fmaxf(acc[s][t][r], 0.f) : acc[s][t][r];
                        f4 *dst = reinterpret_cast<f4 *>(lds + out_base + s * LAT_ACT_BYTES + tile * 1024 + W.voff);
                        dst[0] = v[0];
                        dst[64] = v[1];
                        if (TRAIN) {
                            lat_store_f4(act_rs, act_off(s, store_row + tile), v[0]);
                            lat_store_f4(act_rs, act_off(s, store_row + tile + 1), v[1]);
                            unsigned m = 0;
#pragma unroll
                            for (int t = 0; t < 2; ++t)
#pragma unroll
                                for (int r = 0; r < 4; ++r) m |= (unsigned)min(max(__float_as_int(v[t][r]), 0), 1) << (4 * t + r);
                            // (always issued: a layer without a mask stores out of range - see lat_store_f4)
                            const unsigned mo = (mask_idx >= 0 && okay(s)) ? lat_mask_off(A.act_mask, mask_idx, n32, (unsigned)sample_of(s), g) + (unsigned)(tile >> 1) : LAT_OOB;
                            lat_store_b8(act_rs, mo, m);
                            lat_store_b8(act_rs, (mo != LAT_OOB && (op & LAT_HALF_WORD)) ? mo + 4u : LAT_OOB, 0u);
                        }
                    }
                }
            }
            if (op & LAT_BARRIER) __syncthreads();
        }
        // (the next pass's phase 0 writes the encoder region only, and its barrier stands between this pass's rgb reads and the
        // first activation write of the next pass)
    }
}

template <int S, bool TRAIN>
__global__ __launch_bounds__(LAT_THREADS) void mlp_fwd_lat_kernel(LatTable tab_in_kernarg, FwdArgs A, LatGeom G, LatLds Lo) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    mlp_fwd_lat_body<S, TRAIN>(lat_table_ptr() /* = &tab_in_kernarg, read in place (mlp_lat_device.h) */, A, G, Lo, (int)blockIdx.x, lds);
}
// r06: THREE tiles per CU in one pass as a workgroup of two tiles and a workgroup of one, side by side on the CU (2 x lat_lds(2) fits
// the LDS, 16 waves): each covers the other's layer boundaries - what the paired launches of an even tile count get from two equal
// workgroups (lat_split).  The first n_a workgroups (dispatched first: one per CU) take two tiles each from GA, the others one tile
// from GB; both walk the same table, built for the LDS layout of S = 2.  README.md:23's 64-ray batch is this case: 768 fine tiles.
template <bool TRAIN>
__global__ __launch_bounds__(LAT_THREADS) void mlp_fwd_lat_mixed_kernel(LatTable tab_in_kernarg, FwdArgs A, LatGeom GA, LatGeom GB, LatLds Lo, int n_a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if ((int)blockIdx.x < n_a) mlp_fwd_lat_body<2, TRAIN>(lat_table_ptr(), A, GA, Lo, (int)blockIdx.x, lds);
    else mlp_fwd_lat_body<1, TRAIN>(lat_table_ptr(), A, GB, Lo, (int)blockIdx.x - n_a, lds);
}

static void lat_layer_stream(LatLayer &o, int first_slab, int nkb, int t_out) {
    o.soff = first_slab * SLAB_BYTES;
    o.nkb = nkb;
    o.t_out = t_out;
    int kps = SLAB_TILES / t_out, sh = 0;
    while ((1 << sh) < kps) ++sh;
    o.kps_shift = sh;
}

// the forward stream of `P` as the latency kernels walk it: layers in plan order (mlp_plan.h: make_plan)
static void lat_table_fwd(const Plan &P, const TrainLayout &L, const LatLds &Lo, LatTable &T) {
    T.n = P.nlayers;
    T.stream_bytes = (P.total_slabs + SLAB_PAD) * SLAB_BYTES;
    const int nh = P.n_hidden;
    const int o_buf = Lo.act[(nh + 1) & 1], h_buf = Lo.act[nh & 1];
    const int NONE = 1 << 20;   // "every k-block comes from the first region"
    for (int l = 0; l < P.nlayers; ++l) {
        const Layer &Ly = P.layer[l];
        LatLayer &o = T.l[l];
        lat_layer_stream(o, Ly.first_slab, Ly.nkb, Ly.t_out);
        o.wave0 = 0;
        o.b_base1 = o.b_stride1 = 0;
        o.out_base = -1;
        o.store_row = o.mask_idx = -1;
        o.pad_[0] = o.pad_[1] = 0;
        if (l == 0) {                 // positions_pose_input + relu (:45): reads the position encoding
            o.b_base0 = Lo.pe, o.b_stride0 = Lo.pe_stride, o.b_n0 = NONE;
        } else if (l <= nh + 1) {     // positional_net[l - 1] + relu (:46-50): the previous output and, as a skip layer, the encoding behind
            o.b_base0 = Lo.act[(l + 1) & 1], o.b_stride0 = LAT_ACT_BYTES, o.b_n0 = 16;   // it; additional_linear_layer (:51): no activation
            o.b_base1 = Lo.pe, o.b_stride1 = Lo.pe_stride;
        }
        if (l <= nh + 1) {
            o.out_base = Lo.act[l & 1];
            o.op = (l <= nh ? LAT_RELU : 0) | LAT_BARRIER;
            o.store_row = l <= nh ? L.x[1] + l * 16 : L.o;
            o.mask_idx = l <= nh ? l : -1;
        } else if (l == nh + 2) {     // sigma_out_layer (:52) on wave 4, beside directional_input
            o.wave0 = 4;
            o.b_base0 = o_buf, o.b_stride0 = LAT_ACT_BYTES, o.b_n0 = NONE;
            o.out_base = h_buf;   // (sigma goes into its last KiB)
            o.op = LAT_HEAD_SIGMA;
        } else if (l == nh + 3) {     // directional_input, no activation (:54-57): o and the direction encoding
            o.b_base0 = o_buf, o.b_stride0 = LAT_ACT_BYTES, o.b_n0 = 16;
            o.b_base1 = Lo.pe + Lo.pe_dir, o.b_stride1 = Lo.pe_stride;
            o.out_base = h_buf;
            o.op = LAT_BARRIER;
            o.store_row = L.h1;
        } else if (l == nh + 4) {     // directional_net[0] + relu (:58-59)
            o.b_base0 = h_buf, o.b_stride0 = LAT_ACT_BYTES, o.b_n0 = NONE;
            o.out_base = o_buf;
            o.op = LAT_RELU | LAT_BARRIER | LAT_HALF_WORD;
            o.store_row = L.h2;
            o.mask_idx = nh + 1;
        } else {                      // rgb_out_layer (:60) on wave 5
            o.wave0 = 5;
            o.b_base0 = o_buf, o.b_stride0 = LAT_ACT_BYTES, o.b_n0 = NONE;
            o.out_base = h_buf;
            o.op = LAT_HEAD_RGB;
        }
    }
    lat_table_finish(T);
}

// ------------------------------------------------------------------------------------------------
// dgrad: the forward pass of the transposed network on d raw (mlp_train.hip: mlp_bwd_kernel), through the same interpreter
// ------------------------------------------------------------------------------------------------
// IG: with input gradients (smpl_nerf: the nets back-propagate into the warped samples and their view directions,
// models/smpl_nerf_pipeline.py:49-56) for the default-sized encoders (4 position / 2 direction k-blocks: bwd_pe_tiles).  The
// transposed encoder columns are layers of two / one wave(s): waves 0, 1 keep the running d (position encoding) tiles 0-1 / 2-3 of
// the pass in registers, wave 0 finishes the direction columns on the spot; at the end of the pass wave 1 hands its tiles to wave 0
// through LDS, which runs the encoder backward in the order of mlp_bwd_kernel (same bits).
template <int S, bool IG = false>
__global__ __launch_bounds__(LAT_THREADS) void mlp_bwd_lat_kernel(LatTable tab_in_kernarg, BwdArgs A, LatGeom G, LatLds Lo) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const LatTabPtr tab = lat_table_ptr();
    const int tid = threadIdx.x;
    LatWave W;
    W.start(tab, A.packed_t, tid, G.passes);
    const int lane = W.lane, wave = W.wave, g = lane >> 4;
    const unsigned n32 = (unsigned)A.n;
    const __amdgpu_buffer_rsrc_t dy_rs = lat_rsrc(A.dy, LAT_STORE_RANGE);
    const __amdgpu_buffer_rsrc_t dx_rs = lat_rsrc(IG ? A.d_x : A.dy, LAT_STORE_RANGE), dd_rs = lat_rsrc(IG ? A.d_dirs : A.dy, LAT_STORE_RANGE);
    const int n_layers = tab->n, n_mask = A.n_hidden + 2;
    const f4 zero = f4{0.f, 0.f, 0.f, 0.f};

#pragma clang loop unroll(disable)
    for (int pass = 0; pass < G.passes; ++pass) {
        const int64_t tile0 = G.tile_off + ((int64_t)blockIdx.x * G.passes + pass) * S;
        auto sample_of = [&](int s) { return (tile0 + s) * 16 + (lane & 15); };
        auto okay = [&](int s) { return tile0 + s < G.tile_end && sample_of(s) < A.n; };
        auto dy_off = [&](int s, int row) { return okay(s) ? lat_tile_off(row, n32, (unsigned)sample_of(s), g) : LAT_OOB; };

        // ---- phase 0: d raw of this lane's samples; the d rgb operand of the first layer (+ three k-blocks of zeros behind it: the
        // layer has one k-block) into the second activation buffer, which nobody uses yet; the head gradients as tile-rows for the
        // wgrad kernel (rows 0..2 = rgb, row 0 = sigma); the ReLU sign-mask words of the pass -> LDS ------------------------------
        f4 dr[S];
#pragma unroll
        for (int s = 0; s < S; ++s) dr[s] = *reinterpret_cast<const f4 *>(A.d_raw + min(sample_of(s), A.n - 1) * 4);
        // (IG) the forward inputs of this lane's samples and the running d (position encoding) tiles of this wave
        float xin[IG ? S : 1][3], din[IG ? S : 1][3];
        f4 dpe[IG ? S : 1][2];
        if constexpr (IG) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int64_t sc = min(sample_of(s), A.n - 1);
                const float *dp = A.dirs + (A.dirs_per_sample ? sc : sc / A.spr) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    xin[s][c] = A.x[sc * 3 + c];
                    din[s][c] = A.use_dir ? dp[c] : 0.f;
                }
                dpe[s][0] = dpe[s][1] = zero;
            }
        }
        for (int s = wave; s < S; s += LAT_NW) {
            f4 *blk = reinterpret_cast<f4 *>(lds + Lo.act[1] + s * LAT_ACT_BYTES + W.voff);
            const f4 drgb = g == 0 ? f4{dr[0][0], dr[0][1], dr[0][2], 0.f} : zero, dsig = g == 0 ? f4{dr[0][3], 0.f, 0.f, 0.f} : zero;
            f4 v = drgb, w = dsig;
#pragma unroll
            for (int q = 1; q < S; ++q)
                if (s == q) v = g == 0 ? f4{dr[q][0], dr[q][1], dr[q][2], 0.f} : zero, w = g == 0 ? f4{dr[q][3], 0.f, 0.f, 0.f} : zero;
            blk[0] = v;
            blk[64] = zero;
            blk[128] = zero;
            blk[192] = zero;
            lat_store_f4(dy_rs, dy_off(s, A.dy_rgb), v);
            lat_store_f4(dy_rs, dy_off(s, A.dy_sig), w);
        }
        for (int u = wave; u < S * n_mask; u += LAT_NW) {
            const int s = u / n_mask, idx = u - s * n_mask;
            const int64_t sc = min(sample_of(s), A.n - 1);
            *reinterpret_cast<uint2 *>(lds + Lo.aux + s * Lo.aux_stride + idx * 512 + lane * 8) = *mask_ptr(A.act, A.act_mask, idx, A.n, sc, g);
        }
        __syncthreads();

#pragma clang loop unroll(disable)
        for (int l = 0; l < n_layers; ++l) {
            const LatLayer Ly = lat_layer_at(tab, l);
            const int w0 = Ly.wave0, t_out = Ly.t_out, op = Ly.op;
            if (wave >= w0 && wave < w0 + ((t_out + 1) >> 1)) {
                const int tile = 2 * (wave - w0);
                f4 acc[S][2];
                const bool scale = op & LAT_SCALE_AUX;   // d o = directional_input[:, :W]^T d h1 + sigma_out_layer^T d sigma (:51-52)
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[s][t][r] = scale ? W.aux[t][r] * dr[s][3] : W.aux[t][r];
                lat_run_layer<S>(W, tab, lds, Ly, acc);
                if (IG && (op & LAT_PE_POS)) {          // (waves 0, 1: k-blocks 2 wave, 2 wave + 1 of the position encoding)
#pragma unroll
                    for (int s = 0; s < S; ++s)
#pragma unroll
                        for (int t = 0; t < 2; ++t) dpe[IG ? s : 0][t] += acc[s][t];
                    continue;
                }
                if (IG && (op & LAT_PE_DIR)) {          // (wave 0: both k-blocks of the direction encoding)
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const float ux = din[IG ? s : 0][0], uy = din[IG ? s : 0][1], uz = din[IG ? s : 0][2];
                        const float nrm = sqrtf(ux * ux + uy * uy + uz * uz);
                        const float nx = ux / nrm, ny = uy / nrm, nz = uz / nrm;
                        float gx = 0.f, gy = 0.f, gz = 0.f;
                        const f4 dd[2] = {acc[s][0], acc[s][1]};
                        pe_backward<2>(dd, A.dir_nkb, nx, ny, nz, A.dir_L, A.dir_id, g, gx, gy, gz);
                        gx = sum_over_g(gx);
                        gy = sum_over_g(gy);
                        gz = sum_over_g(gz);
                        // d (u/|u|) -> d u = (g - n (n.g)) / |u|   (models/smpl_nerf_pipeline.py:54-55)
                        const float dot = nx * gx + ny * gy + nz * gz;
                        const unsigned off = (okay(s) && g == 0) ? (unsigned)sample_of(s) * 12u : LAT_OOB;
                        lat_store_b32(dd_rs, off, (gx - nx * dot) / nrm);
                        lat_store_b32(dd_rs, off == LAT_OOB ? off : off + 4u, (gy - ny * dot) / nrm);
                        lat_store_b32(dd_rs, off == LAT_OOB ? off : off + 8u, (gz - nz * dot) / nrm);
                    }
                    continue;
                }
                const bool mask = op & LAT_MASK_BITS;
                const int out_base = Ly.out_base, store_row = Ly.store_row;
                const int mask_at = Lo.aux + (Ly.mask_idx < 0 ? 0 : Ly.mask_idx) * 512 + lane * 8 + (tile >> 1);
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    // d X masked with X > 0 gives d Y of the layer below: the forward's sign bits, this wave's byte of the word
                    const unsigned m = mask ? *reinterpret_cast<const unsigned char *>(lds + mask_at + s * Lo.aux_stride) : 0xffu;
                    f4 v[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[t][r] = ((m >> (4 * t + r)) & 1u) ? acc[s][t][r] : 0.f;
                    if (out_base >= 0) {
                        f4 *dst = reinterpret_cast<f4 *>(lds + out_base + s * LAT_ACT_BYTES + tile * 1024 + W.voff);
                        dst[0] = v[0];
                        dst[64] = v[1];
                    }
                    lat_store_f4(dy_rs, dy_off(s, store_row + tile), v[0]);
                    lat_store_f4(dy_rs, dy_off(s, store_row + tile + 1), v[1]);
                }
            }
            if (op & LAT_BARRIER) __syncthreads();
        }
        if constexpr (IG) {
            // d (position encoding): wave 1's tiles 2, 3 -> LDS -> wave 0, which holds tiles 0, 1; encoder backward in the order of
            // mlp_bwd_kernel (pe_backward over k-blocks 0 .. 3, then the four lane groups)
            __syncthreads();   // (every wave is through its last layer: the activation buffers are free)
            if (wave == 1) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    f4 *blk = reinterpret_cast<f4 *>(lds + Lo.act[0] + s * LAT_ACT_BYTES + W.voff);
                    blk[0] = dpe[s][0];
                    blk[64] = dpe[s][1];
                }
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const f4 *blk = reinterpret_cast<const f4 *>(lds + Lo.act[0] + s * LAT_ACT_BYTES + W.voff);
                    const f4 d4[4] = {dpe[s][0], dpe[s][1], blk[0], blk[64]};
                    float gx = 0.f, gy = 0.f, gz = 0.f;
                    if (A.pos_nkb > 0) pe_backward<4>(d4, A.pos_nkb, xin[s][0], xin[s][1], xin[s][2], A.pos_L, A.pos_id, g, gx, gy, gz);
                    gx = sum_over_g(gx);
                    gy = sum_over_g(gy);
                    gz = sum_over_g(gz);
                    const unsigned off = (okay(s) && g == 0) ? (unsigned)sample_of(s) * 12u : LAT_OOB;
                    lat_store_b32(dx_rs, off, gx);
                    lat_store_b32(dx_rs, off == LAT_OOB ? off : off + 4u, gy);
                    lat_store_b32(dx_rs, off == LAT_OOB ? off : off + 8u, gz);
                }
            }
        }
    }
}

// the transposed stream (mlp_plan.h: make_bwd_plan, with the encoder-column transposes when B was built with input_grad) as the
// latency kernel walks it
static void lat_table_bwd(const Plan &P, const BwdPlan &B, const TrainLayout &L, const LatLds &Lo, LatTable &T) {
    T.n = B.nl;
    T.stream_bytes = (B.total_slabs + SLAB_PAD) * SLAB_BYTES;
    const int nh = P.n_hidden;
    int n_chain = 0;   // layers of the chain proper (every layer but the encoder-column transposes)
    for (int bi = 0; bi < B.nl; ++bi) n_chain += P.layer[B.layer[bi].fwd].seg[B.layer[bi].seg].type == SEG_PE ? 0 : 1;
    int j = 0;         // chain layers seen so far
    for (int bi = 0; bi < B.nl; ++bi) {
        const BwdLayer &Bl = B.layer[bi];
        LatLayer &o = T.l[bi];
        lat_layer_stream(o, Bl.first_slab, Bl.nkb, Bl.t_out);
        o.wave0 = 0;
        o.b_stride0 = LAT_ACT_BYTES;
        o.b_n0 = 1 << 20;
        o.b_base1 = o.b_stride1 = 0;
        o.pad_[0] = o.pad_[1] = 0;
        if (P.layer[Bl.fwd].seg[Bl.seg].type == SEG_PE) {
            // encoder-column transposes: they contract the d Y the chain layer before them left in LDS (the same operand the next chain
            // layer reads), on waves 0 .. t_out / 2 - 1; no output buffer, no barrier
            o.b_base0 = Lo.act[(j - 1) & 1];
            o.out_base = -1;
            o.store_row = -1;
            o.mask_idx = -1;
            o.op = Bl.fwd == nh + 3 ? LAT_PE_DIR : LAT_PE_POS;
            continue;
        }
        o.b_base0 = j == 0 ? Lo.act[1] : Lo.act[(j + 1) & 1];   // (j = 0: the d rgb operand phase 0 leaves in the second buffer)
        o.out_base = bi + 1 < B.nl ? Lo.act[j & 1] : -1;        // (somebody - chain layer or encoder columns - reads it)
        // forward layer whose d Y this layer produces (mlp_train.hip: mlp_bwd_kernel)
        const int fl = j == 0 ? nh + 4 : j == 1 ? nh + 3 : j == 2 ? nh + 1 : nh - (j - 3);
        o.store_row = L.dy[fl];
        o.mask_idx = j == 0 ? nh + 1 : j >= 3 ? fl : -1;
        o.op = LAT_BARRIER | (o.mask_idx >= 0 ? LAT_MASK_BITS : 0) | (j == 2 ? LAT_SCALE_AUX : 0);
        ++j;
    }
    (void)n_chain;
    lat_table_finish(T);
}

bool lat_enabled() {
    static const bool on = [] { const char *e = getenv("SNERF_LAT"); return e ? atoi(e) != 0 : true; }();
    return on;
}
// Splits n16 tiles into a main launch (n_cu workgroups x passes x S tiles) and a remainder launch of one pass; returns the number of
// launches (1 or 2).  The makespan is passes * S + S' tile units against the ideal n16 / n_cu: at most one unit above it.
struct LatLaunch {
    int S, grid, passes;
    int64_t tile_off, tile_end;
    int mixed_a;   // > 0: S = 3 as a mixed launch - mixed_a workgroups of two tiles, grid - mixed_a of one (forward kernels)
};
// pair_s: the largest S' of which TWO workgroups fit a CU's LDS (0: none - the split of the cost model)
static int lat_split(int64_t n16, int n_cu, int s_max, LatLaunch (&out)[2], int pair_s = 0, bool mixed_ok = false) {
    int k = 0;
    int64_t done = 0;
    const int S = (int)std::min<int64_t>(s_max, (n16 + n_cu - 1) / n_cu);
    const int64_t per_round = (int64_t)S * n_cu;
    const int passes = (int)(n16 / per_round);
    // r06: a ONE-pass launch with an even number of tiles per CU runs as TWO workgroups of S / 2 tiles per CU (16 waves: the kernels
    // keep to 128 VGPRs and 2 x lat_lds(2) fits the LDS): each workgroup's MFMAs cover the other's layer boundaries (barrier,
    // descriptor load, first operand read).  Measured (tools/ab/host_profile_render.py, train_loop.py): 128-ray render 0.408 ->
    // 0.358 ms, 128 / 256-ray steps -2 %; with several passes per workgroup the doubled weight stream costs what the overlap gains
    // (800-ray render 2.02 -> 2.07 ms) - those stay one workgroup per CU.  Same tiles, same arithmetic: bit-identical.
    // three tiles per CU in one pass: a two-tile and a one-tile workgroup per CU (mlp_fwd_lat_mixed_kernel)
    auto mixed = [&](int64_t first, int64_t tiles) {
        const int n_a = (int)std::min<int64_t>(n_cu, tiles / 2);
        return LatLaunch{3, n_a + (int)(tiles - 2 * (int64_t)n_a), 1, first, first + tiles, n_a};
    };
    if (passes > 0) {
        if (passes == 1 && S % 2 == 0 && S / 2 <= pair_s) out[k++] = LatLaunch{S / 2, 2 * n_cu, 1, 0, per_round, 0};
        else if (passes == 1 && S == 3 && mixed_ok && pair_s >= 2) out[k++] = mixed(0, per_round);
        else out[k++] = LatLaunch{S, n_cu, passes, 0, per_round * passes, 0};
        done = per_round * passes;
    }
    const int64_t r = n16 - done;
    if (r > 0) {
        int S2 = (int)std::min<int64_t>(s_max, (r + n_cu - 1) / n_cu);
        if (S2 == 3 && mixed_ok && pair_s >= 2) {
            out[k++] = mixed(done, r);
            return k;
        }
        if (S2 % 2 == 0 && S2 / 2 <= pair_s) S2 /= 2;
        out[k++] = LatLaunch{S2, (int)((r + S2 - 1) / S2), 1, done, n16, 0};
    }
    return k;
}

// Which form is faster for a call of n16 tiles?  Measured per launch on MI355X (r05, tools/ab/lat_trace.sh, microseconds):
//   latency kernels    a launch costs `fixed` + per pass of S tiles 4 + per_tile * S: inference 21 / 41, training forward 22 / 43, dgrad 19 / 38
//   throughput kernels one round of the chip: 64-sample tiles (calls of <= 64 x CUs samples) 178 / 191 / 170,
//                      128-sample tiles 295 / 352 / 314 per round of 128 x CUs samples
// (n = 4096: 62 against 178; 16 384: 185 / 182; 20 480: 247 / 295; 32 768: 370 / 295; 51 200 - inference.py's 800 rays, coarse -
// 567 / 590; 153 600: 1611 / 1475.)  The latency form wins below two tiles of 128 per CU wherever the throughput form would run a
// mostly empty round.
enum LatKind { LAT_INFER = 0, LAT_TRAIN_FWD = 1, LAT_DGRAD = 2 };
static double lat_cost(LatKind kind, int64_t n16, int n_cu, int s_max) {
    static const double fixed[3] = {21, 22, 19}, per_tile[3] = {41, 43, 38};
    LatLaunch Q[2];
    const int nq = lat_split(n16, n_cu, s_max, Q);
    double t = 0;
    for (int i = 0; i < nq; ++i) t += fixed[kind] + Q[i].passes * (4.0 + per_tile[kind] * Q[i].S);
    return t;
}
// 0: the throughput kernel alone; 1: the latency kernels alone; 2: the throughput kernel on the first n_main samples - whole rounds of
// 128-sample tiles on every CU - and the latency kernels on the rest (a call of 2.3 rounds is two full rounds + 0.3 of one at the
// latency kernels' granularity of 16 samples instead of a third, mostly empty round: inference.py's 800 rays, 153 600 fine samples =
// 4.69 rounds)
// The dgrad is different: inside a training step the coarse net's backward runs beside the fine net's on a second stream (chunks of up
// to 1024 rays), and the partly empty rounds of one net's throughput dgrad are filled by the other's workgroups - while a latency
// kernel takes a whole CU's LDS.  Measured on whole steps (profiles/r05_lat_step_ab.txt: 128 rays 1.237 ms with the throughput dgrad
// / 1.261 with the latency dgrad, 800 rays 6.527 / 6.606; 64 rays 0.90 / 0.72): the latency dgrad pays up to four tiles per CU
// (the README's 64-ray batches), and never as the remainder of a throughput launch.
LatChoice lat_choose(int kind_i, int64_t n, int n_cu, int s_max) {
    const LatKind kind = (LatKind)kind_i;
    const int64_t max_tiles = kind == LAT_DGRAD ? 4 : 64;
    static const double round64[3] = {178, 191, 170}, round128[3] = {295, 352, 314};
    const int64_t n16 = (n + 15) / 16, per_round = (int64_t)128 * n_cu;
    const double t_thr = n <= (int64_t)64 * n_cu ? round64[kind] : (double)((n + per_round - 1) / per_round) * round128[kind];
    LatChoice best{0, 0};
    double t_best = t_thr;
    if (n16 <= max_tiles * n_cu) {
        const double t = lat_cost(kind, n16, n_cu, s_max);
        if (t < t_best) t_best = t, best = LatChoice{1, 0};
    }
    const int64_t rounds = n / per_round, rem = n - rounds * per_round;
    if (kind != LAT_DGRAD && rounds >= 1 && rem > 0 && rounds <= 64) {
        const double t = rounds * round128[kind] + lat_cost(kind, (rem + 15) / 16, n_cu, s_max);
        if (t < 0.98 * t_best) best = LatChoice{2, rounds * per_round};
    }
    return best;
}

static LatLds lat_lds_fwd(int S, const Plan &P) {
    const bool add_first = P.add_dim && P.layer[0].seg[0].type == SEG_ADD;
    return lat_lds(S, P.pos_nkb, P.dir_nkb, 0, P.add_nkb, add_first ? 1 : 0);
}

template <int S, bool TRAIN>
static int launch_fwd_lat_s(const Plan &P, const TrainLayout &L, const FwdArgs &A, const LatLaunch &Q, hipStream_t s) {
    const LatLds Lo = lat_lds_fwd(S, P);
    LatTable T;
    lat_table_fwd(P, L, Lo, T);
    static LdsRaised raised;   // per device, per instantiation
    if (int rc = raise_dynamic_lds(reinterpret_cast<const void *>(mlp_fwd_lat_kernel<S, TRAIN>), Lo.total, raised, "mlp_fwd_lat")) return rc;
    const LatGeom G{Q.tile_off, Q.tile_end, Q.passes};
    hipLaunchKernelGGL((mlp_fwd_lat_kernel<S, TRAIN>), dim3((unsigned)Q.grid), dim3(LAT_THREADS), Lo.total, s, T, A, G, Lo);
    return check_launch("mlp_fwd_lat");
}

// 0 = launched; 1 = this call is not for the latency kernels (the caller runs the throughput form); < 0 = error
static int lat_s_max_fwd(const Plan &P) {
    int s_max = LAT_MAX_S;
    while (s_max > 1 && lat_lds_fwd(s_max, P).total > 160 * 1024) --s_max;
    return lat_lds_fwd(s_max, P).total > 160 * 1024 ? 0 : s_max;
}
// plain RenderRayNet plans of width 256; per-ray additional inputs of up to 128 k-blocks (what fits the LDS beside one sample tile)
static bool lat_covers(const Plan &P) { return lat_enabled() && P.width == 256 && P.add_nkb <= 100 && P.nlayers == P.n_hidden + 6; }

template <bool TRAIN>
LatChoice lat_choose_fwd(const Plan &P, int64_t n) {
    if (!lat_covers(P)) return LatChoice{0, 0};
    const int n_cu = device_cu_count("mlp_fwd_lat"), s_max = lat_s_max_fwd(P);
    if (n_cu < 1 || !s_max) return LatChoice{0, 0};
    TrainLayout L;
    make_train_layout(P, L);
    // the masked stores go through 2 GiB buffer resources (mlp_lat_device.h: LAT_STORE_RANGE)
    if ((TRAIN ? (int64_t)L.act_rows * 64 : 16) * n >= (int64_t)LAT_STORE_RANGE) return LatChoice{0, 0};
    return lat_choose(TRAIN ? LAT_TRAIN_FWD : LAT_INFER, n, n_cu, s_max);
}
template LatChoice lat_choose_fwd<false>(const Plan &, int64_t);
template LatChoice lat_choose_fwd<true>(const Plan &, int64_t);

// the latency kernels on samples [first_sample, A.n) (first_sample a multiple of 16)
template <bool TRAIN>
int launch_fwd_lat(const Plan &P, const FwdArgs &A, hipStream_t s, int64_t first_sample) {
    const int n_cu = device_cu_count("mlp_fwd_lat");
    if (n_cu < 1) return n_cu;
    const int s_max = lat_s_max_fwd(P);
    if (!lat_covers(P) || !s_max || first_sample % 16) return fail(SNERF_E_BADARG, "mlp_fwd_lat: not a call for the latency kernels");
    TrainLayout L;
    make_train_layout(P, L);
    const int64_t t0 = first_sample / 16, n16 = (A.n + 15) / 16 - t0;
    LatLaunch Q[2];
    int pair_s = 0;
    for (int h = 1; h <= 2; ++h)
        if (2 * lat_lds_fwd(h, P).total <= 160 * 1024) pair_s = h;
    const int nq = lat_split(n16, n_cu, s_max, Q, pair_s, true);
    for (int i = 0; i < nq; ++i) {
        Q[i].tile_off += t0;
        Q[i].tile_end += t0;
        int rc;
        if (Q[i].mixed_a > 0) {
            const LatLds Lo = lat_lds_fwd(2, P);
            LatTable T;
            lat_table_fwd(P, L, Lo, T);
            static LdsRaised raised;   // per device, per instantiation
            if ((rc = raise_dynamic_lds(reinterpret_cast<const void *>(mlp_fwd_lat_mixed_kernel<TRAIN>), Lo.total, raised, "mlp_fwd_lat"))) return rc;
            const int64_t split = Q[i].tile_off + 2 * (int64_t)Q[i].mixed_a;
            const LatGeom GA{Q[i].tile_off, split, 1}, GB{split, Q[i].tile_end, 1};
            hipLaunchKernelGGL((mlp_fwd_lat_mixed_kernel<TRAIN>), dim3((unsigned)Q[i].grid), dim3(LAT_THREADS), Lo.total, s, T, A, GA, GB, Lo, Q[i].mixed_a);
            if ((rc = check_launch("mlp_fwd_lat"))) return rc;
            continue;
        }
        switch (Q[i].S) {
            case 1: rc = launch_fwd_lat_s<1, TRAIN>(P, L, A, Q[i], s); break;
            case 2: rc = launch_fwd_lat_s<2, TRAIN>(P, L, A, Q[i], s); break;
            case 3: rc = launch_fwd_lat_s<3, TRAIN>(P, L, A, Q[i], s); break;
            default: rc = launch_fwd_lat_s<4, TRAIN>(P, L, A, Q[i], s); break;
        }
        if (rc) return rc;
    }
    return 0;
}
template int launch_fwd_lat<false>(const Plan &, const FwdArgs &, hipStream_t, int64_t);
template int launch_fwd_lat<true>(const Plan &, const FwdArgs &, hipStream_t, int64_t);

template <int S, bool IG>
static int launch_bwd_lat_s(const Plan &P, const BwdPlan &B, const TrainLayout &L, const BwdArgs &A, const LatLaunch &Q, hipStream_t s) {
    const LatLds Lo = lat_lds(S, 0, 0, (P.n_hidden + 2) * 512);
    LatTable T;
    lat_table_bwd(P, B, L, Lo, T);
    static LdsRaised raised;
    if (int rc = raise_dynamic_lds(reinterpret_cast<const void *>(mlp_bwd_lat_kernel<S, IG>), Lo.total, raised, "mlp_bwd_lat")) return rc;
    const LatGeom G{Q.tile_off, Q.tile_end, Q.passes};
    hipLaunchKernelGGL((mlp_bwd_lat_kernel<S, IG>), dim3((unsigned)Q.grid), dim3(LAT_THREADS), Lo.total, s, T, A, G, Lo);
    return check_launch("mlp_bwd_lat");
}

static int lat_s_max_bwd(const Plan &P) {
    int s_max = LAT_MAX_S;
    while (s_max > 1 && lat_lds(s_max, 0, 0, (P.n_hidden + 2) * 512).total > 160 * 1024) --s_max;
    return lat_lds(s_max, 0, 0, (P.n_hidden + 2) * 512).total > 160 * 1024 ? 0 : s_max;
}
// input gradients: the default-sized encoders (4 position / 2 direction k-blocks; the kernel's waves 0, 1 hold two tiles each), nets
// that read directions, and d_x / d_dirs within the store range
static bool lat_ig_covers(const Plan &P, int64_t n) {
    const PeTiles pt = bwd_pe_tiles(P);
    return pt.pos == 4 && pt.dir == 2 && P.add_dim == 0 && n * 12 < (int64_t)LAT_STORE_RANGE;
}
// With input gradients inside a step whose two backward chains run side by side (smpl_nerf, small chunks): the throughput kernels of
// the two nets pack - 64 rays are 64 + 192 workgroups of 64 samples, one round of the chip for both, 190 us - while each latency
// kernel takes every CU's LDS and the two run one after the other (74 + 166 us; whole step 0.98 -> 1.02 ms, r05).  There the latency
// form is kept for nets that leave room for the other one (up to 3/4 tile per CU); alone on the chip (autograd path, no auxiliary
// stream: 189 -> 74 us at 4096 samples, 189 -> 166 at 12 288) the rule of the plain dgrad holds.
LatChoice lat_choose_bwd(const Plan &P, int64_t n, bool input_grad, bool beside_another_net) {
    if (!lat_covers(P) || (input_grad && !lat_ig_covers(P, n))) return LatChoice{0, 0};
    const int n_cu = device_cu_count("mlp_bwd_lat"), s_max = lat_s_max_bwd(P);
    if (n_cu < 1 || !s_max) return LatChoice{0, 0};
    if (input_grad && beside_another_net && (n + 15) / 16 > (int64_t)n_cu * 3 / 4) return LatChoice{0, 0};
    BwdPlan B;
    make_bwd_plan(P, B, input_grad);
    TrainLayout L;
    make_train_layout(P, L);
    if (B.nl > LAT_MAX_LAYERS || (int64_t)L.dy_rows * n * 64 >= (int64_t)LAT_STORE_RANGE) return LatChoice{0, 0};
    return lat_choose(LAT_DGRAD, n, n_cu, s_max);
}

// the dgrad of launch_bwd (mlp_train.hip) on samples [first_sample, A.n) with the latency kernel (A.d_x != NULL: with input gradients,
// A.packed_t is then the input_grad stream)
int launch_bwd_lat(const Plan &P, const BwdArgs &A, hipStream_t s, int64_t first_sample) {
    const int n_cu = device_cu_count("mlp_bwd_lat");
    if (n_cu < 1) return n_cu;
    const int s_max = lat_s_max_bwd(P);
    const bool ig = A.d_x != nullptr;
    if (!lat_covers(P) || !s_max || first_sample % 16 || (ig && !lat_ig_covers(P, A.n)))
        return fail(SNERF_E_BADARG, "mlp_bwd_lat: not a call for the latency kernel");
    BwdPlan B;
    make_bwd_plan(P, B, ig);
    TrainLayout L;
    make_train_layout(P, L);
    const int64_t t0 = first_sample / 16, n16 = (A.n + 15) / 16 - t0;
    LatLaunch Q[2];
    int pair_s = 0;
    for (int h = 1; h <= 2; ++h)
        if (2 * lat_lds(h, 0, 0, (P.n_hidden + 2) * 512).total <= 160 * 1024) pair_s = h;
    const int nq = lat_split(n16, n_cu, s_max, Q, pair_s);
    for (int i = 0; i < nq; ++i) {
        Q[i].tile_off += t0;
        Q[i].tile_end += t0;
        int rc;
        if (ig) {
            switch (Q[i].S) {
                case 1: rc = launch_bwd_lat_s<1, true>(P, B, L, A, Q[i], s); break;
                case 2: rc = launch_bwd_lat_s<2, true>(P, B, L, A, Q[i], s); break;
                case 3: rc = launch_bwd_lat_s<3, true>(P, B, L, A, Q[i], s); break;
                default: rc = launch_bwd_lat_s<4, true>(P, B, L, A, Q[i], s); break;
            }
        } else {
            switch (Q[i].S) {
                case 1: rc = launch_bwd_lat_s<1, false>(P, B, L, A, Q[i], s); break;
                case 2: rc = launch_bwd_lat_s<2, false>(P, B, L, A, Q[i], s); break;
                case 3: rc = launch_bwd_lat_s<3, false>(P, B, L, A, Q[i], s); break;
                default: rc = launch_bwd_lat_s<4, false>(P, B, L, A, Q[i], s); break;
            }
        }
        if (rc) return rc;
    }
    return 0;
}

}  // namespace snerf
