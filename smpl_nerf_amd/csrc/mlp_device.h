// Device-side building blocks shared by the forward (mlp.hip) and training (mlp_train.hip) kernels: the two slab pipes - SlabPipeDma
// (L2 -> LDS by global_load_lds into a 4-slot ring: the headline inference kernel of width 256 and every kernel of the widths above
// 256) and SlabPipe (L2 -> registers -> 3-slot LDS ring: training forward and dgrad up to width 256) -, the k-block MFMA step, the
// layer walker and the in-register positional-encoding operands.  See mlp_plan.h for the operand algebra.
#pragma once
#include <type_traits>
#include "snerf_common.h"
#include "mlp_plan.h"

namespace snerf {

typedef float f4 __attribute__((ext_vector_type(4)));

struct TrainLayout;
// defined in mlp_train.hip: split-K wgrad + fixed-order reduce for any (Plan, TrainLayout)
// n_beside: samples of ANOTHER net's weight gradient that runs on a second stream at the same time (train_step.hip: the coarse
// net's backward beside the fine net's), 0 = this launch has the chip to itself - small calls then split their jobs for their
// share of the CUs, n / (n + n_beside), so that both nets' workgroups are resident in one round
int launch_wgrad(const Plan &P, const TrainLayout &L, const float *act, const float *dy, int64_t n, float *gpart,
                 float *flat_grad, hipStream_t s, int wide_nsplit = 0, bool accumulate = false, int64_t n_beside = 0);
// defined in mlp_train.hip / mlp_train_bf16.hip: dgrad + wgrad + reduce of one net on n samples (the bodies of snerf_mlp_bwd_f32 /
// snerf_mlp_bwd_inputs_f32 and their split-precision twins); accumulate: flat_grad += instead of = (train_step.hip)
int launch_bwd(const snerf_mlp_desc *desc, const float *packed_t, const float *act, const float *d_raw, int64_t n, float *dy,
               float *gpart, float *flat_grad, const float *x, const float *dirs, int dirs_per_sample, int spr, float *d_x,
               float *d_dirs, snerf_stream_t stream, bool accumulate = false, bool beside_another_net = false, int64_t n_beside = 0);
// (beside_another_net: the caller runs another net's backward on a second stream at the same time - train_step.hip)
// defined in warp.hip: the body of snerf_warp_bwd_f32
int launch_warp_bwd(const snerf_warp_desc *desc, const float *packed_t, const float *act, const float *d_warp, int64_t n, float *dy,
                    float *gpart, float *flat_grad, snerf_stream_t stream, bool accumulate = false);
int launch_bwd_bf16(const snerf_mlp_desc *desc, const void *packed_t, int nsplit, const float *act, const float *d_raw, int64_t n,
                    float *dy, float *gpart, float *flat_grad, const float *x, const float *dirs, int dirs_per_sample, int spr,
                    float *d_x, float *d_dirs, snerf_stream_t stream, bool accumulate = false);
// defined in mlp_lat.hip: the latency-class kernels of small calls, and which form a call takes (mode 0: the throughput kernel alone,
// 1: the latency kernels alone, 2: the throughput kernel on the first n_main samples - whole rounds of the chip - and the latency
// kernels on the rest)
struct LatChoice {
    int mode;
    int64_t n_main;
};
struct FwdArgs;
struct BwdArgs;
template <bool TRAIN>
LatChoice lat_choose_fwd(const Plan &P, int64_t n);
template <bool TRAIN>
int launch_fwd_lat(const Plan &P, const FwdArgs &A, hipStream_t s, int64_t first_sample);
LatChoice lat_choose_bwd(const Plan &P, int64_t n, bool input_grad, bool beside_another_net = false);
int launch_bwd_lat(const Plan &P, const BwdArgs &A, hipStream_t s, int64_t first_sample);
// defined in mlp.hip: packs params_flat into the slab stream described by `P` (any plan)
int launch_pack(const Plan &P, const float *params_flat, float *packed, hipStream_t s, const char *what);

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
    int add_dim, add_nkb, add_first;
    int use_dir;
    int enc_stride;
    // training only: activation buffer in tile-row-major layout (mlp_plan.h TrainLayout)
    float *act;
    int act_pe, act_add, act_dpe, act_x1, act_o, act_h1, act_h2, act_mask;
    // inference with per-ray additional inputs (FOLD instantiation): fold[(ray * fold_slots + slot) * WIDTH + o] =
    // sum_c W_l[o, add columns] * add[ray][c] for layer 0 (slot 0) and the skip layers in order (mlp.hip: mlp_add_fold_kernel)
    const float *fold;
    int fold_slots;
    int no_fold;         // (host only) the caller asked for the per-sample form: SNERF_FWD_NO_RAY_FOLD
    float *fold_ws;      // (host only) caller's workspace for the per-ray fold table, or null: per-sample form
    int64_t fold_ws_bytes;
    int act_rows;      // f16x3 training forward: the per-layer |X| exponents go behind this many tile-rows of `act`
    // split-bf16 training forward only: encoder / additional k-block counts of the 16-wide (fp32) plan, which
    // defines the activation layout the backward kernels read
    int pos_nkb16, add_nkb16, dir_nkb16;
    int total_slabs;   // slabs of the weight stream (persistent workgroups wrap around)
    int64_t n_tiles;   // sample tiles (NWAVES x 16 samples)
};

#ifndef SNERF_EXP_NOSTORE
#define SNERF_EXP_NOSTORE 0
#endif
// one tile (16 features of this lane's sample) <-> the tile-row-major activation buffer
__device__ __forceinline__ void store_tile(float *buf, int row, int64_t n, int64_t sample, int g, f4 v) {
#if SNERF_EXP_NOSTORE != 1   // (diagnostic build: the training forward without its activation stores - DESIGN 6)
    __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(buf + ((int64_t)row * n + sample) * 16 + 4 * g));
#endif
}
template <int N>
__device__ __forceinline__ void store_tiles(float *buf, int row0, int64_t n, int64_t sample, int g, const f4 (&tiles)[N]) {
#pragma unroll
    for (int t = 0; t < N; ++t) store_tile(buf, row0 + t, n, sample, g, tiles[t]);
}
__device__ __forceinline__ f4 load_tile(const float *buf, int row, int64_t n, int64_t sample, int g) {
    return *reinterpret_cast<const f4 *>(buf + ((int64_t)row * n + sample) * 16 + 4 * g);
}

// ReLU sign mask of one layer output for the split-bf16 dgrad kernel: bit 4 t + r of this lane's 64-bit word is
// (tiles[t][r] > 0).  Mask `idx` lives in tile-row mask_row + idx / 2, floats (idx & 1) * 8 + 2 g of the sample.
__device__ __forceinline__ uint2 *mask_ptr(const float *buf, int mask_row, int idx, int64_t n, int64_t sample, int g) {
    return reinterpret_cast<uint2 *>(const_cast<float *>(buf) + ((int64_t)(mask_row + (idx >> 1)) * n + sample) * 16 +
                                     (idx & 1) * 8 + 2 * g);
}
// 32-tile layers (width 512): mask `idx` is tile-row mask_row + idx, 16 bytes per lane group
__device__ __forceinline__ uint4 *mask_ptr4(const float *buf, int mask_row, int idx, int64_t n, int64_t sample, int g) {
    return reinterpret_cast<uint4 *>(const_cast<float *>(buf) + ((int64_t)(mask_row + idx) * n + sample) * 16 + 4 * g);
}
// ROW_PER_MASK: the layout of a net with 32-tile layers (also for its 16-tile directional branch, which stores two words there)
template <int N, bool ROW_PER_MASK = (N > 16)>
__device__ __forceinline__ void store_mask(float *buf, int mask_row, int idx, int64_t n, int64_t sample, int g,
                                           const f4 (&tiles)[N]) {
    static_assert(N <= 32, "up to four mask words");
    static_assert(N <= 16 || ROW_PER_MASK, "four words need a tile-row of their own");
    constexpr int NWORDS = N <= 16 ? 2 : 4;
    unsigned w[NWORDS];
#pragma unroll
    for (int q = 0; q < NWORDS; ++q) w[q] = 0u;
#pragma unroll
    for (int t = 0; t < N; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // x > 0 <=> its bit pattern, read as a signed integer, is > 0: clamp to {0, 1} and shift into place
            const int bit = min(max(__float_as_int(tiles[t][r]), 0), 1);
            w[t >> 3] |= static_cast<unsigned>(bit) << (((t & 7) << 2) | r);
        }
    // the address is formed here, not hoisted to the top of the tile (where it would cost two registers per layer)
    asm volatile("" : "+v"(sample));
    if constexpr (N <= 16 && ROW_PER_MASK) {
        *reinterpret_cast<uint2 *>(mask_ptr4(buf, mask_row, idx, n, sample, g)) = uint2{w[0], w[1]};
    } else if constexpr (N <= 16) {
#if SNERF_EXP_NOSTORE != 2   // (diagnostic build: without the sign-mask stores)
        *mask_ptr(buf, mask_row, idx, n, sample, g) = uint2{w[0], w[1]};
#else
        asm volatile("" ::"v"(w[0]), "v"(w[1]));
#endif
    } else {   // 32 tiles: four words per lane, one mask per tile-row (TrainLayout::mask)
        *mask_ptr4(buf, mask_row, idx, n, sample, g) = uint4{w[0], w[1], w[2], w[3]};
    }
}

// The 3-slot ring (99 KiB) is dynamic LDS: a launch gets 64 KiB unless the limit is raised per kernel once.
constexpr int RING_BYTES = 3 * SLAB_FLOATS * 4;
#define SNERF_LAUNCH_RING(kernel, grid, block, stream, ...)                                                            \
    do {                                                                                                               \
        static ::snerf::LdsRaised snerf_lds_raised_; /* per device (snerf_common.h) */                                 \
        if (int snerf_rc_ = ::snerf::raise_dynamic_lds(reinterpret_cast<const void *>(kernel), ::snerf::RING_BYTES,    \
                                                       snerf_lds_raised_, #kernel))                                    \
            return snerf_rc_;                                                                                          \
        hipLaunchKernelGGL(kernel, grid, block, ::snerf::RING_BYTES, stream, __VA_ARGS__);                             \
    } while (0)

// ... the 4-slot ring of the LDS-DMA pipe (SlabPipeDma below: the widths above 256), 132 KiB
constexpr int RING4_BYTES = 4 * SLAB_FLOATS * 4;
#define SNERF_LAUNCH_RING4(kernel, grid, block, stream, ...)                                                           \
    do {                                                                                                               \
        static ::snerf::LdsRaised snerf_lds_raised_; /* per device (snerf_common.h) */                                 \
        if (int snerf_rc_ = ::snerf::raise_dynamic_lds(reinterpret_cast<const void *>(kernel), ::snerf::RING4_BYTES,   \
                                                       snerf_lds_raised_, #kernel))                                    \
            return snerf_rc_;                                                                                          \
        hipLaunchKernelGGL(kernel, grid, block, ::snerf::RING4_BYTES, stream, __VA_ARGS__);                            \
    } while (0)
// Which kernels take which pipe (r05 measurements, DESIGN_HISTORY.md; the A/B switches that selected them are gone - the patches
// under tools/ab/ are the record): the widths above 256 (one wave per SIMD) and the 8-wave inference kernel of width 256 stream
// their slabs global -> LDS by DMA into the 4-slot ring (SlabPipeDma, below); the training forward and the dgrad of the widths up
// to 256 measured no gain from it and keep the register-staged 3-slot ring (SlabPipe).

// Streams the slab sequence global -> registers -> LDS ring (3 slots).
template <int NT>
struct SlabPipe {
    static constexpr int NA = SLAB_A_FLOATS / 4 / NT;  // f4 per thread in the A region (NT=256: 4, 512: 2)
    const f4 *g;   // this thread's read cursor in the packed stream
    const f4 *g0;  // ... and its position at slab 0: persistent kernels run the stream once per sample tile and wrap
    int src, total;
    float *ring;
    f4 st[NA], st_aux;
    f4 pa0, pa1;  // first A-operand pair of the next k-block, prefetched (see kblock)
    int tid, rd, wr;

    __device__ __forceinline__ void load() {
#pragma unroll
        for (int i = 0; i < NA; ++i) st[i] = g[i * NT];
        if (tid < 64) st_aux = g[SLAB_A_FLOATS / 4];
        g += SLAB_FLOATS / 4;
        if (++src == total) {
            src = 0;
            g = g0;
        }
    }
    __device__ __forceinline__ void store(int slot) {
        f4 *d = reinterpret_cast<f4 *>(ring + slot * SLAB_FLOATS) + tid;
#pragma unroll
        for (int i = 0; i < NA; ++i) d[i * NT] = st[i];
        if (tid < 64) d[SLAB_A_FLOATS / 4] = st_aux;
    }
    // total_slabs: length of the stream for kernels that run it repeatedly (one pass per sample tile); a one-pass kernel
    // leaves the default and reads on into the zero padding slabs
    __device__ __forceinline__ void prologue(const float *packed, float *ring_, int tid_, int total_slabs = 0x7fffffff) {
        ring = ring_;
        tid = tid_;
        g = g0 = reinterpret_cast<const f4 *>(packed) + tid;
        src = 0;
        total = total_slabs;
        load(); store(0);
        load(); store(1);
        load();
        rd = 0;
        wr = 2;
        __syncthreads();
        const f4 *np = reinterpret_cast<const f4 *>(ring) + (tid & 63);
        pa0 = np[0];
        pa1 = np[64];
    }
    __device__ __forceinline__ const float *acquire() const { return ring + rd * SLAB_FLOATS; }
    __device__ __forceinline__ const float *peek_next() const { return ring + (rd == 2 ? 0 : rd + 1) * SLAB_FLOATS; }
    // the two halves of release(): stage() may run anywhere inside the current slab's MFMAs (slot `wr` has been free since the
    // barrier that ended the previous slab), advance() ends the slab
    __device__ __forceinline__ void stage() {
        store(wr);
        load();
    }
    __device__ __forceinline__ void advance() {
        __syncthreads();
        rd = rd == 2 ? 0 : rd + 1;
        wr = wr == 2 ? 0 : wr + 1;
    }
    __device__ __forceinline__ void release() {
        stage();
        advance();
    }
    __device__ __forceinline__ void drain() {}   // (loads into registers: nothing of this pipe can land after the kernel)
};

// The same interface with the slabs copied global -> LDS by DMA (`global_load_lds`, 16 B per lane: one instruction moves a 1 KiB
// piece) into a 4-slot ring: no staging registers and half the instructions per slab of the register-staged pipe.  For the
// kernels that run ONE wave per SIMD (widths above 256): there nobody issues MFMAs while a wave sits in its refill instructions
// (each blocks the issuing wave for ~60 cycles, DESIGN_HISTORY 3.1), so their number is what the refill costs.
// Ring: slab p consumed, p+1 resident and visible (as above), p+2 landing or landed, p+3 being issued into the slot slab p-1 left
// at the last barrier.  A wave's pieces of a slab are complete before the barrier TWO slabs later (vmcnt leaves only the
// newest slab's pieces in flight), i.e. a slab has two periods to arrive.
template <int NT>
struct SlabPipeDma {
    static constexpr int NW = NT / 64;
    static constexpr int PIECES = SLAB_A_FLOATS / 256;   // 1 KiB pieces of the A region (32); the aux block is one more (wave 0)
    static constexpr int PPW = PIECES / NW;
    static_assert(PIECES % NW == 0, "pieces divide over the waves");
    const float *g, *g0;   // this lane's cursor in the stream (slab `src`, piece wave * PPW, lane's 16 B) and its position at slab 0
    int src, total;
    float *ring;
    f4 pa0, pa1;
    int tid, rd, wr;
    bool w0;

    __device__ __forceinline__ void issue(int slot) {
        float *dst = ring + slot * SLAB_FLOATS + (tid >> 6) * PPW * 256;   // wave-uniform; the hardware adds lane * 16 B
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + i * 256),
                                             (__attribute__((address_space(3))) void *)(dst + i * 256), 16, 0, 0);
        if (w0)   // wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + SLAB_A_FLOATS),
                                             (__attribute__((address_space(3))) void *)(ring + slot * SLAB_FLOATS + SLAB_A_FLOATS), 16, 0, 0);
        g += SLAB_FLOATS;
        if (++src == total) {
            src = 0;
            g = g0;
        }
    }
    __device__ __forceinline__ void prologue(const float *packed, float *ring_, int tid_, int total_slabs = 0x7fffffff) {
        ring = ring_;
        tid = tid_;
        w0 = __builtin_amdgcn_readfirstlane(tid_ >> 6) == 0;
        g = g0 = packed + (tid_ >> 6) * PPW * 256 + (tid_ & 63) * 4;   // (wave 0: the slab's first byte + the lane's 16)
        src = 0;
        total = total_slabs;
        issue(0);
        issue(1);
        issue(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        rd = 0;
        wr = 3;
        __syncthreads();
        const f4 *np = reinterpret_cast<const f4 *>(ring) + (tid & 63);
        pa0 = np[0];
        pa1 = np[64];
    }
    __device__ __forceinline__ const float *acquire() const { return ring + rd * SLAB_FLOATS; }
    __device__ __forceinline__ const float *peek_next() const { return ring + ((rd + 1) & 3) * SLAB_FLOATS; }
    __device__ __forceinline__ void stage() { issue(wr); }
    __device__ __forceinline__ void advance() {
        // everything but this wave's newest slab (PPW pieces, + 1 for wave 0) has landed: the slab issued one period ago is complete
        if (w0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        __syncthreads();
        rd = (rd + 1) & 3;
        wr = (wr + 1) & 3;
    }
    __device__ __forceinline__ void release() {
        stage();
        advance();
    }
    // end of the kernel: the slabs issued ahead of the last one consumed are still landing in this workgroup's LDS (s_endpgm waits
    // for a wave's outstanding memory operations by itself; said here so that it does not rest on that)
    __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};
// the pipe of a kernel of WIDTH features on NT threads
template <int WIDTH, int NT>
using PipeFor = std::conditional_t<(WIDTH > 256), SlabPipeDma<NT>, SlabPipe<NT>>;

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
// One k-block: T_OUT x (ds_read_b128 + 4 MFMA), software-pipelined: the A values of tile pair p+1 are read
// from LDS while the 8 MFMAs of pair p issue, and the first pair of the NEXT k-block (`next`, possibly in the
// next slab) is read during the last pair - so a wave never waits on an LDS read between MFMAs.  Tiles are
// walked in pairs so that consecutive MFMAs never share an accumulator (dependent latency of 16x16x4 is 40
// cycles vs 32 issue).  (pa0, pa1) carry the prefetched first pair from k-block to k-block.
// mid(): called once between two tile pairs early in the block (LayerRun: the staging of the slab after next, so that its LDS
// writes and global loads issue between MFMAs instead of behind the slab's last one)
struct NoMid {
    __device__ __forceinline__ void operator()() const {}
};
template <int T_OUT, class Mid = NoMid>
__device__ __forceinline__ void kblock(const float *a_kb, const float *a_next, f4 b, f4 (&acc)[T_OUT], f4 &pa0, f4 &pa1,
                                       int lane, Mid mid = Mid{}) {
    const f4 *ap = reinterpret_cast<const f4 *>(a_kb) + lane;
    const f4 *np = reinterpret_cast<const f4 *>(a_next) + lane;
    if constexpr (T_OUT == 1) {
        const f4 a = pa0;
        pa0 = np[0];
        pa1 = np[64];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc[0], 0, 0, 0);
    } else {
        f4 a0 = pa0, a1 = pa1;
#pragma unroll
        for (int to = 0; to < T_OUT; to += 2) {
            f4 n0, n1;
            if (to + 2 < T_OUT) {
                n0 = ap[(to + 2) * 64];
                n1 = ap[(to + 3) * 64];
            } else {
                n0 = np[0];
                n1 = np[64];
            }
            // pin the schedule: the two LDS reads of the NEXT pair issue first, then the 8 MFMAs of this pair
            // alternating between the two accumulators (hipcc otherwise sinks the reads to their first use and
            // groups the MFMAs by accumulator, which exposes both the LDS and the 40-cycle dependent latency)
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // 2 DS reads
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], b[r], acc[to], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                acc[to + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[r], b[r], acc[to + 1], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (to == (T_OUT >= 8 ? 2 : 0)) {
                mid();
                __builtin_amdgcn_sched_barrier(0);
            }
            a0 = n0;
            a1 = n1;
        }
        pa0 = a0;
        pa1 = a1;
    }
}

// Walks the k-blocks of one layer through the slab pipe.  A slab is released (staged slab written to LDS,
// next global loads issued, workgroup barrier) as soon as its last k-block has been issued; the first A
// pair of the following slab was read before that barrier - legal because slab p+1 has been resident and
// visible since the barrier that ended slab p-1 (the ring holds p, p+1 and the slot being filled with p+2).
template <int T_OUT, int NT, class PIPE = SlabPipe<NT>>
struct LayerRun {
    static constexpr int KPS = SLAB_TILES / T_OUT;
    PIPE &pipe;
    const float *slab;
    int kbl;  // k-block index inside the current slab
    int lane;

    __device__ __forceinline__ LayerRun(PIPE &p, int lane_) : pipe(p), slab(p.acquire()), kbl(0), lane(lane_) {}
    // bias -> accumulator init (aux block of the layer's first slab: bias[16*to + 4*g + r])
    __device__ __forceinline__ void init(f4 (&acc)[T_OUT]) {
        const f4 *aux = reinterpret_cast<const f4 *>(slab + SLAB_A_FLOATS) + (lane >> 4);
#pragma unroll
        for (int to = 0; to < (T_OUT < 16 ? T_OUT : 16); ++to) acc[to] = aux[to * 4];
        if constexpr (T_OUT > 16) {   // tiles 16 .. : the bias block of the layer's second slab (resident: LayerRun above)
            const f4 *aux2 = reinterpret_cast<const f4 *>(pipe.peek_next() + SLAB_A_FLOATS) + (lane >> 4);
#pragma unroll
            for (int to = 16; to < T_OUT; ++to) acc[to] = aux2[(to - 16) * 4];
        }
    }
    __device__ __forceinline__ void step(f4 b, f4 (&acc)[T_OUT]) {
        const float *cur = slab + kbl * (T_OUT * 256);
        const float *nxt = (kbl + 1 < KPS) ? cur + T_OUT * 256 : pipe.peek_next();
        if constexpr (T_OUT >= 8) {
            const bool last = kbl + 1 == KPS;     // the slab's last k-block carries the staging between its MFMAs
            kblock<T_OUT>(cur, nxt, b, acc, pipe.pa0, pipe.pa1, lane, [&]() __attribute__((always_inline)) {
                if (last) pipe.stage();
            });
            if (++kbl == KPS) {
                pipe.advance();
                slab = pipe.acquire();
                kbl = 0;
            }
        } else {
            kblock<T_OUT>(cur, nxt, b, acc, pipe.pa0, pipe.pa1, lane);
            if (++kbl == KPS) {
                pipe.release();
                slab = pipe.acquire();
                kbl = 0;
            }
        }
    }
    // a k-block of the stream that is not multiplied (FOLD: the additional-input columns arrive as a per-ray vector):
    // advance like step() and re-prime the A-operand prefetch from the block behind it
    __device__ __forceinline__ void skip() {
        const float *cur = slab + kbl * (T_OUT * 256);
        const f4 *np = reinterpret_cast<const f4 *>((kbl + 1 < KPS) ? cur + T_OUT * 256 : pipe.peek_next()) + lane;
        pipe.pa0 = np[0];
        pipe.pa1 = np[64];
        if (++kbl == KPS) {
            pipe.release();
            slab = pipe.acquire();
            kbl = 0;
        }
    }
    // end of the layer: a partially consumed slab is dropped (its tail is padding) and the prefetched pair
    // re-read from the slab the next layer starts in
    __device__ __forceinline__ void finish() {
        if (kbl != 0) {
            const f4 *np = reinterpret_cast<const f4 *>(pipe.peek_next()) + lane;
            pipe.pa0 = np[0];
            pipe.pa1 = np[64];
            pipe.release();
        }
    }
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


}  // namespace snerf
