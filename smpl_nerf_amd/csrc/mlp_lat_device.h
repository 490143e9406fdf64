// Latency-class kernels for small calls (README.md:23 `--batchsize=64`: 4096 + 12 288 samples; inference.py:231: 800 rays):
// device-side building blocks shared by the forward and the dgrad kernel of mlp_lat.hip.
//
// The throughput kernels (mlp.hip, mlp_train.hip) give one wave 16 samples for the whole network: a call of fewer than
// 128 samples per CU is then one wave's serial pass through the weight stream (9856 fp32 MFMAs = 131 us) with three of the
// four matrix pipes of most CUs idle.  Here a 16-sample tile's OUTPUT FEATURES are split over the 8 waves of a workgroup:
//   * wave w owns output tiles 2w, 2w+1 of every 256-wide layer (the 128-wide directional layers: waves 0-3; the two
//     one-tile heads: waves 4 and 5), for the S <= 4 sample tiles of a pass - all four matrix pipes of the CU work on the
//     same 16 S samples, a pass lasts 9856 S / 4 MFMA slots;
//   * the weights are NOT staged in LDS: nobody shares a wave's A tiles, so each wave streams its own tiles from L2 straight
//     into MFMA A operands (buffer_load_dwordx4 with a wave-uniform scalar offset: no vector address arithmetic), through a
//     register FIFO that runs one group of LAT_PF k-blocks ahead of the MFMAs across layer boundaries, barriers and passes;
//     the aux values of a layer (bias) come with it: two more small loads per group;
//   * a layer whose k-block count is not a multiple of LAT_PF is padded with steps whose B operand is a block of zeros in LDS
//     (they add +-0), a one-tile head runs as a two-tile layer whose second tile is discarded: the inner loop has no branch;
//   * the activations are exchanged through LDS once per layer: a wave writes its two output tiles in B-operand layout
//     ([k-block][lane] f4 - the accumulator layout of mlp_plan.h IS that layout), one barrier, every wave reads all k-blocks;
//   * the same packed streams (mlp_plan.h), the same k-block order and the same MFMA sequence per accumulator as the
//     throughput kernels: results are bit-identical, so a sample's value does not depend on the size of the call;
//   * the kernels are interpreters of a layer table (LatTable, built on the host from the Plan / BwdPlan): ONE site runs a
//     layer, whatever its width, inputs and epilogue - so the FIFO lives in one set of registers (with several sites the
//     compiler renames it between them and copies it, behind a vmcnt(0), at every join).
#pragma once
#include "mlp_train_device.h"

namespace snerf {

constexpr int LAT_NW = 8;             // waves per workgroup
constexpr int LAT_THREADS = LAT_NW * 64;
constexpr int LAT_PF = 4;             // steps per group = depth of the A-operand FIFO
constexpr int LAT_MAX_S = 4;          // sample tiles (of 16) per pass
constexpr int LAT_MAX_LAYERS = 44;    // MAX_BWD_LAYERS + margin
constexpr int SLAB_BYTES = SLAB_FLOATS * 4;
constexpr int SLAB_A_BYTES = SLAB_A_FLOATS * 4;
constexpr int LAT_ACT_BYTES = 16 * 1024;   // one activation buffer of one sample tile: 16 k-blocks x 64 lanes x 16 B

// epilogue of a layer (LatLayer::op)
enum LatOp : int {
    LAT_RELU = 1,         // forward: ReLU
    LAT_HEAD_SIGMA = 2,   // forward: row 0 of the tile is sigma -> LDS hand-over
    LAT_HEAD_RGB = 4,     // forward: rows 0..2 are rgb -> [rgb | sigma] store
    LAT_MASK_BITS = 8,    // dgrad: multiply by the forward's ReLU sign bits (mask word `mask_idx`)
    LAT_SCALE_AUX = 16,   // dgrad: the aux block is the sigma head's weight row, scaled by d sigma of the sample
    LAT_BARRIER = 32,     // a workgroup barrier follows the layer
    LAT_HALF_WORD = 64,   // forward: a 128-wide layer fills half of the mask word; the other half is zeroed
    LAT_PE_POS = 128,     // dgrad with input gradients: transposed position-encoding columns of layer 0 / a skip layer - the result is
                          // added to the wave's running d (position encoding) tiles, nothing is stored
    LAT_PE_DIR = 256,     // ... the direction-encoding columns of directional_input: encoder + normalisation backward -> d_dirs
};

// one layer of a packed stream as the kernels walk it (32-bit fields: scalar loads from the kernarg segment)
struct LatLayer {
    int soff;          // byte offset of the layer's first slab in the stream
    int nkb;           // k-blocks
    int t_out;         // output tiles: waves wave0 .. wave0 + ceil(t_out / 2) - 1 own tiles 2 (w - wave0), 2 (w - wave0) + 1
    int wave0;
    int kps_shift;     // log2(k-blocks per slab) = log2(SLAB_TILES / t_out)
    // B operands in LDS: k-blocks [0, n0) at b_base0 + s * b_stride0 + kb KiB, the others at b_base1 + s * b_stride1 + (kb - n0) KiB
    int b_base0, b_stride0, b_n0, b_base1, b_stride1;
    int out_base;      // LDS byte offset of the output buffer (+ s * LAT_ACT_BYTES + tile KiB), or -1
    int op;            // LatOp bits
    int store_row;     // first tile-row of the output in the activation (forward, TRAIN) / dY (dgrad) buffer, or -1
    int mask_idx;      // sign-mask word written (forward, TRAIN) / read (dgrad), or -1
    int pad_[2];       // 64 bytes: a kernel reads a whole descriptor with ONE scalar load (lat_layer_at)
};
static_assert(sizeof(LatLayer) == 64, "LatLayer is read as 16 dwords");
struct LatTable {
    int n;
    int stream_bytes;
    int nseq[LAT_NW];                          // layers wave w takes part in ...
    unsigned seq[LAT_NW][LAT_MAX_LAYERS / 4];  // ... their table indices in stream order, one byte each
    LatLayer l[LAT_MAX_LAYERS];
};
// The kernels index the table with run-time layer numbers.  Indexing the by-value kernel parameter would make the compiler copy
// it to scratch; instead the table is the FIRST kernel parameter and is read in place, through the kernarg segment pointer
// (constant address space: scalar loads with a run-time offset).
typedef const __attribute__((address_space(4))) LatTable *LatTabPtr;
__device__ __forceinline__ LatTabPtr lat_table_ptr() { return (LatTabPtr)__builtin_amdgcn_kernarg_segment_ptr(); }
// the whole descriptor of layer l in one s_load_dwordx16 (field by field, the compiler issues a chain of dependent scalar loads
// with a wait behind each: ~0.5 us at the start of every layer)
typedef int lat_i16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ LatLayer lat_layer_at(LatTabPtr tab, int l) {
    const lat_i16 v = *reinterpret_cast<const __attribute__((address_space(4))) lat_i16 *>(&tab->l[l]);
    return __builtin_bit_cast(LatLayer, v);
}
__device__ __forceinline__ int lat_seq_at(LatTabPtr tab, int wave, int k) {
    return (int)((tab->seq[wave][k >> 2] >> (8 * (k & 3))) & 0xffu);
}
inline bool lat_takes_part_host(const LatLayer &Ly, int w) { return w >= Ly.wave0 && w < Ly.wave0 + (Ly.t_out + 1) / 2; }
// host: fills nseq / seq from the layers' (wave0, t_out)
inline void lat_table_finish(LatTable &T) {
    for (int w = 0; w < LAT_NW; ++w) {
        T.nseq[w] = 0;
        for (int q = 0; q < LAT_MAX_LAYERS / 4; ++q) T.seq[w][q] = 0;
        for (int l = 0; l < T.n; ++l)
            if (lat_takes_part_host(T.l[l], w)) {
                const int k = T.nseq[w]++;
                T.seq[w][k >> 2] |= (unsigned)l << (8 * (k & 3));
            }
    }
}
// tiles [tile_off, tile_end) of 16 samples; workgroup b walks `passes` passes of S tiles: tile_off + (b * passes + p) * S + s
struct LatGeom {
    int64_t tile_off, tile_end;
    int passes;
};

// Position of a wave's A-operand prefetch in the stream: (k-th layer of the wave's sequence, k-block).  A layer is walked in
// GROUPS of LAT_PF steps (its k-blocks padded to a multiple of LAT_PF), so that step number mod LAT_PF - the FIFO slot - is a
// compile-time constant in the unrolled consumer loop, and so that the prefetch, which runs exactly one group ahead, never
// changes layers inside a group: the layer change (scalar loads of the next descriptor) is one rarely taken branch at the
// group's end.  Every group also re-reads the aux values (bias) of the layer the prefetch is in - two more small loads - so the
// consumer finds the aux values of a layer in registers when it enters it, one group later.  All of it is wave-uniform.
struct LatCursor {
    int k, kb, nkb, nkb_pad, base, aux, t_out, kps_shift;
    int so, kbl;       // byte offset of the next step's first A tile; its k-block index inside the slab
    int passes_left;   // passes the prefetch may still enter (the last pass does not wrap around)

    __device__ __forceinline__ void enter(LatTabPtr tab, int wave, int k_) {
        k = k_;
        kb = 0;
        kbl = 0;
        const int l = lat_seq_at(tab, wave, k_);
        nkb = tab->l[l].nkb;
        nkb_pad = (nkb + LAT_PF - 1) & ~(LAT_PF - 1);
        t_out = tab->l[l].t_out;
        kps_shift = tab->l[l].kps_shift;
        const int tile0 = 2 * (wave - tab->l[l].wave0);
        so = base = tab->l[l].soff + tile0 * 1024;
        aux = tab->l[l].soff + SLAB_A_BYTES + tile0 * 64;   // aux[16 * tile + 4 g + r]: 64 B per tile
    }
    __device__ __forceinline__ int aux_offset() const { return passes_left > 0 ? aux : base; }
    // byte offset of step j of the group, then on to the next step.  A padding step (or one behind the last pass) re-reads the
    // layer's first tile: EVERY step issues the same two loads, so that the wait counts of the consumer are compile-time constants.
    __device__ __forceinline__ int step_offset(int j) {
        const int out = (kb + j < nkb && passes_left > 0) ? so : base;
        const int kps_mask = (1 << kps_shift) - 1;
        const bool last = kbl == kps_mask;
        so += last ? SLAB_BYTES - kps_mask * t_out * 1024 : t_out * 1024;
        kbl = last ? 0 : kbl + 1;
        return out;
    }
    __device__ __forceinline__ void group_advance(LatTabPtr tab, int wave) {
        kb += LAT_PF;
        if (kb >= nkb_pad) {
            int nk = k + 1;
            if (nk == tab->nseq[wave]) {
                nk = 0;
                --passes_left;
            }
            enter(tab, wave, nk);
        }
    }
};

__device__ __forceinline__ f4 lat_load_a(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t lat_rsrc(const void *base, unsigned bytes) {
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)hi << 32) | lo), 0,
                                             (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// Stores through a buffer resource of 2 GiB: an offset of LAT_OOB is out of range and the hardware drops the store - the
// sample mask without a branch around a vector-memory instruction (a conditional store would make every wait count behind it
// unknowable at compile time, and the compiler would drain the A-operand FIFO to be safe).
constexpr unsigned LAT_STORE_RANGE = 0x80000000u, LAT_OOB = 0xfffffff0u;
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lat_store_f4(__amdgpu_buffer_rsrc_t rs, unsigned off, f4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), rs, (int)off, 0, 0);
}
__device__ __forceinline__ void lat_store_b32(__amdgpu_buffer_rsrc_t rs, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, (int)off, 0, 0);
}
__device__ __forceinline__ void lat_store_b8(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned v) {
    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)v, rs, (int)off, 0, 0);
}
// byte offset of a tile (16 features of one sample) in a tile-row-major buffer of n samples (mlp_plan.h TrainLayout)
__device__ __forceinline__ unsigned lat_tile_off(int row, unsigned n, unsigned sample, int g) {
    return (((unsigned)row * n + sample) * 16u + 4u * (unsigned)g) * 4u;
}
// byte offset of sign-mask word `idx` of a sample: tile-row mask_row + idx / 2, floats (idx & 1) * 8 + 2 g (mask_ptr, mlp_device.h)
__device__ __forceinline__ unsigned lat_mask_off(int mask_row, int idx, unsigned n, unsigned sample, int g) {
    return (((unsigned)(mask_row + (idx >> 1)) * n + sample) * 16u + (unsigned)((idx & 1) * 8 + 2 * g)) * 4u;
}

// The A-operand FIFO and the wave's place in the workgroup
struct LatWave {
    __amdgpu_buffer_rsrc_t rs;
    int voff;       // lane * 16
    int gvoff;      // (lane >> 4) * 16: lane offset into an aux block
    int wave, lane;
    LatCursor cur;
    f4 ff[LAT_PF][2];
    f4 aux[2];      // aux values (bias) of this lane's rows of the wave's two tiles, for the layer the prefetch is in

    // the two loads of one step into its slot
    __device__ __forceinline__ void fetch(int so, f4 (&slot)[2]) {
        slot[0] = lat_load_a(rs, voff, so);
        slot[1] = lat_load_a(rs, voff, so + 1024);   // (a one-tile layer: the KiB behind its tile, inside the slab; unused)
    }
    __device__ __forceinline__ void fetch_aux() {
        const int so = cur.aux_offset();
        aux[0] = lat_load_a(rs, gvoff, so);
        aux[1] = lat_load_a(rs, gvoff, so + 64);     // (a one-tile layer: the next 16 floats of the aux block; unused)
    }
    __device__ __forceinline__ void start(LatTabPtr tab, const float *packed, int tid, int passes) {
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        voff = lane * 16;
        gvoff = (lane >> 4) * 16;
        rs = lat_rsrc(packed, (unsigned)tab->stream_bytes);
        cur.passes_left = passes;
        cur.enter(tab, wave, 0);
        fetch_aux();
#pragma unroll
        for (int j = 0; j < LAT_PF; ++j) fetch(cur.step_offset(j), ff[j]);
        cur.group_advance(tab, wave);
    }
};

// (r05, measured and withdrawn: the FIFO's loads and their `s_waitcnt vmcnt(8)` written by hand as inline asm - the compiler's own
// wait-count insertion makes the first step of every group wait for all but three of the ten loads in flight, i.e. for loads issued
// one step earlier.  The compiler does not know the asm outputs are still in flight: under register pressure it parks FIFO registers
// in other registers around address arithmetic (v_mov out, v_mov back) and the late-landing load is overwritten by the stale copy -
// wrong gradients and memory faults in two-trainer runs.  The loads stay builtins.)
// two MFMAs on two accumulators, in place (tied operands: the builtin form lets the compiler accumulate out of place and copy back)
__device__ __forceinline__ void lat_mfma2(f4 &c0, f4 &c1, float a0, float a1, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %4, %0\n\tv_mfma_f32_16x16x4_f32 %1, %3, %4, %1" : "+v"(c0), "+v"(c1) : "v"(a0), "v"(a1), "v"(b));
}

// One layer for this wave: its two output tiles x S sample tiles, k-blocks [0, nkb) in whole groups of LAT_PF steps.  `acc` holds
// the initial values (the aux values the FIFO brought: W.aux).  The B operands of step kb + 1 are read from LDS before the MFMAs
// of step kb.  The loop body is branch-free: padding steps (k-blocks >= nkb) multiply the zeros the host's layout puts behind the
// layer's last input region, a one-tile head computes a second tile nobody reads.  The loads stay behind the MFMAs that read their
// slot and nothing crosses the step boundary (a load hoisted above them would need a second set of registers per slot).
// Requires b_n0 % LAT_PF == 0 (a group's k-blocks lie in one input region).
template <int S>
__device__ __forceinline__ void lat_run_layer(LatWave &W, LatTabPtr tab, const char *lds, const LatLayer &Ly, f4 (&acc)[S][2]) {
    static_assert(LAT_PF % 2 == 0, "the B operands ping-pong by step parity");
    const int nkb = Ly.nkb;
    const int b_base0 = Ly.b_base0, b_stride0 = Ly.b_stride0, b_n0 = Ly.b_n0, b_base1 = Ly.b_base1, b_stride1 = Ly.b_stride1;
    // LDS address (this lane) and per-sample-tile stride of k-block kb's B operand
    auto b_addr = [&](int kb, int &stride) __attribute__((always_inline)) {
        const bool first = kb < b_n0;
        stride = first ? b_stride0 : b_stride1;
        return lds + (first ? b_base0 + kb * 1024 : b_base1 + (kb - b_n0) * 1024) + W.voff;
    };
    f4 b[2][S];
    {
        int st;
        const char *p = b_addr(0, st);
#pragma unroll
        for (int s = 0; s < S; ++s) b[0][s] = *reinterpret_cast<const f4 *>(p + s * st);
    }
#pragma clang loop unroll(disable)
    for (int kb0 = 0; kb0 < nkb; kb0 += LAT_PF) {
        W.fetch_aux();
        int st, st_next;
        const char *p = b_addr(kb0, st);
        // (behind the last group: this group's first block once more - the value is not used)
        const char *p_next = b_addr(kb0 + LAT_PF < nkb ? kb0 + LAT_PF : kb0, st_next);
#pragma unroll
        for (int j = 0; j < LAT_PF; ++j) {
#pragma unroll
            for (int s = 0; s < S; ++s)
                b[(j + 1) & 1][s] = j + 1 < LAT_PF ? *reinterpret_cast<const f4 *>(p + (j + 1) * 1024 + s * st)
                                                   : *reinterpret_cast<const f4 *>(p_next + s * st_next);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int s = 0; s < S; ++s) lat_mfma2(acc[s][0], acc[s][1], W.ff[j][0][r], W.ff[j][1][r], b[j & 1][s][r]);
            const int so = W.cur.step_offset(j);
            __builtin_amdgcn_sched_barrier(0);
            W.fetch(so, W.ff[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        W.cur.group_advance(tab, W.wave);
    }
}

}  // namespace snerf
