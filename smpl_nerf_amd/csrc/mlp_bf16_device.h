// Device machinery of the split-bf16 kernels (mlp_bf16.hip: forward / training forward, mlp_train_bf16.hip: dgrad):
// operand splitting, the LDS-DMA weight ring, the hand-laid k-block instruction stream and the layer runner.
// See mlp_bf16.hip for the scheme.
#pragma once
#include <type_traits>

#include "mlp_device.h"

namespace snerf {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__host__ __device__ constexpr int slab16_bytes(int ns) { return ns * 16384 + 1024; }

// mlp_bf16.hip: any kw = 32 Plan -> split-bf16 slab stream
int launch_pack_bf16(const Plan &P, int ns, const float *params_flat, void *packed, hipStream_t s, const char *what, int fmt = 0);

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// cross terms (A part, B part), smallest first.  FIRST[s] = the A part whose first use is term s (s < NS): the A
// loads of a tile are issued in that order, so term s can start once 2(s+1) loads of its pair have landed.
template <int NS> struct Terms;
template <> struct Terms<2> {
    static constexpr int N = 3;
    static constexpr int A[3] = {1, 0, 0};
    static constexpr int B[3] = {0, 1, 0};
    static constexpr int FIRST[2] = {1, 0};
};
template <> struct Terms<3> {
    static constexpr int N = 6;
    static constexpr int A[6] = {2, 0, 1, 1, 0, 0};
    static constexpr int B[6] = {0, 2, 1, 0, 1, 0};
    static constexpr int FIRST[3] = {2, 0, 1};
};

// A-operand loads are issued as inline asm: hipcc waits lgkmcnt(0) in front of every consumer of a ds_read
// inside the (rolled) hidden-layer loop, i.e. right behind the prefetch of the NEXT tile pair, which exposes
// a full LDS round trip every pair.  With the loads opaque to the compiler the counted waits below are the
// only ones.  Rules that keep this sound:
//   * a loaded register is consumed only through a wait asm whose "+v" operands make every consumer depend on
//     the s_waitcnt;
//   * a loaded register stays live (is named by a wait asm) until its load has returned, so the allocator
//     cannot hand it out early;
//   * LDS returns data in order, so lgkmcnt(n) with n = the loads issued after the wanted ones is exact; the
//     compiler's own waits (bias loads in init()) and any younger loads can only make a wait stricter;
//   * all A loads of a tile pair are issued in Terms::FIRST order, tile 0 before tile 1 per part.
template <int OFF>
__device__ __forceinline__ void lds_load_a(bf8 &dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
// the same load pinned between two MFMAs by data dependences: after the MFMA that produced `after`, before the
// next MFMA that accumulates into `before` - one load per MFMA gap instead of a clump of loads per tile pair
// (both waves of a SIMD reach a clump together, and the matrix pipe idles for its length)
template <int OFF>
__device__ __forceinline__ void lds_load_between(bf8 &dst, uint32_t addr, const f4 &after, f4 &before) {
    asm volatile("ds_read_b128 %0, %2 offset:%3" : "=v"(dst), "+v"(before) : "v"(addr), "n"(OFF), "v"(after) : "memory");
}
template <int CNT>
__device__ __forceinline__ void wait_two(bf8 &x, bf8 &y) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(CNT));
}
__device__ __forceinline__ uint32_t lds_addr(const char *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)p;
}
// tiles TO, TO+1 of a k-block, outside the MFMA stream (pipeline start-up).  `s_nop 7` covers the MFMA-SrcC ->
// LDS-write WAR distance the compiler would insert for a ds_read it knows.
template <int NS, int TO>
__device__ __forceinline__ void issue_pair(uint32_t addr, bf8 (&x0)[NS], bf8 (&x1)[NS]) {
    asm volatile("s_nop 7");
    static_for<0, NS>([&](auto i) __attribute__((always_inline)) {
        constexpr int s = Terms<NS>::FIRST[decltype(i)::value];
        lds_load_a<(TO * NS + s) * 1024>(x0[s], addr);
        lds_load_a<((TO + 1) * NS + s) * 1024>(x1[s], addr);
    });
}
template <int NS, int LEFT>
__device__ __forceinline__ void wait_pair(bf8 (&x0)[NS], bf8 (&x1)[NS]) {
    static_assert(LEFT == 0 || LEFT == 4 || LEFT == 6, "lgkmcnt immediates used below");
    if constexpr (NS == 3) {
        if constexpr (LEFT == 6)
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(x0[0]), "+v"(x0[1]), "+v"(x0[2]), "+v"(x1[0]), "+v"(x1[1]), "+v"(x1[2]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0[0]), "+v"(x0[1]), "+v"(x0[2]), "+v"(x1[0]), "+v"(x1[1]), "+v"(x1[2]));
    } else {
        if constexpr (LEFT == 4)
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(x0[0]), "+v"(x0[1]), "+v"(x1[0]), "+v"(x1[1]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0[0]), "+v"(x0[1]), "+v"(x1[0]), "+v"(x1[1]));
    }
}

// ------------------------------------------------------------------------------------------------
// slab pipe with a run-time slab size (dynamic LDS)
// ------------------------------------------------------------------------------------------------
// The slab stream goes L2 -> LDS by DMA (global_load_lds, 16 B per lane = 1 KiB per wave instruction), no VGPR
// round trip: 3-slot ring, slab p is consumed while p+1 has landed and p+2 is in flight - two slab periods of
// latency tolerance (the register-staged pipe of the fp32 kernel has one).
// Issuing a piece costs a wave ~60 issue cycles, and all waves reach the hand-over of a slab together: if every
// wave issued its share there, both waves of each SIMD would be busy with DMA at the same moment and the matrix
// pipe would idle for the length of the clump.  So the two halves of the workgroup (waves 0..3 / 4..7 - wave w
// and w+4 share a SIMD) take turns: slab s is issued entirely by half s & 1, while the other half goes straight
// on with its MFMAs.  The half that issued a slab is also the only one that has to wait for it, two hand-overs
// later, and by then it has nothing younger in flight: `s_waitcnt vmcnt(0)`, then one raw s_barrier publishes
// the slab to the workgroup.
template <int NT, int NS>
struct SlabPipe16 {
    static constexpr int SB = NS * 16384 + 1024;
    static constexpr int NW = NT / 64;
    static constexpr int HALF = NW / 2;
    static constexpr int PER_WAVE = NS * 16 / HALF;  // 1 KiB pieces of the A region per issuing wave
    static_assert(NS * 16 % HALF == 0, "A region must split evenly over the issuing waves");
    const char *gsrc;  // packed + lane*16
    char *ring;
    int wave, rd, next;  // next = running index of the slab to issue (its parity picks the issuing half)
    int src_slab, total;  // stream position of that slab: persistent kernels wrap around after `total` slabs
    // A parts of the first tile pair of the k-block that runs next: issued one tile pair ahead like every other
    // pair, i.e. during the last pair of the previous k-block - across slab and layer boundaries too
    bf8 fa0[NS], fa1[NS];
    uint32_t lane16;
    __device__ __forceinline__ void prefetch_first(const char *at) { issue_pair<NS, 0>(lds_addr(at) + lane16, fa0, fa1); }

    __device__ __forceinline__ bool my_turn() const { return (wave >= HALF) == ((next & 1) != 0); }
    // slab `next` -> ring slot `slot`, by the half whose turn it is
    __device__ __forceinline__ void issue(int slot) {
        if (my_turn()) {
            const char *src = gsrc + (int64_t)src_slab * SB;
            char *dst = ring + slot * SB;
            const int w = wave >= HALF ? wave - HALF : wave;
#pragma unroll
            for (int i = 0; i < PER_WAVE; ++i) {
                const int piece = w * PER_WAVE + i;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + piece * 1024),
                                                 (__attribute__((address_space(3))) void *)(dst + piece * 1024), 16, 0, 0);
            }
            if (w == 0)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + NS * 16384),
                                                 (__attribute__((address_space(3))) void *)(dst + NS * 16384), 16, 0, 0);
        }
        ++next;
        src_slab = src_slab + 1 == total ? 0 : src_slab + 1;
    }
    // total_slabs: length of the stream for kernels that run it repeatedly (one pass per sample tile); a one-pass
    // kernel leaves the default and reads on into the zero padding slabs
    __device__ __forceinline__ void prologue(const void *packed, char *ring_, int tid, int total_slabs = 0x7fffffff) {
        ring = ring_;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        gsrc = reinterpret_cast<const char *>(packed) + (tid & 63) * 16;
        next = 0;
        src_slab = 0;
        total = total_slabs;
        issue(0);
        issue(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // slabs 0, 1 (and the per-sample input loads)
        __builtin_amdgcn_s_barrier();
        rd = 0;
        issue(2);
        lane16 = (tid & 63) * 16;
        prefetch_first(ring);
    }
    __device__ __forceinline__ const char *acquire() const { return ring + rd * SB; }
    // done reading slab `rd`: make the next slab visible, then refill the slot just freed.  The slab published
    // here is slab next-2, issued by the same half that now issues slab `next`.
    __device__ __forceinline__ void release() {
        if (my_turn()) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int freed = rd;
        rd = rd == 2 ? 0 : rd + 1;
        issue(freed);
    }
};

// One 32-wide k-block: per output tile NS ds_read_b128 + Terms<NS>::N MFMAs; tiles in pairs so consecutive
// MFMAs alternate accumulators.  The instruction stream is laid out by hand:
//   * while pair p's MFMAs run, the 2*NS A parts of pair p+1 are issued one per MFMA gap (behind the first 2*NS
//     MFMAs), in the order pair p+1 will first use them; term s of pair p+1 waits lgkmcnt(2*NS-2), i.e. only for
//     its own two parts (the later parts of its pair and the loads already issued for pair p+2 stay in flight);
//   * the pipeline runs across k-blocks, slabs and layers: on entry (fa0, fa1) hold the in-flight first pair of
//     this k-block; in the last pair, once this k-block's loads have all returned, `boundary()` does the slab
//     hand-over if one is due (counted vmcnt wait, barrier, refill of the freed slot) and returns the LDS
//     address of the next k-block, whose first pair then streams in behind the last pair's MFMAs - neither the
//     barrier nor the first LDS round trip of a slab sits in front of an empty matrix pipe;
//   * `make_piece.make(i)`, i in [0, 4), is the VALU work that prepares elements 2i, 2i+1 of the NEXT k-block's
//     B operand (splitting fp32 accumulators into bf16 parts); the pieces are spread over the pairs and float
//     between that pair's MFMAs instead of forming a serial phase between k-blocks.
// `b` holds this k-block's B operand parts (ready on entry).
// operand formats of the split kernels: FMT_BF16 - bf16 parts on v_mfma_f32_16x16x32_bf16; FMT_F16 - two fp16 parts of
// operands pre-scaled by powers of two (per layer for the weights, per sample for the activations) on
// v_mfma_f32_16x16x32_f16 (mlp_bf16.hip, "f16x3").  Both take eight 16-bit values per lane: `bf8` is the container.
enum { FMT_BF16 = 0, FMT_F16 = 1 };
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
template <int FMT>
__device__ __forceinline__ f4 mfma16(const bf8 &a, const bf8 &b, const f4 &c) {
    if constexpr (FMT == FMT_F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int T_OUT, int NS, int FMT, class MakePiece, class Boundary>
__device__ __forceinline__ void kblock16(uint32_t addr, bf8 (&fa0)[NS], bf8 (&fa1)[NS], const bf8 (&b)[NS],
                                         MakePiece make_piece, f4 (&acc)[T_OUT], Boundary boundary) {
    using Tm = Terms<NS>;
    if constexpr (T_OUT == 1) {
        bf8 a[NS];
        wait_pair<NS, 0>(fa0, fa1);
#pragma unroll
        for (int s = 0; s < NS; ++s) a[s] = fa0[s];
        const uint32_t next = boundary();
        issue_pair<NS, 0>(next, fa0, fa1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) make_piece.make(i);
#pragma unroll
        for (int t = 0; t < Tm::N; ++t) acc[0] = mfma16<FMT>(a[Tm::A[t]], b[Tm::B[t]], acc[0]);
        make_piece.touch();
        __builtin_amdgcn_sched_barrier(0);
    } else {
        constexpr int PAIRS = T_OUT / 2;
        bf8 a0[NS], a1[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            a0[s] = fa0[s];
            a1[s] = fa1[s];
        }
        static_for<0, PAIRS>([&](auto pc) __attribute__((always_inline)) {
            constexpr int p = decltype(pc)::value, to = 2 * p;
            constexpr bool LAST = p + 1 == PAIRS;
            constexpr int NEXT_TILE = LAST ? 0 : to + 2;  // of the next k-block when LAST
            bf8 n0[NS], n1[NS];
            uint32_t src = addr;
            if constexpr (LAST) {
                wait_pair<NS, 0>(a0, a1);
                src = boundary();
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr bool has_piece = (p * 4) % PAIRS == 0 || PAIRS < 4;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i * PAIRS / 4 == p) make_piece.make(i);
            static_for<0, Tm::N>([&](auto tc) __attribute__((always_inline)) {
                constexpr int t = decltype(tc)::value;
                if constexpr (t < NS) {
                    constexpr int s = Tm::FIRST[t];
                    wait_two<2 * NS - 2>(a0[s], a1[s]);
                    acc[to] = mfma16<FMT>(a0[Tm::A[t]], b[Tm::B[t]], acc[to]);
                    lds_load_between<(NEXT_TILE * NS + s) * 1024>(n0[s], src, acc[to], acc[to + 1]);
                    acc[to + 1] = mfma16<FMT>(a1[Tm::A[t]], b[Tm::B[t]], acc[to + 1]);
                    lds_load_between<((NEXT_TILE + 1) * NS + s) * 1024>(n1[s], src, acc[to + 1], acc[to]);
                } else {
                    acc[to] = mfma16<FMT>(a0[Tm::A[t]], b[Tm::B[t]], acc[to]);
                    acc[to + 1] = mfma16<FMT>(a1[Tm::A[t]], b[Tm::B[t]], acc[to + 1]);
                }
            });
            if constexpr (has_piece) make_piece.touch();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                a0[s] = n0[s];
                a1[s] = n1[s];
            }
        });
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            fa0[s] = a0[s];
            fa1[s] = a1[s];
        }
    }
}

typedef float f2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

// (v0, v1) -> dword `i` (elements 2i, 2i+1) of the NS packed-bf16 parts.  Per part: one v_cvt_pk_bf16_f32 for both
// values, a shift and a mask to widen the two halves back to fp32, and two scalar subtractions (kept scalar on
// purpose: packed-fp32 VALU beside MFMAs costs matrix-pipe issue slots - MI355X_MICROARCH.md).
typedef __fp16 h2v __attribute__((ext_vector_type(2)));
template <int NS, int FMT = FMT_BF16>
__device__ __forceinline__ void split_pair_into(float v0, float v1, bf8 (&dst)[NS], int i) {
    if constexpr (FMT == FMT_F16) {
        // two fp16 parts, rounded toward zero (v_cvt_pkrtz_f16_f32: the one packed conversion there is); the caller
        // has scaled the values into fp16's range
        static_assert(NS == 2, "f16x3 = two parts");
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const h2v h = __builtin_amdgcn_cvt_pkrtz(v0, v1);
            u4v t = __builtin_bit_cast(u4v, dst[s]);
            t[i] = __builtin_bit_cast(uint32_t, h);
            dst[s] = __builtin_bit_cast(bf8, t);
            if (s == 0) {
                v0 = v0 - (float)h[0];
                v1 = v1 - (float)h[1];
            }
        }
        return;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t u = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2v{v0, v1}, bf2v));
        u4v t = __builtin_bit_cast(u4v, dst[s]);
        t[i] = u;
        dst[s] = __builtin_bit_cast(bf8, t);
        if (s + 1 < NS) {
            v0 = v0 - __builtin_bit_cast(float, u << 16);
            v1 = v1 - __builtin_bit_cast(float, u & 0xffff0000u);
        }
    }
}
// relu as exactly one v_max_f32 (fmaxf adds a canonicalising v_max in front)
__device__ __forceinline__ float relu1(float v) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
    return r;
}
// `delta` (FMT_F16 only): power-of-two rescaling of the source accumulators into this layer's operand scale
template <bool RELU, bool PIN, int T_SRC, int NS, int FMT = FMT_BF16>
__device__ __forceinline__ void split_piece(const f4 (&src)[T_SRC], int kb, int i, bf8 (&b)[NS], int delta = 0) {
    float v0 = src[2 * kb + (i >> 1)][(2 * i) & 3], v1 = src[2 * kb + (i >> 1)][((2 * i) & 3) + 1];
    if constexpr (PIN) asm volatile("" : "+v"(v0), "+v"(v1));
    if constexpr (RELU) {
        v0 = relu1(v0);
        v1 = relu1(v1);
    }
    if constexpr (FMT == FMT_F16) {
        v0 = __builtin_ldexpf(v0, delta);
        v1 = __builtin_ldexpf(v1, delta);
    }
    split_pair_into<NS, FMT>(v0, v1, b, i);
}
struct NoPiece {
    __device__ __forceinline__ void make(int) const {}
    __device__ __forceinline__ void touch() const {}
};
template <bool RELU, int T_SRC, int NS, int FMT = FMT_BF16>
struct NextPieceT {
    const f4 (&src)[T_SRC];
    bf8 (&bn)[NS];
    int kb;  // the k-block being prepared (nothing to do past the last one)
    int delta = 0;
    __device__ __forceinline__ void make(int i) const {
        if (kb < T_SRC / 2) split_piece<RELU, true, T_SRC, NS, FMT>(src, kb, i, bn, delta);
    }
    // keeps the piece's results inside the tile-pair region they were issued in
    __device__ __forceinline__ void touch() const {
        if (kb < T_SRC / 2) {
            if constexpr (NS == 3) asm volatile("" ::"v"(bn[0]), "v"(bn[1]), "v"(bn[2]));
            else asm volatile("" ::"v"(bn[0]), "v"(bn[1]));
        }
    }
};

template <int T_OUT, int NT, int NS, int FMT = FMT_BF16>
struct LayerRun16 {
    static constexpr int KPS = 16 / T_OUT;
    SlabPipe16<NT, NS> &pipe;
    const char *slab;
    int kbl, lane;
    __device__ __forceinline__ LayerRun16(SlabPipe16<NT, NS> &p, int lane_) : pipe(p), slab(p.acquire()), kbl(0), lane(lane_) {}
    // as stored (bf16 modes; f16x3 dgrad: zeros, or fp32 values the caller scales itself)
    __device__ __forceinline__ void init_plain(f4 (&acc)[T_OUT]) {
        const f4 *aux = reinterpret_cast<const f4 *>(slab + NS * 16384) + (lane >> 4);
#pragma unroll
        for (int to = 0; to < T_OUT; ++to) acc[to] = aux[to * 4];
    }
    // bias_exp (FMT_F16): the accumulators of the layer carry the scale 2^bias_exp (weights x operands), so does the bias
    __device__ __forceinline__ void init(f4 (&acc)[T_OUT], int bias_exp = 0) {
        const f4 *aux = reinterpret_cast<const f4 *>(slab + NS * 16384) + (lane >> 4);
#pragma unroll
        for (int to = 0; to < T_OUT; ++to) {
            acc[to] = aux[to * 4];
            if constexpr (FMT == FMT_F16) {   // x 2^bias_exp as a multiplication (|bias_exp| stays far below 126)
                const float sc = __int_as_float((min(max(bias_exp, -126), 127) + 127) << 23);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[to][r] *= sc;
            }
        }
    }
    static constexpr int KB_BYTES = T_OUT * NS * 1024;
    template <class MakePiece>
    __device__ __forceinline__ void step(const bf8 (&b)[NS], MakePiece make_piece, f4 (&acc)[T_OUT]) {
        kblock16<T_OUT, NS, FMT>(lds_addr(slab + kbl * KB_BYTES) + pipe.lane16, pipe.fa0, pipe.fa1, b, make_piece, acc,
                            [&]() __attribute__((always_inline)) -> uint32_t {
                                if (++kbl == KPS) {  // this k-block was the last of its slab
                                    pipe.release();
                                    slab = pipe.acquire();
                                    kbl = 0;
                                }
                                return lds_addr(slab + kbl * KB_BYTES) + pipe.lane16;
                            });
    }
    // operand computed up front (encoder / additional-input k-blocks)
    template <class MakeB>
    __device__ __forceinline__ void step_make(MakeB make_b, f4 (&acc)[T_OUT]) {
        bf8 b[NS];
        make_b(b);
        step(b, NoPiece{}, acc);
    }
    // k-blocks fed by the accumulators of a previous layer: tiles 2kb, 2kb+1 are split just in time
    template <bool RELU, int T_SRC>
    __device__ __forceinline__ void run_hidden(const f4 (&src)[T_SRC], f4 (&acc)[T_OUT], int delta = 0) {
        bf8 bc[NS], bn[NS];
#pragma unroll
        for (int i = 0; i < 4; ++i) split_piece<RELU, false, T_SRC, NS, FMT>(src, 0, i, bc, delta);  // the only split of the layer not hidden behind MFMAs
#pragma unroll
        for (int kb = 0; kb < T_SRC / 2; ++kb) {
            step(bc, NextPieceT<RELU, T_SRC, NS, FMT>{src, bn, kb + 1, delta}, acc);
#pragma unroll
            for (int s = 0; s < NS; ++s) bc[s] = bn[s];
        }
    }
    // every layer starts on a fresh slab: hand over a partly used last slab and restart the first-pair prefetch
    // (the one issued by the last k-block pointed into the unused part of the old slab)
    __device__ __forceinline__ void finish() {
        if (kbl != 0) {
            wait_pair<NS, 0>(pipe.fa0, pipe.fa1);  // its registers stay live until the loads have landed
            pipe.release();
            pipe.prefetch_first(pipe.acquire());
        }
    }
};

// B operand of encoder k-block kb: 4 units (sin, cos pairs) per lane
// `half[h]` = the fp32 values of 16-wide k-block 2kb + h in the fp32 kernel's layout (mlp_device.h pe_operand): what
// the training forward stores for the backward kernels
// ---- f16x3 scaling helpers (shared by the forward and the dgrad kernel) --------------------------------------------
// Unbiased exponent of the largest |value| (after the ReLU, if any) that this lane's sample has in `src`.  On the bit
// patterns: as signed integers the largest positive float wins (ReLU: negatives lose against 0); as unsigned ones a
// negative float, if there is one, wins with the largest magnitude (v_max3_i32 / v_max3_u32: no canonicalising extra
// instruction as with fmaxf).  The four lanes of a sample are 16 and 32 lanes apart: two row swaps.
template <int N>
__device__ __forceinline__ int sample_exp16(const f4 (&src)[N], bool relu) {
    int mp = 0;
    unsigned mu = 0u;
#pragma unroll
    for (int t = 0; t < N; ++t)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            const int b0 = __float_as_int(src[t][r]), b1 = __float_as_int(src[t][r + 1]);
            mp = max(mp, max(b0, b1));
            if (!relu) mu = max(mu, max((unsigned)b0, (unsigned)b1));
        }
    unsigned m = (unsigned)mp;
    if (!relu) m = max(m, mu & 0x7fffffffu);   // (mu is a positive float <= mp when no value is negative)
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    u2v w = __builtin_amdgcn_permlane16_swap(m, m, false, false);
    m = max(w[0], w[1]);
    w = __builtin_amdgcn_permlane32_swap(m, m, false, false);
    m = max(w[0], w[1]);
    return (int)((m >> 23) & 0xffu) - 127;
}
// raise stat[idx] to the largest exponent in this wave (one atomic per wave, and only when it would change the entry)
__device__ __forceinline__ void stat_max16(int *stat, int idx, int e) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) e = max(e, __shfl_xor(e, d, 64));
    if ((threadIdx.x & 63) == 0 && e > stat[idx]) atomicMax(stat + idx, e);
}
// operand scale exponent of a layer whose input has exponent e_src at accumulator scale es: largest input to [2^14, 2^15)
__device__ __forceinline__ int operand_scale16(int e_src, int es, int cap) { return min(14 - (e_src - es), cap); }
// the table of weight exponents (f16x3) sits in pad slab `table_slab` of a packed stream
int launch_wexp(const Plan &P, int ns, const float *params_flat, void *packed, int table_slab, hipStream_t s, const char *what);

template <int NS, int FMT = FMT_BF16>
__device__ __forceinline__ void pe_operand16(const SampleCtx &c, bool is_dir, int L, int ident, int kb, bf8 (&b)[NS],
                                             f4 (&half)[2], int scale_exp = 0) {
    const float x = is_dir ? c.dx : c.px, y = is_dir ? c.dy : c.py, z = is_dir ? c.dz : c.pz;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float s0, c0;
        pe_unit(x, y, z, L, ident, 4 * (4 * kb + u) + c.g, s0, c0);
        half[u >> 1][2 * (u & 1)] = s0;
        half[u >> 1][2 * (u & 1) + 1] = c0;
        if constexpr (FMT == FMT_F16) {
            s0 = __builtin_ldexpf(s0, scale_exp);
            c0 = __builtin_ldexpf(c0, scale_exp);
        }
        split_pair_into<NS, FMT>(s0, c0, b, u);
    }
}
template <int NS, int FMT = FMT_BF16>
__device__ __forceinline__ void add_operand16(const SampleCtx &c, int add_dim, int kb, bf8 (&b)[NS], f4 (&half)[2],
                                              int scale_exp = 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = 2 * j + h, col = 32 * kb + 16 * (e >> 2) + 4 * c.g + (e & 3);
            v[h] = col < add_dim ? c.add[col] : 0.f;
            half[e >> 2][e & 3] = v[h];
            if constexpr (FMT == FMT_F16) v[h] = __builtin_ldexpf(v[h], scale_exp);
        }
        split_pair_into<NS, FMT>(v[0], v[1], b, j);
    }
}
}  // namespace snerf
