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

#include <type_traits>

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
    int wave, rd, next;  // next = slab index to issue
    // A parts of the first tile pair of the k-block that runs next: issued one tile pair ahead like every other
    // pair, i.e. during the last pair of the previous k-block - across slab and layer boundaries too
    bf8 fa0[NS], fa1[NS];
    uint32_t lane16;
    __device__ __forceinline__ void prefetch_first(const char *at) { issue_pair<NS, 0>(lds_addr(at) + lane16, fa0, fa1); }

    __device__ __forceinline__ bool my_turn() const { return (wave >= HALF) == ((next & 1) != 0); }
    // slab `next` -> ring slot `slot`, by the half whose turn it is
    __device__ __forceinline__ void issue(int slot) {
        if (my_turn()) {
            const char *src = gsrc + (int64_t)next * SB;
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
    }
    __device__ __forceinline__ void prologue(const void *packed, char *ring_, int tid) {
        ring = ring_;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        gsrc = reinterpret_cast<const char *>(packed) + (tid & 63) * 16;
        next = 0;
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
template <int T_OUT, int NS, class MakePiece, class Boundary>
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
        for (int t = 0; t < Tm::N; ++t) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[Tm::A[t]], b[Tm::B[t]], acc[0], 0, 0, 0);
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
                    acc[to] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[Tm::A[t]], b[Tm::B[t]], acc[to], 0, 0, 0);
                    lds_load_between<(NEXT_TILE * NS + s) * 1024>(n0[s], src, acc[to], acc[to + 1]);
                    acc[to + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[Tm::A[t]], b[Tm::B[t]], acc[to + 1], 0, 0, 0);
                    lds_load_between<((NEXT_TILE + 1) * NS + s) * 1024>(n1[s], src, acc[to + 1], acc[to]);
                } else {
                    acc[to] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[Tm::A[t]], b[Tm::B[t]], acc[to], 0, 0, 0);
                    acc[to + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[Tm::A[t]], b[Tm::B[t]], acc[to + 1], 0, 0, 0);
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
template <int NS>
__device__ __forceinline__ void split_pair_into(float v0, float v1, bf8 (&dst)[NS], int i) {
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
template <bool RELU, bool PIN, int T_SRC, int NS>
__device__ __forceinline__ void split_piece(const f4 (&src)[T_SRC], int kb, int i, bf8 (&b)[NS]) {
    float v0 = src[2 * kb + (i >> 1)][(2 * i) & 3], v1 = src[2 * kb + (i >> 1)][((2 * i) & 3) + 1];
    if constexpr (PIN) asm volatile("" : "+v"(v0), "+v"(v1));
    if constexpr (RELU) {
        v0 = relu1(v0);
        v1 = relu1(v1);
    }
    split_pair_into<NS>(v0, v1, b, i);
}
struct NoPiece {
    __device__ __forceinline__ void make(int) const {}
    __device__ __forceinline__ void touch() const {}
};
template <bool RELU, int T_SRC, int NS>
struct NextPieceT {
    const f4 (&src)[T_SRC];
    bf8 (&bn)[NS];
    int kb;  // the k-block being prepared (nothing to do past the last one)
    __device__ __forceinline__ void make(int i) const {
        if (kb < T_SRC / 2) split_piece<RELU, true>(src, kb, i, bn);
    }
    // keeps the piece's results inside the tile-pair region they were issued in
    __device__ __forceinline__ void touch() const {
        if (kb < T_SRC / 2) {
            if constexpr (NS == 3) asm volatile("" ::"v"(bn[0]), "v"(bn[1]), "v"(bn[2]));
            else asm volatile("" ::"v"(bn[0]), "v"(bn[1]));
        }
    }
};

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
    static constexpr int KB_BYTES = T_OUT * NS * 1024;
    template <class MakePiece>
    __device__ __forceinline__ void step(const bf8 (&b)[NS], MakePiece make_piece, f4 (&acc)[T_OUT]) {
        kblock16<T_OUT, NS>(lds_addr(slab + kbl * KB_BYTES) + pipe.lane16, pipe.fa0, pipe.fa1, b, make_piece, acc,
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
    __device__ __forceinline__ void run_hidden(const f4 (&src)[T_SRC], f4 (&acc)[T_OUT]) {
        bf8 bc[NS], bn[NS];
#pragma unroll
        for (int i = 0; i < 4; ++i) split_piece<RELU, false>(src, 0, i, bc);  // the only split of the layer not hidden behind MFMAs
#pragma unroll
        for (int kb = 0; kb < T_SRC / 2; ++kb) {
            step(bc, NextPieceT<RELU, T_SRC, NS>{src, bn, kb + 1}, acc);
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
template <int NS>
__device__ __forceinline__ void pe_operand16(const SampleCtx &c, bool is_dir, int L, int ident, int kb, bf8 (&b)[NS],
                                             f4 (&half)[2]) {
    const float x = is_dir ? c.dx : c.px, y = is_dir ? c.dy : c.py, z = is_dir ? c.dz : c.pz;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float s0, c0;
        pe_unit(x, y, z, L, ident, 4 * (4 * kb + u) + c.g, s0, c0);
        half[u >> 1][2 * (u & 1)] = s0;
        half[u >> 1][2 * (u & 1) + 1] = c0;
        split_pair_into<NS>(s0, c0, b, u);
    }
}
template <int NS>
__device__ __forceinline__ void add_operand16(const SampleCtx &c, int add_dim, int kb, bf8 (&b)[NS], f4 (&half)[2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = 2 * j + h, col = 32 * kb + 16 * (e >> 2) + 4 * c.g + (e & 3);
            v[h] = col < add_dim ? c.add[col] : 0.f;
            half[e >> 2][e & 3] = v[h];
        }
        split_pair_into<NS>(v[0], v[1], b, j);
    }
}
// post-activation tiles -> the tile-row-major activation buffer
template <bool RELU, int N>
__device__ __forceinline__ void store_act(float *buf, int row0, int64_t n, int64_t sample, int g, const f4 (&t)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        f4 v = t[i];
        if (RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        store_tile(buf, row0 + i, n, sample, g, v);
    }
}

// TRAIN additionally stores every layer input (post-activation, fp32, the fp32 kernel's layout) for the backward
// kernels of mlp_train.hip.
template <int WIDTH, int NWAVES, int NS, bool TRAIN>
__global__ __launch_bounds__(NWAVES * 64) void mlp_fwd_bf16_kernel(FwdArgs A) {
    constexpr int NT = NWAVES * 64;
    constexpr int T = WIDTH / 16, TD = WIDTH / 32;
    extern __shared__ __attribute__((aligned(16))) char ring[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t sample = ((int64_t)blockIdx.x * NWAVES + wave) * 16 + (lane & 15);
    const bool valid = sample < A.n;
    const int64_t sc = valid ? sample : A.n - 1;
    SampleCtx c;
    c.g = lane >> 4;
    c.enc = nullptr;
    c.add = nullptr;
    // read-once / write-once streams bypass L2 retention: the 2.4 - 3.6 MB weight stream that every workgroup
    // re-reads is what each XCD's 4 MB L2 should keep
    c.px = __builtin_nontemporal_load(A.x + sc * 3 + 0);
    c.py = __builtin_nontemporal_load(A.x + sc * 3 + 1);
    c.pz = __builtin_nontemporal_load(A.x + sc * 3 + 2);
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
    // make the compiler retire the input loads HERE: a later first use would put its s_waitcnt vmcnt(0) inside
    // the slab loop and drain the in-flight weight DMA every time
    asm volatile("" ::"v"(c.px), "v"(c.py), "v"(c.pz), "v"(c.dx), "v"(c.dy), "v"(c.dz));

    SlabPipe16<NT, NS> pipe;
    pipe.prologue(A.packed, ring, tid);

    // two accumulator sets ping-pong between consecutive layers: the finished set feeds the next layer's B
    // operands (split just in time, k-block by k-block) while the other set accumulates
    f4 accA[T], accB[T];
    auto pos_segments = [&](LayerRun16<T, NT, NS> &run, f4(&acc)[T], bool first) __attribute__((always_inline)) {
        auto pe_segment = [&]() __attribute__((always_inline)) {
            for (int kb = 0; kb < A.pos_nkb; ++kb)
                run.step_make([&](bf8(&b)[NS]) __attribute__((always_inline)) {
                    f4 half[2];
                    pe_operand16<NS>(c, false, A.pos_L, A.pos_id, kb, b, half);
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
                    add_operand16<NS>(c, A.add_dim, kb, b, half);
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
    auto hidden = [&](int i, const f4(&src)[T], f4(&dst)[T]) __attribute__((always_inline)) {
        LayerRun16<T, NT, NS> run(pipe, lane);
        run.init(dst);
        run.template run_hidden<true>(src, dst);
        if ((A.skip_mask >> i) & 1u) pos_segments(run, dst, false);
        run.finish();
        if (TRAIN && valid) store_act<true>(A.act, A.act_x1 + (i + 1) * T, A.n, sample, c.g, dst);
    };
    {  // positions_pose_input (its relu is applied when the next layer splits accA)
        LayerRun16<T, NT, NS> run(pipe, lane);
        run.init(accA);
        pos_segments(run, accA, true);
        run.finish();
        if (TRAIN && valid) store_act<true>(A.act, A.act_x1, A.n, sample, c.g, accA);
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
    {  // additional_linear_layer: relu(accA) -> accB (no activation on its output)
        LayerRun16<T, NT, NS> run(pipe, lane);
        run.init(accB);
        run.template run_hidden<true>(accA, accB);
        run.finish();
        if (TRAIN && valid) store_act<false>(A.act, A.act_o, A.n, sample, c.g, accB);
    }
    f4 sig[1];
    {
        LayerRun16<1, NT, NS> run(pipe, lane);
        run.init(sig);
        run.template run_hidden<false>(accB, sig);
        run.finish();
    }
    f4 accd[TD], acce[TD];
    {  // directional_input (no activation)
        LayerRun16<TD, NT, NS> run(pipe, lane);
        run.init(accd);
        run.template run_hidden<false>(accB, accd);
        for (int kb = 0; kb < A.dir_nkb; ++kb)
            run.step_make([&](bf8(&b)[NS]) __attribute__((always_inline)) {
                f4 half[2];
                pe_operand16<NS>(c, true, A.dir_L, A.dir_id, kb, b, half);
                if (TRAIN && valid) {
                    store_tile(A.act, A.act_dpe + 2 * kb, A.n, sample, c.g, half[0]);
                    if (2 * kb + 1 < A.dir_nkb16) store_tile(A.act, A.act_dpe + 2 * kb + 1, A.n, sample, c.g, half[1]);
                }
            }, accd);
        run.finish();
        if (TRAIN && valid) store_act<false>(A.act, A.act_h1, A.n, sample, c.g, accd);
    }
    {  // directional_net[0] (its relu is applied when the rgb head splits acce)
        LayerRun16<TD, NT, NS> run(pipe, lane);
        run.init(acce);
        run.template run_hidden<false>(accd, acce);
        run.finish();
        if (TRAIN && valid) store_act<true>(A.act, A.act_h2, A.n, sample, c.g, acce);
    }
    f4 rgb[1];
    {
        LayerRun16<1, NT, NS> run(pipe, lane);
        run.init(rgb);
        run.template run_hidden<true>(acce, rgb);
        run.finish();
    }
    // the last k-block prefetched past the end of the stream (padding slabs): retire those loads before their
    // registers can be reused
    wait_pair<NS, 0>(pipe.fa0, pipe.fa1);
    if (valid && c.g == 0)
        __builtin_nontemporal_store(f4{rgb[0][0], rgb[0][1], rgb[0][2], sig[0][0]}, reinterpret_cast<f4 *>(A.raw) + sample);
}

static int plan16(const snerf_mlp_desc *desc, Plan &P, const char *what) {
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "%s: desc is null", what);
    if (make_plan(*desc, P, why, 32) != 0) return fail(SNERF_E_BADARG, "%s: %s", what, why);
    if (P.width != 256) return fail(SNERF_E_BADARG, "%s: the split-bf16 path supports width 256 only", what);
    return SNERF_OK;
}

template <int NS, bool TRAIN>
static int launch_bf16(const FwdArgs &A, hipStream_t s) {
    constexpr int NW = 8;
    const int lds = 3 * slab16_bytes(NS);
    static bool attr = false;  // idempotent; a race only repeats the call
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_fwd_bf16_kernel<256, NW, NS, TRAIN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return fail(SNERF_E_LAUNCH, "mlp_fwd_bf16: cannot raise the dynamic LDS limit to %d bytes", lds);
        attr = true;
    }
    const int64_t grid = (A.n + NW * 16 - 1) / (NW * 16);
    if (grid > 0x7fffffffLL) return fail(SNERF_E_BADARG, "mlp_fwd_bf16: n too large");
    hipLaunchKernelGGL((mlp_fwd_bf16_kernel<256, NW, NS, TRAIN>), dim3((unsigned)grid), dim3(NW * 64), lds, s, A);
    return check_launch("mlp_fwd_bf16");
}

// argument checks + FwdArgs shared by the inference and the training forward; act == nullptr: inference
static int fwd_bf16(const snerf_mlp_desc *desc, const void *packed, int nsplit, const float *x, const float *dirs,
                    int dirs_per_sample, const float *add, int64_t n, int samples_per_ray, float *raw, float *act,
                    bool train, snerf_stream_t stream, const char *what) {
    Plan P;
    if (nsplit != 2 && nsplit != 3) return fail(SNERF_E_BADARG, "%s: nsplit must be 2 or 3", what);
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
        A.pos_nkb16 = Q.pos_nkb;
        A.add_nkb16 = Q.add_nkb;
        A.dir_nkb16 = Q.dir_nkb;
        if (nsplit == 3) return launch_bf16<3, true>(A, (hipStream_t)stream);
        return launch_bf16<2, true>(A, (hipStream_t)stream);
    }
    if (nsplit == 3) return launch_bf16<3, false>(A, (hipStream_t)stream);
    return launch_bf16<2, false>(A, (hipStream_t)stream);
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
    return snerf::fwd_bf16(desc, packed, nsplit, x, dirs, dirs_per_sample, add, n, samples_per_ray, raw, nullptr, false,
                           stream, "mlp_fwd_bf16");
}

extern "C" int snerf_mlp_fwd_train_bf16_f32(const snerf_mlp_desc *desc, const void *packed, int nsplit, const float *x,
                                            const float *dirs, int dirs_per_sample, const float *add, int64_t n,
                                            int samples_per_ray, float *raw, float *act, snerf_stream_t stream) {
    return snerf::fwd_bf16(desc, packed, nsplit, x, dirs, dirs_per_sample, add, n, samples_per_ray, raw, act, true, stream,
                           "mlp_fwd_train_bf16");
}
