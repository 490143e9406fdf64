// Micro-benchmark: cycles per v_mfma_f32_16x16x4_f32 in the k-block pattern of the fused kernels (two accumulators alternating,
// two ds_read_b128 of the next tile pair per 8 MFMAs), with and without the LDS reads, 1 or 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_lds.hip -o /tmp/mfma_lds && /tmp/mfma_lds
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
typedef float f4 __attribute__((ext_vector_type(4)));

// 0: MFMA only; 1: + 2 ds_read_b128 per 8 MFMAs (operands used); 2: ds_reads issued but MFMA operands constant;
// 3: as 1 + a workgroup barrier every 128 MFMAs per wave (the slab hand-over); 4: as 3 + 33 KiB of LDS-DMA per period behind
// the period's first k-block (the slab refill); 5: as 3 + the register-staged refill (global loads -> ds_write before the barrier)
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, long long *cyc, int iters, const float *src) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)(i & 7) * 1e-3f;
    __syncthreads();
    f4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
    f4 b = f4{1.f, 0.5f, 0.25f, 0.125f};
    const f4 *ap = reinterpret_cast<const f4 *>(lds) + lane;
    f4 a0 = ap[0], a1 = ap[64];
    long long t0 = __builtin_readcyclecounter();
    f4 st[5];
    const f4 *gsrc = reinterpret_cast<const f4 *>(src) + threadIdx.x;
    if (MODE == 5) {
#pragma unroll
        for (int q = 0; q < 4; ++q) st[q] = gsrc[q * 512];
    }
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 3 && (it & 1) == 0 && it) {
            if (MODE == 5) {   // register-staged hand-over: staged slab -> LDS, next global loads, barrier
                f4 *d = reinterpret_cast<f4 *>(lds + 8192 + ((it >> 1) % 2) * 8448) + threadIdx.x;
#pragma unroll
                for (int q = 0; q < 4; ++q) d[q * 512] = st[q];
#pragma unroll
                for (int q = 0; q < 4; ++q) st[q] = gsrc[q * 512 + ((it >> 1) & 63) * 2112];
            }
            if (MODE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
#pragma unroll
        for (int to = 0; to < 16; to += 2) {
            f4 n0 = a0, n1 = a1;
            if (MODE >= 1) {
                n0 = ap[((to + 2) & 15) * 64];
                n1 = ap[((to + 3) & 15) * 64];
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(MODE == 2 ? b[r] : a0[r], b[r], acc[to], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                acc[to + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(MODE == 2 ? b[r] : a1[r], b[r], acc[to + 1], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 2) { asm volatile("" ::"v"(n0), "v"(n1)); }
            a0 = n0;
            a1 = n1;
            if (MODE == 4 && to == 14 && (it & 1) == 0) {   // the refill of this period, behind its first k-block
                const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
                const char *g = reinterpret_cast<const char *>(src) + lane * 16 + (size_t)((it >> 1) & 63) * 33792;
                char *dst = reinterpret_cast<char *>(lds + 8192 + ((it >> 1) % 2) * 8448);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + (q * 512 + wave * 64) * 16),
                                                     (__attribute__((address_space(3))) void *)(dst + (q * 512 + wave * 64) * 16), 16, 0, 0);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

// The split-precision pattern: v_mfma_f32_16x16x32_bf16 (16 cycles), one ds_read_b128 per MFMA (a 1 KiB A part per product),
// PRODUCTS products per output tile, 16 tiles per k-block = one slab of PARTS x 16 KiB per period.
// MODE 0: MFMA + LDS reads; 1: + barrier per slab; 2: + the slab refill by LDS-DMA, issued by alternating halves (as shipped)
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u4c __attribute__((ext_vector_type(4)));
#ifdef SPLIT_DOT2
// -DSPLIT_DOT2: the residual v - bf16(v) as ONE v_dot2c_f32_bf16 per value (acc = v; acc += pack.lo * -1 + pack.hi * 0) instead of
// shift / mask + subtract; the parts are the same bits (the residual is exactly representable either way)
typedef __bf16 bf2u __attribute__((ext_vector_type(2)));
typedef float f2u __attribute__((ext_vector_type(2)));
typedef unsigned u4u __attribute__((ext_vector_type(4)));
template <int PARTS>
__device__ inline void split_parts(const float (&v)[8], bf8 (&b)[PARTS]) {
    unsigned c_lo = 0x0000bf80u, c_hi = 0xbf800000u;   // (-1, 0) and (0, -1) as bf16 pairs, kept out of the inline-constant encodings
    asm volatile("" : "+s"(c_lo), "+s"(c_hi));
    u4u w[PARTS];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float r0 = v[2 * i], r1 = v[2 * i + 1];
#pragma unroll
        for (int p = 0; p < PARTS; ++p) {
            const bf2u pk = __builtin_convertvector(f2u{r0, r1}, bf2u);
            w[p][i] = __builtin_bit_cast(unsigned, pk);
            if (p + 1 < PARTS) {
                r0 = __builtin_amdgcn_fdot2_f32_bf16(pk, __builtin_bit_cast(bf2u, c_lo), r0, false);
                r1 = __builtin_amdgcn_fdot2_f32_bf16(pk, __builtin_bit_cast(bf2u, c_hi), r1, false);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < PARTS; ++p) b[p] = __builtin_bit_cast(bf8, w[p]);
}
#else
template <int PARTS>
__device__ inline void split_parts(const float (&v)[8], bf8 (&b)[PARTS]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float r = v[e];
#pragma unroll
        for (int p = 0; p < PARTS; ++p) {
            const __bf16 h = (__bf16)r;
            b[p][e] = h;
            r -= (float)h;
        }
    }
}
#endif
// the split of 512 values, for comparing the two forms bit by bit (main: `mfma_lds split-check`)
__global__ void split_check_kernel(const float *in, unsigned *out) {
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = in[threadIdx.x * 8 + e];
    bf8 b[3];
    split_parts<3>(v, b);
    for (int p = 0; p < 3; ++p) {
        const u4c w = __builtin_bit_cast(u4c, b[p]);
        for (int i = 0; i < 4; ++i) out[(threadIdx.x * 3 + p) * 4 + i] = w[i];
    }
}
template <int MODE, int PARTS, int PRODUCTS, bool SPLIT = false>
__global__ __launch_bounds__(512) void kb(float *out, long long *cyc, int iters, const float *src) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < PARTS * 4096; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    f4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
    bf8 b[PARTS];
#pragma unroll
    for (int p = 0; p < PARTS; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) b[p][e] = (__bf16)1.0f;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 1.0f + 1e-3f * (float)(lane + e);
    const bf8 *ap = reinterpret_cast<const bf8 *>(lds) + lane;
    // MODE 3: no workgroup barrier.  Three LDS counters hand the slabs over: landed[parity] (+1 per producing wave once its DMA
    // pieces of a slab are in LDS), done (+1 per wave per slab consumed).  Slab s+2 is issued by half (s & 1) in the MIDDLE of slab s
    // (once every wave has finished slab s-1, whose slot it takes), its landing is signalled in the middle of slab s+1, and it is
    // needed at the start of slab s+2: nobody waits unless somebody is a full half slab behind.  Spins are bounded.
    int *cnt = reinterpret_cast<int *>(reinterpret_cast<char *>(lds) + 3 * PARTS * 16384);
    if (MODE == 4) {
        if (threadIdx.x < 2) cnt[threadIdx.x] = 8;   // half-slabs 0..3 count as resident
        if (threadIdx.x == 2) cnt[2] = 0;
        if (threadIdx.x == 3) cnt[3] = 0;
        __syncthreads();
        if (wave >= 4) __builtin_amdgcn_s_sleep(12);   // ~768 cycles: half of a half-slab period
    }
    if (MODE == 3) {
        if (threadIdx.x < 2) cnt[threadIdx.x] = 4;
        if (threadIdx.x == 2) cnt[2] = 0;
        if (threadIdx.x == 3) cnt[3] = 0;
        __syncthreads();
    }
    const int half = wave >= 4;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (SPLIT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(v[e]));
            split_parts<PARTS>(v, b);
        }
        if (MODE == 3) {
            const int need = 4 * ((it >> 1) + 1);
            int spins = 0;
            while (__hip_atomic_load(&cnt[it & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) { cnt[3] = 1; break; }
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int to = 0; to < 8; ++to) {
                bf8 a[PARTS];
#pragma unroll
                for (int p = 0; p < PARTS; ++p) a[p] = ap[(to * PARTS + p) * 64];
#pragma unroll
                for (int t = 0; t < PRODUCTS; ++t) acc[to] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t % PARTS], b[(t / PARTS + t) % PARTS], acc[to], 0, 0, 0);
            }
            if (half == (it & 1)) {   // producer of slab it + 2
                spins = 0;
                while (__hip_atomic_load(&cnt[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 8 * it) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 22)) { cnt[3] = 1; break; }
                }
                asm volatile("" ::: "memory");
                const char *g = reinterpret_cast<const char *>(src) + lane * 16 + (size_t)(it & 31) * PARTS * 16384;
                char *dst = reinterpret_cast<char *>(lds) + PARTS * 16384 * ((it + 2) % 3);
#pragma unroll
                for (int q = 0; q < PARTS * 4; ++q)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + ((wave & 3) * PARTS * 4 + q) * 1024),
                                                     (__attribute__((address_space(3))) void *)(dst + ((wave & 3) * PARTS * 4 + q) * 1024), 16, 0, 0);
            } else if (it) {          // my pieces of slab it + 1 were issued a slab ago
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(&cnt[half], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#pragma unroll
            for (int to = 8; to < 16; ++to) {
                bf8 a[PARTS];
#pragma unroll
                for (int p = 0; p < PARTS; ++p) a[p] = ap[(to * PARTS + p) * 64];
#pragma unroll
                for (int t = 0; t < PRODUCTS; ++t) acc[to] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t % PARTS], b[(t / PARTS + t) % PARTS], acc[to], 0, 0, 0);
            }
            if (lane == 0) __hip_atomic_fetch_add(&cnt[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            continue;
        }
        if (MODE == 4) {
            // half-slab granularity (6 ring slots of 8 tiles), no barrier, the two halves of the workgroup (= the two waves of every
            // SIMD) started half a half-slab apart: half-slab h+4 is issued by half (h & 1) at the boundary after h (needs everyone
            // past h-2), signalled landed at that half's next boundary, needed at the start of h+4.  Either half may run a whole
            // half-slab ahead of the other without anybody waiting.
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {
                const int h = 2 * it + hs;
                const int need = 4 * ((h >> 1) + 1);
                int spins = 0;
                while (__hip_atomic_load(&cnt[h & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
                    if (++spins > (1 << 22)) { cnt[3] = 1; break; }
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int to = 8 * hs; to < 8 * hs + 8; ++to) {
                    bf8 a[PARTS];
#pragma unroll
                    for (int p = 0; p < PARTS; ++p) a[p] = ap[(to * PARTS + p) * 64];
#pragma unroll
                    for (int t = 0; t < PRODUCTS; ++t) acc[to] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t % PARTS], b[(t / PARTS + t) % PARTS], acc[to], 0, 0, 0);
                }
                if (lane == 0) __hip_atomic_fetch_add(&cnt[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (half == (h & 1)) {   // producer of half-slab h + 4
                    spins = 0;
                    while (__hip_atomic_load(&cnt[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 8 * (h - 1)) {
                        if (++spins > (1 << 22)) { cnt[3] = 1; break; }
                    }
                    asm volatile("" ::: "memory");
                    const char *g = reinterpret_cast<const char *>(src) + lane * 16 + (size_t)(h & 63) * PARTS * 8192;
                    char *dst = reinterpret_cast<char *>(lds) + PARTS * 8192 * ((h + 4) % 6);
#pragma unroll
                    for (int q = 0; q < PARTS * 2; ++q)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + ((wave & 3) * PARTS * 2 + q) * 1024),
                                                         (__attribute__((address_space(3))) void *)(dst + ((wave & 3) * PARTS * 2 + q) * 1024), 16, 0, 0);
                } else if (h) {          // my pieces of half-slab h + 3 were issued a half-slab ago
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_fetch_add(&cnt[half], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            continue;
        }
        if (MODE >= 1 && it) {
            if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (MODE == 2 && ((wave >= 4) == ((it & 1) != 0))) {   // this half's turn: the whole slab, PARTS * 4 pieces per wave
                const char *g = reinterpret_cast<const char *>(src) + lane * 16 + (size_t)(it & 31) * PARTS * 16384;
                char *dst = reinterpret_cast<char *>(lds) + PARTS * 16384 * (1 + (it & 1));
#pragma unroll
                for (int q = 0; q < PARTS * 4; ++q)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + ((wave & 3) * PARTS * 4 + q) * 1024),
                                                     (__attribute__((address_space(3))) void *)(dst + ((wave & 3) * PARTS * 4 + q) * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int to = 0; to < 16; ++to) {
            bf8 a[PARTS];
#pragma unroll
            for (int p = 0; p < PARTS; ++p) a[p] = ap[(to * PARTS + p) * 64];
#pragma unroll
            for (int t = 0; t < PRODUCTS; ++t) acc[to] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t % PARTS], b[(t / PARTS + t) % PARTS], acc[to], 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    if (MODE >= 3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (cnt[3]) s = __builtin_nanf("");   // a bounded spin ran out
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int MODE, int PARTS, int PRODUCTS, bool SPLIT = false>
void runb(const char *name) {
    float *out, *src;
    long long *cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 8 * 256 * 8);
    hipMalloc(&src, 32 * PARTS * 16384 + 65536);
    hipMemset(src, 0, 32 * PARTS * 16384 + 65536);
    const int iters = 20000, lds_bytes = 3 * PARTS * 16384 + 16;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kb<MODE, PARTS, PRODUCTS, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((kb<MODE, PARTS, PRODUCTS, SPLIT>), dim3(256), dim3(512), lds_bytes, 0, out, cyc, iters, src);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((kb<MODE, PARTS, PRODUCTS, SPLIT>), dim3(256), dim3(512), lds_bytes, 0, out, cyc, iters, src);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double tflops = 256.0 * 8 * iters * 16.0 * PRODUCTS * 16384.0 / (ms * 1e-3) / 1e12;
    float probe = 0.f;
    hipMemcpy(&probe, out, 4, hipMemcpyDeviceToHost);
    printf("bf16 16x16x32, %d parts / %d products%s, %-40s wall %.3f ms = %.0f TFLOP/s of products (%.3f of 2500)%s\n", PARTS, PRODUCTS,
           SPLIT ? " + VALU split" : "", name, ms, tflops, tflops / 2500.0, probe != probe ? "  [SPIN LIMIT HIT]" : "");
    hipFree(out);
    hipFree(cyc);
    hipFree(src);
}

// The same slab stream with 32 samples per wave: v_mfma_f32_32x32x16_bf16 (32 cycles), ONE wave per SIMD (4 waves, 128 samples per
// workgroup as shipped; 8 accumulator tiles of 16 registers).  A slab = one 32-wide k-block x 256 outputs x PARTS = 8 output tiles
// x 2 k-halves x PARTS A pieces of 1 KiB, each read once per wave (half the LDS bytes per FLOP of the 16-sample form).
// MODE as kb; the refill is issued by all four waves every slab (there is no other half to alternate with).
// SPLIT: the operand split of the k-block's B values is done on the VALU inside the loop (16 values per lane here, 8 in kb).
typedef float f16v __attribute__((ext_vector_type(16)));
template <int MODE, int PARTS, int PRODUCTS, bool SPLIT>
__global__ __launch_bounds__(256) void kc(float *out, long long *cyc, int iters, const float *src) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < PARTS * 4096; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    f16v acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    bf8 b[2][PARTS];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int p = 0; p < PARTS; ++p)
#pragma unroll
            for (int e = 0; e < 8; ++e) b[h][p][e] = (__bf16)1.0f;
    float v[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[h][e] = 1.0f + 1e-3f * (float)(lane + e + h);
    const bf8 *ap = reinterpret_cast<const bf8 *>(lds) + lane;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 1 && it) {
            if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (MODE == 2) {
                const char *g = reinterpret_cast<const char *>(src) + lane * 16 + (size_t)(it & 31) * PARTS * 16384;
                char *dst = reinterpret_cast<char *>(lds) + PARTS * 16384 * (1 + (it & 1));
#pragma unroll
                for (int q = 0; q < PARTS * 4; ++q)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + (wave * PARTS * 4 + q) * 1024),
                                                     (__attribute__((address_space(3))) void *)(dst + (wave * PARTS * 4 + q) * 1024), 16, 0, 0);
            }
        }
        if (SPLIT) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(v[h][e]));
                split_parts<PARTS>(v[h], b[h]);
            }
        }
#pragma unroll
        for (int to = 0; to < 8; ++to)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bf8 a[PARTS];
#pragma unroll
                for (int p = 0; p < PARTS; ++p) a[p] = ap[((to * 2 + h) * PARTS + p) * 64];
#pragma unroll
                for (int t = 0; t < PRODUCTS; ++t)
                    acc[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t % PARTS], b[h][(t / PARTS + t) % PARTS], acc[to], 0, 0, 0);
            }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[t][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}
template <int MODE, int PARTS, int PRODUCTS, bool SPLIT>
void runc(const char *name) {
    float *out, *src;
    long long *cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 8 * 256 * 8);
    hipMalloc(&src, 32 * PARTS * 16384 + 65536);
    hipMemset(src, 0, 32 * PARTS * 16384 + 65536);
    const int iters = 20000, lds_bytes = 3 * PARTS * 16384;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kc<MODE, PARTS, PRODUCTS, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((kc<MODE, PARTS, PRODUCTS, SPLIT>), dim3(256), dim3(256), lds_bytes, 0, out, cyc, iters, src);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((kc<MODE, PARTS, PRODUCTS, SPLIT>), dim3(256), dim3(256), lds_bytes, 0, out, cyc, iters, src);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double tflops = 256.0 * 4 * iters * 16.0 * PRODUCTS * 32768.0 / (ms * 1e-3) / 1e12;
    printf("bf16 32x32x16, %d parts / %d products%s, %-40s wall %.3f ms = %.0f TFLOP/s of products (%.3f of 2500)\n", PARTS, PRODUCTS,
           SPLIT ? " + VALU split" : "", name, ms, tflops, tflops / 2500.0);
    hipFree(out);
    hipFree(cyc);
    hipFree(src);
}

// The fp32 k-block pattern with CG 16-sample column groups per wave and ONE wave per SIMD (4 waves, CG x 64 samples per
// workgroup): every A tile read from LDS feeds CG MFMAs, and a 33 KiB slab lasts CG x 128 MFMAs per wave.
// MODE 1: MFMA + A reads; 3: + barrier per slab; 4: + the slab refill by LDS-DMA
template <int MODE, int CG>
__global__ __launch_bounds__(256) void k3(float *out, long long *cyc, int iters, const float *src) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)(i & 7) * 1e-3f;
    __syncthreads();
    f4 acc[CG][16];
#pragma unroll
    for (int c = 0; c < CG; ++c)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[c][t] = f4{0.f, 0.f, 0.f, 0.f};
    f4 b[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) b[c] = f4{1.f + c, 0.5f, 0.25f, 0.125f};
    const f4 *ap = reinterpret_cast<const f4 *>(lds) + lane;
    f4 a0 = ap[0], a1 = ap[64];
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 3 && (it & 1) == 0 && it) {
            if (MODE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
#pragma unroll
        for (int to = 0; to < 16; to += 2) {
            f4 n0 = ap[((to + 2) & 15) * 64], n1 = ap[((to + 3) & 15) * 64];
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int c = 0; c < CG; ++c) {
                    acc[c][to] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], b[c][r], acc[c][to], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    acc[c][to + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[r], b[c][r], acc[c][to + 1], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            a0 = n0;
            a1 = n1;
            if (MODE == 4 && (it & 1) == 0 && (to & 1) == 0) {   // the refill of this period: one piece per wave behind each tile pair
                const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
                const int q = to >> 1;                             // 8 rounds x 4 waves = 32 pieces of 1 KiB (+ the bias piece: wave 0, round 0)
                const char *g = reinterpret_cast<const char *>(src) + lane * 16 + (size_t)((it >> 1) & 63) * 33792;
                char *dst = reinterpret_cast<char *>(lds + 8192 + ((it >> 1) % 2) * 8448);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + (q * 4 + wave) * 1024),
                                                 (__attribute__((address_space(3))) void *)(dst + (q * 4 + wave) * 1024), 16, 0, 0);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CG; ++c)
#pragma unroll
        for (int t = 0; t < 16; ++t) s += acc[c][t][0] + acc[c][t][1] + acc[c][t][2] + acc[c][t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE, int CG>
void run3(const char *name) {
    float *out, *src;
    long long *cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 8 * 256 * 8);
    hipMalloc(&src, 64 * 33792 + 65536);
    hipMemset(src, 0, 64 * 33792 + 65536);
    const int iters = 8000, lds_bytes = 32768 + 2 * 33792;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k3<MODE, CG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k3<MODE, CG>), dim3(256), dim3(256), lds_bytes, 0, out, cyc, iters, src);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k3<MODE, CG>), dim3(256), dim3(256), lds_bytes, 0, out, cyc, iters, src);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double tflops = 256.0 * 4 * iters * 64.0 * CG * 2048.0 / (ms * 1e-3) / 1e12;
    printf("fp32 16x16x4, %d column groups per wave, 1 wave/SIMD, %-44s wall %.3f ms = %.1f TFLOP/s (%.3f of 157.3)\n", CG, name, ms, tflops,
           tflops / 157.3);
    hipFree(out);
    hipFree(cyc);
    hipFree(src);
}

template <int MODE>
void run(const char *name, int threads) {
    float *out;
    long long *cyc, h[8 * 256];
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, sizeof(h));
    const int iters = 20000;
    const int lds_bytes = 32768 + 2 * 33792;
    static float *src = nullptr;
    if (!src) {
        hipMalloc(&src, 64 * 33792 + 65536);
        hipMemset(src, 0, 64 * 33792 + 65536);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    }
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), lds_bytes, 0, out, cyc, iters, src);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), lds_bytes, 0, out, cyc, iters, src);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    double mean = 0;
    for (int i = 0; i < 256 * waves; ++i) mean += (double)h[i];
    mean /= 256 * waves;
    const double per_mfma_wave = mean / (iters * 64.0);
    const double tflops = 256.0 * waves * iters * 64.0 * 2048.0 / (ms * 1e-3) / 1e12;
    printf("%-46s waves/SIMD %d: %.2f s_memtime ticks per MFMA per wave; wall %.3f ms = %.1f TFLOP/s (%.3f of 157.3)\n", name,
           waves / 4, per_mfma_wave, ms, tflops, tflops / 157.3);
    hipFree(out);
    hipFree(cyc);
}
static unsigned host_bf16_rne(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
static int split_check() {
    const int n = 64 * 8;
    float h[n];
    unsigned seed = 12345u;
    for (int i = 0; i < n; ++i) {
        seed = seed * 1664525u + 1013904223u;
        const float m = (float)((seed >> 8) & 0xffffff) / 16777216.0f * 2.0f - 1.0f;
        const int e = (int)((seed >> 3) % 40) - 30;
        h[i] = i < 8 ? (i & 1 ? -0.0f : 0.0f) : ldexpf(m, e);
    }
    float *din;
    unsigned *dout, ho[64 * 12];
    hipMalloc(&din, sizeof(h));
    hipMalloc(&dout, sizeof(ho));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(split_check_kernel, dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t)
        for (int e = 0; e < 8; ++e) {
            float r = h[t * 8 + e];
            for (int p = 0; p < 3; ++p) {
                const unsigned want = host_bf16_rne(r);
                const unsigned got = (ho[(t * 3 + p) * 4 + e / 2] >> (16 * (e & 1))) & 0xffffu;
                if (want != got && bad++ < 8) printf("value %d part %d: want %04x got %04x (v = %a)\n", t * 8 + e, p, want, got, h[t * 8 + e]);
                unsigned wb = want << 16;
                float wf;
                memcpy(&wf, &wb, 4);
                r -= wf;
            }
        }
    printf("split check: %d of %d parts differ from round-to-nearest-even bf16 of the exact residuals\n", bad, n * 3);
    return bad != 0;
}
int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "split-check")) return split_check();
    if (argc > 1 && !strcmp(argv[1], "split")) {
        runb<2, 3, 6, true>("+ 48 KiB LDS-DMA refill per slab (halves alternate)");
        runb<2, 2, 3, true>("+ 32 KiB LDS-DMA refill per slab (halves alternate)");
        return 0;
    }
    run<0>("MFMA only", 256);
    run<0>("MFMA only", 512);
    run<1>("+ 2 ds_read_b128 per 8 MFMAs (used)", 256);
    run<1>("+ 2 ds_read_b128 per 8 MFMAs (used)", 512);
    run<2>("+ 2 ds_read_b128 per 8 MFMAs (results unused)", 512);
    run<3>("+ barrier every 128 MFMAs per wave", 512);
    run<4>("+ barrier + 33 KiB LDS-DMA refill per period", 512);
    run<5>("+ barrier + register-staged refill per period", 512);
    run3<1, 1>("MFMA + A reads");
    run3<3, 1>("+ barrier per slab");
    run3<4, 1>("+ barrier + 33 KiB LDS-DMA refill per slab");
    run3<1, 2>("MFMA + A reads");
    run3<3, 2>("+ barrier per slab");
    run3<4, 2>("+ barrier + 33 KiB LDS-DMA refill per slab");
    run3<1, 3>("MFMA + A reads");
    run3<3, 3>("+ barrier per slab");
    run3<4, 3>("+ barrier + 33 KiB LDS-DMA refill per slab");
    runb<0, 3, 6>("MFMA + one ds_read_b128 per product");
    runb<1, 3, 6>("+ barrier per slab");
    runb<2, 3, 6>("+ 48 KiB LDS-DMA refill per slab (halves alternate)");
    runb<0, 2, 3>("MFMA + one ds_read_b128 per product");
    runb<1, 2, 3>("+ barrier per slab");
    runb<2, 2, 3>("+ 32 KiB LDS-DMA refill per slab (halves alternate)");
    runb<2, 3, 6, true>("+ 48 KiB LDS-DMA refill per slab (halves alternate)");
    runb<2, 2, 3, true>("+ 32 KiB LDS-DMA refill per slab (halves alternate)");
    runb<4, 3, 6>("+ 48 KiB refill, half-slab counters, halves de-phased");
    runb<4, 2, 3>("+ 32 KiB refill, half-slab counters, halves de-phased");
    runb<4, 3, 6, true>("+ 48 KiB refill, half-slab counters, halves de-phased");
    runb<4, 2, 3, true>("+ 32 KiB refill, half-slab counters, halves de-phased");
    runb<3, 3, 6>("+ 48 KiB refill, counters instead of the barrier");
    runb<3, 2, 3>("+ 32 KiB refill, counters instead of the barrier");
    runb<3, 3, 6, true>("+ 48 KiB refill, counters instead of the barrier");
    runb<3, 2, 3, true>("+ 32 KiB refill, counters instead of the barrier");
    runc<0, 3, 6, false>("MFMA + one ds_read_b128 per 2 products");
    runc<1, 3, 6, false>("+ barrier per slab");
    runc<2, 3, 6, false>("+ 48 KiB LDS-DMA refill per slab (all 4 waves)");
    runc<2, 3, 6, true>("+ 48 KiB LDS-DMA refill per slab (all 4 waves)");
    runc<0, 2, 3, false>("MFMA + one ds_read_b128 per 1.5 products");
    runc<1, 2, 3, false>("+ barrier per slab");
    runc<2, 2, 3, false>("+ 32 KiB LDS-DMA refill per slab (all 4 waves)");
    runc<2, 2, 3, true>("+ 32 KiB LDS-DMA refill per slab (all 4 waves)");
    return 0;
}
