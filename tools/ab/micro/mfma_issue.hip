// Microbenchmark: how much of the fp32 MFMA peak ONE wave per SIMD can issue (the widths above 256 run that way), beside two.
//   hipcc --offload-arch=gfx950 -O3 tools/ab/micro/mfma_issue.hip -o /tmp/mfma_issue && /tmp/mfma_issue
// Every wave runs ITER x NACC independent v_mfma_f32_16x16x4_f32 (NACC accumulators, round-robin: no dependent issue inside the
// 40-cycle latency), optionally with one ds_read_b128 per 4 MFMAs like the fused kernels' A-operand stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(512) void k(float *out, int iters, int lds_words) {
    extern __shared__ float sm[];
    for (int i = threadIdx.x; i < lds_words; i += blockDim.x) sm[i] = 1.0f / (1 + i);
    __syncthreads();
    f4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    float a = 1.0f + lane, b = 0.5f;
    const f4 *ap = reinterpret_cast<const f4 *>(sm) + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; i += 2) {
            f4 a0, a1;
            if (LDS) {
                a0 = ap[(i & 31) * 64];
                a1 = ap[((i + 1) & 31) * 64];
            } else {
                a0 = f4{a, a, a, a};
                a1 = a0;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], b, acc[i], 0, 0, 0);
                acc[i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[r], b, acc[i + 1], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool LDS>
static void run(const char *name, int threads, float *out) {
    const int iters = 2000, blocks = 256 * 4;
    const int lds = 100 * 1024;   // one workgroup per CU
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<NACC, LDS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<NACC, LDS><<<blocks, threads, lds>>>(out, 10, lds / 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, LDS><<<blocks, threads, lds>>>(out, iters, lds / 4);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * (threads / 64) * iters * NACC * 4 * 2048.0;
    printf("%-44s %2d waves/SIMD  %8.3f ms  %7.1f TFLOP/s = %.3f of 157.3\n", name, threads / 256, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 4 * 512 * sizeof(float));
    run<32, false>("32 accumulators, operands in registers", 256, out);
    run<32, false>("32 accumulators, operands in registers", 512, out);
    run<16, false>("16 accumulators, operands in registers", 256, out);
    run<16, false>("16 accumulators, operands in registers", 512, out);
    run<32, true>("32 accumulators, A from LDS (b128 / 4 MFMA)", 256, out);
    run<32, true>("32 accumulators, A from LDS (b128 / 4 MFMA)", 512, out);
    run<16, true>("16 accumulators, A from LDS (b128 / 4 MFMA)", 512, out);
    return 0;
}
