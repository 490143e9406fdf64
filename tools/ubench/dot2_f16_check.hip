// Does v_dot2c_f32_f16 give v - (float)rtz_f16(v) bit for bit, including fp16-denormal parts?  (the f16x3 operand split)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/dot2_f16_check.hip -o /tmp/dot2_f16_check && /tmp/dot2_f16_check
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
typedef __fp16 h2v __attribute__((ext_vector_type(2)));
typedef _Float16 hf2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *in, float *ref, float *got, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float v0 = in[2 * i], v1 = in[2 * i + 1];
    const h2v h = __builtin_amdgcn_cvt_pkrtz(v0, v1);
    ref[2 * i] = v0 - (float)h[0];
    ref[2 * i + 1] = v1 - (float)h[1];
    unsigned c_lo = 0x0000bc00u, c_hi = 0xbc000000u;   // (-1, 0), (0, -1) as fp16 pairs
    asm volatile("" : "+s"(c_lo), "+s"(c_hi));
    const hf2 hh = __builtin_bit_cast(hf2, h);
    got[2 * i] = __builtin_amdgcn_fdot2(hh, __builtin_bit_cast(hf2, c_lo), v0, false);
    got[2 * i + 1] = __builtin_amdgcn_fdot2(hh, __builtin_bit_cast(hf2, c_hi), v1, false);
}
int main() {
    const int n = 1 << 16;
    static float h[n], r[n], g[n];
    unsigned seed = 777u;
    for (int i = 0; i < n; ++i) {
        seed = seed * 1664525u + 1013904223u;
        const float m = (float)((seed >> 8) & 0xffffff) / 16777216.0f * 2.0f - 1.0f;
        const int e = (int)((seed >> 3) % 44) - 28;   // 2^-28 .. 2^15: covers fp16 denormals (< 2^-14) and values that round to zero
        h[i] = ldexpf(m, e);
    }
    float *din, *dr, *dg;
    hipMalloc(&din, sizeof(h));
    hipMalloc(&dr, sizeof(h));
    hipMalloc(&dg, sizeof(h));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, din, dr, dg, n);
    hipMemcpy(r, dr, sizeof(h), hipMemcpyDeviceToHost);
    hipMemcpy(g, dg, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0, bad_denorm = 0;
    for (int i = 0; i < n; ++i)
        if (memcmp(&r[i], &g[i], 4)) {
            if (fabsf(h[i]) < 6.2e-5f) ++bad_denorm;
            if (bad++ < 6) printf("v = %a: cvt + sub %a, dot2 %a\n", h[i], r[i], g[i]);
        }
    printf("f16 residuals: %d of %d differ (%d of them with |v| in fp16's denormal range)\n", bad, n, bad_denorm);
    return 0;
}
