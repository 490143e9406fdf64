// Times the HBM-class entry points of a libsmplnerf_hip.so (path = argv[1]) at one 128x128 frame (16 384 rays) through the
// C-ABI with HIP events: snerf_sample_pdf_f32 (64 + 128), snerf_composite_fwd_f32 (N = 64, 192), snerf_composite_bwd_f32.
// Used to A/B kernel variants: build each variant to its own .so and pass them one after the other.
//   hipcc -O2 tools/ubench/helpers_bench.cpp -o tools/ubench/helpers_bench -ldl
//   tools/ubench/helpers_bench smpl_nerf_amd/csrc/libsmplnerf_hip.so [more.so ...]
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int (*sample_pdf_fn)(const float *, const float *, const float *, const float *, const float *, int64_t, int, int, int64_t *,
                             float *, float *, float *, void *);
typedef int (*composite_fwd_fn)(const float *, const float *, const float *, int, const float *, int64_t, int, int, float *, float *,
                                float *, void *);
typedef int (*composite_bwd_fn)(const float *, const float *, const float *, int, const float *, int64_t, int, int, const float *,
                                float *, float *, void *);

static float *dev(const std::vector<float> &h) {
    float *p;
    hipMalloc(&p, h.size() * 4);
    hipMemcpy(p, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    return p;
}
static double urand() { return (double)rand() / RAND_MAX; }

template <class F>
static double time_us(F f, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / iters;
}

int main(int argc, char **argv) {
    const int64_t B = argc > 2 && atoi(argv[argc - 1]) > 0 ? atoi(argv[--argc]) : 16384;
    const int Nc = 64, Nf = 128, Nt = Nc + Nf;
    srand(1);
    std::vector<float> z(B * Nc), w(B * Nc), u(Nf), o(B * 3), d(B * 3), raw64(B * Nc * 4), raw192(B * Nt * 4), z192(B * Nt), drgb(B * 3);
    for (int64_t b = 0; b < B; ++b) {
        const double c = urand() * 64, s = 2 + urand() * 6;
        for (int i = 0; i < Nc; ++i) {
            z[b * Nc + i] = 1.f + 3.f * (float)((i + urand()) / Nc);
            w[b * Nc + i] = (float)(exp(-(i - c) * (i - c) / (2 * s * s)) * 0.3 + 1e-4 * urand());
        }
        for (int i = 0; i < Nt; ++i) z192[b * Nt + i] = 1.f + 3.f * (float)((i + urand()) / Nt);
        for (int k = 0; k < 3; ++k) o[b * 3 + k] = (float)(urand() - .5), d[b * 3 + k] = (float)(urand() - .5), drgb[b * 3 + k] = (float)(urand() - .5);
    }
    for (int i = 0; i < Nf; ++i) u[i] = (float)i / (Nf - 1);
    for (auto &v : raw64) v = (float)(urand() * 4 - 1);
    for (auto &v : raw192) v = (float)(urand() * 4 - 1);
    float *dz = dev(z), *dw = dev(w), *du = dev(u), *dor = dev(o), *dd = dev(d), *draw64 = dev(raw64), *draw192 = dev(raw192), *dz192 = dev(z192),
          *ddrgb = dev(drgb);
    float *zf, *pts, *rgb, *wts, *alpha, *d_raw;
    hipMalloc(&zf, B * Nt * 4);
    hipMalloc(&pts, B * Nt * 12);
    hipMalloc(&rgb, B * 12);
    hipMalloc(&wts, B * Nt * 4);
    hipMalloc(&alpha, B * Nt * 4);
    hipMalloc(&d_raw, B * Nt * 16);
    for (int a = 1; a < argc; ++a) {
        void *h = dlopen(argv[a], RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            fprintf(stderr, "%s: %s\n", argv[a], dlerror());
            return 1;
        }
        auto sp = (sample_pdf_fn)dlsym(h, "snerf_sample_pdf_f32");
        auto cf = (composite_fwd_fn)dlsym(h, "snerf_composite_fwd_f32");
        auto cb = (composite_bwd_fn)dlsym(h, "snerf_composite_bwd_f32");
        printf("%s  (B = %ld rays)\n", argv[a], (long)B);
        const double sp_bytes = (double)B * (2 * Nc * 4 + 24 + Nt * 16);
        double t = time_us([&] { sp(dz, dw, du, dor, dd, B, Nc, Nf, nullptr, nullptr, zf, pts, nullptr); }, 200);
        printf("  sample_pdf 64+128            %8.2f us  %6.2f TB/s algorithmic (%.3f of 8)\n", t, sp_bytes / t * 1e-6, sp_bytes / t * 1e-6 / 8);
        for (int N : {64, 192}) {
            const double cbytes = (double)B * (N * 28.0 + 12 + 12);
            const float *r = N == 64 ? draw64 : draw192, *zz = N == 64 ? dz : dz192;
            t = time_us([&] { cf(r, zz, dd, 0, nullptr, B, N, 0, rgb, wts, alpha, nullptr); }, 200);
            printf("  composite_fwd N=%-3d          %8.2f us  %6.2f TB/s algorithmic (%.3f of 8)\n", N, t, cbytes / t * 1e-6, cbytes / t * 1e-6 / 8);
            const double bbytes = (double)B * (N * 36.0 + 24);
            t = time_us([&] { cb(r, zz, dd, 0, nullptr, B, N, 0, ddrgb, d_raw, nullptr, nullptr); }, 200);
            printf("  composite_bwd N=%-3d          %8.2f us  %6.2f TB/s algorithmic (%.3f of 8)\n", N, t, bbytes / t * 1e-6, bbytes / t * 1e-6 / 8);
        }
        dlclose(h);
    }
    return 0;
}
