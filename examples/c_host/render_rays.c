/* A non-Python host of libsmplnerf_hip.so: renders a batch of rays through snerf_render_rays_f32 (the whole
 * NerfPipeline.forward, models/nerf_pipeline.py:14-67) using nothing but the C-ABI of include/smplnerf.h and the
 * HIP runtime for device memory.  Plain C99.
 *
 *   render_rays <in.bin> <out.bin> <precision: 0 | 2 | 3 | 16 (SNERF_SPLIT_F16X3)>
 *
 * in.bin  (little endian): int32 magic 0x534e5246, int32 B, Nc, Nf, white_background, then two snerf_mlp_desc
 *          (coarse, fine; 10 x int32 each), then float32 arrays: params_coarse, params_fine (state_dict order,
 *          snerf_mlp_param_floats each), ray_samples [B,Nc,3], rays_o [B,3], rays_d [B,3], z_vals [B,Nc], u [Nf]
 * out.bin: float32 rgb [B,3], rgb_fine [B,3], samples_fine [B,Nc+Nf,3], densities_fine [B,Nc+Nf]
 *
 * Build (tests/test_gpu_parity.py::test_c_host_example does exactly this):
 *   gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_host/render_rays.c \
 *       smpl_nerf_amd/csrc/libsmplnerf_hip.so -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o render_rays
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "smplnerf.h"

#define HIP_OK(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));              \
            return 2;                                                                      \
        }                                                                                  \
    } while (0)
#define SNERF_OK_OR_DIE(call)                                                              \
    do {                                                                                   \
        int rc_ = (call);                                                                  \
        if (rc_ != SNERF_OK) {                                                             \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, snerf_last_error_string()); \
            return 3;                                                                      \
        }                                                                                  \
    } while (0)

static float *to_device(FILE *f, size_t count, int *ok) {
    float *host = (float *)malloc(count * sizeof(float)), *dev = NULL;
    if (!host || fread(host, sizeof(float), count, f) != count) *ok = 0;
    if (*ok && hipMalloc((void **)&dev, count * sizeof(float)) != hipSuccess) *ok = 0;
    if (*ok && hipMemcpy(dev, host, count * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) *ok = 0;
    free(host);
    return dev;
}

int main(int argc, char **argv) {
    if (argc != 4) {
        fprintf(stderr, "usage: %s in.bin out.bin precision(0|2|3|16)\n", argv[0]);
        return 1;
    }
    const int precision = atoi(argv[3]);
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    int32_t head[5];
    snerf_mlp_desc desc[2];
    if (fread(head, sizeof(int32_t), 5, f) != 5 || head[0] != 0x534e5246 || fread(desc, sizeof(snerf_mlp_desc), 2, f) != 2) return 1;
    const int64_t B = head[1];
    const int Nc = head[2], Nf = head[3], white = head[4], N = Nc + Nf;
    if (snerf_device_count() < 1) {
        fprintf(stderr, "no HIP device\n");
        return 1;
    }
    int ok = 1;
    float *params[2];
    void *packed[2];
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    for (int k = 0; k < 2; ++k) {
        const int64_t np = snerf_mlp_param_floats(&desc[k]);
        if (np < 0) return 3;
        params[k] = to_device(f, (size_t)np, &ok);
        if (!ok) return 2;
        if (precision == 0) {
            HIP_OK(hipMalloc(&packed[k], (size_t)snerf_mlp_packed_floats(&desc[k]) * sizeof(float)));
            SNERF_OK_OR_DIE(snerf_mlp_pack_f32(&desc[k], params[k], (float *)packed[k], stream));
        } else {
            HIP_OK(hipMalloc(&packed[k], (size_t)snerf_mlp_packed_bf16_bytes(&desc[k], precision)));
            SNERF_OK_OR_DIE(snerf_mlp_pack_bf16(&desc[k], params[k], packed[k], precision, stream));
        }
    }
    float *ray_samples = to_device(f, (size_t)(B * Nc * 3), &ok), *rays_o = to_device(f, (size_t)(B * 3), &ok);
    float *rays_d = to_device(f, (size_t)(B * 3), &ok), *z_vals = to_device(f, (size_t)(B * Nc), &ok);
    float *u = Nf ? to_device(f, (size_t)Nf, &ok) : NULL;
    fclose(f);
    if (!ok) return 2;

    const size_t n_out = (size_t)(B * 3 + B * 3 + B * N * 3 + B * N);
    float *out_dev, *out_host = (float *)malloc(n_out * sizeof(float));
    void *workspace;
    HIP_OK(hipMalloc((void **)&out_dev, n_out * sizeof(float)));
    HIP_OK(hipMalloc(&workspace, (size_t)snerf_render_rays_workspace_bytes(B, Nc, Nf)));
    float *rgb = out_dev, *rgb_fine = rgb + B * 3, *samples_fine = rgb_fine + B * 3, *dens = samples_fine + B * N * 3;
    SNERF_OK_OR_DIE(snerf_render_rays_f32(&desc[0], packed[0], &desc[1], packed[1], precision, ray_samples, rays_o, rays_d,
                                          z_vals, u, NULL, NULL, B, Nc, Nf, white, workspace, rgb, rgb_fine, samples_fine,
                                          dens, stream));
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(out_host, out_dev, n_out * sizeof(float), hipMemcpyDeviceToHost));
    FILE *g = fopen(argv[2], "wb");
    if (!g || fwrite(out_host, sizeof(float), n_out, g) != n_out) return 1;
    fclose(g);
    printf("rendered %lld rays (%d + %d samples), library version %d\n", (long long)B, Nc, Nf, snerf_version());
    return 0;
}
