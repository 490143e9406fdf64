/* A non-Python host of the training path: K steps of NerfSolver.train's per-batch body (solver/nerf_solver.py:76-87 - forward,
 * MSE coarse + fine, backward, Adam) through snerf_nerf_train_step_f32, using nothing but the C-ABI of include/smplnerf.h and
 * the HIP runtime for device memory.  Plain C99.  What a host owns is visible here in full: one flat parameter buffer, a
 * flat gradient buffer, Adam's two moment buffers and a step counter per parameter tensor, the two weight streams and the two
 * slot tables per net (packed / built once - the optimiser step keeps the streams current), one workspace.
 *
 *   train_steps <in.bin> <out.bin> <steps> <rays_per_chunk>
 *
 * in.bin  (little endian): int32 magic 0x534e5254, int32 B, Nc, Nf, white_background, float64 lr, then two snerf_mlp_desc
 *          (coarse, fine; 10 x int32 each), then float32 arrays: params_coarse, params_fine (state_dict order), ray_samples
 *          [B,Nc,3], rays_o [B,3], rays_d [B,3], z_vals [B,Nc], rgb_truth [B,3], u [Nf]
 * out.bin: float32 loss [steps], then the parameters after the last step (coarse, fine)
 *
 * Build (tests/test_gpu_round4.py::test_c_host_training_example does exactly this):
 *   gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_host/train_steps.c \
 *       smpl_nerf_amd/csrc/libsmplnerf_hip.so -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o train_steps
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

static int read_to_device(FILE *f, float *dev, size_t count) {
    float *host = (float *)malloc(count * sizeof(float));
    int ok = host && fread(host, sizeof(float), count, f) == count &&
             hipMemcpy(dev, host, count * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    free(host);
    return ok;
}

int main(int argc, char **argv) {
    if (argc != 5) {
        fprintf(stderr, "usage: %s in.bin out.bin steps rays_per_chunk\n", argv[0]);
        return 1;
    }
    const int steps = atoi(argv[3]);
    const int64_t rays_per_chunk = atoll(argv[4]);
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    int32_t head[5];
    double lr;
    snerf_mlp_desc desc[2];
    if (fread(head, sizeof(int32_t), 5, f) != 5 || head[0] != 0x534e5254 || fread(&lr, sizeof(double), 1, f) != 1 ||
        fread(desc, sizeof(snerf_mlp_desc), 2, f) != 2)
        return 1;
    const int64_t B = head[1];
    const int Nc = head[2], Nf = head[3], wb = head[4], N = Nc + Nf;
    const int64_t np[2] = {snerf_mlp_param_floats(&desc[0]), snerf_mlp_param_floats(&desc[1])};
    if (np[0] < 0 || np[1] < 0) return 3;
    const int64_t n_params = np[0] + np[1];
    hipStream_t stream, aux;
    HIP_OK(hipStreamCreate(&stream));
    HIP_OK(hipStreamCreate(&aux));

    /* flat buffers: parameters, gradients, Adam moments; one step counter per parameter TENSOR (torch's state[p]["step"]) -
     * a RenderRayNet of n_layers has 2 * (n_layers + 5) tensors; all tensors step together here, so one range covers them */
    float *params, *grads, *m, *v, *scratch;
    int64_t *step;
    const int n_tensors = 2 * (desc[0].n_layers + 5) + 2 * (desc[1].n_layers + 5);
    HIP_OK(hipMalloc((void **)&params, n_params * sizeof(float)));
    HIP_OK(hipMalloc((void **)&grads, n_params * sizeof(float)));
    HIP_OK(hipMalloc((void **)&m, n_params * sizeof(float)));
    HIP_OK(hipMalloc((void **)&v, n_params * sizeof(float)));
    HIP_OK(hipMalloc((void **)&scratch, 64 * sizeof(float)));
    HIP_OK(hipMalloc((void **)&step, n_tensors * sizeof(int64_t)));
    HIP_OK(hipMemset(m, 0, n_params * sizeof(float)));
    HIP_OK(hipMemset(v, 0, n_params * sizeof(float)));
    HIP_OK(hipMemset(step, 0, n_tensors * sizeof(int64_t)));
    if (!read_to_device(f, params, (size_t)n_params)) return 1;

    /* the batch */
    float *samples, *o, *d, *z, *gt, *u;
    HIP_OK(hipMalloc((void **)&samples, B * Nc * 3 * sizeof(float)));
    HIP_OK(hipMalloc((void **)&o, B * 3 * sizeof(float)));
    HIP_OK(hipMalloc((void **)&d, B * 3 * sizeof(float)));
    HIP_OK(hipMalloc((void **)&z, B * Nc * sizeof(float)));
    HIP_OK(hipMalloc((void **)&gt, B * 3 * sizeof(float)));
    HIP_OK(hipMalloc((void **)&u, (Nf > 0 ? Nf : 1) * sizeof(float)));
    if (!read_to_device(f, samples, (size_t)(B * Nc * 3)) || !read_to_device(f, o, (size_t)(B * 3)) ||
        !read_to_device(f, d, (size_t)(B * 3)) || !read_to_device(f, z, (size_t)(B * Nc)) || !read_to_device(f, gt, (size_t)(B * 3)) ||
        (Nf > 0 && !read_to_device(f, u, (size_t)Nf)))
        return 1;
    fclose(f);

    /* weight streams (packed ONCE) and slot tables per net */
    snerf_adam_net nets[2];
    for (int k = 0; k < 2; ++k) {
        const float *pk = params + (k ? np[0] : 0);
        int64_t packed_t_floats = 0;
        float *packed, *packed_t;
        int32_t *slot_fwd, *slot_t;
        SNERF_OK_OR_DIE(snerf_mlp_train_sizes(&desc[k], 0, NULL, NULL, &packed_t_floats, NULL, NULL));
        HIP_OK(hipMalloc((void **)&packed, snerf_mlp_packed_floats(&desc[k]) * sizeof(float)));
        HIP_OK(hipMalloc((void **)&packed_t, packed_t_floats * sizeof(float)));
        HIP_OK(hipMalloc((void **)&slot_fwd, np[k] * sizeof(int32_t)));
        HIP_OK(hipMalloc((void **)&slot_t, np[k] * sizeof(int32_t)));
        SNERF_OK_OR_DIE(snerf_mlp_pack_f32(&desc[k], pk, packed, stream));
        SNERF_OK_OR_DIE(snerf_mlp_pack_t_f32(&desc[k], pk, packed_t, 0, stream));
        SNERF_OK_OR_DIE(snerf_mlp_stream_slots(&desc[k], slot_fwd, slot_t, 0, stream));
        nets[k].desc = &desc[k];
        nets[k].param_offset = k ? np[0] : 0;
        nets[k].precision = 0;
        nets[k].packed = packed;
        nets[k].packed_t = packed_t;
        nets[k].slot_fwd = slot_fwd;
        nets[k].slot_t = slot_t;
    }

    const int64_t ws_bytes = snerf_nerf_train_workspace_bytes(&desc[0], &desc[1], B, Nc, Nf, rays_per_chunk);
    if (ws_bytes < 0) return 3;
    void *ws;
    float *out, *loss_host = (float *)malloc((size_t)steps * sizeof(float));
    HIP_OK(hipMalloc(&ws, (size_t)ws_bytes));
    HIP_OK(hipMalloc((void **)&out, (size_t)(4 * steps + 6 * B) * sizeof(float))); /* loss[3] per step (padded to 4), rgb, rgb_fine */
    float *rgb = out + 4 * steps, *rgb_fine = rgb + 3 * B;

    snerf_nerf_batch batch = {samples, o, d, z, gt, u, NULL, NULL, NULL, B, Nc, Nf, wb};
    snerf_adam_state adam = {params, grads, m, v, n_params, scratch, lr, 0.9, 0.999, 1e-8, 0.0}; /* solver/nerf_solver.py:11-14 */
    snerf_adam_range range = {0, n_params, step, n_tensors};
    (void)N;
    for (int k = 0; k < steps; ++k)
        SNERF_OK_OR_DIE(snerf_nerf_train_step_f32(&desc[0], nets[0].packed, nets[0].packed_t, &desc[1], nets[1].packed,
                                                  nets[1].packed_t, 0, &batch, rays_per_chunk, ws, grads, grads + np[0],
                                                  out + 4 * k, rgb, rgb_fine, &adam, &range, 1, nets, 2, stream, aux));
    HIP_OK(hipStreamSynchronize(stream));

    float *host = (float *)malloc((size_t)(4 * steps) * sizeof(float)), *p_host = (float *)malloc((size_t)n_params * sizeof(float));
    HIP_OK(hipMemcpy(host, out, (size_t)(4 * steps) * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(p_host, params, (size_t)n_params * sizeof(float), hipMemcpyDeviceToHost));
    for (int k = 0; k < steps; ++k) loss_host[k] = host[4 * k];
    FILE *g = fopen(argv[2], "wb");
    if (!g || fwrite(loss_host, sizeof(float), (size_t)steps, g) != (size_t)steps ||
        fwrite(p_host, sizeof(float), (size_t)n_params, g) != (size_t)n_params)
        return 1;
    fclose(g);
    return 0;
}
