// snerf_render_rays_f32: NerfPipeline.forward (models/nerf_pipeline.py:14-67) for inference as one C-ABI call.
// Pure sequencing of the library's own entry points on the caller's stream; the workspace holds what the
// reference keeps as temporaries (raw, weights, z_fine) - [B*N, 84] encodings and [B*N, 256] activations never
// exist in HBM.
#include "snerf_common.h"

namespace snerf {

struct RenderWs {
    int64_t raw_c, weights_c, alpha_c, z_samples, z_fine, raw_f, total;
};
static int64_t align16(int64_t b) { return (b + 15) & ~int64_t(15); }
static RenderWs render_ws(int64_t B, int Nc, int Nf) {
    RenderWs w{};
    const int64_t N = Nc + Nf;
    int64_t off = 0;
    w.raw_c = off, off += align16(B * Nc * 4 * 4);
    w.weights_c = off, off += align16(B * Nc * 4);
    w.alpha_c = off, off += align16(B * Nc * 4);
    w.z_samples = off, off += align16(B * (Nf > 0 ? Nf : 1) * 4);
    w.z_fine = off, off += align16(B * N * 4);
    w.raw_f = off, off += align16(B * N * 4 * 4);
    w.total = off;
    return w;
}

// add: per-ray additional inputs [n / spr, add_dim] or null; fold_ws / fold_bytes: room for the per-ray fold of the fp32
// kernel (snerf_mlp_fold_workspace_bytes; 0 bytes = the per-sample form)
static int mlp(const snerf_mlp_desc *desc, const void *packed, int precision, const float *x, const float *dirs,
               int64_t n, int spr, float *raw, snerf_stream_t stream, const float *add = nullptr, void *fold_ws = nullptr,
               int64_t fold_bytes = 0) {
    if (precision == 0) {
        if (add && fold_bytes > 0)
            return snerf_mlp_fwd_ws_f32(desc, reinterpret_cast<const float *>(packed), x, dirs, 0, add, n, spr, raw, fold_ws, fold_bytes,
                                        stream);
        return snerf_mlp_fwd_f32(desc, reinterpret_cast<const float *>(packed), x, dirs, 0, add, n, spr, raw, stream);
    }
    return snerf_mlp_fwd_bf16_f32(desc, packed, precision, x, dirs, 0, add, n, spr, raw, stream);
}

static int mlp_per_sample_dirs(const snerf_mlp_desc *desc, const void *packed, int precision, const float *x,
                               const float *sdirs, int64_t n, int spr, float *raw, snerf_stream_t stream) {
    if (precision == 0)
        return snerf_mlp_fwd_f32(desc, reinterpret_cast<const float *>(packed), x, sdirs, 1, nullptr, n, spr, raw, stream);
    return snerf_mlp_fwd_bf16_f32(desc, packed, precision, x, sdirs, 1, nullptr, n, spr, raw, stream);
}
static int warp(const snerf_warp_desc *desc, const void *packed, int precision, const float *x, const float *pose_enc,
                const float *o, int64_t n, int spr, float *w, float *warped, float *sdirs, void *fold_ws, int64_t fold_bytes,
                snerf_stream_t stream) {
    if (precision == 0)
        return snerf_warp_fwd_ws_f32(desc, reinterpret_cast<const float *>(packed), x, pose_enc, o, n, spr, w, warped, sdirs,
                                     fold_ws, fold_bytes, stream);
    return snerf_warp_fwd_bf16_f32(desc, packed, x, pose_enc, o, n, spr, w, warped, sdirs, stream);
}

struct SmplWs {
    int64_t base, warp_c, warped_c, sdirs_c, sdirs_f, fold, fold_bytes, total;
};
static SmplWs smpl_ws(int64_t B, int Nc, int Nf) {
    SmplWs w{};
    const int64_t N = Nc + Nf;
    int64_t off = align16(render_ws(B, Nc, Nf).total);
    w.base = 0;
    w.warp_c = off, off += align16(B * Nc * 3 * 4);
    w.warped_c = off, off += align16(B * Nc * 3 * 4);
    w.sdirs_c = off, off += align16(B * Nc * 3 * 4);
    w.sdirs_f = off, off += align16(B * N * 3 * 4);
    w.fold_bytes = B * 256 * 4;   // per-ray pose fold of the fp32 warp kernel: one row of <= 256 floats per ray (warp.hip)
    w.fold = off, off += align16(w.fold_bytes);
    w.total = off;
    return w;
}

}  // namespace snerf

extern "C" int64_t snerf_render_rays_workspace_bytes(int64_t B, int Nc, int Nf) {
    using namespace snerf;
    if (B < 0 || Nc < 1 || Nf < 0) return fail(SNERF_E_BADARG, "render_rays_workspace_bytes: bad B/Nc/Nf");
    return render_ws(B, Nc, Nf).total;
}

namespace snerf {
static int render_rays_impl(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const snerf_mlp_desc *desc_fine,
                            const void *packed_fine, int precision, const float *ray_samples, const float *rays_o,
                            const float *rays_d, const float *z_vals, const float *additional, const float *u,
                            const float *noise_coarse, const float *noise_fine, int64_t B, int Nc, int Nf, int white_background,
                            void *workspace, float *rgb, float *rgb_fine, float *samples_fine, float *densities_fine,
                            snerf_stream_t stream);
// fold tables of the two passes of snerf_render_rays_add_f32 (they are not live at the same time: one region, the larger)
static int64_t add_fold_bytes(const snerf_mlp_desc *dc, const snerf_mlp_desc *df, int64_t B, int Nc, int Nf, int64_t &fc, int64_t &ff) {
    fc = snerf_mlp_fold_workspace_bytes(dc, B * Nc, Nc);
    ff = Nf > 0 ? snerf_mlp_fold_workspace_bytes(df, B * (Nc + Nf), Nc + Nf) : 0;
    if (fc < 0 || ff < 0) return -1;
    return fc > ff ? fc : ff;
}
}  // namespace snerf

extern "C" int snerf_render_rays_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse,
                                     const snerf_mlp_desc *desc_fine, const void *packed_fine, int precision,
                                     const float *ray_samples, const float *rays_o, const float *rays_d,
                                     const float *z_vals, const float *u, const float *noise_coarse,
                                     const float *noise_fine, int64_t B, int Nc, int Nf, int white_background,
                                     void *workspace, float *rgb, float *rgb_fine, float *samples_fine,
                                     float *densities_fine, snerf_stream_t stream) {
    using namespace snerf;
    if (desc_coarse && Nf >= 0 && ((desc_coarse->add_dim) || (Nf > 0 && desc_fine && desc_fine->add_dim)))
        return fail(SNERF_E_BADARG, "render_rays: nets with additional inputs go through snerf_render_rays_add_f32");
    return render_rays_impl(desc_coarse, packed_coarse, desc_fine, packed_fine, precision, ray_samples, rays_o, rays_d, z_vals, nullptr, u,
                            noise_coarse, noise_fine, B, Nc, Nf, white_background, workspace, rgb, rgb_fine, samples_fine,
                            densities_fine, stream);
}

extern "C" int64_t snerf_render_rays_add_workspace_bytes(const snerf_mlp_desc *desc_coarse, const snerf_mlp_desc *desc_fine, int64_t B,
                                                         int Nc, int Nf) {
    using namespace snerf;
    if (!desc_coarse || (Nf > 0 && !desc_fine) || B < 0 || Nc < 1 || Nf < 0)
        return fail(SNERF_E_BADARG, "render_rays_add_workspace_bytes: bad arguments");
    int64_t fc, ff;
    const int64_t fold = add_fold_bytes(desc_coarse, desc_fine, B, Nc, Nf, fc, ff);
    if (fold < 0) return SNERF_E_BADARG;
    return align16(render_ws(B, Nc, Nf).total) + align16(fold);
}

extern "C" int snerf_render_rays_add_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse,
                                         const snerf_mlp_desc *desc_fine, const void *packed_fine, int precision,
                                         const float *ray_samples, const float *rays_o, const float *rays_d, const float *z_vals,
                                         const float *additional, const float *u, const float *noise_coarse,
                                         const float *noise_fine, int64_t B, int Nc, int Nf, int white_background, void *workspace,
                                         float *rgb, float *rgb_fine, float *samples_fine, float *densities_fine,
                                         snerf_stream_t stream) {
    using namespace snerf;
    if (!desc_coarse || (Nf > 0 && !desc_fine)) return fail(SNERF_E_BADARG, "render_rays_add: null descriptor");
    if (!desc_coarse->add_dim || (Nf > 0 && desc_fine->add_dim != desc_coarse->add_dim))
        return fail(SNERF_E_BADARG, "render_rays_add: both nets must read the same per-ray additional inputs (add_dim > 0)");
    if (!additional && B > 0) return fail(SNERF_E_BADARG, "render_rays_add: additional is null");
    return render_rays_impl(desc_coarse, packed_coarse, desc_fine, packed_fine, precision, ray_samples, rays_o, rays_d, z_vals, additional,
                            u, noise_coarse, noise_fine, B, Nc, Nf, white_background, workspace, rgb, rgb_fine, samples_fine,
                            densities_fine, stream);
}

static int snerf::render_rays_impl(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const snerf_mlp_desc *desc_fine,
                                   const void *packed_fine, int precision, const float *ray_samples, const float *rays_o,
                                   const float *rays_d, const float *z_vals, const float *additional, const float *u,
                                   const float *noise_coarse, const float *noise_fine, int64_t B, int Nc, int Nf,
                                   int white_background, void *workspace, float *rgb, float *rgb_fine, float *samples_fine,
                                   float *densities_fine, snerf_stream_t stream) {
    using namespace snerf;
    if (precision != 0 && precision != 2 && precision != 3 && precision != SNERF_SPLIT_F16X3)
        return fail(SNERF_E_BADARG, "render_rays: precision must be 0 (fp32), 2 (bf16x3), 3 (bf16x6) or 16 (f16x3)");
    if (B < 0 || Nc < 1 || Nf < 0) return fail(SNERF_E_BADARG, "render_rays: bad B/Nc/Nf");
    if (B == 0) return SNERF_OK;
    if (!desc_coarse || !packed_coarse || !ray_samples || !rays_d || !z_vals || !workspace || !rgb || !rgb_fine ||
        !samples_fine || !densities_fine)
        return fail(SNERF_E_BADARG, "render_rays: null pointer");
    if (Nf > 0 && (!desc_fine || !packed_fine || !rays_o || !u))
        return fail(SNERF_E_BADARG, "render_rays: the fine pass needs desc_fine, packed_fine, rays_o and u");
    if (!aligned(workspace, 16)) return fail(SNERF_E_ALIGN, "render_rays: workspace must be 16-byte aligned");
    const RenderWs w = render_ws(B, Nc, Nf);
    char *ws = reinterpret_cast<char *>(workspace);
    int64_t fold_c = 0, fold_f = 0;
    void *fold_ws = nullptr;
    if (additional) {
        if (add_fold_bytes(desc_coarse, desc_fine, B, Nc, Nf, fold_c, fold_f) < 0) return SNERF_E_BADARG;
        fold_ws = ws + align16(w.total);
    }
    float *raw_c = reinterpret_cast<float *>(ws + w.raw_c), *weights_c = reinterpret_cast<float *>(ws + w.weights_c);
    float *alpha_c = reinterpret_cast<float *>(ws + w.alpha_c), *z_samples = reinterpret_cast<float *>(ws + w.z_samples);
    float *z_fine = reinterpret_cast<float *>(ws + w.z_fine), *raw_f = reinterpret_cast<float *>(ws + w.raw_f);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // coarse net on the given samples, then compositing (:29-42)
    if ((rc = mlp(desc_coarse, packed_coarse, precision, ray_samples, rays_d, B * Nc, Nc, raw_c, stream, additional, fold_ws, fold_c)))
        return rc;
    if (Nf == 0) {  // run_fine = 0 (:43-44): (rgb, rgb, ray_samples, alpha)
        if ((rc = snerf_composite_fwd_f32(raw_c, z_vals, rays_d, 0, noise_coarse, B, Nc, white_background, rgb, weights_c,
                                          densities_fine, stream)))
            return rc;
        if (hipMemcpyAsync(rgb_fine, rgb, B * 3 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess ||
            hipMemcpyAsync(samples_fine, ray_samples, B * Nc * 3 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
            return fail(SNERF_E_LAUNCH, "render_rays: device copy failed");
        return SNERF_OK;
    }
    if ((rc = snerf_composite_fwd_f32(raw_c, z_vals, rays_d, 0, noise_coarse, B, Nc, white_background, rgb, weights_c,
                                      alpha_c, stream)))
        return rc;
    // hierarchical samples (:47) and the fine net on them (:49-65)
    if ((rc = snerf_sample_pdf_f32(z_vals, weights_c, u, rays_o, rays_d, B, Nc, Nf, nullptr, z_samples, z_fine,
                                   samples_fine, stream)))
        return rc;
    const int N = Nc + Nf;
    if ((rc = mlp(desc_fine, packed_fine, precision, samples_fine, rays_d, B * N, N, raw_f, stream, additional, fold_ws, fold_f))) return rc;
    return snerf_composite_fwd_f32(raw_f, z_fine, rays_d, 0, noise_fine, B, N, white_background, rgb_fine, nullptr,
                                   densities_fine, stream);
}

extern "C" int64_t snerf_render_rays_smpl_workspace_bytes(int64_t B, int Nc, int Nf) {
    using namespace snerf;
    if (B < 0 || Nc < 1 || Nf < 1) return fail(SNERF_E_BADARG, "render_rays_smpl_workspace_bytes: bad B/Nc/Nf");
    return smpl_ws(B, Nc, Nf).total;
}

extern "C" int snerf_render_rays_smpl_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse,
                                          const snerf_mlp_desc *desc_fine, const void *packed_fine,
                                          const snerf_warp_desc *desc_warp, const void *packed_warp, int precision,
                                          const float *ray_samples, const float *rays_o, const float *rays_d,
                                          const float *z_vals, const float *pose_enc, const float *u,
                                          const float *noise_coarse, const float *noise_fine, int64_t B, int Nc, int Nf,
                                          int white_background, void *workspace, float *rgb, float *rgb_fine,
                                          float *warp_fine, float *samples_fine, float *warped_fine,
                                          float *densities_fine, snerf_stream_t stream) {
    using namespace snerf;
    if (precision != 0 && precision != 2 && precision != 3 && precision != SNERF_SPLIT_F16X3)
        return fail(SNERF_E_BADARG, "render_rays_smpl: precision must be 0 (fp32), 2 (bf16x3), 3 (bf16x6) or 16 (f16x3)");
    if (B < 0 || Nc < 1 || Nf < 1) return fail(SNERF_E_BADARG, "render_rays_smpl: bad B/Nc/Nf");
    if (B == 0) return SNERF_OK;
    if (!desc_coarse || !packed_coarse || !desc_fine || !packed_fine || !desc_warp || !packed_warp || !ray_samples ||
        !rays_o || !rays_d || !z_vals || !pose_enc || !u || !workspace || !rgb || !rgb_fine || !warp_fine || !samples_fine ||
        !warped_fine || !densities_fine)
        return fail(SNERF_E_BADARG, "render_rays_smpl: null pointer");
    if (desc_coarse->add_dim || desc_fine->add_dim)
        return fail(SNERF_E_BADARG, "render_rays_smpl: nets with additional inputs go through snerf_mlp_fwd_* directly");
    if (!aligned(workspace, 16)) return fail(SNERF_E_ALIGN, "render_rays_smpl: workspace must be 16-byte aligned");
    const RenderWs w = render_ws(B, Nc, Nf);
    const SmplWs sw = smpl_ws(B, Nc, Nf);
    char *ws = reinterpret_cast<char *>(workspace);
    auto f = [&](int64_t off) { return reinterpret_cast<float *>(ws + off); };
    float *raw_c = f(w.raw_c), *weights_c = f(w.weights_c), *alpha_c = f(w.alpha_c), *z_samples = f(w.z_samples);
    float *z_fine = f(w.z_fine), *raw_f = f(w.raw_f);
    float *warp_c = f(sw.warp_c), *warped_c = f(sw.warped_c), *sdirs_c = f(sw.sdirs_c), *sdirs_f = f(sw.sdirs_f);
    const int N = Nc + Nf;
    int rc;
    // coarse: warp the given samples, net on (x', x' - o), compositing scaled per sample (:38-63)
    if ((rc = warp(desc_warp, packed_warp, precision, ray_samples, pose_enc, rays_o, B * Nc, Nc, warp_c, warped_c, sdirs_c, ws + sw.fold,
                   sw.fold_bytes, stream)))
        return rc;
    if ((rc = mlp_per_sample_dirs(desc_coarse, packed_coarse, precision, warped_c, sdirs_c, B * Nc, Nc, raw_c, stream))) return rc;
    if ((rc = snerf_composite_fwd_f32(raw_c, z_vals, sdirs_c, 1, noise_coarse, B, Nc, white_background, rgb, weights_c, alpha_c,
                                      stream)))
        return rc;
    // hierarchical samples on the un-warped ray (:68), then the fine stage (:71-98)
    if ((rc = snerf_sample_pdf_f32(z_vals, weights_c, u, rays_o, rays_d, B, Nc, Nf, nullptr, z_samples, z_fine, samples_fine,
                                   stream)))
        return rc;
    if ((rc = warp(desc_warp, packed_warp, precision, samples_fine, pose_enc, rays_o, B * N, N, warp_fine, warped_fine, sdirs_f,
                   ws + sw.fold, sw.fold_bytes, stream)))
        return rc;
    if ((rc = mlp_per_sample_dirs(desc_fine, packed_fine, precision, warped_fine, sdirs_f, B * N, N, raw_f, stream))) return rc;
    return snerf_composite_fwd_f32(raw_f, z_fine, rays_d, 0, noise_fine, B, N, white_background, rgb_fine, nullptr,
                                   densities_fine, stream);
}
