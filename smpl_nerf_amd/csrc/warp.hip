// WarpFieldNet forward fused with its surroundings in SmplNerfPipeline (a7):
//   warp   = linear2(relu(linear1([PE(x) | PE(pose)])))            models/warp_field_net.py:17-21
//   x'     = x + warp                                               models/smpl_nerf_pipeline.py:48-49 / :77-79
//   sdir   = x' - o   (per-sample view direction)                   models/smpl_nerf_pipeline.py:52-53 / :82-83
// Same machinery as the RenderRayNet kernel (mlp_device.h): one wave owns 16 samples, the 100 -> 256
// layer runs on v_mfma_f32_16x16x4_f32 with the encoding evaluated in registers (the pose encoding is a
// per-ray constant read as "additional input"), the 256 -> 3 head is one padded tile.  52 736 FLOP per
// sample (4 % of a RenderRayNet evaluation); HBM: 12 B in, 36 B out per sample.
#include "mlp_device.h"

namespace snerf {

inline int make_warp_plan(const snerf_warp_desc &d, Plan &P, const char *&why) {
    why = "";
    if (d.width != 256 && d.width != 128) { why = "width must be 256 or 128"; return -1; }
    if (d.pos_freqs < 0 || d.pos_freqs > 16) { why = "bad encoder frequencies"; return -1; }
    if (d.pose_dim < 0 || d.pose_dim > 4096) { why = "bad pose_dim"; return -1; }
    const int pid = d.pos_identity ? 1 : 0;
    P.width = d.width;
    P.n_hidden = 0;
    P.pos_dim = 3 * (pid + 2 * d.pos_freqs);
    P.dir_dim = 0;
    P.add_dim = d.pose_dim;
    if (P.pos_dim + P.add_dim == 0) { why = "empty input"; return -1; }
    P.pos_nkb = pe_nkb(d.pos_freqs, pid);
    P.dir_nkb = 0;
    P.add_nkb = (d.pose_dim + 15) / 16;
    Layer &L0 = P.layer[0];
    L0.n_out = d.width;
    L0.t_out = d.width / 16;
    L0.nseg = 0;
    int col = 0;
    L0.seg[L0.nseg++] = Seg{SEG_PE, col, P.pos_dim, P.pos_nkb, d.pos_freqs, pid};
    col += P.pos_dim;
    if (P.add_dim) {
        L0.seg[L0.nseg++] = Seg{SEG_ADD, col, P.add_dim, P.add_nkb, 0, 0};
        col += P.add_dim;
    }
    L0.n_in = col;
    L0.nkb = P.pos_nkb + P.add_nkb;
    L0.first_slab = 0;
    L0.nslab = L0.nkb;  // t_out = 16 or 8 ...
    {
        const int kps = 16 / L0.t_out;
        L0.nslab = (L0.nkb + kps - 1) / kps;
    }
    L0.w_off = 0;
    L0.b_off = (int64_t)L0.n_out * L0.n_in;
    Layer &L1 = P.layer[1];
    L1.n_out = 3;
    L1.t_out = 1;
    L1.nseg = 1;
    L1.seg[0] = Seg{SEG_HIDDEN, 0, d.width, d.width / 16, 0, 0};
    L1.n_in = d.width;
    L1.nkb = d.width / 16;
    L1.first_slab = L0.nslab;
    L1.nslab = (L1.nkb + 15) / 16;
    L1.w_off = L0.b_off + L0.n_out;
    L1.b_off = L1.w_off + (int64_t)3 * d.width;
    P.nlayers = 2;
    P.total_slabs = L0.nslab + L1.nslab;
    P.param_floats = L1.b_off + 3;
    return 0;
}

struct WarpArgs {
    const float *packed;
    const float *x;     // [n,3] or null (encoded mode: every input column comes from `add`)
    const float *add;   // pose encoding [n/spr, add_dim]
    const float *o;     // [n/spr, 3] ray origins or null
    float *warp, *warped, *sdirs;  // [n,3] each, warped/sdirs nullable
    int64_t n;
    int spr;
    int pos_L, pos_id, pos_nkb, add_dim, add_nkb;
};

template <int WIDTH, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void warp_fwd_kernel(WarpArgs A) {
    constexpr int NT = NWAVES * 64;
    constexpr int T = WIDTH / 16;
    __shared__ __attribute__((aligned(16))) float ring[3 * SLAB_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t sample = ((int64_t)blockIdx.x * NWAVES + wave) * 16 + (lane & 15);
    const bool valid = sample < A.n;
    const int64_t sc = valid ? sample : A.n - 1;
    const int64_t ray = sc / A.spr;
    SampleCtx c;
    c.g = lane >> 4;
    c.enc = nullptr;
    c.px = c.py = c.pz = c.dx = c.dy = c.dz = 0.f;
    if (A.x) {
        c.px = A.x[sc * 3 + 0];
        c.py = A.x[sc * 3 + 1];
        c.pz = A.x[sc * 3 + 2];
    }
    c.add = A.add_dim ? A.add + ray * A.add_dim : nullptr;

    SlabPipe<NT> pipe;
    pipe.prologue(A.packed, ring, tid);
    f4 in[T], acc[T];
    {
        LayerRun<T, NT> run(pipe, lane);
        run.init(acc);
        for (int kb = 0; kb < A.pos_nkb; ++kb) run.step(pe_operand<false>(c, false, A.pos_L, A.pos_id, kb, 0), acc);
        for (int kb = 0; kb < A.add_nkb; ++kb) run.step(add_operand(c, A.add_dim, kb), acc);
        run.finish();
        relu_into(in, acc);
    }
    f4 w[1];
    {
        LayerRun<1, NT> run(pipe, lane);
        run.init(w);
#pragma unroll
        for (int kb = 0; kb < T; ++kb) run.step(in[kb], w);
        run.finish();
    }
    if (valid && c.g == 0) {
        float *wp = A.warp + sample * 3;
        wp[0] = w[0][0];
        wp[1] = w[0][1];
        wp[2] = w[0][2];
        if (A.warped) {
            const float wx = __fadd_rn(c.px, w[0][0]), wy = __fadd_rn(c.py, w[0][1]), wz = __fadd_rn(c.pz, w[0][2]);
            float *q = A.warped + sample * 3;
            q[0] = wx;
            q[1] = wy;
            q[2] = wz;
            if (A.sdirs) {
                const float *op = A.o + ray * 3;
                float *s = A.sdirs + sample * 3;
                s[0] = __fsub_rn(wx, op[0]);
                s[1] = __fsub_rn(wy, op[1]);
                s[2] = __fsub_rn(wz, op[2]);
            }
        }
    }
}

}  // namespace snerf

extern "C" int64_t snerf_warp_param_floats(const snerf_warp_desc *desc) {
    using namespace snerf;
    Plan P;
    const char *why;
    if (!desc || make_warp_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "warp: bad descriptor");
    return P.param_floats;
}

extern "C" int64_t snerf_warp_packed_floats(const snerf_warp_desc *desc) {
    using namespace snerf;
    Plan P;
    const char *why;
    if (!desc || make_warp_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "warp: bad descriptor");
    return (int64_t)(P.total_slabs + SLAB_PAD) * SLAB_FLOATS;
}

extern "C" int snerf_warp_pack_f32(const snerf_warp_desc *desc, const float *params_flat, float *packed,
                                   snerf_stream_t stream) {
    using namespace snerf;
    Plan P;
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "warp_pack: desc is null");
    if (make_warp_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "warp_pack: %s", why);
    if (!params_flat || !packed) return fail(SNERF_E_BADARG, "warp_pack: null pointer");
    if (!aligned(packed, 16)) return fail(SNERF_E_ALIGN, "warp_pack: packed must be 16-byte aligned");
    return launch_pack(P, params_flat, packed, (hipStream_t)stream, "warp_pack");
}

extern "C" int snerf_warp_fwd_f32(const snerf_warp_desc *desc, const float *packed, const float *x,
                                  const float *pose_enc, const float *o, int64_t n, int samples_per_ray,
                                  float *warp, float *warped, float *sdirs, snerf_stream_t stream) {
    using namespace snerf;
    Plan P;
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "warp_fwd: desc is null");
    if (make_warp_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "warp_fwd: %s", why);
    if (n < 0 || samples_per_ray < 1) return fail(SNERF_E_BADARG, "warp_fwd: bad n/samples_per_ray");
    if (n == 0) return SNERF_OK;
    if (!packed || !warp) return fail(SNERF_E_BADARG, "warp_fwd: null pointer");
    if (P.pos_dim && !x) return fail(SNERF_E_BADARG, "warp_fwd: x is null");
    if (P.add_dim && !pose_enc) return fail(SNERF_E_BADARG, "warp_fwd: pose_enc is null");
    if (warped && !x) return fail(SNERF_E_BADARG, "warp_fwd: warped requested without x");
    if (sdirs && (!warped || !o)) return fail(SNERF_E_BADARG, "warp_fwd: sdirs needs warped and o");
    if (!aligned(packed, 16)) return fail(SNERF_E_ALIGN, "warp_fwd: packed must be 16-byte aligned");
    WarpArgs A{};
    A.packed = packed;
    A.x = P.pos_dim ? x : (warped ? x : nullptr);
    A.add = pose_enc;
    A.o = o;
    A.warp = warp;
    A.warped = warped;
    A.sdirs = sdirs;
    A.n = n;
    A.spr = samples_per_ray;
    A.pos_L = desc->pos_freqs;
    A.pos_id = desc->pos_identity ? 1 : 0;
    A.pos_nkb = P.pos_nkb;
    A.add_dim = P.add_dim;
    A.add_nkb = P.add_nkb;
    constexpr int NW = 4;
    const int64_t grid = (n + NW * 16 - 1) / (NW * 16);
    if (grid > 0x7fffffffLL) return fail(SNERF_E_BADARG, "warp_fwd: n too large");
    if (P.width == 256)
        hipLaunchKernelGGL((warp_fwd_kernel<256, NW>), dim3((unsigned)grid), dim3(NW * 64), 0, (hipStream_t)stream, A);
    else
        hipLaunchKernelGGL((warp_fwd_kernel<128, NW>), dim3((unsigned)grid), dim3(NW * 64), 0, (hipStream_t)stream, A);
    return check_launch("warp_fwd");
}
