// WarpFieldNet forward (warp.hip) on the bf16 matrix cores: the 100 -> 256 -> 3 net with split-bf16 operands (always
// three parts = six products: the warp moves the sample BEFORE the 2^9 band of the position encoding, so it has to be
// fp32-class accurate whatever mode the RenderRayNet kernels run in), fused with x' = x + warp and sdir = x' - o.
// Machinery of mlp_bf16.hip (mlp_bf16_device.h): persistent workgroups, LDS-DMA weight ring, hand-laid k-block stream.
#include <stdlib.h>

#include "mlp_bf16_device.h"
#include "warp_plan.h"

namespace snerf {

constexpr int WNS = 3;

template <int WIDTH, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void warp_fwd_bf16_kernel(WarpArgs A, int total_slabs, int64_t n_tiles) {
    constexpr int NT = NWAVES * 64;
    constexpr int T = WIDTH / 16;
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SlabPipe16<NT, WNS> pipe;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t sample = (tile * NWAVES + wave) * 16 + (lane & 15);
        const bool valid = sample < A.n;
        const int64_t sc = valid ? sample : A.n - 1;
        const int64_t ray = sc / A.spr;
        SampleCtx c;
        c.g = lane >> 4;
        c.enc = nullptr;
        c.dx = c.dy = c.dz = 0.f;
        c.px = A.x[sc * 3 + 0];
        c.py = A.x[sc * 3 + 1];
        c.pz = A.x[sc * 3 + 2];
        c.add = A.add_dim ? A.add + ray * A.add_dim : nullptr;
        asm volatile("" ::"v"(c.px), "v"(c.py), "v"(c.pz));  // retire the loads before the weight DMA (cf. mlp_fwd_bf16_kernel)
        if (tile == blockIdx.x) pipe.prologue(A.packed, ring, tid, total_slabs);

        f4 acc[T];
        {  // linear1 + relu (its relu is applied when the head splits acc)
            LayerRun16<T, NT, WNS> run(pipe, lane);
            run.init(acc);
            for (int kb = 0; kb < A.pos_nkb; ++kb)
                run.step_make([&](bf8(&b)[WNS]) __attribute__((always_inline)) {
                    f4 half[2];
                    pe_operand16<WNS>(c, false, A.pos_L, A.pos_id, kb, b, half);
                }, acc);
            for (int kb = 0; kb < A.add_nkb; ++kb)
                run.step_make([&](bf8(&b)[WNS]) __attribute__((always_inline)) {
                    f4 half[2];
                    add_operand16<WNS>(c, A.add_dim, kb, b, half);
                }, acc);
            run.finish();
        }
        f4 w[1];
        {  // linear2: one padded tile, rows 0..2
            LayerRun16<1, NT, WNS> run(pipe, lane);
            run.init(w);
            run.template run_hidden<true>(acc, w);
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
    if (blockIdx.x < n_tiles) wait_pair<WNS, 0>(pipe.fa0, pipe.fa1);
}

static int warp_plan32(const snerf_warp_desc *desc, Plan &P, const char *what) {
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "%s: desc is null", what);
    if (make_warp_plan(*desc, P, why, 32) != 0) return fail(SNERF_E_BADARG, "%s: %s", what, why);
    if (desc->width != 256) return fail(SNERF_E_BADARG, "%s: the split-bf16 path supports width 256 only", what);
    return SNERF_OK;
}

}  // namespace snerf

extern "C" int64_t snerf_warp_packed_bf16_bytes(const snerf_warp_desc *desc) {
    using namespace snerf;
    Plan P;
    int rc = warp_plan32(desc, P, "warp_packed_bf16_bytes");
    if (rc) return rc;
    return (int64_t)(P.total_slabs + SLAB_PAD) * slab16_bytes(WNS);
}

extern "C" int snerf_warp_pack_bf16(const snerf_warp_desc *desc, const float *params_flat, void *packed,
                                    snerf_stream_t stream) {
    using namespace snerf;
    Plan P;
    int rc = warp_plan32(desc, P, "warp_pack_bf16");
    if (rc) return rc;
    if (!params_flat || !packed) return fail(SNERF_E_BADARG, "warp_pack_bf16: null pointer");
    if (!aligned(packed, 16)) return fail(SNERF_E_ALIGN, "warp_pack_bf16: packed must be 16-byte aligned");
    return launch_pack_bf16(P, WNS, params_flat, packed, (hipStream_t)stream, "warp_pack_bf16");
}

extern "C" int snerf_warp_fwd_bf16_f32(const snerf_warp_desc *desc, const void *packed, const float *x,
                                       const float *pose_enc, const float *o, int64_t n, int samples_per_ray,
                                       float *warp, float *warped, float *sdirs, snerf_stream_t stream) {
    using namespace snerf;
    Plan P;
    int rc = warp_plan32(desc, P, "warp_fwd_bf16");
    if (rc) return rc;
    if (n < 0 || samples_per_ray < 1) return fail(SNERF_E_BADARG, "warp_fwd_bf16: bad n/samples_per_ray");
    if (n == 0) return SNERF_OK;
    if (!packed || !x || !warp) return fail(SNERF_E_BADARG, "warp_fwd_bf16: null pointer");
    if (P.add_dim && !pose_enc) return fail(SNERF_E_BADARG, "warp_fwd_bf16: pose_enc is null");
    if (sdirs && (!warped || !o)) return fail(SNERF_E_BADARG, "warp_fwd_bf16: sdirs needs warped and o");
    if (!aligned(packed, 16)) return fail(SNERF_E_ALIGN, "warp_fwd_bf16: packed must be 16-byte aligned");
    WarpArgs A{};
    A.packed = reinterpret_cast<const float *>(packed);
    A.x = x;
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
    constexpr int NW = 8;
    const int lds = 3 * slab16_bytes(WNS);
    static LdsRaised raised;   // per device
    if ((rc = raise_dynamic_lds(reinterpret_cast<const void *>(warp_fwd_bf16_kernel<256, NW>), lds, raised, "warp_fwd_bf16"))) return rc;
    const int n_cu = device_cu_count("warp_fwd_bf16");
    if (n_cu < 1) return n_cu;
    const int64_t n_tiles = (n + NW * 16 - 1) / (NW * 16);
    const int64_t grid = n_tiles < n_cu ? n_tiles : n_cu;
    hipLaunchKernelGGL((warp_fwd_bf16_kernel<256, NW>), dim3((unsigned)grid), dim3(NW * 64), lds, (hipStream_t)stream, A,
                       P.total_slabs, n_tiles);
    return check_launch("warp_fwd_bf16");
}
