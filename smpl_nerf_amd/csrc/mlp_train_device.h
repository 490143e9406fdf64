// Shared between the fp32 dgrad kernel (mlp_train.hip) and the split-bf16 one (mlp_train_bf16.hip): kernel
// arguments and the backward of the positional encoding.
#pragma once
#include "mlp_device.h"

namespace snerf {

struct BwdArgs {
    const float *packed_t;
    const float *act;
    const float *d_raw;  // [n,4]
    float *dy;
    int64_t n;
    int n_hidden;
    int act_x1, act_h2, act_mask;
    int dy_rows;       // f16x3 dgrad: the per-layer |dY| exponents go behind this many tile-rows of `dy`
    int dy_sig, dy_din, dy_dn0, dy_rgb;  // dy of forward layer l <= nh+1 is l*T
    // input gradients (INPUT_GRAD kernels only)
    const float *x, *dirs;  // forward inputs: positions [n,3], directions [n/spr,3] or [n,3]
    float *d_x, *d_dirs;    // [n,3] each: d loss / d position, d loss / d (un-normalised) direction
    int dirs_per_sample, spr;
    unsigned skip_mask;
    int pos_L, pos_id, pos_nkb, dir_L, dir_id, dir_nkb, use_dir;
    int total_slabs;   // split-bf16 kernel: slabs of the transposed stream (persistent workgroups wrap around)
    int64_t n_tiles;   // split-bf16 kernel: 128-sample tiles
};

// Backward of the positional encoding (utils.py:127-131) for the slots this lane holds: dpe[kb][2u], [2u+1] are
// the gradients of (first, second) of unit p = 4*(2kb+u)+g.  Adds this lane's share to (dx, dy, dz).
template <int NKB>
__device__ __forceinline__ void pe_backward(const f4 (&dpe)[NKB], int nkb, float x, float y, float z, int L, int ident,
                                            int g, float &dx, float &dy, float &dz) {
    const int nid = ident ? 3 : 0;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        if (kb >= nkb) break;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = 4 * (2 * kb + u) + g;
            const float d0 = dpe[kb][2 * u], d1 = dpe[kb][2 * u + 1];
            float val = 0.f;
            int c = -1;
            if (p < nid) {
                c = p;
                val = d0;
            } else if (p - nid < 3 * L) {
                const int pp = p - nid, k = pp / 3;
                c = pp - 3 * k;
                const float v = c == 0 ? x : (c == 1 ? y : z);
                float sn, cs;
                sincosf(ldexpf(v, k), &sn, &cs);
                val = ldexpf(cs * d0 - sn * d1, k);  // d/dv sin(2^k v) = 2^k cos, d/dv cos(2^k v) = -2^k sin
            }
            dx += c == 0 ? val : 0.f;
            dy += c == 1 ? val : 0.f;
            dz += c == 2 ? val : 0.f;
        }
    }
}
__device__ __forceinline__ float sum_over_g(float v) {  // the 4 lanes (g = 0..3) that share a sample
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

struct WgradArgs {
    const float *act;
    const float *dy;
    float *part;  // [G][gp_floats]
    int64_t n;
    int64_t chunk;  // samples per K-split, multiple of 16
    const int *xstat, *ystat;  // f16x3 wide wgrad: exponents of the largest |X| entering / |dY| leaving forward layer l
    int fold;                  // fp32 step: narrow pairs that share an operand with a wide job ride with it (wgrad_kind)
};


// A (layer, input segment) pair is "wide" when it fills the 16-wave workgroup of mlp_wgrad_kernel with real work
// (the 256x256 layers and the 128x256 one); the narrow pairs - encoder columns, heads, the 128x128 layer: 15 % of
// the FLOPs - go to mlp_wgrad_direct_kernel, whose independent single-wave workgroups have no per-stage barrier.
// (Measured: sending the narrow pairs with >= 8 output tiles through mlp_wgrad_kernel's LDS stages instead, which
// would read dY and X from HBM once per job rather than 2-4 times, is 0.3 ms per 10^6 samples SLOWER - the re-reads
// of concurrently running blocks hit L2.)
__host__ __device__ inline bool wgrad_wide(const Layer &Ly, int s) { return Ly.t_out >= 8 && Ly.seg[s].nkb >= 16; }
// Folding (fp32 steps).  The narrow pairs are bound by HBM, not by the matrix pipe: a single-wave 4x4-tile job needs 2 KB of
// operands per 16 MFMAs and the 20 narrow jobs of the default net move 8.7 KB per sample - while most of those bytes are
// ALREADY staged in LDS by a wide job: the direction-encoding columns of directional_input contract the same d Y rows as
// the layer's 256-column job (only 2 more X tile-rows), and the sigma head contracts the same X rows (`o`) as that job (one
// more d Y tile-row).  mlp_wgrad_kernel computes those as extra accumulator tiles of that (half-length) job: no extra d Y / X
// traffic beyond the few extra rows, no extra barrier - and 6 of the 20 narrow jobs (2 KB per sample) disappear.
//   xseg fold: the narrow segment (<= 4 k-blocks) of directional_input, which also has a wide segment -> extra X rows
//   sigma fold: the 1-row sigma head -> extra d Y row of directional_input's hidden-segment job (same X rows, same width)
__host__ __device__ inline int wgrad_first_wide_seg(const Layer &Ly) {
    for (int s = 0; s < Ly.nseg; ++s)
        if (wgrad_wide(Ly, s)) return s;
    return -1;
}
// Only directional_input's job (8 output tiles = half the MFMAs of a 256 x 256 job for the same operand traffic) carries
// folded tiles.  (Measured r03: letting the 16-tile skip layers carry their position-encoding columns as well - +25 % MFMAs
// in that one job - made the wide kernel 17 % SLOWER: it breaks the whole-rounds schedule of 9 equal jobs x 113 chunks on
// 256 CUs.  Those columns stay direct jobs.)
__host__ __device__ inline int wgrad_fold_xseg(const Plan &P, int l, int fold = 1) {   // folded segment of layer l, or -1
    const Layer &Ly = P.layer[l];
    if (wgrad_first_wide_seg(Ly) < 0) return -1;
    if (!(Ly.t_out == 8 && l == P.n_hidden + 3 && P.nlayers == P.n_hidden + 6)) return -1;
    for (int s = 0; s < Ly.nseg; ++s)
        if (!wgrad_wide(Ly, s) && Ly.seg[s].nkb >= 1 && Ly.seg[s].nkb <= 4) return s;
    return -1;
}
__host__ __device__ inline bool wgrad_fold_sigma(const Plan &P) {
    const int nh = P.n_hidden;
    if (P.nlayers != nh + 6) return false;   // a RenderRayNet plan (the warp net's two-layer plan has no heads)
    const Layer &Ls = P.layer[nh + 2], &Ld = P.layer[nh + 3];
    return Ls.t_out == 1 && Ls.nseg == 1 && wgrad_first_wide_seg(Ld) == 0 && Ld.seg[0].nkb == Ls.seg[0].nkb && Ld.seg[0].nkb == 16 &&
           Ld.t_out == 8;
}
// (Measured r03: the 8-tile job - half the MFMAs of a 16-tile job per sample - lasts 0.7, not 0.5, of a 16-tile
// workgroup; giving it double chunks to "equalise" made the launch 19 % slower.  Per stage a workgroup pays ~20 % of a
// 16-tile stage that does not overlap with its MFMAs.)
// how the pair (layer l, segment s) is computed: 0 = wide job, 1 = rides with a wide job, 2 = direct narrow job
__host__ __device__ inline int wgrad_kind(const Plan &P, int l, int s, int fold) {
    if (wgrad_wide(P.layer[l], s)) return 0;
    if (fold) {
        if (wgrad_fold_xseg(P, l, fold) == s) return 1;
        if (l == P.n_hidden + 2 && wgrad_fold_sigma(P)) return 1;
    }
    return 2;
}
// wide jobs of a pair: groups of <= 16 output tiles x groups of <= 16 input k-blocks (a 512 x 512 layer: 2 x 2 jobs)
__host__ __device__ inline int wgrad_wide_jobs(const Layer &Ly, int s) {
    return wgrad_wide(Ly, s) ? ((Ly.t_out + 15) / 16) * ((Ly.seg[s].nkb + 15) / 16) : 0;
}
__host__ __device__ inline int wgrad_jobs(const Plan &P) {
    int jobs = 0;
    for (int l = 0; l < P.nlayers; ++l)
        for (int s = 0; s < P.layer[l].nseg; ++s) jobs += wgrad_wide_jobs(P.layer[l], s);
    return jobs;
}
__host__ __device__ inline int wgrad_direct_jobs(const Plan &P, int fold = 0) {  // narrow jobs: 4x4-tile blocks
    int jobs = 0;
    for (int l = 0; l < P.nlayers; ++l)
        for (int s = 0; s < P.layer[l].nseg; ++s)
            if (wgrad_kind(P, l, s, fold) == 2) jobs += ((P.layer[l].t_out + 3) / 4) * ((P.layer[l].seg[s].nkb + 3) / 4);
    return jobs;
}

// f16x3 statistics (STAT_INTS ints behind the rows of act / dy): exponent of the largest |X| entering forward layer l
// through its hidden segment at [l], of the largest encoder / additional-input column at [STAT_ENC]; of the largest |dY|
// of forward layer l at [l].  Encoded directions are <= 1.
constexpr int STAT_ENC = STAT_INTS - 1;
__host__ __device__ inline int xstat_index(const Plan &P, int l, int s) {   // -1: exponent 0 (direction encoding)
    const Seg &sg = P.layer[l].seg[s];
    if (sg.type == SEG_HIDDEN) return l == P.n_hidden + 2 ? P.n_hidden + 3 : l;   // the sigma head reads what directional_input reads
    if (sg.type == SEG_PE && l == P.n_hidden + 3) return -1;
    return STAT_ENC;
}
// mlp_train_bf16.hip: the narrow jobs with two fp16 parts (f16x3 training)
int launch_wgrad_direct_f16(const Plan &P, const TrainLayout &L, const WgradArgs &W, int jobs, int G, hipStream_t s);
// mlp_train_bf16.hip: the wide jobs with split-bf16 operands (nsplit parts each)
int launch_wgrad_wide_bf16(const Plan &P, const TrainLayout &L, const WgradArgs &W, int jobs, int G, int nsplit, hipStream_t s);

}  // namespace snerf
