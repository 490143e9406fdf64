// Plan of the 2-layer WarpFieldNet (models/warp_field_net.py:8-22) for the kernels of warp.hip / warp_bf16.hip.
#pragma once
#include "mlp_device.h"

namespace snerf {

// kw = 16: fp32 stream (warp.hip); kw = 32: split-bf16 stream (warp_bf16.hip, width 256 only)
inline int make_warp_plan(const snerf_warp_desc &d, Plan &P, const char *&why, int kw = 16) {
    why = "";
    if (d.width != 256 && d.width != 128) { why = "width must be 256 or 128"; return -1; }
    if (d.pos_freqs < 0 || d.pos_freqs > 16) { why = "bad encoder frequencies"; return -1; }
    if (d.pose_dim < 0 || d.pose_dim > 4096) { why = "bad pose_dim"; return -1; }
    const int pid = d.pos_identity ? 1 : 0;
    P.width = d.width;
    P.kw = kw;
    P.n_hidden = 0;
    P.pos_dim = 3 * (pid + 2 * d.pos_freqs);
    P.dir_dim = 0;
    P.add_dim = d.pose_dim;
    if (P.pos_dim + P.add_dim == 0) { why = "empty input"; return -1; }
    P.pos_nkb = kw == 16 ? pe_nkb(d.pos_freqs, pid) : pe_nkb32(d.pos_freqs, pid);
    P.dir_nkb = 0;
    P.add_nkb = (d.pose_dim + kw - 1) / kw;
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
        const int kps = slab_tiles(kw) / L0.t_out;
        L0.nslab = (L0.nkb + kps - 1) / kps;
    }
    L0.w_off = 0;
    L0.b_off = (int64_t)L0.n_out * L0.n_in;
    Layer &L1 = P.layer[1];
    L1.n_out = 3;
    L1.t_out = 1;
    L1.nseg = 1;
    L1.seg[0] = Seg{SEG_HIDDEN, 0, d.width, d.width / kw, 0, 0};
    L1.n_in = d.width;
    L1.nkb = d.width / kw;
    L1.first_slab = L0.nslab;
    L1.nslab = (L1.nkb + slab_tiles(kw) - 1) / slab_tiles(kw);
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
    float *act;  // training: tile-row-major [pe | pose | h] (see warp_train_layout)
};

}  // namespace snerf
