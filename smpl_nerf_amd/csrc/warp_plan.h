// Kernel arguments of the 2-layer WarpFieldNet (models/warp_field_net.py:8-22) for warp.hip / warp_bf16.hip; its plan
// (make_warp_plan) lives with the other host-side layout logic in mlp_plan.h.
#pragma once
#include "mlp_device.h"

namespace snerf {

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
    int l1_slab;        // first slab of linear2 in the stream (resident kernel)
    int64_t n_tiles;    // sample tiles (resident kernel: persistent workgroups)
    const float *ray_bias;   // resident inference kernel: [n/spr][WIDTH] = linear1.bias + linear1[:, pose columns] . pose_enc, or null
};

}  // namespace snerf
