// Host-side layout logic of libsmplnerf_hip.so (csrc/mlp_plan.h, warp_plan.h: slab stream, activation / gradient
// tile-row layouts, transposed stream) compiled for the CPU with -fsanitize=address,undefined and swept over the
// descriptor space ("sanitizer build").  Every offset the kernels index with is checked against the
// buffer sizes the C-ABI reports, so an out-of-range slot computed on the host shows up here, not as a stray GPU write.
#define __host__
#define __device__
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct snerf_mlp_desc {
    int32_t n_layers, width, pos_freqs, pos_identity, dir_freqs, dir_identity, add_dim;
    uint32_t skip_mask;
    int32_t use_dir, add_first;
};
struct snerf_warp_desc {
    int32_t width, pos_freqs, pos_identity, pose_dim;
};
#include "../../smpl_nerf_amd/csrc/mlp_plan.h"

using namespace snerf;

#define CHECK(c)                                                                  \
    do {                                                                          \
        if (!(c)) {                                                               \
            std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__);   \
            std::exit(1);                                                         \
        }                                                                         \
    } while (0)

static long checked = 0;

static void check_plan(const snerf_mlp_desc &d, int kw) {
    Plan P;
    const char *why = "";
    if (make_plan(d, P, why, kw) != 0) return;
    ++checked;
    const int tiles = slab_tiles(kw);
    CHECK(P.nlayers == d.n_layers + 5 && P.nlayers <= MAX_LAYERS);
    int64_t off = 0;
    int slab = 0;
    std::vector<char> hit((size_t)P.param_floats, 0);
    for (int l = 0; l < P.nlayers; ++l) {
        const Layer &L = P.layer[l];
        CHECK(L.w_off == off && L.b_off == off + (int64_t)L.n_out * L.n_in);
        off = L.b_off + L.n_out;
        CHECK(L.first_slab == slab && L.t_out * 16 >= L.n_out && tiles % L.t_out == 0);
        const int kps = tiles / L.t_out;
        CHECK(L.nslab == (L.nkb + kps - 1) / kps);
        slab += L.nslab;
        int nkb = 0, cols = 0;
        for (int s = 0; s < L.nseg; ++s) {
            nkb += L.seg[s].nkb;
            CHECK(L.seg[s].col_off == cols);
            cols += L.seg[s].ncols;
        }
        CHECK(nkb == L.nkb && cols == L.n_in);
        // every weight column is fed by exactly one input slot
        std::vector<int> seen((size_t)L.n_in, 0);
        for (int kb = 0; kb < L.nkb; ++kb)
            for (int g = 0; g < 4; ++g)
                for (int e = 0; e < (kw == 16 ? 4 : 8); ++e) {
                    const int c = kw == 16 ? slot_to_col(L, kb, g, e) : slot_to_col32(L, kb, g, e);
                    CHECK(c >= -1 && c < L.n_in);
                    if (c >= 0) {
                        ++seen[(size_t)c];
                        for (int r = 0; r < L.n_out; ++r) hit[(size_t)(L.w_off + (int64_t)r * L.n_in + c)] = 1;
                    }
                }
        for (int c = 0; c < L.n_in; ++c) CHECK(seen[(size_t)c] == 1);
        for (int r = 0; r < L.n_out; ++r) hit[(size_t)(L.b_off + r)] = 1;
    }
    CHECK(off == P.param_floats && slab == P.total_slabs);
    for (char h : hit) CHECK(h == 1);   // the pack kernels read every parameter exactly where the plan says
    if (kw != 16) return;
    // training layouts (16-wide plan)
    TrainLayout T;
    make_train_layout(P, T);
    CHECK(T.act_rows > 0 && T.dy_rows > 0 && T.gp_floats > 0);
    for (int l = 0; l < P.nlayers; ++l) {
        CHECK(T.dy[l] >= 0 && T.dy[l] + P.layer[l].t_out <= T.dy_rows);
        CHECK(T.gp[l] >= 0 && T.gp[l] + P.layer[l].t_out * P.layer[l].nkb * 256 + P.layer[l].t_out * 16 <= T.gp_floats);
        for (int s = 0; s < P.layer[l].nseg; ++s) {
            const int row = seg_act_row(P, T, l, s);
            CHECK(row >= 0 && row + P.layer[l].seg[s].nkb <= T.mask);   // operand rows end before the mask rows
        }
    }
    CHECK(T.mask + (T.nh + 3) / 2 == T.act_rows);
    for (int ig = 0; ig < 2; ++ig)
        for (int kwb = 16; kwb <= 32; kwb += 16) {
            BwdPlan B;
            make_bwd_plan(P, B, ig != 0, kwb);
            CHECK(B.nl > 0 && B.nl <= MAX_BWD_LAYERS);
            int s2 = 0;
            for (int i = 0; i < B.nl; ++i) {
                const BwdLayer &b = B.layer[i];
                CHECK(b.first_slab == s2 && b.fwd >= 0 && b.fwd < P.nlayers && slab_tiles(kwb) % b.t_out == 0);
                CHECK(b.seg >= 0 && b.seg < P.layer[b.fwd].nseg);
                if (P.layer[b.fwd].seg[b.seg].type == SEG_PE) CHECK(P.layer[b.fwd].seg[b.seg].nkb <= b.t_out || kwb == 32 || !ig);
                s2 += b.nslab;
            }
            CHECK(s2 == B.total_slabs && bwd_total_slabs(P, ig != 0, kwb) == s2);
        }
}

int main() {
    for (int n_layers : {1, 2, 3, 8, 16})
        for (int width : {2, 7, 30, 64, 100, 128, 200, 250, 256})   // other widths than 64 / 128 / 256 run zero-padded (make_plan)
            for (int pL : {0, 4, 10, 16})
                for (int pid = 0; pid < 2; ++pid)
                    for (int dL : {0, 4, 16})
                        for (int add : {0, 2, 69})
                            for (uint32_t skip : {0u, 1u << 1, (1u << 0) | (1u << 4)})
                                for (int use_dir = 0; use_dir < 2; ++use_dir)
                                    for (int add_first = 0; add_first < 2; ++add_first)
                                        for (int kw = 16; kw <= 32; kw += 16) {
                                            if (kw == 32 && width != 256) continue;
                                            snerf_mlp_desc d{n_layers, width, pL, pid, dL, 1 - pid, add, skip & ((1u << (n_layers - 1)) - 1u),
                                                             use_dir, add_first};
                                            check_plan(d, kw);
                                        }
    for (int width : {1, 64, 100, 128, 200, 256})
        for (int pL = 0; pL <= 16; ++pL)
            for (int pid = 0; pid < 2; ++pid)
                for (int pose : {0, 2, 40, 69, 100})
                    for (int kw = 16; kw <= 32; kw += 16) {
                        snerf_warp_desc d{width, pL, pid, pose};
                        Plan P;
                        const char *why = "";
                        if (make_warp_plan(d, P, why, kw) != 0) continue;
                        ++checked;
                        CHECK(P.nlayers == 2 && P.layer[1].first_slab == P.layer[0].nslab);
                        CHECK(P.total_slabs == P.layer[0].nslab + P.layer[1].nslab);
                        CHECK(P.param_floats == (int64_t)width * (P.pos_dim + P.add_dim) + width + 3 * width + 3);
                    }
    std::printf("plans checked: %ld\n", checked);
    return checked > 3000 ? 0 : 2;
}
