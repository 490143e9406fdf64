// Host+device description of how a RenderRayNet (models/render_ray_net.py:8-61) is laid out as a
// stream of 17 KiB "slabs" of MFMA A-operands that the fused kernel consumes in order.
//
// MFMA used: v_mfma_f32_16x16x4_f32 (exact fp32).  The kernel computes the TRANSPOSED layer
//     out^T[feature, sample] = W[feature, k] * in^T[k, sample]
// so W is the A operand (lane l supplies W[16*to + (l&15)][k-slot l>>4]) and the activations are
// the B operand (lane l supplies feature k-slot l>>4 of sample l&15).  The accumulator of output
// tile `to` leaves feature 16*to + 4*(l>>4) + r of sample l&15 in register r of lane l - which is
// exactly the B-operand layout of k-step (kb=to, r) of the next layer if that k-step is defined to
// contract features {16*kb + 4*g + r : g = 0..3}.  Activations therefore never leave registers
// between layers; only the weights move, pre-permuted here into consumption order.
//
// k-block kb  = 4 k-steps r=0..3 = 16 input features; one ds_read_b128 per (kb, output tile) gives a
//               lane its 4 A values: W[16*to + (l&15)][col(kb, g=l>>4, r)], r = 0..3.
// slab        = SLAB_TILES x 256 floats of A tiles ([k-block in slab][to][lane][r]) + 256 floats of bias
//               (valid in the first slab of a layer).  k-blocks per slab = SLAB_TILES / t_out.
// Input segments of a layer (columns of its weight matrix), each a whole number of k-blocks:
//   HIDDEN  col = col_off + 16*kb + 4*g + r                     (previous layer's accumulator)
//   PE      encoder output of a 3-vector (utils.py:114-131), two "units" per k-block per lane:
//           unit p = 4*(2*kb + (r>>1)) + g;  identity units p < 3*id carry (x_p, -);
//           p' = p - 3*id -> frequency k = p'/3, channel c = p'%3 carries (sin, cos)(2^k x_c):
//           col = col_off + {3*id + 6*k + c, 3*id + 6*k + 3 + c}[r&1]
//   ADD     per-ray additional inputs, col = col_off + 16*kb + 4*g + r
#pragma once
#include <stdint.h>

namespace snerf {

// fp32 streams: SLAB_TILES A tiles ((k-block, output tile) pairs, 1 KiB each) per slab = 32 MFMAs of 16x16x4 per tile
// and wave between two workgroup barriers.  (16 tiles = one barrier per 64 MFMAs per wave left 8 % of the wave cycles
// parked at the hand-over; 32 halves that.  The split-bf16 streams keep 16 tiles of NS parts.)
constexpr int SLAB_TILES = 32;
__host__ __device__ constexpr int slab_tiles(int kw) { return kw == 16 ? SLAB_TILES : 16; }
constexpr int SLAB_A_FLOATS = SLAB_TILES * 256;
constexpr int SLAB_AUX_FLOATS = 256;
constexpr int SLAB_FLOATS = SLAB_A_FLOATS + SLAB_AUX_FLOATS;  // 8448 floats = 33 KiB
constexpr int SLAB_PAD = 3;                                    // zero slabs after the stream (prefetch overrun)
constexpr int MAX_WIDTH = 512;   // --netwidth (config_parser.py:20): trunks of 64, 128, 256 or 512 features (others run zero-padded)
constexpr int MAX_LAYERS = 21;  // n_layers <= 16, + 5 fixed layers
constexpr int STAT_INTS = 32;    // per-layer statistics behind the activation / dY rows of a training step (f16x3)

enum SegType : int { SEG_HIDDEN = 0, SEG_PE = 1, SEG_ADD = 2 };

struct Seg {
    int type;
    int col_off;   // first column of this segment in the layer's weight matrix
    int ncols;     // real columns
    int nkb;       // k-blocks (16 slots each)
    int L, ident;  // PE only
};

struct Layer {
    int64_t w_off, b_off;  // offsets into params_flat
    int n_out, n_in;       // weight is [n_out, n_in] row-major
    int t_out;             // output tiles of 16 rows (n_out padded)
    int nseg;
    Seg seg[3];
    int nkb;         // total k-blocks
    int first_slab;  // index of the layer's first slab in the stream
    int nslab;
};

struct Plan {
    int nlayers;
    int total_slabs;
    int width, n_hidden;  // n_hidden = n_layers - 1 positional_net layers
    int pos_nkb, dir_nkb, add_nkb;
    int kw;  // features per k-block: 16 (fp32 MFMA 16x16x4 path) or 32 (split-bf16 MFMA 16x16x32 path)
    int pos_dim, dir_dim, add_dim;
    int64_t param_floats;
    Layer layer[MAX_LAYERS];
};

__host__ __device__ inline int pe_units(int L, int ident) { return 3 * (ident ? 1 : 0) + 3 * L; }
__host__ __device__ inline int pe_nkb(int L, int ident) {
    const int per_lane = (pe_units(L, ident) + 3) / 4;  // units per lane group g
    return (2 * per_lane + 3) / 4;
}
// column (within the encoder output) feeding slot (kb, g, r) of a PE segment; -1 = zero padding
__host__ __device__ inline int pe_slot_col(int L, int ident, int kb, int g, int r) {
    const int p = 4 * (2 * kb + (r >> 1)) + g;
    const int nid = ident ? 3 : 0;
    if (p < nid) return (r & 1) ? -1 : p;
    const int pp = p - nid;
    if (pp >= 3 * L) return -1;
    const int k = pp / 3, c = pp - 3 * k;
    return nid + 6 * k + ((r & 1) ? 3 : 0) + c;
}

// ---- 32-wide k-blocks (split-bf16 path): a lane holds 8 values per k-block, slot (kb, g, e), e = 0..7 ----
//   HIDDEN / ADD  col = 32*kb + 16*(e>>2) + 4*g + (e&3)   (accumulators of tiles 2kb, 2kb+1 side by side)
//   PE            unit p = 4*(4*kb + (e>>1)) + g, (first, second) = e&1
__host__ __device__ inline int pe_nkb32(int L, int ident) {
    const int per_lane = (pe_units(L, ident) + 3) / 4;
    return (per_lane + 3) / 4;
}
__host__ __device__ inline int pe_slot_col32(int L, int ident, int kb, int g, int e) {
    const int p = 4 * (4 * kb + (e >> 1)) + g;
    const int nid = ident ? 3 : 0;
    if (p < nid) return (e & 1) ? -1 : p;
    const int pp = p - nid;
    if (pp >= 3 * L) return -1;
    const int k = pp / 3, c = pp - 3 * k;
    return nid + 6 * k + ((e & 1) ? 3 : 0) + c;
}

// Builds the plan; returns 0 or a negative SNERF_E_* with `why` set.
inline int make_plan(const snerf_mlp_desc &d, Plan &P, const char *&why, int kw = 16) {
    why = "";
    P.kw = kw;
    if (d.n_layers < 1 || d.n_layers > 16) { why = "n_layers must be in [1,16]"; return -1; }
    if (d.width < 2 || d.width > MAX_WIDTH) {
        why = "width must be in [2, 512] (the layer chain lives in registers: two accumulator sets of width x 16 samples per wave)";
        return -1;
    }
    if (d.pos_freqs < 0 || d.pos_freqs > 16 || d.dir_freqs < 0 || d.dir_freqs > 16) { why = "bad encoder frequencies"; return -1; }
    if (d.add_dim < 0 || d.add_dim > 4096) { why = "bad add_dim"; return -1; }
    // W / WD: the layer widths of the parameters (RenderRayNet: width and width // 2).  The kernels exist for trunks of 64, 128 and
    // 256 features: any other width runs embedded in the next larger one - the extra output rows and input slots of every
    // layer are zero padding like the rows 3 .. 15 of the rgb head (their activations are relu(0) = 0, nothing reads them,
    // no gradient is scattered from them), so the results are those of the unpadded network.
    const int W = d.width, WD = W / 2;
    // Above 256 (config_parser.py:20 --netwidth): kernels for 320, 384, 448 and 512 features (20 .. 32 tiles), one wave per SIMD
    // with the whole 512-register file (two accumulator sets of up to 32 tiles); a slab then is one k-block of all output tiles
    // (its other tiles are padding) and the bias block (256 floats per slab) of a layer spans its first two slabs.
    const int WK = W <= 64 ? 64 : W <= 128 ? 128 : W <= 256 ? 256 : (W + 63) / 64 * 64, WDK = WK / 2;
    const int pid = d.pos_identity ? 1 : 0, did = d.dir_identity ? 1 : 0;
    P.width = WK;
    P.n_hidden = d.n_layers - 1;
    P.pos_dim = 3 * (pid + 2 * d.pos_freqs);
    P.dir_dim = 3 * (did + 2 * d.dir_freqs);
    P.add_dim = d.add_dim;
    if (P.pos_dim + P.add_dim == 0) { why = "empty position input"; return -1; }
    if (d.use_dir && P.dir_dim == 0) { why = "empty direction encoding"; return -1; }
    P.pos_nkb = kw == 16 ? pe_nkb(d.pos_freqs, pid) : pe_nkb32(d.pos_freqs, pid);
    P.dir_nkb = d.use_dir ? (kw == 16 ? pe_nkb(d.dir_freqs, did) : pe_nkb32(d.dir_freqs, did)) : 0;
    P.add_nkb = (d.add_dim + kw - 1) / kw;
    const int pin = P.pos_dim + P.add_dim;
    int nl = 0, slab = 0;
    int64_t off = 0;
    // n_out / hidden_cols: sizes in the parameters; n_out_k / hidden_k: the (padded) sizes the kernel walks
    auto add_layer = [&](int n_out, int n_out_k, bool hidden, int hidden_cols, int hidden_k, int extra /*0 none, 1 pos(+add), 2 dir*/) {
        Layer &Ly = P.layer[nl++];
        Ly.n_out = n_out;
        Ly.t_out = (n_out_k + 15) / 16;
        Ly.nseg = 0;
        int col = 0;
        if (hidden) {
            Seg &s = Ly.seg[Ly.nseg++];
            s = Seg{SEG_HIDDEN, col, hidden_cols, hidden_k / kw, 0, 0};
            col += hidden_cols;
        }
        if (extra == 1) {
            if (P.add_dim && d.add_first) {
                Seg &a = Ly.seg[Ly.nseg++];
                a = Seg{SEG_ADD, col, P.add_dim, P.add_nkb, 0, 0};
                col += P.add_dim;
            }
            Seg &s = Ly.seg[Ly.nseg++];
            s = Seg{SEG_PE, col, P.pos_dim, P.pos_nkb, d.pos_freqs, pid};
            col += P.pos_dim;
            if (P.add_dim && !d.add_first) {
                Seg &a = Ly.seg[Ly.nseg++];
                a = Seg{SEG_ADD, col, P.add_dim, P.add_nkb, 0, 0};
                col += P.add_dim;
            }
        } else if (extra == 2) {
            Seg &s = Ly.seg[Ly.nseg++];
            s = Seg{SEG_PE, col, P.dir_dim, P.dir_nkb, d.dir_freqs, did};
            col += P.dir_dim;
        }
        Ly.n_in = col;
        Ly.nkb = 0;
        for (int i = 0; i < Ly.nseg; ++i) Ly.nkb += Ly.seg[i].nkb;
        const int kps = slab_tiles(kw) / Ly.t_out;
        Ly.first_slab = slab;
        Ly.nslab = (Ly.nkb + kps - 1) / kps;
        slab += Ly.nslab;
        Ly.w_off = off;
        off += (int64_t)Ly.n_out * Ly.n_in;
        Ly.b_off = off;
        off += Ly.n_out;
    };
    add_layer(W, WK, false, 0, 0, 1);                            // positions_pose_input   (:19)
    for (int i = 0; i < d.n_layers - 1; ++i)                     // positional_net[i]      (:21-25)
        add_layer(W, WK, true, W, WK, ((d.skip_mask >> i) & 1u) ? 1 : 0);
    add_layer(W, WK, true, W, WK, 0);                            // additional_linear_layer (:27)
    add_layer(1, 1, true, W, WK, 0);                             // sigma_out_layer        (:28)
    add_layer(WD, WDK, true, W, WK, d.use_dir ? 2 : 0);          // directional_input      (:31-34)
    add_layer(WD, WDK, true, WD, WDK, 0);                        // directional_net[0]     (:38-39)
    add_layer(3, 3, true, WD, WDK, 0);                           // rgb_out_layer          (:40)
    (void)pin;
    P.nlayers = nl;
    P.total_slabs = slab;
    P.param_floats = off;
    for (int l = 0; l < nl; ++l)
        if (P.layer[l].t_out > 16 && P.layer[l].nslab < 2) {
            why = "width above 256 needs at least 17 input columns (two k-blocks) in every layer";
            return -1;
        }
    return 0;
}

// column of `Ly`'s weight matrix that feeds input slot (kb, g, r) (kb counted over the whole layer),
// or -1 for padding.  Shared by the weight packer, the transposed packer and the gradient scatter.
__host__ __device__ inline int slot_to_col(const Layer &Ly, int kb, int g, int r) {
    for (int s = 0; s < Ly.nseg; ++s) {
        const Seg &sg = Ly.seg[s];
        if (kb < sg.nkb) {
            int c;
            if (sg.type == SEG_PE) {
                c = pe_slot_col(sg.L, sg.ident, kb, g, r);
            } else {
                c = 16 * kb + 4 * g + r;
                if (c >= sg.ncols) c = -1;
            }
            return c < 0 ? -1 : sg.col_off + c;
        }
        kb -= sg.nkb;
    }
    return -1;
}

// Index into params_flat of the parameter that element `e` of the layer's slab number `sl` holds in the fp32 forward
// stream, or -1 for zero padding.  One definition for the packer (mlp.hip: mlp_pack_kernel) and for the slot tables through
// which the optimiser step refreshes the stream in place (train_step.hip).
__host__ __device__ inline int64_t fwd_slab_src(const Layer &Ly, int sl, int e) {
    if (e < SLAB_A_FLOATS) {
        const int kps = SLAB_TILES / Ly.t_out;
        const int per_kb = Ly.t_out * 256;
        const int kbl = e / per_kb;
        int rem = e - kbl * per_kb;
        const int to = rem >> 8;
        rem &= 255;
        const int l = rem >> 2, r = rem & 3;
        const int i = l & 15, g = l >> 4;
        const int row = 16 * to + i;
        const int kb = sl * kps + kbl;
        if (kbl < kps && kb < Ly.nkb && row < Ly.n_out) {
            const int col = slot_to_col(Ly, kb, g, r);
            if (col >= 0) return Ly.w_off + (int64_t)row * Ly.n_in + col;
        }
        return -1;
    }
    // bias block: 256 floats per slab - a layer of more than 16 output tiles continues in its second slab
    const int jj = sl * SLAB_AUX_FLOATS + (e - SLAB_A_FLOATS);
    return (sl < (Ly.t_out + 15) / 16 && jj < Ly.n_out) ? Ly.b_off + jj : -1;
}

// 32-wide counterpart of slot_to_col
__host__ __device__ inline int slot_to_col32(const Layer &Ly, int kb, int g, int e) {
    for (int s = 0; s < Ly.nseg; ++s) {
        const Seg &sg = Ly.seg[s];
        if (kb < sg.nkb) {
            int c;
            if (sg.type == SEG_PE) {
                c = pe_slot_col32(sg.L, sg.ident, kb, g, e);
            } else {
                c = 32 * kb + 16 * (e >> 2) + 4 * g + (e & 3);
                if (c >= sg.ncols) c = -1;
            }
            return c < 0 ? -1 : sg.col_off + c;
        }
        kb -= sg.nkb;
    }
    return -1;
}

// ---- training buffers ------------------------------------------------------------------------------
// Activations and their gradients are kept in HBM between the forward, the dgrad and the wgrad
// kernels in TILE-ROW-MAJOR layout: a tile-row is [n samples][16 features] fp32 (n*64 B); feature
// 16*t + 4*g + r of sample s lives at ((row0 + t)*n + s)*16 + 4*g + r.  A wave stores/loads a tile
// of its 16 samples as one fully coalesced 1 KiB access (f4 per lane), and the wgrad kernel reads
// 4 samples x 16 features = 256 contiguous bytes straight into an MFMA A/B operand register.
struct TrainLayout {
    int T, TD, nh;
    int pe, add, dpe;    // encoder inputs in slot order (pos_nkb / add_nkb / dir_nkb tile-rows)
    int x[18];           // x[i], i = 1..nh+1: input of positional_net[i-1] / additional (post-ReLU), T rows each
    int o, h1, h2;       // additional out (T), directional_input out (TD), directional_net[0] out post-ReLU (TD)
    int mask;            // ReLU sign masks, 8 bytes per (sample, lane group): x[1..nh+1] then h2, two per tile-row
                         // (width 512: 16 bytes, one per tile-row)
    int act_rows;
    int dy[MAX_LAYERS];  // dY of forward layer l (plan order): t_out tile-rows
    int dy_rows;
    int gp[MAX_LAYERS];  // offset of layer l in the slot-ordered gradient: dW [t_out][nkb][64 lanes][4] then db [t_out*16]
    int gp_floats;
    short xrow[MAX_LAYERS][3];  // first activation tile-row of input segment s of forward layer l (wgrad B operand)
};

inline void make_train_layout(const Plan &P, TrainLayout &L) {
    L.T = P.width / 16;
    L.TD = P.width / 32;
    L.nh = P.n_hidden;
    int r = 0;
    L.pe = r; r += P.pos_nkb;
    L.add = r; r += P.add_nkb;
    L.dpe = r; r += P.dir_nkb;
    for (int i = 1; i <= L.nh + 1; ++i) { L.x[i] = r; r += L.T; }
    L.o = r; r += L.T;
    L.h1 = r; r += L.TD;
    L.h2 = r; r += L.TD;
    L.mask = r; r += L.T > 16 ? L.nh + 2 : (L.nh + 2 + 1) / 2;
    L.act_rows = r;
    int d = 0, g = 0;
    for (int l = 0; l < P.nlayers; ++l) {
        L.dy[l] = d;
        d += P.layer[l].t_out;
        L.gp[l] = g;
        g += P.layer[l].t_out * P.layer[l].nkb * 256 + P.layer[l].t_out * 16;
    }
    L.dy_rows = d;
    L.gp_floats = g;
    const int nh = L.nh;
    for (int l = 0; l < P.nlayers; ++l)
        for (int sgi = 0; sgi < P.layer[l].nseg; ++sgi) {
            const Seg &sg = P.layer[l].seg[sgi];
            int row;
            if (sg.type == SEG_ADD) row = L.add;
            else if (sg.type == SEG_PE) row = (l == nh + 3) ? L.dpe : L.pe;
            else if (l >= 1 && l <= nh + 1) row = L.x[l];        // positional_net[l-1] / additional
            else if (l == nh + 2 || l == nh + 3) row = L.o;      // sigma head, directional_input
            else if (l == nh + 4) row = L.h1;
            else row = L.h2;                                     // rgb head
            L.xrow[l][sgi] = (short)row;
        }
}

// first activation tile-row of input segment `s` of forward layer `l`
__host__ __device__ inline int seg_act_row(const Plan &, const TrainLayout &L, int l, int s) { return L.xrow[l][s]; }

// ---- backward (dgrad) weight stream ---------------------------------------------------------------
// The dgrad pass is the forward pass of the transposed network: dX^T[in feature, sample] = W^T * dY^T,
// with the same in-register chaining.  Its slab stream holds W^T tiles in reverse layer order:
//   A[(kb, to, lane (i,g), r)] = W_fwd[16*kb + 4*g + r][16*to + i]   (hidden input columns only)
struct BwdLayer {
    int fwd;      // forward layer (plan order) whose weight is transposed
    int seg;      // input segment of that layer whose columns are produced (0 = hidden columns)
    int t_out;    // tiles of input features produced
    int nkb;      // k-blocks over the forward layer's output rows
    int aux_fwd;  // forward layer whose weight row 0 is shipped in the aux block (sigma head), or -1
    int first_slab, nslab;
};
constexpr int MAX_BWD_LAYERS = 40;  // 4 + n_hidden + encoder-column transposes (<= n_hidden + 2)
struct BwdPlan {
    int nl, total_slabs;
    BwdLayer layer[MAX_BWD_LAYERS];
};
// input_grad: also emit the transposes of the encoder-input columns (direction encoding of directional_input,
// position encoding of the skip layers and of layer 0), which the dgrad kernel turns into d x / d dir.
inline int pe_seg_of(const Layer &Ly) {
    for (int s = 0; s < Ly.nseg; ++s)
        if (Ly.seg[s].type == SEG_PE) return s;
    return -1;
}
// kw = 16: the fp32 stream (k-blocks of 16 forward output rows); kw = 32: the split-bf16 stream (k-blocks of 32;
// the output tiles stay 16 wide).  P is always the 16-wide plan.
// Output tiles of the encoder-column transposes the input-gradient dgrad variants are compiled for: the default encoders
// (<= 4 position / 2 direction k-blocks: L = 10 / 4 without identity columns) or the wide variant (<= 8 / 8: identity
// columns, up to 16 frequencies).  The packer and the kernel launch make the same choice from the plan.
struct PeTiles {
    int pos, dir;
};
inline PeTiles bwd_pe_tiles(const Plan &P) {
    return (P.pos_nkb <= 4 && P.dir_nkb <= 2) ? PeTiles{4, 2} : PeTiles{8, 8};
}
inline void make_bwd_plan(const Plan &P, BwdPlan &B, bool input_grad = false, int kw = 16) {
    const int T = P.width / 16, TD = P.width / 32, nh = P.n_hidden;
    const int kdiv = kw / 16;
    int nl = 0, slab = 0;
    auto add = [&](int fwd, int seg, int t_out, int nkb16, int aux) {
        const int nkb = (nkb16 + kdiv - 1) / kdiv;
        BwdLayer &b = B.layer[nl++];
        b = BwdLayer{fwd, seg, t_out, nkb, aux, slab, 0};
        const int kps = slab_tiles(kw) / t_out;
        b.nslab = (nkb + kps - 1) / kps;
        slab += b.nslab;
    };
    const PeTiles pt = bwd_pe_tiles(P);
    auto add_pe = [&](int fwd, int nkb_out_rows) {
        const int s = pe_seg_of(P.layer[fwd]);
        if (input_grad && s >= 0 && P.layer[fwd].seg[s].nkb > 0) {
            // padded to the tile count the kernel variant is compiled for (a power of two, so that it divides the slab)
            add(fwd, s, fwd == nh + 3 ? pt.dir : pt.pos, nkb_out_rows, -1);
        }
    };
    add(nh + 5, 0, TD, 1, -1);       // rgb head^T : d rgb (3) -> d h2
    add(nh + 4, 0, TD, TD, -1);      // directional_net[0]^T
    add_pe(nh + 3, TD);              // directional_input^T, direction-encoding columns
    add(nh + 3, 0, T, TD, nh + 2);   // directional_input^T (hidden columns) + sigma head row via aux
    add(nh + 1, 0, T, T, -1);        // additional_linear_layer^T -> d Y of forward layer nh
    add_pe(nh, T);
    for (int i = nh; i >= 1; --i) {
        add(i, 0, T, T, -1);         // positional_net[i-1]^T (hidden columns) -> d Y of forward layer i-1
        add_pe(i - 1, T);            // ... whose position-encoding columns follow if it has any (skip / layer 0)
    }
    B.nl = nl;
    B.total_slabs = slab;
}
// The same for the fp32 transposed (dgrad) stream: element `e` of slab number `sl` of backward layer `Bl`
// (mlp_train.hip: mlp_pack_t_kernel; train_step.hip: slot tables).
__host__ __device__ inline int64_t bwd_slab_src(const Plan &P, const BwdLayer &Bl, int sl, int e) {
    const Layer &Ly = P.layer[Bl.fwd];
    if (e < SLAB_A_FLOATS) {
        const int kps = SLAB_TILES / Bl.t_out;
        const int per_kb = Bl.t_out * 256;
        const int kbl = e / per_kb;
        int rem = e - kbl * per_kb;
        const int to = rem >> 8;
        rem &= 255;
        const int l = rem >> 2, r = rem & 3;
        const int i = l & 15, g = l >> 4;
        const int kb = sl * kps + kbl;
        const int row = 16 * kb + 4 * g + r;  // forward output feature (contraction index)
        // forward input column produced by output row (to, i) of the transpose
        const Seg &sg = Ly.seg[Bl.seg];
        int col = -1;
        if (sg.type == SEG_PE) {
            if (to < sg.nkb) {  // slot (i>>2, i&3) of encoder k-block `to`
                const int c = pe_slot_col(sg.L, sg.ident, to, i >> 2, i & 3);
                if (c >= 0) col = sg.col_off + c;
            }
        } else if (16 * to + i < sg.ncols) {
            col = sg.col_off + 16 * to + i;
        }
        if (kbl < kps && kb < Bl.nkb && row < Ly.n_out && col >= 0) return Ly.w_off + (int64_t)row * Ly.n_in + col;
        return -1;
    }
    if (sl < (Bl.t_out + 15) / 16 && Bl.aux_fwd >= 0) {   // (more than 16 tiles: continued in the second slab, like the bias block)
        const Layer &La = P.layer[Bl.aux_fwd];
        const int jj = sl * SLAB_AUX_FLOATS + (e - SLAB_A_FLOATS);
        if (jj < La.seg[0].ncols) return La.w_off + jj;  // row 0 of the sigma head
    }
    return -1;
}
inline int bwd_total_slabs(const Plan &P, bool input_grad = false, int kw = 16) {
    BwdPlan B;
    make_bwd_plan(P, B, input_grad, kw);
    return B.total_slabs;
}
// K-splits (sample chunks) of the wgrad kernels for n samples.  wgrad_chunks: the most any job is split into = what the
// partial buffer is sized for (chunks of >= 128 samples, at most 128 of them); wgrad_chunks_1k: chunks of >= 1024 samples -
// the narrow jobs' split, and the wide jobs' wherever that already fills the chip (mlp_train.hip: launch_wgrad).
inline int wgrad_chunks(int64_t n) {
    int64_t g = (n + 127) / 128;
    return (int)(g < 1 ? 1 : (g > 128 ? 128 : g));
}
inline int wgrad_chunks_1k(int64_t n) {
    int64_t g = (n + 1023) / 1024;
    return (int)(g < 1 ? 1 : (g > 128 ? 128 : g));
}

// ---- WarpFieldNet (models/warp_field_net.py:8-22): linear1 [width, PE(x) | pose] + ReLU, linear2 [3, width] -------------
// kw = 16: fp32 stream (warp.hip); kw = 32: split-bf16 stream (warp_bf16.hip, width 256 only)
inline int make_warp_plan(const snerf_warp_desc &d, Plan &P, const char *&why, int kw = 16) {
    why = "";
    if (d.width < 1 || d.width > 256) { why = "width must be in [1, 256]"; return -1; }
    if (d.pos_freqs < 0 || d.pos_freqs > 16) { why = "bad encoder frequencies"; return -1; }
    if (d.pose_dim < 0 || d.pose_dim > 4096) { why = "bad pose_dim"; return -1; }
    const int pid = d.pos_identity ? 1 : 0;
    const int WK = d.width <= 128 ? 128 : 256;   // other widths run zero-padded inside the next kernel width (make_plan)
    P.width = WK;
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
    L0.t_out = WK / 16;
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
    L1.seg[0] = Seg{SEG_HIDDEN, 0, d.width, WK / kw, 0, 0};
    L1.n_in = d.width;
    L1.nkb = WK / kw;
    L1.first_slab = L0.nslab;
    L1.nslab = (L1.nkb + slab_tiles(kw) - 1) / slab_tiles(kw);
    L1.w_off = L0.b_off + L0.n_out;
    L1.b_off = L1.w_off + (int64_t)3 * d.width;
    P.nlayers = 2;
    P.total_slabs = L0.nslab + L1.nslab;
    P.param_floats = L1.b_off + 3;
    return 0;
}

}  // namespace snerf
