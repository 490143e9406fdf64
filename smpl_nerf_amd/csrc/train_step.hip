// The per-batch body of NerfSolver.train (solver/nerf_solver.py:76-87) as ONE C-ABI call:
//
//     rgb, rgb_fine, ... = pipeline(batch)            models/nerf_pipeline.py:14-67 (activations saved)
//     loss = MSE(rgb, gt) + MSE(rgb_fine, gt)         solver/nerf_solver.py:48-52
//     loss.backward()                                 autograd through compositing and both RenderRayNets
//     Adam.step()                                     solver/nerf_solver.py:31-33, 87 (torch.optim.Adam, one group)
//
// Pure sequencing of the library's own kernels on the caller's stream plus three small ones that live here: the MSE
// value / gradient, the Adam update over the flat parameter buffer, and the slot tables through which that update writes
// every new weight straight into the MFMA-ordered forward stream and the transposed dgrad stream (no re-pack launches).
// No allocation, no host synchronisation, no step-dependent host scalars (the step counter and the bias corrections live on
// the device): the call can be captured into a HIP graph.
//
// Ray chunks.  The loss is a mean over rays, so d loss / d rgb of a ray does not depend on the other rays of the batch:
// the batch is walked in chunks of `rays_per_chunk` rays - forward, loss gradient, backward per chunk, parameter gradients
// summed over the chunks in chunk order - and the saved layer inputs / d Y buffers (21 KB per sample) are sized by the
// chunk, not by the batch.  Nothing is recomputed (the autograd form needs the whole forward before any backward; the
// reference keeps 2.6 MB per ray alive).
#include <mutex>
#include <thread>
#include <vector>

#include "mlp_device.h"

namespace snerf {

static int64_t align256(int64_t b) { return (b + 255) & ~int64_t(255); }

// ------------------------------------------------------------------------------------------------
// slot tables: parameter i of params_flat -> its element of the forward / transposed fp32 stream (or -1)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fwd_slot_table_kernel(Plan P, int32_t *__restrict__ slot) {
    const int slab = blockIdx.x;
    int li = 0;
    while (li + 1 < P.nlayers && slab >= P.layer[li + 1].first_slab) ++li;
    const Layer &Ly = P.layer[li];
    const int sl = slab - Ly.first_slab;
    for (int e = threadIdx.x; e < SLAB_FLOATS; e += 256) {
        const int64_t src = fwd_slab_src(Ly, sl, e);
        if (src >= 0) slot[src] = slab * SLAB_FLOATS + e;
    }
}
__global__ __launch_bounds__(256) void bwd_slot_table_kernel(Plan P, BwdPlan B, int32_t *__restrict__ slot) {
    const int slab = blockIdx.x;
    int bi = 0;
    while (bi + 1 < B.nl && slab >= B.layer[bi + 1].first_slab) ++bi;
    const BwdLayer &Bl = B.layer[bi];
    const int sl = slab - Bl.first_slab;
    for (int e = threadIdx.x; e < SLAB_FLOATS; e += 256) {
        const int64_t src = bwd_slab_src(P, Bl, sl, e);
        if (src >= 0) slot[src] = slab * SLAB_FLOATS + e;
    }
}

// ------------------------------------------------------------------------------------------------
// loss: MSE(rgb, gt) + MSE(rgb_fine, gt), mean over all B * 3 values each (nn.MSELoss(), solver/nerf_solver.py:48-52)
// ------------------------------------------------------------------------------------------------
// One workgroup per chunk (a chunk is a few thousand rays): d rgb = (2 / (3 B)) (rgb - gt) like mse_loss_backward, the sums
// of squares in fp64, accumulated over the chunks in chunk order (deterministic).  fine == nullptr is run_fine = 0: the
// pipeline returns `rgb` twice (models/nerf_pipeline.py:43-44), so the loss is 2 MSE(rgb) and both gradients land on rgb.
constexpr int MSE_THREADS = 1024;
__global__ __launch_bounds__(MSE_THREADS) void mse_grad_kernel(const float *__restrict__ rgb_c, const float *__restrict__ rgb_f,
                                                               const float *__restrict__ gt, int64_t count, float norm,
                                                               float *__restrict__ d_c, float *__restrict__ d_f,
                                                               double *__restrict__ acc, int first, int last, double inv_total,
                                                               float *__restrict__ loss_out) {
    __shared__ double s_sum[2][MSE_THREADS / WAVE];
    double sc = 0.0, sf = 0.0;
    for (int64_t i = threadIdx.x; i < count; i += MSE_THREADS) {
        const float t = gt[i];
        const float ec = __fsub_rn(rgb_c[i], t);
        const float gc = __fmul_rn(norm, ec);
        sc += (double)ec * (double)ec;
        if (rgb_f) {
            const float ef = __fsub_rn(rgb_f[i], t);
            d_f[i] = __fmul_rn(norm, ef);
            d_c[i] = gc;
            sf += (double)ef * (double)ef;
        } else {
            d_c[i] = __fadd_rn(gc, gc);
        }
    }
    sc = wave_sum(sc);
    sf = wave_sum(sf);
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    if (lane == 0) s_sum[0][wave] = sc, s_sum[1][wave] = sf;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tc = 0.0, tf = 0.0;
        for (int w = 0; w < MSE_THREADS / WAVE; ++w) tc += s_sum[0][w], tf += s_sum[1][w];
        if (!rgb_f) tf = tc;
        tc += first ? 0.0 : acc[0];
        tf += first ? 0.0 : acc[1];
        acc[0] = tc;
        acc[1] = tf;
        if (last) {
            const float lc = (float)(tc * inv_total), lf = (float)(tf * inv_total);
            loss_out[0] = __fadd_rn(lc, lf);
            loss_out[1] = lc;
            loss_out[2] = lf;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam, amsgrad = False, maximize = False; solver/nerf_solver.py:11-14, 31-33)
// ------------------------------------------------------------------------------------------------
// The statements of torch's single-tensor update in their order, in fp32 with the Python scalars rounded to fp32 where
// torch hands them to a kernel, and with the fused multiply-adds torch's CPU kernels (the reference's optimiser runs there)
// contain - add(alpha), lerp_ and addcmul_ are `fmadd`s in ATen's vectorised code; established by bit comparison with
// torch.optim.Adam on the CPU (tools/ab/adam_debug.py; tests/test_gpu_round4.py holds both moments to equality; the parameters
// land within one unit in the last place - torch's vectorised CPU sqrt is not correctly rounded on every host, IEEE sqrt here):
//     grad = fma(weight_decay, param, grad)                        (only if weight_decay != 0)
//     exp_avg     = fma(1 - beta1, grad - exp_avg, exp_avg)        lerp_
//     exp_avg_sq  = fma((1 - beta2) * grad, grad, exp_avg_sq * beta2)
//     denom       = sqrt(exp_avg_sq) / sqrt(1 - beta2^t) + eps
//     param       = param + (-(lr / (1 - beta1^t)) * exp_avg) / denom
// t and the two step-dependent scalars live on the device (a captured graph replays with the right values).  torch keeps one
// step counter per parameter TENSOR (a tensor whose .grad is None is skipped and does not count): a range carries the counters
// of its tensors, all equal on entry, all incremented here.
constexpr int ADAM_MAX_RANGES = 32;
struct AdamRanges {
    int n;
    int64_t *step[ADAM_MAX_RANGES];
    int n_steps[ADAM_MAX_RANGES];
};
__global__ __launch_bounds__(64) void adam_scalars_kernel(AdamRanges R, double lr, double beta1, double beta2, float *__restrict__ scal) {
    const int r = blockIdx.x;
    const int64_t t = R.step[r][0] + 1;     // (read by every lane before any lane writes: one wave, in lock-step)
    for (int i = threadIdx.x; i < R.n_steps[r]; i += 64) R.step[r][i] = t;
    if (threadIdx.x == 0) {
        const double bc1 = 1.0 - pow(beta1, (double)t), bc2 = 1.0 - pow(beta2, (double)t);
        scal[2 * r + 0] = (float)(-(lr / bc1));
        scal[2 * r + 1] = (float)sqrt(bc2);
    }
}

constexpr int ADAM_MAX_NETS = 8;
struct AdamNet {
    int64_t begin, end;          // this net's kernel-ordered parameters inside the flat buffer
    float *packed, *packed_t;    // fp32 streams refreshed in place (nullable)
    const int32_t *slot_fwd, *slot_t;
};
struct AdamArgs {
    float *p;
    const float *g;
    float *m, *v;
    const float *scal;
    float w1, beta2, w2, eps, wd;
    int n_nets;
    AdamNet net[ADAM_MAX_NETS];
};

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs A, int64_t begin, int64_t end, int range) {
    const int64_t i = begin + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= end) return;
    const float neg_step = A.scal[2 * range], bc2_sqrt = A.scal[2 * range + 1];
    float p = A.p[i], g = A.g[i];
    if (A.wd != 0.f) g = __fmaf_rn(A.wd, p, g);
    float m = A.m[i], v = A.v[i];
    m = __fmaf_rn(A.w1, __fsub_rn(g, m), m);
    v = __fmaf_rn(__fmul_rn(A.w2, g), g, __fmul_rn(v, A.beta2));
    const float denom = __fadd_rn(__fdiv_rn(sqrtf(v), bc2_sqrt), A.eps);
    p = __fadd_rn(p, __fdiv_rn(__fmul_rn(neg_step, m), denom));
    A.m[i] = m;
    A.v[i] = v;
    A.p[i] = p;
#pragma unroll
    for (int k = 0; k < ADAM_MAX_NETS; ++k) {
        if (k >= A.n_nets) break;
        const AdamNet &N = A.net[k];
        if (i >= N.begin && i < N.end) {
            const int64_t j = i - N.begin;
            if (N.packed) {
                const int s = N.slot_fwd[j];
                if (s >= 0) N.packed[s] = p;
            }
            if (N.packed_t) {
                const int s = N.slot_t[j];
                if (s >= 0) N.packed_t[s] = p;
            }
        }
    }
}

static bool split_code(int precision) { return precision == 2 || precision == 3 || precision == SNERF_SPLIT_F16X3; }

}  // namespace snerf

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
extern "C" int snerf_mlp_stream_slots(const snerf_mlp_desc *desc, int32_t *slot_fwd, int32_t *slot_t, int input_grad,
                                      snerf_stream_t stream) {
    using namespace snerf;
    Plan P;
    const char *why;
    if (!desc) return fail(SNERF_E_BADARG, "mlp_stream_slots: desc is null");
    if (make_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "mlp_stream_slots: %s", why);
    hipStream_t s = (hipStream_t)stream;
    if (slot_fwd) {
        if (hipMemsetAsync(slot_fwd, 0xff, (size_t)P.param_floats * sizeof(int32_t), s) != hipSuccess)
            return fail(SNERF_E_LAUNCH, "mlp_stream_slots: memset failed");
        hipLaunchKernelGGL(fwd_slot_table_kernel, dim3(P.total_slabs), dim3(256), 0, s, P, slot_fwd);
        if (int rc = check_launch("mlp_stream_slots(fwd)")) return rc;
    }
    if (slot_t) {
        BwdPlan B;
        make_bwd_plan(P, B, input_grad != 0);
        if (hipMemsetAsync(slot_t, 0xff, (size_t)P.param_floats * sizeof(int32_t), s) != hipSuccess)
            return fail(SNERF_E_LAUNCH, "mlp_stream_slots: memset failed");
        hipLaunchKernelGGL(bwd_slot_table_kernel, dim3(B.total_slabs), dim3(256), 0, s, P, B, slot_t);
        if (int rc = check_launch("mlp_stream_slots(t)")) return rc;
    }
    return SNERF_OK;
}

extern "C" int snerf_adam_step_f32(const snerf_adam_state *st, const snerf_adam_range *ranges_host, int n_ranges,
                                   const snerf_adam_net *nets_host, int n_nets, snerf_stream_t stream) {
    using namespace snerf;
    if (!st || !st->params || !st->grads || !st->exp_avg || !st->exp_avg_sq || !st->scratch)
        return fail(SNERF_E_BADARG, "adam_step: null pointer in the optimiser state");
    if (st->n_params < 0 || n_ranges < 0 || n_ranges > ADAM_MAX_RANGES || n_nets < 0 || n_nets > ADAM_MAX_NETS)
        return fail(SNERF_E_BADARG, "adam_step: bad n_params / n_ranges / n_nets (at most %d ranges, %d nets)", ADAM_MAX_RANGES,
                    ADAM_MAX_NETS);
    if ((n_ranges && !ranges_host) || (n_nets && !nets_host)) return fail(SNERF_E_BADARG, "adam_step: null range / net array");
    if (n_ranges == 0) return SNERF_OK;
    AdamRanges R{};
    R.n = n_ranges;
    for (int r = 0; r < n_ranges; ++r) {
        const snerf_adam_range &g = ranges_host[r];
        if (g.begin < 0 || g.end > st->n_params || g.begin > g.end || !g.step || g.n_steps < 1)
            return fail(SNERF_E_BADARG, "adam_step: range %d: outside the parameter buffer, or no step counters", r);
        R.step[r] = g.step;
        R.n_steps[r] = g.n_steps;
    }
    if (!(st->lr >= 0.0) || !(st->eps >= 0.0) || !(st->beta1 >= 0.0 && st->beta1 < 1.0) || !(st->beta2 >= 0.0 && st->beta2 < 1.0) ||
        !(st->weight_decay >= 0.0))
        return fail(SNERF_E_BADARG, "adam_step: invalid hyper-parameters (torch.optim.Adam's own checks)");
    hipStream_t s = (hipStream_t)stream;
    AdamArgs A{};
    A.p = st->params;
    A.g = st->grads;
    A.m = st->exp_avg;
    A.v = st->exp_avg_sq;
    A.scal = st->scratch;
    A.w1 = (float)(1.0 - st->beta1);
    A.beta2 = (float)st->beta2;
    A.w2 = (float)(1.0 - st->beta2);
    A.eps = (float)st->eps;
    A.wd = (float)st->weight_decay;
    Plan P;
    const char *why;
    for (int k = 0; k < n_nets; ++k) {
        const snerf_adam_net &N = nets_host[k];
        if (!N.desc || make_plan(*N.desc, P, why) != 0) return fail(SNERF_E_BADARG, "adam_step: net %d has a bad descriptor", k);
        if (N.param_offset < 0 || N.param_offset + P.param_floats > st->n_params)
            return fail(SNERF_E_BADARG, "adam_step: net %d does not lie inside the flat parameter buffer", k);
        if (N.precision != 0 && !split_code(N.precision)) return fail(SNERF_E_BADARG, "adam_step: net %d: bad precision code", k);
        AdamNet &D = A.net[A.n_nets];
        D.begin = N.param_offset;
        D.end = N.param_offset + P.param_floats;
        if (N.precision == 0) {   // fp32 streams: refreshed in place by the update kernel
            D.packed = reinterpret_cast<float *>(N.packed);
            D.packed_t = reinterpret_cast<float *>(N.packed_t);
            D.slot_fwd = N.slot_fwd;
            D.slot_t = N.slot_t;
            if ((D.packed && !D.slot_fwd) || (D.packed_t && !D.slot_t))
                return fail(SNERF_E_BADARG, "adam_step: net %d: an fp32 stream needs its slot table (snerf_mlp_stream_slots)", k);
            if (D.packed || D.packed_t) ++A.n_nets;
        }
    }
    hipLaunchKernelGGL(adam_scalars_kernel, dim3(n_ranges), dim3(64), 0, s, R, st->lr, st->beta1, st->beta2, st->scratch);
    if (int rc = check_launch("adam_step(scalars)")) return rc;
    for (int r = 0; r < n_ranges; ++r) {
        const int64_t b = ranges_host[r].begin, e = ranges_host[r].end;
        if (b == e) continue;
        const int64_t grid = (e - b + 255) / 256;
        if (grid > 0x7fffffffLL) return fail(SNERF_E_BADARG, "adam_step: range too large");
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)grid), dim3(256), 0, s, A, b, e, r);
        if (int rc = check_launch("adam_step")) return rc;
    }
    // split-precision streams hold pre-split parts (and, f16x3, per-layer scales): re-packed from the new parameters
    for (int k = 0; k < n_nets; ++k) {
        const snerf_adam_net &N = nets_host[k];
        if (N.precision == 0) continue;
        const float *pf = st->params + N.param_offset;
        int rc;
        if (N.packed && (rc = snerf_mlp_pack_bf16(N.desc, pf, N.packed, N.precision, stream))) return rc;
        if (N.packed_t && (rc = snerf_mlp_pack_t_bf16(N.desc, pf, N.packed_t, N.precision, 0, stream))) return rc;
    }
    return SNERF_OK;
}

namespace snerf {

struct TrainWs {
    int64_t raw_c, weights_c, z_fine, pts_f, raw_f, d_rgb_c, d_rgb_f, d_raw, act_c, act_f, dy, gpart, loss_acc;
    int64_t d_raw2, dy2, gpart2;   // second set for the coarse net's backward when it runs beside the fine net's (small chunks)
    int64_t contract;              // scratch of the d loss / d additional-inputs contraction (nets with add_dim > 0)
    bool concurrent;
    int64_t total;
};

// Chunks of at most this many samples (coarse + fine of a ray) leave much of the chip idle in some kernel (a 64-ray batch is 96
// workgroups of 128 samples; at 800 rays the coarse passes fill 1.56 rounds of the CUs): there the backward of the coarse
// net - independent of the fine net's: the hierarchical samples are detached, utils.py:260 - runs beside it on the caller's
// auxiliary stream.  A pure size rule, so that the workspace size does not depend on the device.  Measured (tools/ab/
// conc_ab.sh, ms per step without / with): 128 rays 1.49 / 1.33, 256: 2.42 / 2.29-2.35, 512: 4.00 / 3.96, 800: 6.60 / 6.45-6.48,
// 1024: 7.80 / 7.83, 2048: 15.29 / 15.21 - above 1024 rays nothing is left to fill, and the second d Y buffer would cost memory.
constexpr int64_t CONCURRENT_MAX_FINE_SAMPLES = 1024 * 256;
// the same rule for the smpl_nerf step (its coarse chain carries the warp net's backward as well).  Measured r05 with the rule
// lifted (coarse chain beside the fine chain at every chunk size): 4096 rays 34.00 -> 33.72 ms per step, 2048 rays the same
// fraction, for 1.8 GB more workspace - both chains are bound by the matrix pipe there, nothing is left to fill.
constexpr int64_t SMPL_CONCURRENT_MAX_FINE_SAMPLES = CONCURRENT_MAX_FINE_SAMPLES;

// fork / join events of the concurrent backward: one pair per host thread and device (include/smplnerf.h "State"), kept in a
// registry so that snerf_shutdown() can destroy them
struct EventPair {
    std::thread::id tid;
    int dev;
    hipEvent_t ev[2];
};
static std::mutex &event_mutex() {
    static std::mutex m;
    return m;
}
static std::vector<EventPair> &event_registry() {
    static std::vector<EventPair> v;
    return v;
}
static int fork_join_events(hipEvent_t &fork, hipEvent_t &join) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return fail(SNERF_E_LAUNCH, "nerf_train: cannot query the current device");
    const std::thread::id me = std::this_thread::get_id();
    std::lock_guard<std::mutex> lock(event_mutex());
    for (const EventPair &p : event_registry())
        if (p.tid == me && p.dev == dev) {
            fork = p.ev[0];
            join = p.ev[1];
            return SNERF_OK;
        }
    EventPair p{me, dev, {nullptr, nullptr}};
    for (int k = 0; k < 2; ++k)
        if (hipEventCreateWithFlags(&p.ev[k], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (k == 1) (void)hipEventDestroy(p.ev[0]);
            return fail(SNERF_E_LAUNCH, "nerf_train: cannot create an event");
        }
    event_registry().push_back(p);
    fork = p.ev[0];
    join = p.ev[1];
    return SNERF_OK;
}
void destroy_fork_join_events() {
    std::lock_guard<std::mutex> lock(event_mutex());
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (const EventPair &p : event_registry()) {
        (void)hipSetDevice(p.dev);
        (void)hipEventDestroy(p.ev[0]);
        (void)hipEventDestroy(p.ev[1]);
    }
    event_registry().clear();
    if (have_cur) (void)hipSetDevice(cur);
    (void)hipGetLastError();
}

static int train_ws(const snerf_mlp_desc *dc, const snerf_mlp_desc *df, int64_t chunk, int Nc, int Nf, TrainWs &w, bool two_streams = true,
                    int64_t concurrent_max = CONCURRENT_MAX_FINE_SAMPLES) {
    const int64_t N = Nc + Nf;
    int64_t act_c = 0, dy_c = 0, gp_c = 0, act_f = 0, dy_f = 0, gp_f = 0;
    int rc;
    if ((rc = snerf_mlp_train_sizes(dc, chunk * Nc, &act_c, &dy_c, nullptr, &gp_c, nullptr))) return rc;
    if (Nf > 0 && (rc = snerf_mlp_train_sizes(df, chunk * N, &act_f, &dy_f, nullptr, &gp_f, nullptr))) return rc;
    int64_t off = 0;
    auto take = [&](int64_t floats) {
        const int64_t o = off;
        off += align256(floats * 4);
        return o;
    };
    w.raw_c = take(chunk * Nc * 4);
    w.weights_c = take(chunk * Nc);
    w.z_fine = take(Nf > 0 ? chunk * N : 0);
    w.pts_f = take(Nf > 0 ? chunk * N * 3 : 0);
    w.raw_f = take(Nf > 0 ? chunk * N * 4 : 0);
    w.d_rgb_c = take(chunk * 3);
    w.d_rgb_f = take(chunk * 3);
    w.d_raw = take(chunk * N * 4);
    w.act_c = take(act_c);
    w.act_f = take(act_f);
    w.dy = take(dy_c > dy_f ? dy_c : dy_f);
    w.gpart = take(gp_c > gp_f ? gp_c : gp_f);
    w.loss_acc = take(4);
    w.concurrent = two_streams && Nf > 0 && chunk * N <= concurrent_max;
    w.d_raw2 = take(w.concurrent ? chunk * Nc * 4 : 0);
    w.dy2 = take(w.concurrent ? dy_c : 0);
    w.gpart2 = take(w.concurrent ? gp_c : 0);
    // d loss / d additional inputs (snerf_*_ig_f32): the per-ray partial sums of the contraction (contract.hip)
    int64_t cs = 0;
    if (dc->add_dim > 0) {
        cs = snerf_dy_contract_scratch_floats(chunk * Nc, dc->add_dim, Nc);
        if (Nf > 0) {
            const int64_t cf = snerf_dy_contract_scratch_floats(chunk * N, dc->add_dim, (int)N);
            cs = cf > cs ? cf : cs;
        }
        if (cs < 0) return (int)cs;
    }
    w.contract = take(cs);
    w.total = off;
    return SNERF_OK;
}

// d loss / d additional inputs of one net for one chunk: the stored d Y_l of layer 0 and of every skip layer contracted with the
// weight columns that read the additional inputs, summed over the samples of a ray (contract.hip; what autograd does through
// models/render_ray_net.py:43-50 and the `expand` of the pose rows, models/append_smpl_params_pipeline.py:29-52).  out: [rays, add_dim]
// rows of this chunk; overwrite: the first contraction of the chunk overwrites, the others accumulate.
static int contract_additional(const snerf_mlp_desc *desc, const float *params, const float *dy, int64_t n, int spr, float *out,
                               bool overwrite, float *scratch, snerf_stream_t stream) {
    Plan P;
    const char *why;
    if (make_plan(*desc, P, why) != 0) return fail(SNERF_E_BADARG, "nerf_train_grads: %s", why);
    TrainLayout L;
    make_train_layout(P, L);
    bool first = overwrite;
    for (int l = 0; l < P.nlayers; ++l)
        for (int sg = 0; sg < P.layer[l].nseg; ++sg) {
            const Seg &seg = P.layer[l].seg[sg];
            if (seg.type != SEG_ADD) continue;
            const Layer &Ly = P.layer[l];
            const int rc = snerf_dy_contract_f32(dy, n, L.dy[l], Ly.n_out, params + Ly.w_off, Ly.n_in, seg.col_off, seg.ncols, spr, out,
                                                 seg.ncols, 0, first ? 0 : 1, scratch, stream);
            if (rc) return rc;
            first = false;
        }
    return SNERF_OK;
}

static int64_t effective_chunk(int64_t B, int64_t rays_per_chunk) {
    if (rays_per_chunk <= 0 || rays_per_chunk > B) return B > 0 ? B : 1;
    return rays_per_chunk;
}

}  // namespace snerf

extern "C" int64_t snerf_nerf_train_workspace_bytes(const snerf_mlp_desc *desc_coarse, const snerf_mlp_desc *desc_fine, int64_t B,
                                                    int Nc, int Nf, int64_t rays_per_chunk) {
    using namespace snerf;
    if (!desc_coarse || B < 0 || Nc < 1 || Nf < 0 || (Nf > 0 && !desc_fine))
        return fail(SNERF_E_BADARG, "nerf_train_workspace_bytes: bad arguments");
    TrainWs w{};
    if (int rc = train_ws(desc_coarse, desc_fine, effective_chunk(B, rays_per_chunk), Nc, Nf, w)) return rc;
    return w.total;
}

namespace snerf {
// dp_comm.hip
int dp_allreduce_avg(snerf_comm_t comm, float *buf, int64_t begin, int64_t end, int64_t skip_begin, int64_t skip_end, hipStream_t stream,
                     const char *what);
}

// comm != NULL (the data-parallel step): the flat gradient buffer flat_g[0 .. flat_n), which holds grad_coarse / grad_fine as
// segments, is averaged over the ranks behind the last chunk's backward - the coarse net's segment on the auxiliary stream beside
// the fine net's backward when the two run concurrently, the rest on `stream` behind the join
static int nerf_train_grads_impl(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                 const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine,
                                 int precision, const snerf_nerf_batch *batch, int64_t rays_per_chunk, void *workspace,
                                 float *grad_coarse, float *grad_fine, float *loss, float *rgb, float *rgb_fine,
                                 snerf_stream_t stream, snerf_stream_t aux_stream, snerf_comm_t comm, float *flat_g, int64_t flat_n,
                                 const snerf_input_grads *ig = nullptr) {
    using namespace snerf;
    if (precision != 0 && !split_code(precision))
        return fail(SNERF_E_BADARG, "nerf_train_grads: precision must be 0 (fp32), 2 (bf16x3), 3 (bf16x6) or 16 (f16x3)");
    if (!batch) return fail(SNERF_E_BADARG, "nerf_train_grads: batch is null");
    const int64_t B = batch->B;
    const int Nc = batch->Nc, Nf = batch->Nf, N = Nc + Nf;
    if (B < (comm ? 0 : 1) || Nc < 1 || Nf < 0) return fail(SNERF_E_BADARG, "nerf_train_grads: need B >= 1, Nc >= 1, Nf >= 0");
    if (comm && B == 0) {
        // a rank whose shard ran out (RayBatchLoader allows unequal shards) still takes part in the step's collectives, in the
        // order every other rank issues them, with a zero gradient
        if (!flat_g || flat_n < 1 || !desc_coarse || !grad_coarse || !loss) return fail(SNERF_E_BADARG, "nerf_train_step_dp: null pointer");
        const int64_t pc = snerf_mlp_param_floats(desc_coarse);
        const int64_t c0 = grad_coarse - flat_g;
        if (pc < 1 || c0 < 0 || c0 + pc > flat_n) return fail(SNERF_E_BADARG, "nerf_train_step_dp: grad_coarse must lie inside adam->grads[0 .. n_params)");
        hipStream_t s0 = (hipStream_t)stream;
        if (hipMemsetAsync(flat_g, 0, (size_t)flat_n * sizeof(float), s0) != hipSuccess ||
            hipMemsetAsync(loss, 0, 3 * sizeof(float), s0) != hipSuccess)
            return fail(SNERF_E_LAUNCH, "nerf_train_step_dp: memset failed");
        if (int rc0 = dp_allreduce_avg(comm, flat_g, c0, c0 + pc, 0, 0, s0, "nerf_train_step_dp")) return rc0;
        return dp_allreduce_avg(comm, flat_g, 0, flat_n, c0, c0 + pc, s0, "nerf_train_step_dp");
    }
    if (!desc_coarse || !packed_coarse || !packed_t_coarse || !batch->ray_samples || !batch->rays_d || !batch->z_vals ||
        !batch->rgb_truth || !workspace || !grad_coarse || !loss || !rgb || !rgb_fine)
        return fail(SNERF_E_BADARG, "nerf_train_grads: null pointer");
    if (Nf > 0 && (!desc_fine || !packed_fine || !packed_t_fine || !grad_fine || !batch->rays_o || !batch->u))
        return fail(SNERF_E_BADARG, "nerf_train_grads: the fine pass needs desc_fine, its streams, grad_fine, rays_o and u");
    const int add_dim = desc_coarse->add_dim;
    if (Nf > 0 && desc_fine->add_dim != add_dim)
        return fail(SNERF_E_BADARG, "nerf_train_grads: both nets must read the same per-ray additional inputs (add_dim %d vs %d)", add_dim,
                    desc_fine->add_dim);
    if (add_dim && !batch->additional) return fail(SNERF_E_BADARG, "nerf_train_grads: the nets have additional inputs but batch->additional is null");
    float *d_add = ig ? ig->d_additional : nullptr;
    if (d_add && (!add_dim || !ig->params_coarse || (Nf > 0 && !ig->params_fine)))
        return fail(SNERF_E_BADARG, "nerf_train_grads: d_additional needs nets with additional inputs and their parameters (params_coarse / params_fine)");
    if (!aligned(workspace, 256)) return fail(SNERF_E_ALIGN, "nerf_train_grads: workspace must be 256-byte aligned");
    if (precision != 0 && (desc_coarse->width != 256 || (Nf > 0 && desc_fine->width != 256)))
        return fail(SNERF_E_BADARG, "nerf_train_grads: the split-precision kernels exist for width 256");
    const int64_t chunk = effective_chunk(B, rays_per_chunk);
    TrainWs w{};
    int rc;
    if ((rc = train_ws(desc_coarse, desc_fine, chunk, Nc, Nf, w))) return rc;
    char *ws = reinterpret_cast<char *>(workspace);
    auto f = [&](int64_t off) { return reinterpret_cast<float *>(ws + off); };
    float *raw_c = f(w.raw_c), *weights_c = f(w.weights_c), *z_fine = f(w.z_fine), *pts_f = f(w.pts_f), *raw_f = f(w.raw_f);
    float *d_rgb_c = f(w.d_rgb_c), *d_rgb_f = f(w.d_rgb_f), *d_raw = f(w.d_raw), *act_c = f(w.act_c), *act_f = f(w.act_f);
    float *dy = f(w.dy), *gpart = f(w.gpart), *cscratch = f(w.contract);
    // small chunks with an auxiliary stream: the coarse net's backward beside the fine net's, on its own scratch buffers
    const bool concurrent = w.concurrent && aux_stream && aux_stream != stream;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    if (concurrent && (rc = fork_join_events(ev_fork, ev_join))) return rc;
    float *d_raw_c = concurrent ? f(w.d_raw2) : d_raw, *dy_c = concurrent ? f(w.dy2) : dy, *gpart_c = concurrent ? f(w.gpart2) : gpart;
    const snerf_stream_t stream_c = concurrent ? aux_stream : stream;
    // (data-parallel step) the coarse net's gradient as a segment [cb0, cb1) of the flat buffer
    int64_t cb0 = 0, cb1 = 0;
    bool coarse_bucket = false;
    if (comm) {
        if (!flat_g || flat_n < 1) return fail(SNERF_E_BADARG, "nerf_train_step_dp: no flat gradient buffer");
        const int64_t pc = snerf_mlp_param_floats(desc_coarse);
        cb0 = grad_coarse - flat_g;
        cb1 = cb0 + pc;
        coarse_bucket = pc > 0 && cb0 >= 0 && cb1 <= flat_n;
        if (!coarse_bucket) return fail(SNERF_E_BADARG, "nerf_train_step_dp: grad_coarse must lie inside adam->grads[0 .. n_params)");
        const int64_t pf = Nf > 0 ? snerf_mlp_param_floats(desc_fine) : 0;
        if (Nf > 0 && (grad_fine - flat_g < 0 || grad_fine - flat_g + pf > flat_n))
            return fail(SNERF_E_BADARG, "nerf_train_step_dp: grad_fine must lie inside adam->grads[0 .. n_params)");
    }
    double *loss_acc = reinterpret_cast<double *>(ws + w.loss_acc);
    hipStream_t s = (hipStream_t)stream;
    const int wb = batch->white_background ? 1 : 0;
    const float norm = (float)(2.0 / (3.0 * (double)B));      // mse_loss_backward: 2 / numel
    const double inv_total = 1.0 / (3.0 * (double)B);
    auto fwd_train = [&](const snerf_mlp_desc *d, const void *packed, const float *x, const float *dirs, const float *add, int64_t n,
                         int spr, float *raw, float *act) {
        if (precision == 0)
            return snerf_mlp_fwd_train_f32(d, reinterpret_cast<const float *>(packed), x, dirs, 0, add, n, spr, raw, act, stream);
        return snerf_mlp_fwd_train_bf16_f32(d, packed, precision, x, dirs, 0, add, n, spr, raw, act, stream);
    };
    auto bwd = [&](const snerf_mlp_desc *d, const void *packed_t, const float *act, const float *d_raw_, int64_t n, float *dy_,
                   float *gpart_, float *grad, bool accumulate, snerf_stream_t st, int64_t n_beside) {
        if (precision == 0)
            return launch_bwd(d, reinterpret_cast<const float *>(packed_t), act, d_raw_, n, dy_, gpart_, grad, nullptr, nullptr, 0, 1,
                              nullptr, nullptr, st, accumulate, false, n_beside);
        return launch_bwd_bf16(d, packed_t, precision, act, d_raw_, n, dy_, gpart_, grad, nullptr, nullptr, 0, 1, nullptr, nullptr,
                               st, accumulate);
    };
    // (the weight-gradient launches of a chunk small enough for the concurrent backward split their jobs for each net's share of the
    // chip - by the size rule alone, with or without an auxiliary stream: the summation order, hence every bit of the gradients,
    // does not depend on the streams the caller brings)
    for (int64_t r0 = 0; r0 < B; r0 += chunk) {
        const int64_t b = (B - r0 < chunk) ? B - r0 : chunk;
        const float *x = batch->ray_samples + r0 * Nc * 3, *d = batch->rays_d + r0 * 3, *z = batch->z_vals + r0 * Nc;
        const float *gt = batch->rgb_truth + r0 * 3;
        const float *nz_c = batch->noise_coarse ? batch->noise_coarse + r0 * Nc : nullptr;
        const float *nz_f = batch->noise_fine ? batch->noise_fine + r0 * N : nullptr;
        float *rgb_c = rgb + r0 * 3, *rgb_fo = rgb_fine + r0 * 3;
        const float *add = add_dim ? batch->additional + r0 * add_dim : nullptr;   // per-ray constants (append_smpl_params_pipeline.py:29-52)
        // forward (models/nerf_pipeline.py:29-65) with every layer input saved
        if ((rc = fwd_train(desc_coarse, packed_coarse, x, d, add, b * Nc, Nc, raw_c, act_c))) return rc;
        if ((rc = snerf_composite_fwd_f32(raw_c, z, d, 0, nz_c, b, Nc, wb, rgb_c, Nf > 0 ? weights_c : nullptr, nullptr, stream))) return rc;
        if (Nf > 0) {
            if ((rc = snerf_sample_pdf_f32(z, weights_c, batch->u, batch->rays_o + r0 * 3, d, b, Nc, Nf, nullptr, nullptr, z_fine,
                                           pts_f, stream)))
                return rc;
            if ((rc = fwd_train(desc_fine, packed_fine, pts_f, d, add, b * N, N, raw_f, act_f))) return rc;
            if ((rc = snerf_composite_fwd_f32(raw_f, z_fine, d, 0, nz_f, b, N, wb, rgb_fo, nullptr, nullptr, stream))) return rc;
        }
        // loss value and d loss / d rgb (solver/nerf_solver.py:48-52, 85-86)
        hipLaunchKernelGGL(mse_grad_kernel, dim3(1), dim3(MSE_THREADS), 0, s, rgb_c, Nf > 0 ? rgb_fo : nullptr, gt, b * 3, norm,
                           d_rgb_c, d_rgb_f, loss_acc, r0 == 0 ? 1 : 0, r0 + b >= B ? 1 : 0, inv_total, loss);
        if ((rc = check_launch("nerf_train_grads(mse)"))) return rc;
        // backward: compositing, then dgrad + wgrad + reduce of each net (the hierarchical samples are detached, utils.py:260)
        if (concurrent && (hipEventRecord(ev_fork, s) != hipSuccess || hipStreamWaitEvent((hipStream_t)aux_stream, ev_fork, 0) != hipSuccess))
            return fail(SNERF_E_LAUNCH, "nerf_train_grads: cannot fork onto the auxiliary stream");
        if (Nf > 0) {
            if ((rc = snerf_composite_bwd_f32(raw_f, z_fine, d, 0, nz_f, b, N, wb, d_rgb_f, d_raw, nullptr, stream))) return rc;
            if ((rc = bwd(desc_fine, packed_t_fine, act_f, d_raw, b * N, dy, gpart, grad_fine, r0 > 0, stream, w.concurrent ? b * Nc : 0))) return rc;
            if (d_add && (rc = contract_additional(desc_fine, ig->params_fine, dy, b * N, N, d_add + r0 * add_dim, true, cscratch, stream))) return rc;
        }
        if ((rc = snerf_composite_bwd_f32(raw_c, z, d, 0, nz_c, b, Nc, wb, d_rgb_c, d_raw_c, nullptr, stream_c))) return rc;
        if ((rc = bwd(desc_coarse, packed_t_coarse, act_c, d_raw_c, b * Nc, dy_c, gpart_c, grad_coarse, r0 > 0, stream_c, w.concurrent ? b * N : 0))) return rc;
        // The collectives of the step are the same two on every rank, whatever its batch size, chunking and streams (RCCL matches
        // collectives by issue order and size; ADVICE r05: the concurrent form depends on the rank's own B): first the coarse net's
        // segment [cb0, cb1), then the rest of the flat buffer - both on `stream`, behind the join.
        const bool last = r0 + b >= B;
        if (concurrent && (hipEventRecord(ev_join, (hipStream_t)aux_stream) != hipSuccess || hipStreamWaitEvent(s, ev_join, 0) != hipSuccess))
            return fail(SNERF_E_LAUNCH, "nerf_train_grads: cannot join the auxiliary stream");
        // (r06, late: both on `stream`, behind the join.  r05 put the coarse bucket on the auxiliary stream beside the fine net's
        // backward - two collectives of ONE communicator in flight on two streams.  That has never run on more than one rank, it buys
        // the overlap of a 2.4 MB all-reduce, and it is the one pattern of the step RCCL does not promise to order for us: one stream.)
        if (comm && last && (rc = dp_allreduce_avg(comm, flat_g, cb0, cb1, 0, 0, s, "nerf_train_step_dp"))) return rc;
        if (comm && last && (rc = dp_allreduce_avg(comm, flat_g, 0, flat_n, cb0, cb1, s, "nerf_train_step_dp"))) return rc;
        // the coarse net's share of d loss / d additional inputs: behind the join (the two backwards may have run side by side; the
        // coarse net's d Y buffer is its own then), added to the fine net's
        if (d_add && (rc = contract_additional(desc_coarse, ig->params_coarse, dy_c, b * Nc, Nc, d_add + r0 * add_dim, Nf == 0, cscratch, stream)))
            return rc;
    }
    if (Nf == 0 && rgb_fine != rgb &&
        hipMemcpyAsync(rgb_fine, rgb, (size_t)B * 3 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return fail(SNERF_E_LAUNCH, "nerf_train_grads: device copy failed");
    return SNERF_OK;
}

extern "C" int snerf_nerf_train_grads_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                          const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine,
                                          int precision, const snerf_nerf_batch *batch, int64_t rays_per_chunk, void *workspace,
                                          float *grad_coarse, float *grad_fine, float *loss, float *rgb, float *rgb_fine,
                                          snerf_stream_t stream, snerf_stream_t aux_stream) {
    return nerf_train_grads_impl(desc_coarse, packed_coarse, packed_t_coarse, desc_fine, packed_fine, packed_t_fine, precision, batch,
                                 rays_per_chunk, workspace, grad_coarse, grad_fine, loss, rgb, rgb_fine, stream, aux_stream, nullptr, nullptr, 0);
}

extern "C" int snerf_nerf_train_grads_ig_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                             const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine,
                                             int precision, const snerf_nerf_batch *batch, int64_t rays_per_chunk, void *workspace,
                                             float *grad_coarse, float *grad_fine, float *loss, float *rgb, float *rgb_fine,
                                             const snerf_input_grads *input_grads, snerf_stream_t stream, snerf_stream_t aux_stream) {
    return nerf_train_grads_impl(desc_coarse, packed_coarse, packed_t_coarse, desc_fine, packed_fine, packed_t_fine, precision, batch,
                                 rays_per_chunk, workspace, grad_coarse, grad_fine, loss, rgb, rgb_fine, stream, aux_stream, nullptr, nullptr, 0,
                                 input_grads);
}

extern "C" int snerf_nerf_train_step_ig_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                            const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine,
                                            int precision, const snerf_nerf_batch *batch, int64_t rays_per_chunk, void *workspace,
                                            float *grad_coarse, float *grad_fine, float *loss, float *rgb, float *rgb_fine,
                                            const snerf_adam_state *adam, const snerf_adam_range *ranges_host, int n_ranges,
                                            const snerf_adam_net *nets_host, int n_nets, const snerf_input_grads *input_grads,
                                            snerf_stream_t stream, snerf_stream_t aux_stream) {
    // (the contraction reads the parameters: it runs inside the gradient half, before Adam moves them)
    int rc = nerf_train_grads_impl(desc_coarse, packed_coarse, packed_t_coarse, desc_fine, packed_fine, packed_t_fine, precision, batch,
                                   rays_per_chunk, workspace, grad_coarse, grad_fine, loss, rgb, rgb_fine, stream, aux_stream, nullptr, nullptr,
                                   0, input_grads);
    if (rc) return rc;
    return snerf_adam_step_f32(adam, ranges_host, n_ranges, nets_host, n_nets, stream);
}

extern "C" int snerf_nerf_train_step_dp_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                            const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine,
                                            int precision, const snerf_nerf_batch *batch, int64_t rays_per_chunk, void *workspace,
                                            float *grad_coarse, float *grad_fine, float *loss, float *rgb, float *rgb_fine,
                                            const snerf_adam_state *adam, const snerf_adam_range *ranges_host, int n_ranges,
                                            const snerf_adam_net *nets_host, int n_nets, snerf_comm_t comm, snerf_stream_t stream,
                                            snerf_stream_t aux_stream) {
    using namespace snerf;
    if (!comm) return fail(SNERF_E_BADARG, "nerf_train_step_dp: comm is null (the single-GPU step is snerf_nerf_train_step_f32)");
    if (!adam || !adam->grads || adam->n_params < 1) return fail(SNERF_E_BADARG, "nerf_train_step_dp: adam / adam->grads is null");
    int rc = nerf_train_grads_impl(desc_coarse, packed_coarse, packed_t_coarse, desc_fine, packed_fine, packed_t_fine, precision, batch,
                                   rays_per_chunk, workspace, grad_coarse, grad_fine, loss, rgb, rgb_fine, stream, aux_stream, comm,
                                   const_cast<float *>(adam->grads), adam->n_params);
    if (rc) return rc;
    return snerf_adam_step_f32(adam, ranges_host, n_ranges, nets_host, n_nets, stream);
}

extern "C" int snerf_nerf_train_step_dp_ig_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                               const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine,
                                               int precision, const snerf_nerf_batch *batch, int64_t rays_per_chunk, void *workspace,
                                               float *grad_coarse, float *grad_fine, float *loss, float *rgb, float *rgb_fine,
                                               const snerf_adam_state *adam, const snerf_adam_range *ranges_host, int n_ranges,
                                               const snerf_adam_net *nets_host, int n_nets, const snerf_input_grads *input_grads,
                                               snerf_comm_t comm, snerf_stream_t stream, snerf_stream_t aux_stream) {
    using namespace snerf;
    if (!comm) return fail(SNERF_E_BADARG, "nerf_train_step_dp_ig: comm is null (the single-GPU step is snerf_nerf_train_step_ig_f32)");
    if (!adam || !adam->grads || adam->n_params < 1) return fail(SNERF_E_BADARG, "nerf_train_step_dp_ig: adam / adam->grads is null");
    // (the contraction reads the parameters and the stored d Y: inside the gradient half, before the average and before Adam)
    int rc = nerf_train_grads_impl(desc_coarse, packed_coarse, packed_t_coarse, desc_fine, packed_fine, packed_t_fine, precision, batch,
                                   rays_per_chunk, workspace, grad_coarse, grad_fine, loss, rgb, rgb_fine, stream, aux_stream, comm,
                                   const_cast<float *>(adam->grads), adam->n_params, input_grads);
    if (rc) return rc;
    return snerf_adam_step_f32(adam, ranges_host, n_ranges, nets_host, n_nets, stream);
}

extern "C" int snerf_nerf_train_step_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                         const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine,
                                         int precision, const snerf_nerf_batch *batch, int64_t rays_per_chunk, void *workspace,
                                         float *grad_coarse, float *grad_fine, float *loss, float *rgb, float *rgb_fine,
                                         const snerf_adam_state *adam, const snerf_adam_range *ranges_host, int n_ranges,
                                         const snerf_adam_net *nets_host, int n_nets, snerf_stream_t stream, snerf_stream_t aux_stream) {
    int rc = snerf_nerf_train_grads_f32(desc_coarse, packed_coarse, packed_t_coarse, desc_fine, packed_fine, packed_t_fine, precision,
                                        batch, rays_per_chunk, workspace, grad_coarse, grad_fine, loss, rgb, rgb_fine, stream, aux_stream);
    if (rc) return rc;
    return snerf_adam_step_f32(adam, ranges_host, n_ranges, nets_host, n_nets, stream);
}

// ------------------------------------------------------------------------------------------------
// SmplNerfSolver.train's per-batch body (solver/smpl_nerf_solver.py:76-89 with the default loss; models/smpl_nerf_pipeline.py:16-100)
// ------------------------------------------------------------------------------------------------
namespace snerf {

// out = a + b (+ c): d loss / d warp = what arrives at the warped samples through the net's positions, through its
// per-sample view directions x' - o (and, in the coarse stage, through the compositing's distance scale |x' - o|)
__global__ __launch_bounds__(256) void sum3_kernel(const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ c,
                                                   int64_t n, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = __fadd_rn(a[i], b[i]);
    if (c) v = __fadd_rn(v, c[i]);
    out[i] = v;
}

struct SmplTrainWs {
    TrainWs base;
    int64_t warp_c, warped_c, sdirs_c, warp_f, warped_f, sdirs_f, act_wc, act_wf, d_x, d_dirs, d_cdirs, d_warp, dy_w, gpart_w;
    int64_t d_x2, d_dirs2, d_warp2, dy_w2, gpart_w2, grad_warp2;   // the coarse chain's own set (base.concurrent)
    int64_t total;
};

// a += b, element by element (the coarse chain's warp-net gradient joins the fine chain's in the order of the sequential form)
__global__ __launch_bounds__(256) void add_inplace_kernel(float *__restrict__ a, const float *__restrict__ b, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = __fadd_rn(a[i], b[i]);
}

static int smpl_train_ws(const snerf_mlp_desc *dc, const snerf_mlp_desc *df, const snerf_warp_desc *dw, int64_t chunk, int Nc, int Nf,
                         SmplTrainWs &w) {
    int rc;
    // small chunks: the coarse chain (compositing, net, warp net) has its own scratch set and may run beside the fine chain; both end
    // in the warp net's gradient, so the coarse chain's goes to a buffer of its own that is added behind the join
    if ((rc = train_ws(dc, df, chunk, Nc, Nf, w.base, true, SMPL_CONCURRENT_MAX_FINE_SAMPLES))) return rc;
    const int64_t N = Nc + Nf, nmax = chunk * (Nf > 0 ? N : Nc);
    int64_t act_c = 0, act_f = 0, dy_c = 0, dy_f = 0, gp_c = 0, gp_f = 0;
    if ((rc = snerf_warp_train_sizes(dw, chunk * Nc, &act_c, &dy_c, nullptr, &gp_c))) return rc;
    if (Nf > 0 && (rc = snerf_warp_train_sizes(dw, chunk * N, &act_f, &dy_f, nullptr, &gp_f))) return rc;
    int64_t off = align256(w.base.total);
    auto take = [&](int64_t floats) {
        const int64_t o = off;
        off += align256(floats * 4);
        return o;
    };
    w.warp_c = take(chunk * Nc * 3);
    w.warped_c = take(chunk * Nc * 3);
    w.sdirs_c = take(chunk * Nc * 3);
    w.warp_f = take(Nf > 0 ? chunk * N * 3 : 0);
    w.warped_f = take(Nf > 0 ? chunk * N * 3 : 0);
    w.sdirs_f = take(Nf > 0 ? chunk * N * 3 : 0);
    w.act_wc = take(act_c);
    w.act_wf = take(act_f);
    w.d_x = take(nmax * 3);
    w.d_dirs = take(nmax * 3);
    w.d_cdirs = take(chunk * Nc * 3);
    w.d_warp = take(nmax * 3);
    w.dy_w = take(dy_c > dy_f ? dy_c : dy_f);
    w.gpart_w = take(gp_c > gp_f ? gp_c : gp_f);
    const bool two = w.base.concurrent;
    const int64_t nw = snerf_warp_param_floats(dw);
    if (nw < 0) return (int)nw;
    w.d_x2 = take(two ? chunk * Nc * 3 : 0);
    w.d_dirs2 = take(two ? chunk * Nc * 3 : 0);
    w.d_warp2 = take(two ? chunk * Nc * 3 : 0);
    w.dy_w2 = take(two ? dy_c : 0);
    w.gpart_w2 = take(two ? gp_c : 0);
    w.grad_warp2 = take(two ? nw : 0);
    w.total = off;
    return SNERF_OK;
}

}  // namespace snerf

extern "C" int64_t snerf_smpl_nerf_train_workspace_bytes(const snerf_mlp_desc *desc_coarse, const snerf_mlp_desc *desc_fine,
                                                         const snerf_warp_desc *desc_warp, int64_t B, int Nc, int Nf,
                                                         int64_t rays_per_chunk) {
    using namespace snerf;
    if (!desc_coarse || !desc_warp || B < 0 || Nc < 1 || Nf < 0 || (Nf > 0 && !desc_fine))
        return fail(SNERF_E_BADARG, "smpl_nerf_train_workspace_bytes: bad arguments");
    SmplTrainWs w{};
    if (int rc = smpl_train_ws(desc_coarse, desc_fine, desc_warp, effective_chunk(B, rays_per_chunk), Nc, Nf, w)) return rc;
    return w.total;
}

// aux_stream (may be NULL): small chunks run the coarse chain there beside the fine chain on `stream` - same results either way
static int smpl_nerf_train_grads_impl(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                      const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine,
                                      const snerf_warp_desc *desc_warp, const float *packed_warp, const float *packed_t_warp,
                                      int precision, const snerf_nerf_batch *batch, const float *pose_enc, int64_t rays_per_chunk,
                                      void *workspace, float *grad_coarse, float *grad_fine, float *grad_warp, float *loss, float *rgb,
                                      float *rgb_fine, snerf_stream_t stream, snerf_stream_t aux_stream) {
    using namespace snerf;
    if (precision != 0 && !split_code(precision))
        return fail(SNERF_E_BADARG, "smpl_nerf_train_grads: precision must be 0 (fp32), 2 (bf16x3), 3 (bf16x6) or 16 (f16x3)");
    if (!batch) return fail(SNERF_E_BADARG, "smpl_nerf_train_grads: batch is null");
    const int64_t B = batch->B;
    const int Nc = batch->Nc, Nf = batch->Nf, N = Nc + Nf;
    if (B < 1 || Nc < 1 || Nf < 0) return fail(SNERF_E_BADARG, "smpl_nerf_train_grads: need B >= 1, Nc >= 1, Nf >= 0");
    if (!desc_coarse || !packed_coarse || !packed_t_coarse || !desc_warp || !packed_warp || !packed_t_warp || !batch->ray_samples ||
        !batch->rays_o || !batch->rays_d || !batch->z_vals || !batch->rgb_truth || !pose_enc || !workspace || !grad_coarse || !grad_warp ||
        !loss || !rgb || !rgb_fine)
        return fail(SNERF_E_BADARG, "smpl_nerf_train_grads: null pointer");
    if (Nf > 0 && (!desc_fine || !packed_fine || !packed_t_fine || !grad_fine || !batch->u))
        return fail(SNERF_E_BADARG, "smpl_nerf_train_grads: the fine pass needs desc_fine, its streams, grad_fine and u");
    if (desc_coarse->add_dim || (Nf > 0 && desc_fine->add_dim)) return fail(SNERF_E_BADARG, "smpl_nerf_train_grads: nets with additional inputs are not covered");
    if (!desc_coarse->use_dir || (Nf > 0 && !desc_fine->use_dir))
        return fail(SNERF_E_BADARG, "smpl_nerf_train_grads: the nets read per-sample view directions (use_dir = 1)");
    if (!aligned(workspace, 256)) return fail(SNERF_E_ALIGN, "smpl_nerf_train_grads: workspace must be 256-byte aligned");
    if (precision != 0 && (desc_coarse->width != 256 || (Nf > 0 && desc_fine->width != 256)))
        return fail(SNERF_E_BADARG, "smpl_nerf_train_grads: the split-precision kernels exist for width 256");
    const int64_t chunk = effective_chunk(B, rays_per_chunk);
    SmplTrainWs w{};
    int rc;
    if ((rc = smpl_train_ws(desc_coarse, desc_fine, desc_warp, chunk, Nc, Nf, w))) return rc;
    char *ws = reinterpret_cast<char *>(workspace);
    auto f = [&](int64_t off) { return reinterpret_cast<float *>(ws + off); };
    const TrainWs &b0 = w.base;
    float *raw_c = f(b0.raw_c), *weights_c = f(b0.weights_c), *z_fine = f(b0.z_fine), *pts_f = f(b0.pts_f), *raw_f = f(b0.raw_f);
    float *d_rgb_c = f(b0.d_rgb_c), *d_rgb_f = f(b0.d_rgb_f), *d_raw = f(b0.d_raw), *act_c = f(b0.act_c), *act_f = f(b0.act_f);
    float *dy = f(b0.dy), *gpart = f(b0.gpart);
    double *loss_acc = reinterpret_cast<double *>(ws + b0.loss_acc);
    float *warp_c = f(w.warp_c), *warped_c = f(w.warped_c), *sdirs_c = f(w.sdirs_c), *warp_f = f(w.warp_f), *warped_f = f(w.warped_f);
    float *sdirs_f = f(w.sdirs_f), *act_wc = f(w.act_wc), *act_wf = f(w.act_wf), *d_x = f(w.d_x), *d_dirs = f(w.d_dirs);
    float *d_cdirs = f(w.d_cdirs), *d_warp = f(w.d_warp), *dy_w = f(w.dy_w), *gpart_w = f(w.gpart_w);
    hipStream_t s = (hipStream_t)stream;
    // small chunks with an auxiliary stream: the coarse chain beside the fine chain, on its own scratch set
    const bool concurrent = b0.concurrent && aux_stream && aux_stream != stream;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    if (concurrent && (rc = fork_join_events(ev_fork, ev_join))) return rc;
    const snerf_stream_t stream_c = concurrent ? aux_stream : stream;
    float *d_raw_c = concurrent ? f(b0.d_raw2) : d_raw, *dy_c = concurrent ? f(b0.dy2) : dy, *gpart_c = concurrent ? f(b0.gpart2) : gpart;
    float *d_x_c = concurrent ? f(w.d_x2) : d_x, *d_dirs_c = concurrent ? f(w.d_dirs2) : d_dirs, *d_warp_c = concurrent ? f(w.d_warp2) : d_warp;
    float *dy_w_c = concurrent ? f(w.dy_w2) : dy_w, *gpart_w_c = concurrent ? f(w.gpart_w2) : gpart_w;
    float *grad_warp_c = concurrent ? f(w.grad_warp2) : grad_warp;
    const int64_t n_warp = snerf_warp_param_floats(desc_warp);
    if (n_warp < 0) return (int)n_warp;
    const int wb = batch->white_background ? 1 : 0;
    const float norm = (float)(2.0 / (3.0 * (double)B));
    const double inv_total = 1.0 / (3.0 * (double)B);
    const int pose_dim = desc_warp->pose_dim;
    auto fwd_train = [&](const snerf_mlp_desc *d, const void *packed, const float *x, const float *sd, int64_t n, int spr, float *raw,
                         float *act) {
        if (precision == 0)
            return snerf_mlp_fwd_train_f32(d, reinterpret_cast<const float *>(packed), x, sd, 1, nullptr, n, spr, raw, act, stream);
        return snerf_mlp_fwd_train_bf16_f32(d, packed, precision, x, sd, 1, nullptr, n, spr, raw, act, stream);
    };
    auto bwd_inputs = [&](const snerf_mlp_desc *d, const void *packed_t, const float *act, const float *d_raw_, const float *x,
                          const float *sd, int64_t n, int spr, float *dy_, float *gpart_, float *grad, float *d_x_, float *d_dirs_,
                          bool accumulate, snerf_stream_t st) {
        if (precision == 0)
            return launch_bwd(d, reinterpret_cast<const float *>(packed_t), act, d_raw_, n, dy_, gpart_, grad, x, sd, 1, spr, d_x_, d_dirs_,
                              st, accumulate, concurrent, b0.concurrent ? (spr == Nc ? n / Nc * N : n / N * Nc) : 0);
        return launch_bwd_bf16(d, packed_t, precision, act, d_raw_, n, dy_, gpart_, grad, x, sd, 1, spr, d_x_, d_dirs_, st, accumulate);
    };
    auto sum3 = [&](const float *a, const float *b, const float *c, int64_t n, float *out, snerf_stream_t st) {
        hipLaunchKernelGGL(sum3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)st, a, b, c, n, out);
        return check_launch("smpl_nerf_train_grads(sum)");
    };
    for (int64_t r0 = 0; r0 < B; r0 += chunk) {
        const int64_t b = (B - r0 < chunk) ? B - r0 : chunk;
        const float *x = batch->ray_samples + r0 * Nc * 3, *o = batch->rays_o + r0 * 3, *d = batch->rays_d + r0 * 3;
        const float *z = batch->z_vals + r0 * Nc, *gt = batch->rgb_truth + r0 * 3, *pe = pose_enc + r0 * pose_dim;
        const float *nz_c = batch->noise_coarse ? batch->noise_coarse + r0 * Nc : nullptr;
        const float *nz_f = batch->noise_fine ? batch->noise_fine + r0 * N : nullptr;
        float *rgb_c = rgb + r0 * 3, *rgb_fo = rgb_fine + r0 * 3;
        const bool acc = r0 > 0;
        // coarse stage: warp the given samples, net on (x', x' - o), compositing scaled per sample by |x' - o| (:38-63)
        if ((rc = snerf_warp_fwd_train_f32(desc_warp, packed_warp, x, pe, o, b * Nc, Nc, warp_c, warped_c, sdirs_c, act_wc, stream))) return rc;
        if ((rc = fwd_train(desc_coarse, packed_coarse, warped_c, sdirs_c, b * Nc, Nc, raw_c, act_c))) return rc;
        if ((rc = snerf_composite_fwd_f32(raw_c, z, sdirs_c, 1, nz_c, b, Nc, wb, rgb_c, Nf > 0 ? weights_c : nullptr, nullptr, stream))) return rc;
        if (Nf > 0) {   // hierarchical samples on the un-warped ray (:68), then the fine stage (:71-98)
            if ((rc = snerf_sample_pdf_f32(z, weights_c, batch->u, o, d, b, Nc, Nf, nullptr, nullptr, z_fine, pts_f, stream))) return rc;
            if ((rc = snerf_warp_fwd_train_f32(desc_warp, packed_warp, pts_f, pe, o, b * N, N, warp_f, warped_f, sdirs_f, act_wf, stream))) return rc;
            if ((rc = fwd_train(desc_fine, packed_fine, warped_f, sdirs_f, b * N, N, raw_f, act_f))) return rc;
            if ((rc = snerf_composite_fwd_f32(raw_f, z_fine, d, 0, nz_f, b, N, wb, rgb_fo, nullptr, nullptr, stream))) return rc;
        }
        hipLaunchKernelGGL(mse_grad_kernel, dim3(1), dim3(MSE_THREADS), 0, s, rgb_c, Nf > 0 ? rgb_fo : nullptr, gt, b * 3, norm,
                           d_rgb_c, d_rgb_f, loss_acc, r0 == 0 ? 1 : 0, r0 + b >= B ? 1 : 0, inv_total, loss);
        if ((rc = check_launch("smpl_nerf_train_grads(mse)"))) return rc;
        if (concurrent && (hipEventRecord(ev_fork, s) != hipSuccess || hipStreamWaitEvent((hipStream_t)aux_stream, ev_fork, 0) != hipSuccess))
            return fail(SNERF_E_LAUNCH, "smpl_nerf_train_grads: cannot fork onto the auxiliary stream");
        bool warp_acc = acc;
        if (Nf > 0) {   // fine: the compositing is scaled by the ray direction (an input), the net back-propagates into x' and x' - o
            if ((rc = snerf_composite_bwd_f32(raw_f, z_fine, d, 0, nz_f, b, N, wb, d_rgb_f, d_raw, nullptr, stream))) return rc;
            if ((rc = bwd_inputs(desc_fine, packed_t_fine, act_f, d_raw, warped_f, sdirs_f, b * N, N, dy, gpart, grad_fine, d_x, d_dirs, acc,
                                 stream)))
                return rc;
            if ((rc = sum3(d_x, d_dirs, nullptr, b * N * 3, d_warp, stream))) return rc;
            if ((rc = launch_warp_bwd(desc_warp, packed_t_warp, act_wf, d_warp, b * N, dy_w, gpart_w, grad_warp, stream, warp_acc))) return rc;
            warp_acc = true;
        }
        // coarse: the compositing's distance scale |x' - o| depends on the warp as well (:63)
        if ((rc = snerf_composite_bwd_f32(raw_c, z, sdirs_c, 1, nz_c, b, Nc, wb, d_rgb_c, d_raw_c, d_cdirs, stream_c))) return rc;
        if ((rc = bwd_inputs(desc_coarse, packed_t_coarse, act_c, d_raw_c, warped_c, sdirs_c, b * Nc, Nc, dy_c, gpart_c, grad_coarse, d_x_c,
                             d_dirs_c, acc, stream_c)))
            return rc;
        if ((rc = sum3(d_x_c, d_dirs_c, d_cdirs, b * Nc * 3, d_warp_c, stream_c))) return rc;
        if ((rc = launch_warp_bwd(desc_warp, packed_t_warp, act_wc, d_warp_c, b * Nc, dy_w_c, gpart_w_c, grad_warp_c, stream_c,
                                  concurrent ? false : warp_acc)))
            return rc;
        if (concurrent) {   // (concurrent implies Nf > 0: grad_warp holds the fine chain's sum by now)
            if (hipEventRecord(ev_join, (hipStream_t)aux_stream) != hipSuccess || hipStreamWaitEvent(s, ev_join, 0) != hipSuccess)
                return fail(SNERF_E_LAUNCH, "smpl_nerf_train_grads: cannot join the auxiliary stream");
            hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((n_warp + 255) / 256)), dim3(256), 0, s, grad_warp, grad_warp_c, n_warp);
            if ((rc = check_launch("smpl_nerf_train_grads(add)"))) return rc;
        }
    }
    if (Nf == 0 && rgb_fine != rgb &&
        hipMemcpyAsync(rgb_fine, rgb, (size_t)B * 3 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return fail(SNERF_E_LAUNCH, "smpl_nerf_train_grads: device copy failed");
    return SNERF_OK;
}

extern "C" int snerf_smpl_nerf_train_grads_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                               const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine,
                                               const snerf_warp_desc *desc_warp, const float *packed_warp, const float *packed_t_warp,
                                               int precision, const snerf_nerf_batch *batch, const float *pose_enc,
                                               int64_t rays_per_chunk, void *workspace, float *grad_coarse, float *grad_fine,
                                               float *grad_warp, float *loss, float *rgb, float *rgb_fine, snerf_stream_t stream) {
    return smpl_nerf_train_grads_impl(desc_coarse, packed_coarse, packed_t_coarse, desc_fine, packed_fine, packed_t_fine, desc_warp,
                                      packed_warp, packed_t_warp, precision, batch, pose_enc, rays_per_chunk, workspace, grad_coarse,
                                      grad_fine, grad_warp, loss, rgb, rgb_fine, stream, nullptr);
}

extern "C" int snerf_smpl_nerf_train_grads_aux_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse,
                                                   const void *packed_t_coarse, const snerf_mlp_desc *desc_fine, const void *packed_fine,
                                                   const void *packed_t_fine, const snerf_warp_desc *desc_warp, const float *packed_warp,
                                                   const float *packed_t_warp, int precision, const snerf_nerf_batch *batch,
                                                   const float *pose_enc, int64_t rays_per_chunk, void *workspace, float *grad_coarse,
                                                   float *grad_fine, float *grad_warp, float *loss, float *rgb, float *rgb_fine,
                                                   snerf_stream_t stream, snerf_stream_t aux_stream) {
    return smpl_nerf_train_grads_impl(desc_coarse, packed_coarse, packed_t_coarse, desc_fine, packed_fine, packed_t_fine, desc_warp,
                                      packed_warp, packed_t_warp, precision, batch, pose_enc, rays_per_chunk, workspace, grad_coarse,
                                      grad_fine, grad_warp, loss, rgb, rgb_fine, stream, aux_stream);
}

namespace snerf {
// a rank of a data-parallel step whose shard ran out (B == 0): zero gradient, zero loss; the caller goes on to the collective
static int dp_empty_batch(const snerf_adam_state *adam, float *loss, snerf_stream_t stream) {
    if (!loss) return fail(SNERF_E_BADARG, "smpl_nerf_train_step_dp: loss is null");
    if (hipMemsetAsync(const_cast<float *>(adam->grads), 0, (size_t)adam->n_params * sizeof(float), (hipStream_t)stream) != hipSuccess ||
        hipMemsetAsync(loss, 0, 3 * sizeof(float), (hipStream_t)stream) != hipSuccess)
        return fail(SNERF_E_LAUNCH, "smpl_nerf_train_step_dp: memset failed");
    return SNERF_OK;
}
}  // namespace snerf

// comm (may be NULL): the flat gradient buffer is averaged over the ranks between the backward and Adam
extern "C" int snerf_smpl_nerf_train_step_aux_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                                  const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine,
                                                  const snerf_warp_desc *desc_warp, float *packed_warp, float *packed_t_warp, int precision,
                                                  const snerf_nerf_batch *batch, const float *pose_enc, int64_t rays_per_chunk,
                                                  void *workspace, float *grad_coarse, float *grad_fine, float *grad_warp, float *loss,
                                                  float *rgb, float *rgb_fine, const snerf_adam_state *adam,
                                                  const snerf_adam_range *ranges_host, int n_ranges, const snerf_adam_net *nets_host,
                                                  int n_nets, int64_t warp_param_offset, snerf_comm_t comm, snerf_stream_t stream,
                                                  snerf_stream_t aux_stream) {
    using namespace snerf;
    if (comm && (!adam || !adam->grads || adam->n_params < 1))
        return fail(SNERF_E_BADARG, "smpl_nerf_train_step_aux: the data-parallel step needs adam->grads");
    int rc = (comm && batch && batch->B == 0)
                 ? dp_empty_batch(adam, loss, stream)
                 : smpl_nerf_train_grads_impl(desc_coarse, packed_coarse, packed_t_coarse, desc_fine, packed_fine, packed_t_fine, desc_warp,
                                              packed_warp, packed_t_warp, precision, batch, pose_enc, rays_per_chunk, workspace, grad_coarse,
                                              grad_fine, grad_warp, loss, rgb, rgb_fine, stream, aux_stream);
    if (rc) return rc;
    if (comm && (rc = dp_allreduce_avg(comm, const_cast<float *>(adam->grads), 0, adam->n_params, 0, 0, (hipStream_t)stream,
                                       "smpl_nerf_train_step_aux")))
        return rc;
    if ((rc = snerf_adam_step_f32(adam, ranges_host, n_ranges, nets_host, n_nets, stream))) return rc;
    return snerf_warp_repack_f32(desc_warp, adam ? adam->params : nullptr, adam ? adam->n_params : 0, warp_param_offset, packed_warp,
                                 packed_t_warp, stream);
}

extern "C" int snerf_smpl_nerf_train_step_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                              const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine,
                                              const snerf_warp_desc *desc_warp, float *packed_warp, float *packed_t_warp, int precision,
                                              const snerf_nerf_batch *batch, const float *pose_enc, int64_t rays_per_chunk,
                                              void *workspace, float *grad_coarse, float *grad_fine, float *grad_warp, float *loss,
                                              float *rgb, float *rgb_fine, const snerf_adam_state *adam,
                                              const snerf_adam_range *ranges_host, int n_ranges, const snerf_adam_net *nets_host,
                                              int n_nets, int64_t warp_param_offset, snerf_stream_t stream) {
    int rc = snerf_smpl_nerf_train_grads_f32(desc_coarse, packed_coarse, packed_t_coarse, desc_fine, packed_fine, packed_t_fine, desc_warp,
                                             packed_warp, packed_t_warp, precision, batch, pose_enc, rays_per_chunk, workspace, grad_coarse,
                                             grad_fine, grad_warp, loss, rgb, rgb_fine, stream);
    if (rc) return rc;
    if ((rc = snerf_adam_step_f32(adam, ranges_host, n_ranges, nets_host, n_nets, stream))) return rc;
    return snerf_warp_repack_f32(desc_warp, adam ? adam->params : nullptr, adam ? adam->n_params : 0, warp_param_offset, packed_warp,
                                 packed_t_warp, stream);
}

extern "C" int snerf_smpl_nerf_train_step_dp_f32(const snerf_mlp_desc *desc_coarse, const void *packed_coarse, const void *packed_t_coarse,
                                                 const snerf_mlp_desc *desc_fine, const void *packed_fine, const void *packed_t_fine,
                                                 const snerf_warp_desc *desc_warp, float *packed_warp, float *packed_t_warp, int precision,
                                                 const snerf_nerf_batch *batch, const float *pose_enc, int64_t rays_per_chunk,
                                                 void *workspace, float *grad_coarse, float *grad_fine, float *grad_warp, float *loss,
                                                 float *rgb, float *rgb_fine, const snerf_adam_state *adam,
                                                 const snerf_adam_range *ranges_host, int n_ranges, const snerf_adam_net *nets_host,
                                                 int n_nets, int64_t warp_param_offset, snerf_comm_t comm, snerf_stream_t stream) {
    using namespace snerf;
    if (!comm) return fail(SNERF_E_BADARG, "smpl_nerf_train_step_dp: comm is null (the single-GPU step is snerf_smpl_nerf_train_step_f32)");
    if (!adam || !adam->grads || adam->n_params < 1) return fail(SNERF_E_BADARG, "smpl_nerf_train_step_dp: adam / adam->grads is null");
    int rc = (batch && batch->B == 0)
                 ? dp_empty_batch(adam, loss, stream)
                 : snerf_smpl_nerf_train_grads_f32(desc_coarse, packed_coarse, packed_t_coarse, desc_fine, packed_fine, packed_t_fine, desc_warp,
                                                   packed_warp, packed_t_warp, precision, batch, pose_enc, rays_per_chunk, workspace,
                                                   grad_coarse, grad_fine, grad_warp, loss, rgb, rgb_fine, stream);
    if (rc) return rc;
    if ((rc = dp_allreduce_avg(comm, const_cast<float *>(adam->grads), 0, adam->n_params, 0, 0, (hipStream_t)stream, "smpl_nerf_train_step_dp")))
        return rc;
    if ((rc = snerf_adam_step_f32(adam, ranges_host, n_ranges, nets_host, n_nets, stream))) return rc;
    return snerf_warp_repack_f32(desc_warp, adam->params, adam->n_params, warp_param_offset, packed_warp, packed_t_warp, stream);
}

extern "C" int snerf_warp_repack_f32(const snerf_warp_desc *desc_warp, const float *params, int64_t n_params, int64_t warp_param_offset,
                                     float *packed_warp, float *packed_t_warp, snerf_stream_t stream) {
    using namespace snerf;
    const int64_t nw = snerf_warp_param_floats(desc_warp);
    if (nw < 0) return (int)nw;
    if (!params || warp_param_offset < 0 || warp_param_offset + nw > n_params)
        return fail(SNERF_E_BADARG, "warp_repack: the warp net does not lie inside the flat parameter buffer");
    int rc;
    if (packed_warp && (rc = snerf_warp_pack_f32(desc_warp, params + warp_param_offset, packed_warp, stream))) return rc;
    if (packed_t_warp && (rc = snerf_warp_pack_t_f32(desc_warp, params + warp_param_offset, packed_t_warp, stream))) return rc;
    return SNERF_OK;
}
