// Input gradients that are contractions of a stored layer-output gradient with weight columns:
//
//     out[s, c] (+)= sum_f dY_l[s, f] * W_l[f, col0 + c]            (what autograd leaves in the .grad of x when y = x W^T)
//
// - the per-ray additional inputs of the pose- / vertex-conditioned nets (models/render_ray_net.py:43-50 read them in layer
//   0 and the skip layers; models/append_vertices_pipeline.py:30-58, append_smpl_params_pipeline.py: the estimator / pose
//   upstream receives the gradient), summed over the samples of a ray because the pipelines expand one row per ray;
// - already-encoded input rows of RenderRayNet.forward(x) / WarpFieldNet.forward(x) (the reference's modules back-propagate
//   into their input);
// - the per-ray pose rows of the fused warp stage (models/smpl_nerf_pipeline.py:40-48).
//
// dY_l is what the dgrad kernels leave in `dy` (tile-row-major, snerf_mlp_dy_layout): a wave reads the 16 features x 16
// samples of a tile as one coalesced 1 KiB access, straight into the B operand of v_mfma_f32_16x16x4_f32; the weight
// columns (transposed, <= 64 per pass) sit in LDS as the A operand.  HBM-bound: 64 B per sample and k-block of dY against
// 4 x (tiles of 16 columns) MFMAs - the kernel reads every dY_l it contracts once per 128 output columns.
#include <type_traits>

#include "mlp_device.h"

namespace snerf {

constexpr int CT_WAVES = 4;              // (output columns per pass: CT_TILES tiles of 16 - 4 or 8, a template parameter)
constexpr int CT_PITCH_PAD = 4;          // W^T rows in LDS are n_feat_pad + 4 floats apart (16-byte aligned, bank-spread)

struct ContractArgs {
    const float *dy;       // tile-row-major [rows][n][16]
    int64_t n;
    int first_row;         // tile-row of feature 0 of dY_l
    int n_feat;            // features of dY_l = rows of W
    const float *w;        // [n_feat, w_stride] row-major
    int w_stride, col0, ncols;
    float *out;            // per-sample mode: [n, out_stride], columns out_col0 ..; per-ray mode: partial rows [n_rows_partial, ncols]
    int64_t out_stride;
    int out_col0;
    int accumulate;        // per-sample mode: out += instead of =
    int reduce16;          // per-ray mode with samples_per_ray % 16 == 0: one partial row per 16-sample tile (lane-reduced)
    int per_ray;           // per-ray mode (partials; dy_contract_reduce_kernel finishes)
    int64_t n_tiles;
};

// KB: k-blocks of 16 features the kernel walks (16 / 8 / 4: layers of up to 256 / 128 / 64 features; a layer with fewer
// real k-blocks re-reads its last tile-row against zero weights) - a compile-time count, so that the loads of a tile are
// straight-line code (a guarded load per k-block made every load its own basic block with a full wait behind it).
template <int CT_TILES, int KB>
__global__ __launch_bounds__(CT_WAVES * 64) void dy_contract_kernel(ContractArgs A, int c_begin) {
    constexpr int CT_COLS = CT_TILES * 16;
    extern __shared__ __attribute__((aligned(16))) float wt[];   // [CT_COLS][pitch]: W^T of this pass's columns
    const int kb_n = (A.n_feat + 15) / 16;                       // real k-blocks (<= KB)
    constexpr int pitch = KB * 16 + CT_PITCH_PAD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int nc = min(CT_COLS, A.ncols - c_begin);
    for (int e = tid; e < CT_COLS * KB * 16; e += CT_WAVES * 64) {
        const int f = e / CT_COLS, c = e - f * CT_COLS;        // consecutive threads read consecutive columns of one W row
        wt[c * pitch + f] = (c < nc && f < A.n_feat) ? A.w[(int64_t)f * A.w_stride + A.col0 + c_begin + c] : 0.f;
    }
    __syncthreads();
    // (straight-line MFMA code over all CT_TILES column tiles - columns past `nc` meet zero weights: a run-time tile count
    // inside the unrolled loops turned the kernel into 900 branches and 70 000 accumulator moves; the launcher picks the
    // smallest instantiation that covers a pass instead)
    // A wave's d Y tile (16 samples x up to 256 features = 16 KiB) is fetched as up to 16 independent 1 KiB loads, and the NEXT
    // tile's loads are issued before the current tile's MFMAs: with one load in flight per wave the kernel ran at the latency
    // of HBM (1.2 TB/s), not at its bandwidth.
    f4 cur[KB], nxt[KB];
    const int64_t stride = (int64_t)gridDim.x * CT_WAVES;
    int64_t tile = (int64_t)blockIdx.x * CT_WAVES + wave;
    auto fetch = [&](int64_t tl, f4 (&dst)[KB]) __attribute__((always_inline)) {
        const int64_t s0 = tl * 16 + i;
        const int64_t sc = s0 < A.n ? s0 : A.n - 1;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) dst[kb] = load_tile(A.dy, A.first_row + min(kb, kb_n - 1), A.n, sc, g);
    };
    if (tile < A.n_tiles) fetch(tile, nxt);
    for (; tile < A.n_tiles; tile += stride) {
        const int64_t sample = tile * 16 + i;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) cur[kb] = nxt[kb];
        fetch(min(tile + stride, A.n_tiles - 1), nxt);          // (the last iteration re-reads its own tile: no branch around the loads)
        f4 acc[CT_TILES];
#pragma unroll
        for (int t = 0; t < CT_TILES; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            f4 a[CT_TILES];
#pragma unroll
            for (int t = 0; t < CT_TILES; ++t) a[t] = *reinterpret_cast<const f4 *>(wt + (16 * t + i) * pitch + 16 * kb + 4 * g);
            // k-steps outermost: consecutive MFMAs go to different accumulators (no back-to-back dependent issue)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < CT_TILES; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][r], cur[kb][r], acc[t], 0, 0, 0);
        }
        // acc[t][r] = out^T[column 16 t + 4 g + r][sample lane & 15]
        if (!A.per_ray) {
            if (sample < A.n) {
                float *q = A.out + sample * A.out_stride + A.out_col0 + c_begin;
#pragma unroll
                for (int t = 0; t < CT_TILES; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 16 * t + 4 * g + r;
                        if (c < nc) q[c] = A.accumulate ? q[c] + acc[t][r] : acc[t][r];
                    }
            }
        } else if (A.reduce16) {     // the 16 samples of the tile belong to one ray: sum them in a fixed order
#pragma unroll
            for (int t = 0; t < CT_TILES; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = sample < A.n ? acc[t][r] : 0.f;
                    v += __shfl_xor(v, 1, 64);
                    v += __shfl_xor(v, 2, 64);
                    v += __shfl_xor(v, 4, 64);
                    v += __shfl_xor(v, 8, 64);
                    const int c = 16 * t + 4 * g + r;
                    if (i == 0 && c < nc) A.out[tile * A.ncols + c_begin + c] = v;
                }
        } else if (sample < A.n) {   // one partial row per sample
            float *q = A.out + sample * A.ncols + c_begin;
#pragma unroll
            for (int t = 0; t < CT_TILES; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * t + 4 * g + r;
                    if (c < nc) q[c] = acc[t][r];
                }
        }
    }
}

// out[ray, out_col0 + c] (+)= sum of the ray's `rows_per_ray` consecutive partial rows, in row order
__global__ __launch_bounds__(256) void dy_contract_reduce_kernel(const float *__restrict__ part, int64_t n_rays, int rows_per_ray,
                                                                 int ncols, float *__restrict__ out, int64_t out_stride, int out_col0,
                                                                 int accumulate) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_rays * ncols) return;
    const int64_t ray = e / ncols;
    const int c = (int)(e - ray * ncols);
    const float *p = part + ray * rows_per_ray * ncols + c;
    float sum = 0.f;
    for (int k = 0; k < rows_per_ray; ++k) sum += p[(int64_t)k * ncols];
    float *q = out + ray * out_stride + out_col0 + c;
    *q = accumulate ? *q + sum : sum;
}

}  // namespace snerf

extern "C" int64_t snerf_dy_contract_scratch_floats(int64_t n, int ncols, int samples_per_ray) {
    using namespace snerf;
    if (n < 0 || ncols < 1 || samples_per_ray < 0) return fail(SNERF_E_BADARG, "dy_contract_scratch_floats: bad arguments");
    if (samples_per_ray == 0) return 0;                                     // per-sample output: no partials
    return (samples_per_ray % 16 == 0 ? n / 16 : n) * (int64_t)ncols;
}

extern "C" int snerf_dy_contract_f32(const float *dy, int64_t n, int first_row, int n_feat, const float *w, int w_stride, int col0,
                                     int ncols, int samples_per_ray, float *out, int64_t out_stride, int out_col0, int accumulate,
                                     float *scratch, snerf_stream_t stream) {
    using namespace snerf;
    if (n_feat > 256 && n_feat <= 512 && w_stride >= 1) {   // a 512-feature layer (--netwidth above 256): its two 16-tile halves, the second added
        if (int rc = snerf_dy_contract_f32(dy, n, first_row, 256, w, w_stride, col0, ncols, samples_per_ray, out, out_stride, out_col0,
                                           accumulate, scratch, stream))
            return rc;
        return snerf_dy_contract_f32(dy, n, first_row + 16, n_feat - 256, w ? w + (int64_t)256 * w_stride : w, w_stride, col0, ncols,
                                     samples_per_ray, out, out_stride, out_col0, 1, scratch, stream);
    }
    if (n < 0 || first_row < 0 || n_feat < 1 || n_feat > 256 || w_stride < 1 || col0 < 0 || ncols < 1 || col0 + ncols > w_stride ||
        samples_per_ray < 0 || out_stride < 1 || out_col0 < 0 || out_col0 + ncols > out_stride)
        return fail(SNERF_E_BADARG, "dy_contract: bad shape arguments (n_feat <= 512, columns inside the weight / output rows)");
    if (samples_per_ray > 0 && n % samples_per_ray != 0) return fail(SNERF_E_BADARG, "dy_contract: n is not a multiple of samples_per_ray");
    if (n == 0) return SNERF_OK;
    if (!dy || !w || !out || (samples_per_ray > 0 && !scratch)) return fail(SNERF_E_BADARG, "dy_contract: null pointer");
    if (!aligned(dy, 16)) return fail(SNERF_E_ALIGN, "dy_contract: dy must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    ContractArgs A{};
    A.dy = dy;
    A.n = n;
    A.first_row = first_row;
    A.n_feat = n_feat;
    A.w = w;
    A.w_stride = w_stride;
    A.col0 = col0;
    A.ncols = ncols;
    A.per_ray = samples_per_ray > 0 ? 1 : 0;
    A.reduce16 = (samples_per_ray > 0 && samples_per_ray % 16 == 0) ? 1 : 0;
    A.out = A.per_ray ? scratch : out;
    A.out_stride = out_stride;
    A.out_col0 = out_col0;
    A.accumulate = accumulate ? 1 : 0;
    A.n_tiles = (n + 15) / 16;
    const int n_cu = device_cu_count("dy_contract");
    if (n_cu < 1) return n_cu;
    const int64_t wgs = (A.n_tiles + CT_WAVES - 1) / CT_WAVES;
    // passes of up to 8 column tiles (128 columns: 69 pose columns or 84 encoded columns cost ONE read of d Y), each with the
    // smallest instantiation (1 / 2 / 4 / 6 / 8 tiles) that covers it
    auto run = [&](auto tiles_c, auto kb_c, int c0) -> int {
        constexpr int TILES = decltype(tiles_c)::value, KB = decltype(kb_c)::value;
        constexpr int lds = TILES * 16 * (KB * 16 + CT_PITCH_PAD) * (int)sizeof(float);
        static LdsRaised raised;
        if (int rc = raise_dynamic_lds(reinterpret_cast<const void *>(dy_contract_kernel<TILES, KB>), lds, raised, "dy_contract")) return rc;
        const int64_t per_cu = lds <= 72 * 1024 ? 2 : 1;
        const unsigned grid = (unsigned)(wgs < per_cu * n_cu ? wgs : per_cu * n_cu);
        hipLaunchKernelGGL((dy_contract_kernel<TILES, KB>), dim3(grid), dim3(CT_WAVES * 64), lds, s, A, c0);
        return check_launch("dy_contract");
    };
    auto run_kb = [&](auto tiles_c, int c0) -> int {
        using std::integral_constant;
        if (n_feat > 128) return run(tiles_c, integral_constant<int, 16>{}, c0);
        if (n_feat > 64) return run(tiles_c, integral_constant<int, 8>{}, c0);
        return run(tiles_c, integral_constant<int, 4>{}, c0);
    };
    int rc = SNERF_OK;
    for (int c0 = 0; c0 < ncols && !rc; c0 += 128) {
        using std::integral_constant;
        const int tiles = (min(128, ncols - c0) + 15) / 16;
        if (tiles <= 1) rc = run_kb(integral_constant<int, 1>{}, c0);
        else if (tiles <= 2) rc = run_kb(integral_constant<int, 2>{}, c0);
        else if (tiles <= 4) rc = run_kb(integral_constant<int, 4>{}, c0);
        else if (tiles <= 6) rc = run_kb(integral_constant<int, 6>{}, c0);
        else rc = run_kb(integral_constant<int, 8>{}, c0);
    }
    if (rc) return rc;
    if (A.per_ray) {
        const int64_t n_rays = n / samples_per_ray, total = n_rays * ncols;
        const int rows = A.reduce16 ? samples_per_ray / 16 : samples_per_ray;
        if ((total + 255) / 256 > 0x7fffffffLL) return fail(SNERF_E_BADARG, "dy_contract: too many rays");
        hipLaunchKernelGGL(dy_contract_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, scratch, n_rays, rows, ncols,
                           out, out_stride, out_col0, accumulate ? 1 : 0);
        if (int rc = check_launch("dy_contract(reduce)")) return rc;
    }
    return SNERF_OK;
}
