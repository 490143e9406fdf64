// nn.Linear as stand-alone fp32 MFMA GEMMs - the layer-by-layer path for widths the fused kernels do not cover.
//
// config_parser.py:20,24,30 accept ANY --netwidth / --netwidth_fine / --netwidth_warp.  The fused kernels keep a layer chain in
// registers (mlp_plan.h: RenderRayNet up to 512 features, WarpFieldNet up to 256); above that the nets run one nn.Linear at a
// time with the activations in HBM (models/render_ray_net.py:42-61 and models/warp_field_net.py:17-21 are exactly that on the
// reference's side): y = act(x W^T + b) and its two gradients, on the weights IN THE REFERENCE'S OWN LAYOUT ([out, in] row-major
// slices of the flat parameter vector: no packed stream), any leading dimensions, so a skip layer reads its two input blocks from
// two tensors into one accumulator (accumulate = 1) and a column block of a wider weight matrix is addressed by ldw.
//
//   forward      y[n, m]   (+)= x[n, k] w[m, k]^T (+ b) (relu)        A = x   (row, red) = x[row ldx + red],  B = w  (col, red) = w[col ldw + red]
//   dgrad        dx[n, k]  (+)= dy[n, m] w[m, k]                      A = dy,                                  B = w  (col, red) = w[red ldw + col]
//   wgrad        dw[m, k]  (+)= sum_s dy[s, m] x[s, k], db = sum_s dy A = dy  (row, red) = dy[red ldy + row], B = x  (col, red) = x[red ldx + col]
//
// One kernel, v_mfma_f32_16x16x4_f32 (exact fp32 like every other kernel of the library): a workgroup of 4 waves owns a 64 x 64 tile
// of C, each wave 32 x 32 (2 x 2 MFMA tiles); the reduction runs in steps of 16 through LDS ([row][16 + 4 pad] floats: the MFMA
// operand reads are bank-conflict-free), the next step's global loads in flight in registers while the current one multiplies.
// The wgrad reduces over the samples: split over gridDim.z slices whose partial tiles land in a caller scratch and are summed in
// slice order by a second kernel (deterministic; no atomics).
#include <algorithm>

#include "snerf_common.h"

namespace snerf {

typedef float lf4 __attribute__((ext_vector_type(4)));

constexpr int LIN_BM = 64, LIN_BN = 64, LIN_BK = 16, LIN_LD = LIN_BK + 4, LIN_THREADS = 256;

struct LinArgs {
    const float *a, *b;   // operands (see the table above)
    float *c;             // output, or the partial buffer of a split reduction
    const float *bias;    // per output column (forward), or null
    int64_t M, N, K;      // C is M x N, reduction length K
    int64_t lda, ldb, ldc;
    int a_t, b_t;         // 0: element (row, red) at p[row ld + red]; 1: at p[red ld + row]
    int relu, accumulate;
    int64_t k_per_slice;  // reduction elements per gridDim.z slice (multiple of LIN_BK)
    int64_t slice_stride; // floats between the partial outputs of consecutive slices (0: no split)
};

// one operand tile (64 rows x 16 reduction elements) from global memory into registers: 4 floats per thread
template <bool TRANS>
__device__ __forceinline__ lf4 lin_load(const float *p, int64_t ld, int64_t row0, int64_t rows, int64_t k0, int64_t kend, int tid, bool vec) {
    lf4 v = lf4{0.f, 0.f, 0.f, 0.f};
    if (!TRANS) {   // reduction index contiguous: thread t holds (row t / 4, red 4 (t % 4) .. + 3)
        const int64_t r = row0 + (tid >> 2), k = k0 + 4 * (tid & 3);
        if (r < rows) {
            const float *q = p + r * ld + k;
            if (vec && k + 3 < kend) v = *reinterpret_cast<const lf4 *>(q);
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (k + i < kend) v[i] = q[i];
            }
        }
    } else {        // row index contiguous: thread t holds (rows 4 (t % 16) .. + 3, red t / 16)
        const int64_t r = row0 + 4 * (tid & 15), k = k0 + (tid >> 4);
        if (k < kend) {
            const float *q = p + k * ld + r;
            if (vec && r + 3 < rows) v = *reinterpret_cast<const lf4 *>(q);
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (r + i < rows) v[i] = q[i];
            }
        }
    }
    return v;
}
template <bool TRANS>
__device__ __forceinline__ void lin_stage(float *s, lf4 v, int tid) {
    if (!TRANS) {
        *reinterpret_cast<lf4 *>(s + (tid >> 2) * LIN_LD + 4 * (tid & 3)) = v;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) s[(4 * (tid & 15) + i) * LIN_LD + (tid >> 4)] = v[i];
    }
}

template <bool AT, bool BT>
__global__ __launch_bounds__(LIN_THREADS) void linear_gemm_kernel(LinArgs G) {
    __shared__ __attribute__((aligned(16))) float sa[2][LIN_BM * LIN_LD], sb[2][LIN_BN * LIN_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;               // this wave's 32 x 32 quadrant
    const int64_t m0 = (int64_t)blockIdx.x * LIN_BM, n0 = (int64_t)blockIdx.y * LIN_BN;   // (rows on x: a frame has millions of samples)
    const int64_t kb = (int64_t)blockIdx.z * G.k_per_slice, ke = min(G.K, kb + G.k_per_slice);
    // 16-byte loads where the addresses allow them (base and leading dimension multiples of 4 floats)
    const bool va = ((reinterpret_cast<uintptr_t>(G.a) & 15) == 0) && (G.lda % 4 == 0);
    const bool vb = ((reinterpret_cast<uintptr_t>(G.b) & 15) == 0) && (G.ldb % 4 == 0);
    lf4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = lf4{0.f, 0.f, 0.f, 0.f};
    const int nsteps = kb < ke ? (int)((ke - kb + LIN_BK - 1) / LIN_BK) : 0;
    lf4 ra = lf4{0.f, 0.f, 0.f, 0.f}, rb = ra;
    if (nsteps > 0) {
        ra = lin_load<AT>(G.a, G.lda, m0, G.M, kb, ke, tid, va);
        rb = lin_load<BT>(G.b, G.ldb, n0, G.N, kb, ke, tid, vb);
        lin_stage<AT>(sa[0], ra, tid);
        lin_stage<BT>(sb[0], rb, tid);
    }
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        if (st + 1 < nsteps) {   // the next step's operands: in flight while this step multiplies
            const int64_t k0 = kb + (int64_t)(st + 1) * LIN_BK;
            ra = lin_load<AT>(G.a, G.lda, m0, G.M, k0, ke, tid, va);
            rb = lin_load<BT>(G.b, G.ldb, n0, G.N, k0, ke, tid, vb);
        }
        const float *pa = sa[cur] + (32 * wm + (lane & 15)) * LIN_LD + (lane >> 4);
        const float *pb = sb[cur] + (32 * wn + (lane & 15)) * LIN_LD + (lane >> 4);
#pragma unroll
        for (int kk = 0; kk < LIN_BK / 4; ++kk) {
            const float a0 = pa[4 * kk], a1 = pa[16 * LIN_LD + 4 * kk];
            const float b0 = pb[4 * kk], b1 = pb[16 * LIN_LD + 4 * kk];
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (st + 1 < nsteps) {
            lin_stage<AT>(sa[cur ^ 1], ra, tid);
            lin_stage<BT>(sb[cur ^ 1], rb, tid);
        }
        __syncthreads();
    }
    // D layout: acc[i][j][r] = C[m0 + 32 wm + 16 i + 4 (lane >> 4) + r][n0 + 32 wn + 16 j + (lane & 15)]
    float *c = G.c + (int64_t)blockIdx.z * G.slice_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t col = n0 + 32 * wn + 16 * j + (lane & 15);
            if (col >= G.N) continue;
            const float bv = G.bias ? G.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = m0 + 32 * wm + 16 * i + 4 * (lane >> 4) + r;
                if (row >= G.M) continue;
                float v = acc[i][j][r];
                float *q = c + row * G.ldc + col;
                if (G.accumulate) v = __fadd_rn(*q, v);
                if (G.bias) v = __fadd_rn(v, bv);
                if (G.relu) v = fmaxf(v, 0.f);
                *q = v;
            }
        }
}

// out[e] (+)= sum over the slices, in slice order; e walks an M x N matrix with leading dimension ldc
__global__ __launch_bounds__(256) void linear_slice_sum_kernel(const float *__restrict__ part, int slices, int64_t slice_stride, int64_t M,
                                                               int64_t N, int64_t ldp, float *__restrict__ out, int64_t ldc, int accumulate) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= M * N) return;
    const int64_t r = e / N, c = e - r * N;
    float s = 0.f;
    for (int z = 0; z < slices; ++z) s = __fadd_rn(s, part[(int64_t)z * slice_stride + r * ldp + c]);
    float *q = out + r * ldc + c;
    *q = accumulate ? __fadd_rn(*q, s) : s;
}

// column sums of dy [n, m] (the bias gradient): a workgroup owns 64 columns and one slice of the rows
__global__ __launch_bounds__(256) void linear_colsum_kernel(const float *__restrict__ dy, int64_t n, int64_t m, int64_t ldy, int64_t rows_per_slice,
                                                            float *__restrict__ part, int64_t slice_stride) {
    __shared__ float s[4][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + col;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_slice, r1 = min(n, r0 + rows_per_slice);
    float v = 0.f;
    if (c < m)
        for (int64_t r = r0 + grp; r < r1; r += 4) v = __fadd_rn(v, dy[r * ldy + c]);
    s[grp][col] = v;
    __syncthreads();
    if (grp == 0 && c < m) part[(int64_t)blockIdx.y * slice_stride + c] = __fadd_rn(__fadd_rn(s[0][col], s[1][col]), __fadd_rn(s[2][col], s[3][col]));
}

// ReLU backward in place: dy[i] = y[i] > 0 ? dy[i] : 0 over an n x m matrix (both with their leading dimension)
__global__ __launch_bounds__(256) void relu_mask_kernel(float *__restrict__ dy, const float *__restrict__ y, int64_t n, int64_t m, int64_t lddy, int64_t ldy) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * m) return;
    const int64_t r = e / m, c = e - r * m;
    if (!(y[r * ldy + c] > 0.f)) dy[r * lddy + c] = 0.f;
}

static int lin_launch(const LinArgs &G, int slices, hipStream_t s, const char *what) {
    const int64_t gx = (G.M + LIN_BM - 1) / LIN_BM, gy = (G.N + LIN_BN - 1) / LIN_BN;
    if (gx > 0x7fffffffLL || gy > 65535) return fail(SNERF_E_BADARG, "%s: matrix too large", what);
    if (gx == 0 || gy == 0) return SNERF_OK;
    const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)slices);
    if (!G.a_t && !G.b_t) hipLaunchKernelGGL((linear_gemm_kernel<false, false>), grid, dim3(LIN_THREADS), 0, s, G);
    else if (!G.a_t && G.b_t) hipLaunchKernelGGL((linear_gemm_kernel<false, true>), grid, dim3(LIN_THREADS), 0, s, G);
    else if (G.a_t && G.b_t) hipLaunchKernelGGL((linear_gemm_kernel<true, true>), grid, dim3(LIN_THREADS), 0, s, G);
    else hipLaunchKernelGGL((linear_gemm_kernel<true, false>), grid, dim3(LIN_THREADS), 0, s, G);
    return check_launch(what);
}

// slices of the wgrad's reduction over n samples for an m x k gradient: enough workgroups for about two rounds of the chip,
// slices of at least 256 samples
static int wgrad_slices(int64_t n, int64_t m, int64_t k, int n_cu) {
    const int64_t tiles = ((m + LIN_BM - 1) / LIN_BM) * ((k + LIN_BN - 1) / LIN_BN);
    int64_t want = (2 * (int64_t)n_cu + tiles - 1) / std::max<int64_t>(tiles, 1);
    want = std::min<int64_t>(want, (n + 255) / 256);
    return (int)std::max<int64_t>(1, std::min<int64_t>(want, 256));
}

}  // namespace snerf

extern "C" int snerf_linear_fwd_f32(const float *x, int64_t n, int k, int64_t ldx, const float *w, int64_t ldw, int m, const float *bias,
                                    int accumulate, int relu, float *y, int64_t ldy, snerf_stream_t stream) {
    using namespace snerf;
    if (n < 0 || k < 0 || m < 1 || ldx < k || ldw < k || ldy < m) return fail(SNERF_E_BADARG, "linear_fwd: bad sizes / leading dimensions");
    if (n == 0) return SNERF_OK;
    if (!x || !w || !y) return fail(SNERF_E_BADARG, "linear_fwd: null pointer");
    LinArgs G{x, w, y, bias, n, m, k, ldx, ldw, ldy, 0, 0, relu ? 1 : 0, accumulate ? 1 : 0, (k + LIN_BK - 1) / LIN_BK * LIN_BK, 0};
    if (G.k_per_slice == 0) G.k_per_slice = LIN_BK;
    return lin_launch(G, 1, (hipStream_t)stream, "linear_fwd");
}

extern "C" int snerf_linear_bwd_input_f32(const float *dy, int64_t n, int m, int64_t lddy, const float *w, int64_t ldw, int k, int accumulate,
                                          float *dx, int64_t lddx, snerf_stream_t stream) {
    using namespace snerf;
    if (n < 0 || k < 1 || m < 1 || lddy < m || ldw < k || lddx < k) return fail(SNERF_E_BADARG, "linear_bwd_input: bad sizes / leading dimensions");
    if (n == 0) return SNERF_OK;
    if (!dy || !w || !dx) return fail(SNERF_E_BADARG, "linear_bwd_input: null pointer");
    // dx[n, k] = dy[n, m] w[m, k]: reduction over m; B's (col = input feature, red = output feature) sits at w[red ldw + col]
    LinArgs G{dy, w, dx, nullptr, n, k, m, lddy, ldw, lddx, 0, 1, 0, accumulate ? 1 : 0, ((int64_t)m + LIN_BK - 1) / LIN_BK * LIN_BK, 0};
    return lin_launch(G, 1, (hipStream_t)stream, "linear_bwd_input");
}

extern "C" int64_t snerf_linear_bwd_weight_scratch_floats(int64_t n, int m, int k) {
    using namespace snerf;
    if (n < 0 || m < 1 || k < 0) return fail(SNERF_E_BADARG, "linear_bwd_weight_scratch_floats: bad sizes");
    // (sized for the largest split any device takes: 256 slices, and never more than one per 256 samples)
    const int64_t slices = std::max<int64_t>(1, std::min<int64_t>(256, (n + 255) / 256));
    return slices * ((int64_t)m * k + m);
}

extern "C" int snerf_linear_bwd_weight_f32(const float *dy, int64_t n, int m, int64_t lddy, const float *x, int64_t ldx, int k, int accumulate,
                                           float *dw, int64_t lddw, float *db, float *scratch, snerf_stream_t stream) {
    using namespace snerf;
    if (n < 0 || k < 0 || m < 1 || lddy < m || ldx < k || lddw < k) return fail(SNERF_E_BADARG, "linear_bwd_weight: bad sizes / leading dimensions");
    if (!dy || (k > 0 && (!x || !dw)) || !scratch) return fail(SNERF_E_BADARG, "linear_bwd_weight: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int n_cu = device_cu_count("linear_bwd_weight");
    if (n_cu < 1) return n_cu;
    if (n == 0) {   // an empty batch: zero gradients (unless they accumulate)
        if (!accumulate) {
            if (k > 0 && hipMemset2DAsync(dw, (size_t)lddw * 4, 0, (size_t)k * 4, (size_t)m, s) != hipSuccess) return fail(SNERF_E_LAUNCH, "linear_bwd_weight: memset failed");
            if (db && hipMemsetAsync(db, 0, (size_t)m * 4, s) != hipSuccess) return fail(SNERF_E_LAUNCH, "linear_bwd_weight: memset failed");
        }
        return SNERF_OK;
    }
    const int slices = wgrad_slices(n, m, k, n_cu);
    const int64_t per = (((n + slices - 1) / slices) + LIN_BK - 1) / LIN_BK * LIN_BK;
    const int used = (int)((n + per - 1) / per);
    // scratch: [used][m, k] partials of the weight gradient, then [cs_used][m] partials of the bias gradient - the column sums take
    // their own, finer split (they stream all of d Y through (m / 64) x slices workgroups: with the GEMM's 4 slices of a 768 x 768
    // layer that was 48 workgroups on 256 CUs, 6 % of the layer-by-layer training step)
    const int64_t stride = (int64_t)m * k;
    if (k > 0) {
        // dw[m, k] = sum_s dy[s, m] x[s, k]: A (row = output feature, red = sample) at dy[red lddy + row], B (col, red) at x[red ldx + col]
        LinArgs G{dy, x, scratch, nullptr, m, k, n, lddy, ldx, k, 1, 1, 0, 0, per, stride};
        if (int rc = lin_launch(G, used, s, "linear_bwd_weight")) return rc;
        const int64_t e = (int64_t)m * k;
        hipLaunchKernelGGL(linear_slice_sum_kernel, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, s, scratch, used, stride, (int64_t)m, (int64_t)k, (int64_t)k,
                           dw, lddw, accumulate ? 1 : 0);
        if (int rc = check_launch("linear_bwd_weight(sum)")) return rc;
    }
    if (db) {
        const int64_t col_groups = (m + 63) / 64;
        int64_t cs = std::max<int64_t>(1, std::min<int64_t>((4 * (int64_t)n_cu + col_groups - 1) / col_groups, std::min<int64_t>(256, (n + 255) / 256)));
        const int64_t cs_per = (n + cs - 1) / cs;
        const int cs_used = (int)((n + cs_per - 1) / cs_per);
        float *bias_part = scratch + (int64_t)used * stride;
        hipLaunchKernelGGL(linear_colsum_kernel, dim3((unsigned)col_groups, (unsigned)cs_used), dim3(256), 0, s, dy, n, (int64_t)m, lddy, cs_per,
                           bias_part, (int64_t)m);
        if (int rc = check_launch("linear_bwd_weight(bias)")) return rc;
        hipLaunchKernelGGL(linear_slice_sum_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, bias_part, cs_used, (int64_t)m, (int64_t)1,
                           (int64_t)m, (int64_t)m, db, (int64_t)m, accumulate ? 1 : 0);
        if (int rc = check_launch("linear_bwd_weight(bias sum)")) return rc;
    }
    return SNERF_OK;
}

extern "C" int snerf_relu_bwd_f32(float *dy, const float *y, int64_t n, int m, int64_t lddy, int64_t ldy, snerf_stream_t stream) {
    using namespace snerf;
    if (n < 0 || m < 1 || lddy < m || ldy < m) return fail(SNERF_E_BADARG, "relu_bwd: bad sizes");
    if (n == 0) return SNERF_OK;
    if (!dy || !y) return fail(SNERF_E_BADARG, "relu_bwd: null pointer");
    const int64_t e = n * m;
    if ((e + 255) / 256 > 0x7fffffffLL) return fail(SNERF_E_BADARG, "relu_bwd: too large");
    hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, y, n, (int64_t)m, lddy, ldy);
    return check_launch("relu_bwd");
}
