// Adaptive grids of the efficient-KAN layer: what KANLinear.update_grid (node_classification_clean/ekan.py:164-211)
// and the dense b_splines (:79-112) need on the device.  KAGNN itself never calls update_grid,
// so this is the widening row, built for correctness and bounded memory rather than tuned.
//
// The reference refits the coefficients after moving the knots by materialising the old layer's per-feature
// outputs [N, in, out] and running a batched least-squares solve over A = bases_new(x) [in, N, C]
// (curve2coeff, ekan.py:114-144).  Here nothing of size N*in*C or N*in*out exists: with A_o = bases_old(x),
//   A^T (A_o W) = (A^T A_o) W ,
// so the normal equations of feature f need only two C x C Gram matrices, G = A^T A and X = A^T A_o, which one
// streaming pass over x accumulates in fp64 on v_mfma_f64_16x16x4_f64 (deterministic: per-wave partials, summed in
// a fixed order).  A second tiny kernel solves G S = X W per feature (fp64 Cholesky; a basis no sample touches has a
// zero pivot and gets coefficient 0, the minimum-norm choice).  C = G + k <= 16.
#include "common.h"

namespace kagnn {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ dense bases (b_splines, ekan.py:79-112)
template <int K>
__global__ void kan_bsplines_kernel(const float* __restrict__ x, long ldx, long N, int in, int C,
                                    const float* __restrict__ grid, int nknots, float* __restrict__ bases) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= N * in) return;
    const long n = i / in; const int f = (int)(i - n * in);
    float Nv[K + 1], dummy[K + 1];
    const int m = bspline_generic<K, false>(x[n * ldx + f], grid + (long)f * nknots, nknots, Nv, dummy);
    float* b = bases + i * C;
    for (int c = 0; c < C; ++c) b[c] = pick_basis<K>(Nv, m, c);
}

int kan_bsplines(const float* x, long ldx, long N, const float* grid, int in, int G, int K, float* bases,
                 hipStream_t st) {
    const int nk = G + 2 * K + 1, C = G + K;
    if (N * in == 0) return KAGNN_OK;
    const int blocks = cdiv(N * in, 256);
#define L(KK) kan_bsplines_kernel<KK><<<blocks, 256, 0, st>>>(x, ldx, N, in, C, grid, nk, bases)
    switch (K) {
        case 1: L(1); break;
        case 2: L(2); break;
        case 3: L(3); break;
        case 4: L(4); break;
        default: return fail(KAGNN_ERR_UNSUPPORTED, "%s: spline_order must be 1..4", "kan_bsplines");
    }
#undef L
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// ------------------------------------------------------------------ Gram matrices
// workgroup (chunk, f): four waves walk disjoint row ranges of feature f, 4 rows per MFMA (k-lane l>>4 <-> row),
// lane l feeds basis i = l&15 of its row.  slab[f][s][which][16][16], which = 0: A^T A, 1: A^T A_old.
template <int K>
__global__ __launch_bounds__(256) void kan_grid_gram_kernel(
    const float* __restrict__ x, long ldx, long N, int in, int C, const float* __restrict__ grid_old,
    const float* __restrict__ grid_new, int nknots, long rows_per_wave, double* __restrict__ slab) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int f = blockIdx.y, i = lane & 15, k = lane >> 4;
    const long s = (long)blockIdx.x * 4 + wave, S = (long)gridDim.x * 4;
    const long rbeg = s * rows_per_wave, rend = min(N, rbeg + rows_per_wave);
    const float* tn = grid_new + (long)f * nknots;
    const float* to = grid_old + (long)f * nknots;
    f64x4 gn = {0.0, 0.0, 0.0, 0.0}, gx = {0.0, 0.0, 0.0, 0.0};
    for (long n0 = rbeg; n0 < rend; n0 += 4) {
        const long n = n0 + k;
        const float xv = x[min(n, N - 1) * ldx + f];
        float Nn[K + 1], No[K + 1], dummy[K + 1];
        const int mn = bspline_generic<K, false>(xv, tn, nknots, Nn, dummy);
        const int mo = bspline_generic<K, false>(xv, to, nknots, No, dummy);
        const bool live = n < rend && i < C;
        const double a = live ? (double)pick_basis<K>(Nn, mn, i) : 0.0;
        const double b = live ? (double)pick_basis<K>(No, mo, i) : 0.0;
        gn = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, gn, 0, 0, 0);
        gx = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, gx, 0, 0, 0);
    }
    // C/D of the f64 form: col = lane & 15, row = (lane >> 4) + 4 * reg
    double* o = slab + (((long)f * S + s) * 2) * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        o[(k + 4 * r) * 16 + i] = gn[r];
        o[256 + (k + 4 * r) * 16 + i] = gx[r];
    }
}

// gram[f][which][16][16] = sum_s slab[f][s][which][..] in slab order
__global__ void kan_grid_gram_reduce_kernel(const double* __restrict__ slab, long S, double* __restrict__ gram) {
    const int f = blockIdx.x, t = threadIdx.x;            // 512 threads = 2 x 256 entries
    const double* p = slab + (long)f * S * 512 + t;
    double a = 0.0;
    for (long s = 0; s < S; ++s) a += p[s * 512];
    gram[(long)f * 512 + t] = a;
}

// ------------------------------------------------------------------ per-feature solve  G S = X W
// W[c][o] = spline_weight[o][f][c] * scaler[o][f]  (scaled_spline_weight, ekan.py:146-152); the fitted S is
// written to new_sw[o][f][c] unscaled, as ekan.py:211 does.
__global__ __launch_bounds__(256) void kan_grid_solve_kernel(const double* __restrict__ gram, int in, int out, int C,
                                                             const float* __restrict__ sw,
                                                             const float* __restrict__ sc,
                                                             float* __restrict__ new_sw) {
    __shared__ double L[16][17];
    __shared__ double X[16][17];
    __shared__ int dead[16];
    const int f = blockIdx.x, t = threadIdx.x;
    const double* g = gram + (long)f * 512;
    { const int r = t >> 4, c = t & 15; L[r][c] = g[r * 16 + c]; X[r][c] = g[256 + r * 16 + c]; }
    __syncthreads();
    if (t == 0) {                                         // Cholesky, in place in the lower triangle
        double dmax = 0.0;
        for (int j = 0; j < C; ++j) dmax = fmax(dmax, L[j][j]);
        const double tiny = dmax * 1e-13;
        for (int j = 0; j < C; ++j) {
            double d = L[j][j];
            for (int q = 0; q < j; ++q) d -= L[j][q] * L[j][q];
            const bool ok = d > tiny;
            dead[j] = ok ? 0 : 1;
            const double piv = ok ? sqrt(d) : 1.0;
            L[j][j] = piv;
            for (int r = j + 1; r < C; ++r) {
                double v = L[r][j];
                for (int q = 0; q < j; ++q) v -= L[r][q] * L[j][q];
                L[r][j] = ok ? v / piv : 0.0;
            }
        }
    }
    __syncthreads();
    for (int o = t; o < out; o += blockDim.x) {
        const long of = (long)o * in + f;
        const double scale = sc ? (double)sc[of] : 1.0;
        double w[16], y[16];
        for (int c = 0; c < C; ++c) w[c] = (double)sw[of * C + c] * scale;
        for (int r = 0; r < C; ++r) {                     // rhs = X W, then forward substitution
            double v = 0.0;
            for (int c = 0; c < C; ++c) v += X[r][c] * w[c];
            for (int q = 0; q < r; ++q) v -= L[r][q] * y[q];
            y[r] = dead[r] ? 0.0 : v / L[r][r];
        }
        for (int r = C - 1; r >= 0; --r) {                // back substitution with L^T
            double v = y[r];
            for (int q = r + 1; q < C; ++q) v -= L[q][r] * y[q];
            y[r] = dead[r] ? 0.0 : v / L[r][r];
        }
        for (int c = 0; c < C; ++c) new_sw[of * C + c] = (float)y[c];
    }
}

static void gram_plan(long N, int in, int* nbx, long* rpw) {
    int nb = (int)max(1L, min((long)cdiv(N, 1024), (long)max(1, 4096 / in)));
    long r = (N + (long)nb * 4 - 1) / ((long)nb * 4);
    r = max(4L, (r + 3) & ~3L);
    nb = (int)max(1L, (long)cdiv(cdiv(N, r), 4));
    *nbx = nb; *rpw = r;
}

size_t kan_grid_refit_ws_bytes(long N, int in) {
    int nb; long rpw;
    gram_plan(N, in, &nb, &rpw);
    return ((size_t)in * nb * 4 + (size_t)in) * 512 * sizeof(double);
}

int kan_grid_refit(const float* x, long ldx, long N, const float* grid_old, const float* grid_new, int in, int out,
                   int G, int K, const float* sw, const float* sc, float* new_sw, void* ws, size_t ws_bytes,
                   hipStream_t st) {
    const int C = G + K, nk = G + 2 * K + 1;
    if (C > 16) return fail(KAGNN_ERR_UNSUPPORTED, "%s: grid_size + spline_order must be <= 16", "kan_grid_refit");
    if (ws_bytes < kan_grid_refit_ws_bytes(N, in)) return fail(KAGNN_ERR_ARG, "%s: workspace too small", "kan_grid_refit");
    int nb; long rpw;
    gram_plan(N, in, &nb, &rpw);
    double* gram = (double*)ws;
    double* slab = gram + (size_t)in * 512;
    dim3 grid(nb, in);
#define L(KK) kan_grid_gram_kernel<KK><<<grid, 256, 0, st>>>(x, ldx, N, in, C, grid_old, grid_new, nk, rpw, slab)
    switch (K) {
        case 1: L(1); break;
        case 2: L(2); break;
        case 3: L(3); break;
        case 4: L(4); break;
        default: return fail(KAGNN_ERR_UNSUPPORTED, "%s: spline_order must be 1..4", "kan_grid_refit");
    }
#undef L
    KAGNN_LAUNCH_CHECK();
    kan_grid_gram_reduce_kernel<<<in, 512, 0, st>>>(slab, (long)nb * 4, gram);
    KAGNN_LAUNCH_CHECK();
    kan_grid_solve_kernel<<<in, 256, 0, st>>>(gram, in, out, C, sw, sc, new_sw);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

}  // namespace kagnn
