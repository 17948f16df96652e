/* kan_ref.c -- CPU restatement (plain C, fp64 arithmetic) of the KAN-GNN hot path.
 * TEST INFRASTRUCTURE ONLY: built by oracle/build_c.py into oracle/_build/libkagnn_ref.so and used
 * by tests/ as a second, independent checker ("what would exact arithmetic give?") next to the
 * torch-op oracle oracle/kan_oracle.py.  Never linked into or called by the product.
 *
 * Each function restates the algorithm of the reference file:line it cites
 * (paths relative to the reference repository root).  Parity status: the KAN math is pinned through
 * tests/test_oracle_c.py against the golden vectors generated from the reference's own layers; the
 * aggregation follows torch_geometric semantics restated in SURVEY.md 3.1 -- parity unpinned.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Cox-de Boor recursion on the stored knot vector, node_classification_clean/ekan.py:79-112.
 * x: n scalars of ONE feature, knots: G+2k+1 values, out: [n][G+k] (fp64). */
static void bases_nint(double x, const float* t, int m0 /* #intervals = #knots-1 */, int k, double* b) {
    for (int j = 0; j < m0; ++j) b[j] = (x >= (double)t[j] && x < (double)t[j + 1]) ? 1.0 : 0.0;  /* :95 */
    if (!(x == x) || isinf(x)) { for (int j = 0; j < m0; ++j) b[j] = NAN; }   /* reference: (x-t)*0 = NaN */
    for (int p = 1; p <= k; ++p) {                                             /* :96-105 */
        for (int j = 0; j < m0 - p; ++j) {
            const double l = (x - t[j]) / ((double)t[j + p] - t[j]) * b[j];
            const double r = ((double)t[j + p + 1] - x) / ((double)t[j + p + 1] - t[j + 1]) * b[j + 1];
            b[j] = l + r;
        }
    }
}

static void bases_1d(double x, const float* t, int G, int k, double* b /* G+2k */) {
    bases_nint(x, t, G + 2 * k, k, b);
}

void kagnn_ref_bspline_bases(const float* x, int64_t n, int in, const float* grid /* [in][G+2k+1] */,
                             int G, int k, double* out /* [n][in][G+k] */) {
    const int M = G + 2 * k + 1, C = G + k;
    double* b = (double*)malloc(sizeof(double) * (size_t)(M + 1));
    for (int64_t r = 0; r < n; ++r)
        for (int f = 0; f < in; ++f) {
            bases_1d((double)x[r * in + f], grid + (size_t)f * M, G, k, b);
            memcpy(out + ((size_t)r * in + f) * C, b, sizeof(double) * (size_t)C);
        }
    free(b);
}

static double silu(double x) { return x / (1.0 + exp(-x)); }
static double silu_grad(double x) { const double s = 1.0 / (1.0 + exp(-x)); return s * (1.0 + x * (1.0 - s)); }

/* KANLinear.forward, ekan.py:154-162 with scaled_spline_weight :146-152.  scaler may be NULL. */
void kagnn_ref_kan_linear_fwd(const float* x, int64_t n, int in, int out, int G, int k,
                              const float* grid, const float* bw, const float* sw, const float* sc,
                              double* y /* [n][out] */) {
    const int M = G + 2 * k + 1, C = G + k;
    double* b = (double*)malloc(sizeof(double) * (size_t)(M + 1));
    for (int64_t r = 0; r < n; ++r) {
        double* yr = y + (size_t)r * out;
        for (int o = 0; o < out; ++o) yr[o] = 0.0;
        for (int f = 0; f < in; ++f) {
            const double xv = x[r * in + f];
            bases_1d(xv, grid + (size_t)f * M, G, k, b);
            const double s = silu(xv);
            for (int o = 0; o < out; ++o) {
                const size_t of = (size_t)o * in + f;
                double acc = s * bw[of];
                const double scale = sc ? sc[of] : 1.0;
                for (int c = 0; c < C; ++c) acc += b[c] * ((double)sw[of * C + c] * scale);
                yr[o] += acc;
            }
        }
    }
    free(b);
}

/* derivative of the order-k bases by central differences of the SAME recursion in fp64 would lose
 * digits; use the exact identity  d/dx B_{j,k} = k/(t_{j+k}-t_j) B_{j,k-1} - k/(t_{j+k+1}-t_{j+1}) B_{j+1,k-1}
 * (what autograd of ekan.py:96-105 evaluates to). */
static void dbases_1d(double x, const float* t, int G, int k, double* d /* G+k */, double* tmp) {
    bases_nint(x, t, G + 2 * k, k - 1, tmp);       /* order k-1 on the SAME knot vector: G+k+1 bases */
    const int C = G + k;
    for (int j = 0; j < C; ++j) {
        const double a = (double)k / ((double)t[j + k] - t[j]) * tmp[j];
        const double b = (double)k / ((double)t[j + k + 1] - t[j + 1]) * tmp[j + 1];
        d[j] = a - b;
    }
}

/* autograd backward of KANLinear.forward: gx, g_base_weight, g_spline_weight, g_spline_scaler (fp64) */
void kagnn_ref_kan_linear_bwd(const float* x, const float* gy, int64_t n, int in, int out, int G, int k,
                              const float* grid, const float* bw, const float* sw, const float* sc,
                              double* gx, double* gbw, double* gsw, double* gsc) {
    const int M = G + 2 * k + 1, C = G + k;
    double* b = (double*)malloc(sizeof(double) * (size_t)(M + 1));
    double* d = (double*)malloc(sizeof(double) * (size_t)(M + 1));
    double* tmp = (double*)malloc(sizeof(double) * (size_t)(M + 1));
    memset(gbw, 0, sizeof(double) * (size_t)out * in);
    memset(gsw, 0, sizeof(double) * (size_t)out * in * C);
    if (gsc) memset(gsc, 0, sizeof(double) * (size_t)out * in);
    for (int64_t r = 0; r < n; ++r)
        for (int f = 0; f < in; ++f) {
            const double xv = x[r * in + f];
            bases_1d(xv, grid + (size_t)f * M, G, k, b);
            dbases_1d(xv, grid + (size_t)f * M, G, k, d, tmp);
            const double s = silu(xv), sg = silu_grad(xv);
            double acc = 0.0;
            for (int o = 0; o < out; ++o) {
                const size_t of = (size_t)o * in + f;
                const double g = gy[r * out + o];
                const double scale = sc ? sc[of] : 1.0;
                gbw[of] += g * s;
                acc += g * bw[of] * sg;
                double dots = 0.0;
                for (int c = 0; c < C; ++c) {
                    gsw[of * C + c] += g * b[c] * scale;
                    dots += b[c] * sw[of * C + c];
                    acc += g * d[c] * sw[of * C + c] * scale;
                }
                if (gsc) gsc[of] += g * dots;
            }
            gx[r * in + f] = acc;
        }
    free(b); free(d); free(tmp);
}

/* stable counting sort of the edge list by key: perm = argsort(key, stable), col = val[perm],
 * rowptr = exclusive prefix sum of the histogram.  Returns -1 on a bad id. */
int kagnn_ref_csr_build(const int64_t* key, const int64_t* val, int64_t e, int64_t n,
                        int64_t* rowptr, int64_t* col, int64_t* perm) {
    memset(rowptr, 0, sizeof(int64_t) * (size_t)(n + 1));
    for (int64_t i = 0; i < e; ++i) {
        if (key[i] < 0 || key[i] >= n || val[i] < 0 || val[i] >= n) return -1;
        rowptr[key[i] + 1]++;
    }
    for (int64_t i = 0; i < n; ++i) rowptr[i + 1] += rowptr[i];
    int64_t* cur = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    memcpy(cur, rowptr, sizeof(int64_t) * (size_t)n);
    for (int64_t i = 0; i < e; ++i) {
        const int64_t p = cur[key[i]]++;
        perm[p] = i;
        col[p] = val[i];
    }
    free(cur);
    return 0;
}

/* out[i] = self_scale*x[i] + sum_{e: dst=i} w[e]*x[src[e]]  -- MessagePassing.propagate(aggr='add'),
 * edges walked in their original order like scatter_add_.  w may be NULL. */
void kagnn_ref_aggregate(const float* x, int64_t n, int f, const int64_t* src, const int64_t* dst,
                         int64_t e, const float* w, double self_scale, double* out) {
    for (int64_t i = 0; i < n * f; ++i) out[i] = self_scale * x[i];
    for (int64_t k = 0; k < e; ++k) {
        const double wk = w ? w[k] : 1.0;
        const float* xs = x + (size_t)src[k] * f;
        double* od = out + (size_t)dst[k] * f;
        for (int j = 0; j < f; ++j) od[j] += wk * xs[j];
    }
}
