// Shared device/host helpers for libkagnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/kagnn_hip.h"

namespace kagnn {

// ---------------------------------------------------------------- error plumbing
extern thread_local char g_err[512];
inline int fail(int code, const char* fmt, const char* a = "", long b = 0, long c = 0) {
    snprintf(g_err, sizeof(g_err), fmt, a, b, c);
    return code;
}
#define KAGNN_CHECK_ARG(cond, msg)                                                       \
    do {                                                                                 \
        if (!(cond)) return kagnn::fail(KAGNN_ERR_ARG, "%s: argument check failed: " msg, __func__); \
    } while (0)
#define KAGNN_HIP(...)                                                                   \
    do {                                                                                 \
        hipError_t e_ = (__VA_ARGS__);                                                   \
        if (e_ != hipSuccess) {                                                          \
            snprintf(kagnn::g_err, sizeof(kagnn::g_err), "%s: HIP error at line %d -> %s", __func__, __LINE__, \
                     hipGetErrorString(e_));                                             \
            return KAGNN_ERR_HIP;                                                        \
        }                                                                                \
    } while (0)
#define KAGNN_LAUNCH_CHECK() KAGNN_HIP(hipGetLastError())

// Deferred slab reductions of the weight gradient (kan_split_bwd.hip).  A weight-gradient call is {row-slab kernel, slab
// reduction + unpack}; the second is a ~8 us launch per KANLinear, and the message-passing stack of a graph-level model runs
// nconv * L of them per step on a mini-batch whose whole step is ~1 ms.  An entry point that owns a workspace ARENA for the slabs of
// several layers sets g_dw_defer for its duration: a weight-gradient call whose slab lies inside the arena records its reduction
// instead of launching it, and dw_defer_flush runs all recorded ones in ONE launch (same per-element code path: same bits).
struct DwReduceItem {
    const float* slab; long NS; int in, out, C, SG; long inP, outP;
    const float* sw; const float* sc; float* g_bw; float* g_sw; float* g_sc;
};
constexpr int kDwDeferMax = 16;
struct DwDefer {
    unsigned char* arena; size_t arena_bytes, used;
    int n;
    DwReduceItem item[kDwDeferMax];
};
extern thread_local DwDefer* g_dw_defer;
int dw_defer_flush(hipStream_t st);                 // launches what g_dw_defer holds (no-op when empty) and empties it

// Column moments whose finish is left to the consumer (round 6: the graph-level mini-batches are launch-bound).  A caller that runs
// forward-with-moments and the BatchNorm1d right behind it (kagnn_gine_kan_stack_fwd) sets g_mom_defer for the forward call: the
// forward kernel then leaves its <= kMomDeferMaxP per-workgroup partial rows where they are, skips moments_finish_kernel, and
// records them here; the norm's apply kernel folds them itself (bn_apply_from_partial_moments_kernel: the same merge order, the
// same bits, one launch fewer per convolution).
constexpr int kMomDeferMaxP = 32;
struct MomDefer { const float* partial; int P; };
extern thread_local MomDefer* g_mom_defer;

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting: a process-wide "configured" flag leaves the kernel
// at the 64 KB default on the second GPU a process drives and its launch fails there.  One bit per device ordinal.
// Use:  `if (auto first = first_use_on_this_device(seen)) { hipFuncSetAttribute(...); }` -- true until the calling thread's current
// device (the one the launch goes to) has been configured at this call site.  The bit is set when the guard LEAVES the if
// statement, i.e. after the attribute calls: a second host thread on the same device that arrives meanwhile configures again
// (idempotent) instead of launching a > 64 KB kernel before the first thread's call has completed.
struct FirstUseGuard {
    unsigned long long* seen; unsigned long long bit; bool go;
    explicit operator bool() const { return go; }
    FirstUseGuard(unsigned long long* s, unsigned long long b, bool g) : seen(s), bit(b), go(g) {}
    FirstUseGuard(const FirstUseGuard&) = delete;
    FirstUseGuard& operator=(const FirstUseGuard&) = delete;
    ~FirstUseGuard() { if (go && bit) __atomic_fetch_or(seen, bit, __ATOMIC_RELEASE); }
};
inline FirstUseGuard first_use_on_this_device(unsigned long long& seen) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) return FirstUseGuard(&seen, 0, true);      // unknown: configure again
    const unsigned long long bit = 1ull << d;
    return FirstUseGuard(&seen, bit, !(__atomic_load_n(&seen, __ATOMIC_ACQUIRE) & bit));
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kMaxKnots = 48;   // G + 2k + 1 <= 32 + 8 + 1
constexpr int kMaxOrder = 4;

__host__ __device__ inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// D-fragment row of v_mfma_f32_32x32x*: reg r (0..15), lane-half hi -> row in the 32x32 tile
__device__ __forceinline__ int mfma32_row(int reg, int hi) { return (reg & 3) + 8 * (reg >> 2) + 4 * hi; }

__device__ __forceinline__ float sigmoidf_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float siluf(float x) { return x * sigmoidf_fast(x); }
__device__ __forceinline__ float silu_gradf(float x) {
    float s = sigmoidf_fast(x);
    return s * (1.0f + x * (1.0f - s));
}

// ---------------------------------------------------------------- local uniform B-spline
// For x in knot span m (knots[m] <= x < knots[m+1], the half-open test of ekan.py:95) the only
// non-zero order-K bases are j = m-K .. m; N[r] = B_{m-K+r,K}(x), r = 0..K, via the Cox-de Boor
// recursion (ekan.py:96-105) restricted to that span: with u = (x-knots[m])/h,
//   N^p_r = ((u+p-r)/p) N^{p-1}_{r-1} + ((r+1-u)/p) N^{p-1}_r .
// dN[r] = d/dx of the same = (N^{K-1}_{r-1} - N^{K-1}_r) / h.
// Outside [knots[0], knots[last]) everything is 0; non-finite x gives NaN like the reference.
struct SplineGeom {
    int nknots;     // G + 2K + 1
    float g0;       // knots[0]
    float inv_h;    // 1 / knot spacing
};

// geometry from the knot table itself (no host scalars, no device->host read-back)
__device__ __forceinline__ SplineGeom geom_from_knots(const float* knots /* LDS */, int nknots) {
    SplineGeom g;
    g.nknots = nknots;
    g.g0 = knots[0];
    g.inv_h = (float)(nknots - 1) / (knots[nknots - 1] - knots[0]);
    return g;
}

template <int K, bool DERIV>
__device__ __forceinline__ int bspline_local(float x, const float* __restrict__ knots /* LDS */,
                                             const SplineGeom& g, float (&N)[K + 1],
                                             float (&dN)[K + 1]) {
    const int last = g.nknots - 2;                 // last valid span index
    float t = (x - g.g0) * g.inv_h;
    float tc = fminf(fmaxf(t, 0.0f), (float)last);  // NaN -> 0
    int m = (int)tc;
    float tl = knots[m], tr = knots[m + 1];
    // the arithmetic guess can be one span off right at a knot: settle it with the reference's
    // own comparisons against the stored fp32 knots
    if (x >= tr && m < last) {
        ++m; tl = tr; tr = knots[m + 1];
    } else if (x < tl && m > 0) {
        --m; tr = tl; tl = knots[m];
    }
    const bool inside = (x >= tl) && (x < tr);
    const float u = (x - tl) * g.inv_h;
    float n[K + 1];
    float prev[K + 1];
    n[0] = 1.0f;
#pragma unroll
    for (int r = 1; r <= K; ++r) n[r] = 0.0f;
#pragma unroll
    for (int p = 1; p <= K; ++p) {
#pragma unroll
        for (int r = 0; r <= K; ++r) prev[r] = n[r];
        const float ip = 1.0f / (float)p;
#pragma unroll
        for (int r = 0; r <= p; ++r) {
            float a = (r >= 1) ? (u + (float)(p - r)) * ip * prev[r - 1] : 0.0f;
            float b = (r <= p - 1) ? ((float)(r + 1) - u) * ip * prev[r] : 0.0f;
            n[r] = a + b;
        }
        if (DERIV && p == K) {
#pragma unroll
            for (int r = 0; r <= K; ++r) {
                float lo = (r >= 1) ? prev[r - 1] : 0.0f;
                float hi = (r <= K - 1) ? prev[r] : 0.0f;
                dN[r] = (lo - hi) * g.inv_h;
            }
        }
    }
    const bool finite = fabsf(x) <= 3.4028234e38f;   // false for NaN and +-Inf
    const float nanv = __builtin_nanf("");
#pragma unroll
    for (int r = 0; r <= K; ++r) {
        N[r] = finite ? (inside ? n[r] : 0.0f) : nanv;
        if (DERIV) dN[r] = finite ? (inside ? dN[r] : 0.0f) : nanv;
    }
    return m;
}

// ---------------------------------------------------------------- local B-spline on a per-feature, non-uniform
// knot row (the grids KANLinear.update_grid writes, ekan.py:164-211).  Same contract as bspline_local: the span
// m holding x (half-open test of ekan.py:95, knots assumed increasing) and the K+1 bases j = m-K .. m that can
// be non-zero there, by the Cox-de Boor recursion of ekan.py:96-105 with the real knot differences:
//   B_{j,p} = (x - t_j) / (t_{j+p} - t_j) B_{j,p-1} + (t_{j+p+1} - x) / (t_{j+p+1} - t_{j+1}) B_{j+1,p-1} .
// w[q] = t[m-K+q] (indices clamped: they only feed bases j < 0 or j >= G+K, which pick_basis never selects).
template <int K, bool DERIV>
__device__ __forceinline__ int bspline_generic(float x, const float* __restrict__ t /* this feature's row */,
                                               int nknots, float (&N)[K + 1], float (&dN)[K + 1]) {
    int cnt = 0;
    for (int j = 0; j < nknots; ++j) cnt += (x >= t[j]) ? 1 : 0;
    const bool inside = cnt >= 1 && cnt < nknots;          // t[0] <= x < t[last]
    const int m = min(max(cnt - 1, 0), nknots - 2);
    float w[2 * K + 2];
#pragma unroll
    for (int q = 0; q < 2 * K + 2; ++q) w[q] = t[min(max(m - K + q, 0), nknots - 1)];
    float n[K + 1], prev[K + 1];
    n[0] = 1.0f;
#pragma unroll
    for (int r = 1; r <= K; ++r) n[r] = 0.0f;
#pragma unroll
    for (int p = 1; p <= K; ++p) {
#pragma unroll
        for (int r = 0; r <= K; ++r) prev[r] = n[r];
#pragma unroll
        for (int r = 0; r <= p; ++r) {
            const int q0 = K - p + r;                       // w[q0] = t_j for j = m-p+r
            const float a = (r >= 1) ? (x - w[q0]) / (w[q0 + p] - w[q0]) * prev[r - 1] : 0.0f;
            const float b = (r <= p - 1) ? (w[q0 + p + 1] - x) / (w[q0 + p + 1] - w[q0 + 1]) * prev[r] : 0.0f;
            n[r] = a + b;
        }
        if (DERIV && p == K) {
#pragma unroll
            for (int r = 0; r <= K; ++r) {
                const float lo = (r >= 1) ? prev[r - 1] / (w[r + K] - w[r]) : 0.0f;
                const float hi = (r <= K - 1) ? prev[r] / (w[r + K + 1] - w[r + 1]) : 0.0f;
                dN[r] = (float)K * (lo - hi);
            }
        }
    }
    const bool finite = fabsf(x) <= 3.4028234e38f;
    const float nanv = __builtin_nanf("");
#pragma unroll
    for (int r = 0; r <= K; ++r) {
        N[r] = finite ? (inside ? n[r] : 0.0f) : nanv;
        if (DERIV) dN[r] = finite ? (inside ? dN[r] : 0.0f) : nanv;
    }
    return m;
}

// PF: per-feature knot rows in global memory (knots_g is [in][nknots]); else the one shared uniform row in LDS
template <int K, bool DERIV, bool PF>
__device__ __forceinline__ int eval_basis(float x, const float* s_knots, const SplineGeom& g,
                                          const float* __restrict__ knots_g, int f, float (&N)[K + 1],
                                          float (&dN)[K + 1]) {
    if constexpr (PF) return bspline_generic<K, DERIV>(x, knots_g + (long)f * g.nknots, g.nknots, N, dN);
    else return bspline_local<K, DERIV>(x, s_knots, g, N, dN);
}

// value of basis index c given the local set: N[c - (m-K)] if 0 <= c-(m-K) <= K else 0
template <int K>
__device__ __forceinline__ float pick_basis(const float (&N)[K + 1], int m, int c) {
    const int d = c - (m - K);
    float a = 0.0f;
#pragma unroll
    for (int r = 0; r <= K; ++r) a = (d == r) ? N[r] : a;
    return a;
}

// arguments of the sum-aggregation kernels (aggregate.hip)
struct AggArgs {
    const float* x; long ldx;
    float* out; long ldo;
    const int* rowptr; const int* col; const float* ew;
    long N; int F;
    float self_scale;
    const float* in_scale; const float* out_scale; const float* bias;
    int skip_self; int hub_threshold;
    const float* addend; long lda;     // optional [N, F] matrix added to the result row by row, LAST (after scale and bias): e.g. a
                                       // second gradient of the same activation (the skip branch), saving the separate sum pass
    // optional per-COLUMN affine of the GATHERED matrix (round 4): the rows being gathered are a[c] * x[.][c] + b[c] without that
    // matrix existing -- the output of a training-mode BatchNorm1d whose normalising pass is folded into this aggregation
    // (reference node_classification_clean/models.py:198-200: x = bns[i](convs[i](x)) feeding the next GINConv):
    //   out_i = a * (self * x_i + sum_j w_ij x_j) + (self + sum_j w_ij) * b        (GIN form: unit weights => count = deg_i + self)
    const float* col_scale = nullptr; const float* col_shift = nullptr;
    // optional COLUMN STATISTICS of the result (round 4): when the result is the gradient g arriving at a training-mode
    // BatchNorm1d (the transposed aggregation of the NEXT convolution's backward produces it, reference models.py:198-200), the
    // norm's backward needs sum_n g and sum_n g * xhat, xhat = (st_y - st_mean) * st_rstd, st_y = the norm's input.  The row
    // kernel forms both products on the row it is about to store and leaves one partial row pair per workgroup (hub rows: per
    // hub segment slot, from the merge kernel) in st_partial[.][2][F]; bn.hip folds them in a fixed order.  Saves the norm's
    // own statistics pass over g and y.
    const float* st_y = nullptr; long st_ldy = 0; const float* st_mean = nullptr; const float* st_rstd = nullptr;
    float* st_partial = nullptr;
};

// BatchNorm1d backward, training mode, as ONE expression per element of the incoming gradient g (y = the norm's input):
//   gy = k (g - mean(g) - xhat mean(g xhat)),  xhat = (y - m) q,  k = gamma q     ==>  gy = A g + B (y - m) + C
// with per column  A = k,  B = -k q mean(g xhat),  C = -k mean(g)  (bn_finish_table_kernel, bn.hip).  The deviation y - m is
// formed first, as the stand-alone pass always did (near-constant columns: |m| >> sigma).  Used by bn_bwd_apply_kernel and by
// the input-gradient kernel that applies it to the rows it loads (kan_split_dx_kernel<..., BNB>): same bits either way.
struct BnBack {
    const float* y; long ldy;          // the norm's input (= the layer's output), [N, out]
    const float* tab; int ldt;         // [4][ldt]: m | A | B | C per column
    float* gy_out; long ldo;           // the transformed rows, for the weight-gradient kernel that runs next
};
#ifdef __HIPCC__
__device__ __forceinline__ float bn_bwd_value(float g, float y, float m, float A, float B, float C) {
    return __builtin_fmaf(A, g, __builtin_fmaf(B, __fsub_rn(y, m), C));
}
// the three per-column constants from the norm's saved statistics and the two column sums of its backward
__device__ __forceinline__ void bn_bwd_consts(float q /* rstd */, float gamma, float sum_g, float sum_gxhat, float inv_n, float& A,
                                              float& B, float& C) {
    const float k = __fmul_rn(q, gamma);
    A = k;
    B = -__fmul_rn(__fmul_rn(k, q), __fmul_rn(sum_gxhat, inv_n));
    C = -__fmul_rn(k, __fmul_rn(sum_g, inv_n));
}
#endif

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace kagnn
