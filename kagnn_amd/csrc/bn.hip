// BatchNorm1d over node rows ([N, F] activations, statistics per feature column) -- the epilogue that
// follows every KAN convolution in the node models (reference node_classification_clean/models.py:195-202,
// torch.nn.BatchNorm1d semantics: biased variance for the normalisation, unbiased for running_var,
// running = (1-momentum)*running + momentum*batch).  HBM-bound: the forward reads x twice and writes y,
// the backward reads (x, gy) twice and writes gx; column sums go through per-workgroup partials that are
// combined in a fixed order (deterministic, no atomics).
//
// Fused forms ("BatchNorm1d + dropout ... fused into the conv output"):
//   * the forward statistics can arrive as column moments (mean, sum of squared deviations) that the producing
//     KANLinear forward accumulated in its epilogue (kan_sparse_fwd.hip, MOM) -- the statistics pass over x is skipped;
//   * dropout (reference models.py:201, F.dropout after the norm) is applied by the normalising kernel itself and
//     regenerated from (seed, row, column) in the backward: no mask tensor, no extra pass.  Counter-based hash, so the
//     mask is a pure function of the seed; it is NOT torch's Philox stream (same distribution, different bits).
#include <atomic>
#include "common.h"

namespace kagnn {

// thread -> (row slot, 4 consecutive columns); a workgroup covers RS rows per iteration
struct BnShape { int cl; int rs; };                       // cl = float4 column groups per row (<= 256), rs = 256 / cl
static inline BnShape bn_shape(int F) { BnShape s; s.cl = min(256, cdiv(F, 4)); s.rs = 256 / s.cl; return s; }

__device__ __forceinline__ void ld4c(const float* row, int c, int F, bool vec, float (&v)[4]) {
    if (vec) { const float4 t = *reinterpret_cast<const float4*>(row + c); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (c + i < F) ? row[c + i] : 0.0f;
    }
}
__device__ __forceinline__ void st4c(float* row, int c, int F, bool vec, const float (&v)[4]) {
    if (vec) *reinterpret_cast<float4*>(row + c) = make_float4(v[0], v[1], v[2], v[3]);
    else {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (c + i < F) row[c + i] = v[i];
    }
}

// keep-and-scale factors of elements (n, c..c+3): two 32-bit hashes -> four 16-bit uniforms, keep when u < thr
struct DropArgs { unsigned lo, hi, thr; float scale; };          // thr = round((1-p) * 65536); thr >= 65536: no dropout
__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ void keep4(const DropArgs& d, long n, int c, float (&k)[4]) {
    const unsigned a = mix32(mix32((unsigned)n ^ d.lo) + (unsigned)(n >> 32) + (unsigned)(c >> 2) * 0x9e3779b9U + d.hi);
    const unsigned b = mix32(a ^ 0x85ebca6bU);
    k[0] = (a & 0xffffU) < d.thr ? d.scale : 0.0f;
    k[1] = (a >> 16) < d.thr ? d.scale : 0.0f;
    k[2] = (b & 0xffffU) < d.thr ? d.scale : 0.0f;
    k[3] = (b >> 16) < d.thr ? d.scale : 0.0f;
}
static DropArgs drop_args(float p, unsigned long long seed) {
    DropArgs d;
    d.lo = (unsigned)seed; d.hi = (unsigned)(seed >> 32);
    const float keep = 1.0f - p;
    d.thr = p > 0.0f ? (unsigned)lrintf(fminf(fmaxf(keep, 0.0f), 1.0f) * 65536.0f) : 65536u;
    d.scale = (p > 0.0f && keep > 0.0f) ? 1.0f / keep : (p > 0.0f ? 0.0f : 1.0f);
    return d;
}

// bn_colsum_kernel<1>'s optional tail (see its end): ticket = slot of g_bn_tickets (-1: no tail, the caller launches bn_finish_*)
struct BnTail { int ticket = -1; const float* gamma = nullptr; float* sum_gy = nullptr; float* sum_gyx = nullptr; float* tab = nullptr; int ldt = 0; };
constexpr int kBnTickets = 64;
__device__ unsigned g_bn_tickets[kBnTickets];           // zero at module load; every user leaves its slot at zero
static int bn_next_ticket() {                            // launches in flight at once on different streams draw different slots
    static std::atomic<unsigned> next{0};
    return (int)(next.fetch_add(1u, std::memory_order_relaxed) % kBnTickets);
}

// partial[b][0][f] = sum_n a(n,f), partial[b][1][f] = sum_n b(n,f) over the rows of workgroup b, where
//   MODE 0 (forward statistics):  a = x - shift_f,  b = (x - shift_f)^2      (shift_f = x[0][f]: no cancellation)
//   MODE 1 (backward sums):       a = gy,           b = gy * (x - mean_f) * rstd_f
template <int MODE>
__global__ __launch_bounds__(256) void bn_colsum_kernel(const float* __restrict__ x, long ldx,
                                                        const float* __restrict__ gy, long ldgy, long N, int F,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, int cl, int rs,
                                                        long rows_per_block, float* __restrict__ partial, DropArgs dr,
                                                        BnTail tail = BnTail{}) {
    extern __shared__ float s_red[];                    // [rs][2][4*cl]
    const int cg = threadIdx.x % cl, slot = threadIdx.x / cl;
    const bool vec = ((F & 3) == 0) && ((ldx & 3) == 0) && (MODE == 0 || (ldgy & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && (MODE == 0 || (reinterpret_cast<uintptr_t>(gy) & 15) == 0);
    const long r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
    for (int c0 = 0; c0 < F; c0 += 4 * cl) {            // F > 1024 takes more than one pass; uniform trip count (barriers inside)
        const int c = c0 + 4 * cg;
        float m[4], q[4], a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ci = min(c + i, F - 1);
            m[i] = MODE == 0 ? x[ci] : mean[ci];
            q[i] = MODE == 0 ? 1.0f : rstd[ci];
        }
        if (slot < rs && c < F) {
            for (long n = r0 + slot; n < r1; n += rs) {
                float xv[4];
                ld4c(x + n * ldx, c, F, vec, xv);
                if (MODE == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float d = xv[i] - m[i]; a[i] += d; b[i] = fmaf(d, d, b[i]); }
                } else {
                    float gv[4];
                    ld4c(gy + n * ldgy, c, F, vec, gv);
                    if (dr.thr < 65536u) {                  // the gradient of the dropped-out output
                        float k[4];
                        keep4(dr, n, c, k);
#pragma unroll
                        for (int i = 0; i < 4; ++i) gv[i] *= k[i];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) { a[i] += gv[i]; b[i] = fmaf(gv[i], (xv[i] - m[i]) * q[i], b[i]); }
                }
            }
        }
        // combine the row slots in a fixed order
        __syncthreads();
        if (slot < rs) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { s_red[(slot * 2 + 0) * 4 * cl + 4 * cg + i] = a[i]; s_red[(slot * 2 + 1) * 4 * cl + 4 * cg + i] = b[i]; }
        }
        __syncthreads();
        if (slot == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float ta = 0.f, tb = 0.f;
                for (int s = 0; s < rs; ++s) { ta += s_red[(s * 2 + 0) * 4 * cl + 4 * cg + i]; tb += s_red[(s * 2 + 1) * 4 * cl + 4 * cg + i]; }
                if (c + i < F) { partial[(blockIdx.x * 2L + 0) * F + c + i] = ta; partial[(blockIdx.x * 2L + 1) * F + c + i] = tb; }
            }
        }
    }
    if constexpr (MODE == 1) {
        // round 6 (the graph-level mini-batches are launch-bound): the workgroup that finishes LAST folds all partial rows and writes
        // the sums + the per-column table -- bn_finish_table_kernel's work, in ITS order (32 row groups: group rg adds rows rg, rg + 32, ..,
        // then the groups are added 0 .. 31), so the same bits whichever workgroup it is -- instead of a second launch.
        if (tail.ticket < 0) return;
        __shared__ int s_last;
        __shared__ float s_p[32][2][33];
        __threadfence();                                 // this workgroup's partial row is visible device-wide before its ticket
        __syncthreads();
        if (threadIdx.x == 0) s_last = atomicAdd(&g_bn_tickets[tail.ticket], 1u) == gridDim.x - 1;
        __syncthreads();
        if (!s_last) return;
        __threadfence();
        const volatile float* vp = partial;             // (written by other workgroups of this launch: never through a stale cache line)
        const long B = gridDim.x;
        const int c = threadIdx.x & 31, rq = threadIdx.x >> 5;        // 32 columns x 8 threads, each standing for 4 of the 32 row groups
        for (int cb = 0; cb < cdiv(tail.ldt, 32); ++cb) {
            const int f = cb * 32 + c;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rg = 4 * rq + j;
                float a = 0.0f, b = 0.0f;
                if (f < F)
                    for (long w = rg; w < B; w += 32) { a += vp[(w * 2 + 0) * F + f]; b += vp[(w * 2 + 1) * F + f]; }
                s_p[rg][0][c] = a; s_p[rg][1][c] = b;
            }
            __syncthreads();
            if (rq == 0 && f < tail.ldt) {
                float A = 0.0f, Bc = 0.0f, C = 0.0f, m = 0.0f;
                if (f < F) {
                    float ta = 0.f, tb = 0.f;
#pragma unroll
                    for (int g = 0; g < 32; ++g) { ta += s_p[g][0][c]; tb += s_p[g][1][c]; }
                    tail.sum_gy[f] = ta;        // g_bias
                    tail.sum_gyx[f] = tb;       // g_weight
                    m = mean[f];
                    bn_bwd_consts(rstd[f], tail.gamma ? tail.gamma[f] : 1.0f, ta, tb, 1.0f / (float)N, A, Bc, C);
                }
                tail.tab[f] = m; tail.tab[tail.ldt + f] = A; tail.tab[2 * tail.ldt + f] = Bc; tail.tab[3 * tail.ldt + f] = C;
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) g_bn_tickets[tail.ticket] = 0u;      // ready for the next launch that draws this slot
    }
}

// combine the partials (fixed order) and finish the statistics:  32 row groups per 32 columns
//   MODE 0: mean, rstd (biased variance) -> save_mean / save_rstd, running stats update
//   MODE 1: g_bias = sum gy, g_weight = sum gy*xhat; also left in sums[0][f], sums[1][f] for the gx pass
//   MODE 2: column moments only: mean, sum of squared deviations (col_moments below)
template <int MODE>
__global__ __launch_bounds__(1024) void bn_finish_kernel(const float* __restrict__ partial, long B, int F, long N,
                                                        const float* __restrict__ x_row0, float eps, float momentum,
                                                        float* __restrict__ out_a, float* __restrict__ out_b,
                                                        float* __restrict__ running_mean,
                                                        float* __restrict__ running_var) {
    __shared__ float s_p[32][2][33];
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;       // 32 columns x 32 row groups
    const int f = blockIdx.x * 32 + c;
    float a = 0.0f, b = 0.0f;
    if (f < F)
        for (long w = rg; w < B; w += 32) { a += partial[(w * 2 + 0) * F + f]; b += partial[(w * 2 + 1) * F + f]; }
    s_p[rg][0][c] = a; s_p[rg][1][c] = b;
    __syncthreads();
    if (rg == 0 && f < F) {
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) { ta += s_p[g][0][c]; tb += s_p[g][1][c]; }
        if (MODE == 2) {
            const float d = ta / (float)N;
            out_a[f] = x_row0[f] + d;
            out_b[f] = fmaxf(tb - ta * d, 0.0f);
        } else if (MODE == 0) {
            const float inv_n = 1.0f / (float)N;
            const float d = ta * inv_n;                                  // mean - shift
            const float mean = x_row0[f] + d;
            const float var = fmaxf(tb * inv_n - d * d, 0.0f);           // biased
            out_a[f] = mean;
            out_b[f] = rsqrtf(var + eps);
            if (running_mean) {
                const float unb = N > 1 ? var * ((float)N / (float)(N - 1)) : var;
                running_mean[f] = fmaf(momentum, mean - running_mean[f], running_mean[f]);
                running_var[f] = fmaf(momentum, unb - running_var[f], running_var[f]);
            }
        } else {
            out_a[f] = ta;        // g_bias
            out_b[f] = tb;        // g_weight
        }
    }
}

// y = (x - mean) * rstd * gamma + beta          (gamma/beta may be null: affine=False)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, long ldx, long N, int F,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ y, long ldy, int cl, int rs, DropArgs dr) {
    const int cg = threadIdx.x % cl, slot = threadIdx.x / cl;
    if (slot >= rs) return;
    const bool vec = ((F & 3) == 0) && ((ldx & 3) == 0) && ((ldy & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
    for (int c = 4 * cg; c < F; c += 4 * cl) {          // no barriers below: a per-thread trip count is fine
        float sc[4], sh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ci = min(c + i, F - 1);
            sc[i] = rstd[ci] * (gamma ? gamma[ci] : 1.0f);
            sh[i] = fmaf(-mean[ci], sc[i], beta ? beta[ci] : 0.0f);
        }
        for (long n = blockIdx.x * (long)rs + slot; n < N; n += (long)gridDim.x * rs) {
            float v[4];
            ld4c(x + n * ldx, c, F, vec, v);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaf(v[i], sc[i], sh[i]);
            if (dr.thr < 65536u) {
                float k[4];
                keep4(dr, n, c, k);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] *= k[i];
            }
            st4c(y + n * ldy, c, F, vec, v);
        }
    }
}

// statistics from a producer's column moments + bn_apply_kernel in one launch (round 5: the graph-level mini-batches are launch-bound;
// two launches until then): every thread derives mean / rstd of its columns from the moments (mean, M2 / N biased), then applies
// bn_apply_kernel's expressions; workgroup 0's first row slot also stores save_mean / save_rstd and updates the running statistics (one
// thread per column).  Same bits as the two launches.
__global__ __launch_bounds__(256) void bn_apply_from_moments_kernel(const float* __restrict__ x, long ldx, long N, int F,
                                                                    const float* __restrict__ col_mean, const float* __restrict__ col_m2,
                                                                    float eps, float momentum, float* __restrict__ save_mean,
                                                                    float* __restrict__ save_rstd, float* __restrict__ running_mean,
                                                                    float* __restrict__ running_var,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    float* __restrict__ y, long ldy, int cl, int rs, DropArgs dr) {
    const int cg = threadIdx.x % cl, slot = threadIdx.x / cl;
    if (slot >= rs) return;
    const bool vec = ((F & 3) == 0) && ((ldx & 3) == 0) && ((ldy & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
    const bool writer = blockIdx.x == 0 && slot == 0;
    for (int c = 4 * cg; c < F; c += 4 * cl) {
        float sc[4], sh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ci = min(c + i, F - 1);
            const float mean = col_mean[ci], var = fmaxf(col_m2[ci] / (float)N, 0.0f);
            const float rstd = rsqrtf(var + eps);
            if (writer && c + i < F) {
                save_mean[ci] = mean;
                save_rstd[ci] = rstd;
                if (running_mean) {
                    const float unb = N > 1 ? var * ((float)N / (float)(N - 1)) : var;
                    running_mean[ci] = fmaf(momentum, mean - running_mean[ci], running_mean[ci]);
                    running_var[ci] = fmaf(momentum, unb - running_var[ci], running_var[ci]);
                }
            }
            sc[i] = rstd * (gamma ? gamma[ci] : 1.0f);
            sh[i] = fmaf(-mean, sc[i], beta ? beta[ci] : 0.0f);
        }
        for (long n = blockIdx.x * (long)rs + slot; n < N; n += (long)gridDim.x * rs) {
            float v[4];
            ld4c(x + n * ldx, c, F, vec, v);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaf(v[i], sc[i], sh[i]);
            if (dr.thr < 65536u) {
                float k[4];
                keep4(dr, n, c, k);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] *= k[i];
            }
            st4c(y + n * ldy, c, F, vec, v);
        }
    }
}

// (count, mean, M2) of two disjoint row sets -> of their union (Chan et al.); b is folded into a
__device__ __forceinline__ void chan_merge(float& na, float& ma, float& qa, float nb, float mb, float qb) {
    if (nb <= 0.0f) return;
    const float n = na + nb, d = mb - ma, w = nb / n;
    ma = fmaf(d, w, ma);
    qa += qb + d * d * na * w;
    na = n;
}

// bn_apply_from_moments_kernel whose moments arrive as the producer's P <= 32 per-workgroup partial rows [P][{mean, M2, count}][F]
// (common.h: MomDefer): every workgroup folds them first -- for P <= 32 moments_finish_kernel's two-level order IS the sequential
// merge over the rows (each of its 32 row groups holds one row, and merging a row into the empty set copies it exactly), so the
// column statistics carry the same bits -- then the same expressions as above.  Saves the finish launch; round 6.
__global__ __launch_bounds__(256) void bn_apply_from_partial_moments_kernel(const float* __restrict__ x, long ldx, long N, int F,
                                                                            const float* __restrict__ partial, int P,
                                                                            float eps, float momentum, float* __restrict__ save_mean,
                                                                            float* __restrict__ save_rstd, float* __restrict__ running_mean,
                                                                            float* __restrict__ running_var,
                                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                            float* __restrict__ y, long ldy, int cl, int rs) {
    extern __shared__ float s_mom[];                    // [2][F]: mean, M2
    for (int f = threadIdx.x; f < F; f += 256) {
        float n = 0.0f, m = 0.0f, q = 0.0f;
        for (int w = 0; w < P; ++w)
            chan_merge(n, m, q, partial[((long)w * 3 + 2) * F + f], partial[((long)w * 3 + 0) * F + f], partial[((long)w * 3 + 1) * F + f]);
        s_mom[f] = m; s_mom[F + f] = q;
    }
    __syncthreads();
    const int cg = threadIdx.x % cl, slot = threadIdx.x / cl;
    if (slot >= rs) return;
    const bool vec = ((F & 3) == 0) && ((ldx & 3) == 0) && ((ldy & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
    const bool writer = blockIdx.x == 0 && slot == 0;
    for (int c = 4 * cg; c < F; c += 4 * cl) {
        float sc[4], sh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ci = min(c + i, F - 1);
            const float mean = s_mom[ci], var = fmaxf(s_mom[F + ci] / (float)N, 0.0f);
            const float rstd = rsqrtf(var + eps);
            if (writer && c + i < F) {
                save_mean[ci] = mean;
                save_rstd[ci] = rstd;
                if (running_mean) {
                    const float unb = N > 1 ? var * ((float)N / (float)(N - 1)) : var;
                    running_mean[ci] = fmaf(momentum, mean - running_mean[ci], running_mean[ci]);
                    running_var[ci] = fmaf(momentum, unb - running_var[ci], running_var[ci]);
                }
            }
            sc[i] = rstd * (gamma ? gamma[ci] : 1.0f);
            sh[i] = fmaf(-mean, sc[i], beta ? beta[ci] : 0.0f);
        }
        for (long n = blockIdx.x * (long)rs + slot; n < N; n += (long)gridDim.x * rs) {
            float v[4];
            ld4c(x + n * ldx, c, F, vec, v);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaf(v[i], sc[i], sh[i]);
            st4c(y + n * ldy, c, F, vec, v);
        }
    }
}

// training: gx = gamma*rstd * (gy - sum_gy/N - xhat * sum_gy_xhat/N);   eval: gx = gamma*rstd*gy
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, long ldx,
                                                           const float* __restrict__ gy, long ldgy, long N, int F,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ sum_gy,
                                                           const float* __restrict__ sum_gyx, int training,
                                                           float* __restrict__ gx, long ldgx, int cl, int rs,
                                                           DropArgs dr) {
    const int cg = threadIdx.x % cl, slot = threadIdx.x / cl;
    if (slot >= rs) return;
    const bool vec = ((F & 3) == 0) && ((ldx & 3) == 0) && ((ldgy & 3) == 0) && ((ldgx & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(gy) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(gx) & 15) == 0);
    const float inv_n = 1.0f / (float)N;
    for (int c = 4 * cg; c < F; c += 4 * cl) {
        float m[4], cA[4], cB[4], cC[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ci = min(c + i, F - 1);
            m[i] = mean[ci];
            if (training) bn_bwd_consts(rstd[ci], gamma ? gamma[ci] : 1.0f, sum_gy[ci], sum_gyx[ci], inv_n, cA[i], cB[i], cC[i]);
            else { cA[i] = rstd[ci] * (gamma ? gamma[ci] : 1.0f); cB[i] = 0.0f; cC[i] = 0.0f; }
        }
        for (long n = blockIdx.x * (long)rs + slot; n < N; n += (long)gridDim.x * rs) {
            float xv[4], gv[4], o[4];
            ld4c(x + n * ldx, c, F, vec, xv);
            ld4c(gy + n * ldgy, c, F, vec, gv);
            if (dr.thr < 65536u) {
                float kp[4];
                keep4(dr, n, c, kp);
#pragma unroll
                for (int i = 0; i < 4; ++i) gv[i] *= kp[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = bn_bwd_value(gv[i], xv[i], m[i], cA[i], cB[i], cC[i]);   // (the input-gradient kernel's fused form: same expression, common.h)
            st4c(gx + n * ldgx, c, F, vec, o);
        }
    }
}

// bn_finish_kernel<1> + the table m | A | B | C per column for kernels that apply the backward to rows they load themselves (BnBack,
// common.h) in one launch: the thread that finishes column f's two sums also writes its table entries (until round 5 a launch of
// its own reading the sums back: same expressions on the same values, same bits); columns F .. ldt - 1 of the table are zero.
// Grid: cdiv(ldt, 32).
__global__ __launch_bounds__(1024) void bn_finish_table_kernel(const float* __restrict__ partial, long B, int F, long N,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, float* __restrict__ sum_gy,
                                                              float* __restrict__ sum_gyx, float* __restrict__ tab, int ldt) {
    __shared__ float s_p[32][2][33];
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;       // 32 columns x 32 row groups
    const int f = blockIdx.x * 32 + c;
    float a = 0.0f, b = 0.0f;
    if (f < F)
        for (long w = rg; w < B; w += 32) { a += partial[(w * 2 + 0) * F + f]; b += partial[(w * 2 + 1) * F + f]; }
    s_p[rg][0][c] = a; s_p[rg][1][c] = b;
    __syncthreads();
    if (rg == 0 && f < ldt) {
        float A = 0.0f, Bc = 0.0f, C = 0.0f, m = 0.0f;
        if (f < F) {
            float ta = 0.f, tb = 0.f;
#pragma unroll
            for (int g = 0; g < 32; ++g) { ta += s_p[g][0][c]; tb += s_p[g][1][c]; }
            sum_gy[f] = ta;        // g_bias
            sum_gyx[f] = tb;       // g_weight
            m = mean[f];
            bn_bwd_consts(rstd[f], gamma ? gamma[f] : 1.0f, ta, tb, 1.0f / (float)N, A, Bc, C);
        }
        tab[f] = m; tab[ldt + f] = A; tab[2 * ldt + f] = Bc; tab[3 * ldt + f] = C;
    }
}

// rstd from a variance vector (eval mode: running_var)
__global__ void bn_rstd_kernel(const float* __restrict__ var, int F, float eps, float* __restrict__ rstd) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < F) rstd[f] = rsqrtf(var[f] + eps);
}

// training statistics from column moments (mean, M2 = sum of squared deviations) a producer kernel left behind: inside
// bn_apply_from_moments_kernel (above), and ...
// ... plus the per-column affine of the normalisation, for consumers that apply it to the rows they load instead of reading a
// normalised matrix (bn_stats_affine): a = gamma * rstd, b = beta - mean * a  -- the expressions of bn_apply_kernel
__global__ void bn_affine_kernel(const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, int F, float* __restrict__ affine) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const float sc = rstd[f] * (gamma ? gamma[f] : 1.0f);
    affine[f] = sc;
    affine[F + f] = fmaf(-mean[f], sc, beta ? beta[f] : 0.0f);
}

__global__ void bn_from_moments_affine_kernel(const float* __restrict__ col_mean, const float* __restrict__ col_m2, long N, int F,
                                              float eps, float momentum, float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                              float* __restrict__ running_mean, float* __restrict__ running_var,
                                              const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ affine) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const float mean = col_mean[f], var = fmaxf(col_m2[f] / (float)N, 0.0f);
    const float rstd = rsqrtf(var + eps);
    save_mean[f] = mean;
    save_rstd[f] = rstd;
    if (running_mean) {
        const float unb = N > 1 ? var * ((float)N / (float)(N - 1)) : var;
        running_mean[f] = fmaf(momentum, mean - running_mean[f], running_mean[f]);
        running_var[f] = fmaf(momentum, unb - running_var[f], running_var[f]);
    }
    const float sc = rstd * (gamma ? gamma[f] : 1.0f);
    affine[f] = sc;
    affine[F + f] = fmaf(-mean, sc, beta ? beta[f] : 0.0f);
}

// partial[p][{mean, M2, count}][F] of P producer workgroups, merged in a fixed order: 32 row groups x 32 columns
__global__ __launch_bounds__(1024) void moments_finish_kernel(const float* __restrict__ partial, int P, int F,
                                                             float* __restrict__ col_mean, float* __restrict__ col_m2) {
    __shared__ float s_p[32][3][33];
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int f = blockIdx.x * 32 + c;
    float n = 0.0f, m = 0.0f, q = 0.0f;
    if (f < F)
        for (int w = rg; w < P; w += 32)
            chan_merge(n, m, q, partial[((long)w * 3 + 2) * F + f], partial[((long)w * 3 + 0) * F + f], partial[((long)w * 3 + 1) * F + f]);
    s_p[rg][0][c] = m; s_p[rg][1][c] = q; s_p[rg][2][c] = n;
    __syncthreads();
    if (rg == 0 && f < F) {
        n = 0.0f; m = 0.0f; q = 0.0f;
        for (int g = 0; g < 32; ++g) chan_merge(n, m, q, s_p[g][2][c], s_p[g][0][c], s_p[g][1][c]);
        col_mean[f] = m;
        col_m2[f] = q;
    }
}

int moments_finish(const float* partial, int P, int F, float* col_mean, float* col_m2, hipStream_t st) {
    moments_finish_kernel<<<cdiv(F, 32), 1024, 0, st>>>(partial, P, F, col_mean, col_m2);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// ------------------------------------------------------------------ host side
struct BnPlan { int blocks; long rpb; size_t partial_bytes; };
static BnPlan bn_plan(long N, int F) {
    BnPlan p;
    const BnShape s = bn_shape(F);
    long b = min(512L, max(1L, N / (4L * s.rs)));         // >= 4 iterations per workgroup; few partial rows keep bn_finish short
    p.rpb = (N + b - 1) / b;
    p.rpb = ((p.rpb + s.rs - 1) / s.rs) * s.rs;
    p.blocks = (int)max(1L, (N + p.rpb - 1) / p.rpb);
    p.partial_bytes = (size_t)p.blocks * 2 * F * sizeof(float);
    return p;
}

size_t bn_ws_bytes(long N, int F) { return bn_plan(N, F).partial_bytes + 2 * (size_t)F * sizeof(float); }

// column moments of a row block by the statistics pass of the norm (producers without a fused epilogue)
int col_moments(const float* x, long ldx, long N, int F, float* col_mean, float* col_m2, void* ws, size_t ws_bytes,
                hipStream_t st) {
    if (ws_bytes < bn_ws_bytes(N, F)) return fail(KAGNN_ERR_ARG, "%s: workspace too small", "col_moments");
    const BnShape s = bn_shape(F);
    const BnPlan p = bn_plan(N, F);
    float* partial = static_cast<float*>(ws);
    const size_t lds = (size_t)s.rs * 2 * 4 * s.cl * sizeof(float);
    bn_colsum_kernel<0><<<p.blocks, 256, lds, st>>>(x, ldx, nullptr, 0, N, F, nullptr, nullptr, s.cl, s.rs, p.rpb, partial, DropArgs{0, 0, 65536u, 1.0f});
    KAGNN_LAUNCH_CHECK();
    bn_finish_kernel<2><<<cdiv(F, 32), 1024, 0, st>>>(partial, p.blocks, F, N, x, 0.f, 0.f, col_mean, col_m2, nullptr, nullptr);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

int bn_fwd(const float* x, long ldx, long N, int F, const float* gamma, const float* beta, float* running_mean,
           float* running_var, float momentum, float eps, int training, const float* col_mean, const float* col_m2,
           float dropout_p, unsigned long long dropout_seed, float* y, long ldy, float* save_mean,
           float* save_rstd, void* ws, size_t ws_bytes, hipStream_t st) {
    if (ws_bytes < bn_ws_bytes(N, F)) return fail(KAGNN_ERR_ARG, "%s: workspace too small", "bn_fwd");
    const BnShape s = bn_shape(F);
    const BnPlan p = bn_plan(N, F);
    const DropArgs dr = drop_args(training ? dropout_p : 0.0f, dropout_seed);
    if (training && col_mean) {          // statistics and normalisation in one launch
        const int grid = (int)min(4096L, max(1L, (long)cdiv(N, s.rs)));
        bn_apply_from_moments_kernel<<<grid, 256, 0, st>>>(x, ldx, N, F, col_mean, col_m2, eps, momentum, save_mean, save_rstd, running_mean,
                                                          running_var, gamma, beta, y, ldy, s.cl, s.rs, dr);
        KAGNN_LAUNCH_CHECK();
        return KAGNN_OK;
    } else if (training) {
        float* partial = static_cast<float*>(ws);
        const size_t lds = (size_t)s.rs * 2 * 4 * s.cl * sizeof(float);
        bn_colsum_kernel<0><<<p.blocks, 256, lds, st>>>(x, ldx, nullptr, 0, N, F, nullptr, nullptr, s.cl, s.rs, p.rpb, partial, dr);
        KAGNN_LAUNCH_CHECK();
        bn_finish_kernel<0><<<cdiv(F, 32), 1024, 0, st>>>(partial, p.blocks, F, N, x, eps, momentum, save_mean, save_rstd,
                                                         running_mean, running_var);
        KAGNN_LAUNCH_CHECK();
    } else {
        KAGNN_HIP(hipMemcpyAsync(save_mean, running_mean, (size_t)F * sizeof(float), hipMemcpyDeviceToDevice, st));
        bn_rstd_kernel<<<cdiv(F, 256), 256, 0, st>>>(running_var, F, eps, save_rstd);
        KAGNN_LAUNCH_CHECK();
    }
    const int grid = (int)min(4096L, max(1L, (long)cdiv(N, s.rs)));
    bn_apply_kernel<<<grid, 256, 0, st>>>(x, ldx, N, F, save_mean, save_rstd, gamma, beta, y, ldy, s.cl, s.rs, dr);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// training-mode forward whose statistics are the producer's deferred partial moments (common.h: MomDefer): ONE launch
int bn_fwd_partial_moments(const float* x, long ldx, long N, int F, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float momentum, float eps, const float* partial, int P, float* y, long ldy,
                           float* save_mean, float* save_rstd, hipStream_t st) {
    if (P < 1 || P > kMomDeferMaxP) return fail(KAGNN_ERR_ARG, "%s: 1 .. 32 partial rows", "bn_fwd_partial_moments");
    const BnShape s = bn_shape(F);
    const int grid = (int)min(4096L, max(1L, (long)cdiv(N, s.rs)));
    bn_apply_from_partial_moments_kernel<<<grid, 256, 2 * (size_t)F * sizeof(float), st>>>(x, ldx, N, F, partial, P, eps, momentum, save_mean,
                                                                                          save_rstd, running_mean, running_var, gamma, beta,
                                                                                          y, ldy, s.cl, s.rs);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

int bn_stats_affine(const float* col_mean, const float* col_m2, long N, int F, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps, float* save_mean, float* save_rstd,
                    float* affine, hipStream_t st) {
    // (one launch: the statistics expressions of bn_apply_from_moments_kernel followed by bn_affine_kernel's, on values still in registers)
    bn_from_moments_affine_kernel<<<cdiv(F, 256), 256, 0, st>>>(col_mean, col_m2, N, F, eps, momentum, save_mean, save_rstd, running_mean,
                                                                running_var, gamma, beta, affine);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

constexpr int kBnTailMaxBlocks = 128;        // (more partial rows: the 1024-thread finish launch folds them faster than one workgroup would)
static bool bn_tail_enabled() {              // OPT-IN (KAGNN_BN_TAIL=1; bit-identical; read per call for the test): the ticket's device-wide
    const char* e = getenv("KAGNN_BN_TAIL");  // fence + atomic in EVERY workgroup and the 256-thread fold made the pass 26 us against
    return e != nullptr && atoi(e) != 0;      // 6.4 + 4.7 for the two launches (profiles/r06_experiments.md 3)
}

// the statistics half of the training backward: g_beta = sum g, g_gamma = sum g xhat, and the per-column table for a kernel
// that applies the backward to the rows itself (kan_split_dx_kernel<..., BNB>); workspace as bn_bwd
int bn_bwd_stats(const float* x, long ldx, const float* gy, long ldgy, long N, int F, const float* gamma, const float* save_mean,
                 const float* save_rstd, float* g_gamma, float* g_beta, float* tab, int ldt, void* ws, size_t ws_bytes, hipStream_t st) {
    if (ws_bytes < bn_ws_bytes(N, F)) return fail(KAGNN_ERR_ARG, "%s: workspace too small", "bn_bwd_stats");
    const BnShape s = bn_shape(F);
    const BnPlan p = bn_plan(N, F);
    const DropArgs dr = drop_args(0.0f, 0);
    float* partial = static_cast<float*>(ws);
    float* sums = reinterpret_cast<float*>(static_cast<char*>(ws) + p.partial_bytes);
    float* sg = g_beta ? g_beta : sums;
    float* sgx = g_gamma ? g_gamma : sums + F;
    const size_t lds = (size_t)s.rs * 2 * 4 * s.cl * sizeof(float);
    if (p.blocks <= kBnTailMaxBlocks && bn_tail_enabled()) {      // few partial rows (mini-batches): the last workgroup finishes -- one launch
        BnTail tail;
        tail.ticket = bn_next_ticket(); tail.gamma = gamma; tail.sum_gy = sg; tail.sum_gyx = sgx; tail.tab = tab; tail.ldt = ldt;
        bn_colsum_kernel<1><<<p.blocks, 256, lds, st>>>(x, ldx, gy, ldgy, N, F, save_mean, save_rstd, s.cl, s.rs, p.rpb, partial, dr, tail);
        KAGNN_LAUNCH_CHECK();
        return KAGNN_OK;
    }
    bn_colsum_kernel<1><<<p.blocks, 256, lds, st>>>(x, ldx, gy, ldgy, N, F, save_mean, save_rstd, s.cl, s.rs, p.rpb, partial, dr);
    KAGNN_LAUNCH_CHECK();
    bn_finish_table_kernel<<<cdiv(ldt, 32), 1024, 0, st>>>(partial, p.blocks, F, N, save_mean, save_rstd, gamma, sg, sgx, tab, ldt);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// ---- the norm's backward statistics arriving from ELSEWHERE (round 4): the aggregation that produced the incoming gradient g
// left partial row pairs [B][2][F] of sum g and sum g * xhat (aggregate.hip, AggArgs::st_*).  Two launches fold them in a fixed
// order: one workgroup per 64 partial row pairs adds them (32 MB at 1M rows: one row pair per 16 rows; 977 workgroups), then
// bn_finish_kernel<1> adds those 977 rows.
constexpr int kStatsFoldRows = 64;               // partial rows per workgroup of the first folding launch
__global__ __launch_bounds__(256) void stats_fold_kernel(const float* __restrict__ partial, long B, int F2 /* 2 * F */,
                                                         float* __restrict__ stage) {
    __shared__ float s_red[256];
    const long b0 = (long)blockIdx.x * kStatsFoldRows, b1 = min(B, b0 + kStatsFoldRows);
    for (int c0 = 0; c0 < F2; c0 += 64) {                 // 64 columns x 4 row slots per pass (uniform trip count: barriers inside)
        const int c = c0 + (threadIdx.x & 63), slot = threadIdx.x >> 6;
        float v[kStatsFoldRows / 4];
#pragma unroll
        for (int k = 0; k < kStatsFoldRows / 4; ++k) {    // all 16 loads of a thread in flight, then added in row order
            const long b = b0 + slot + 4 * k;
            v[k] = (c < F2 && b < b1) ? partial[b * F2 + c] : 0.0f;
        }
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < kStatsFoldRows / 4; ++k) acc += v[k];
        s_red[threadIdx.x] = acc;
        __syncthreads();
        if (slot == 0 && c < F2) stage[(long)blockIdx.x * F2 + c] = (s_red[threadIdx.x] + s_red[threadIdx.x + 64]) + (s_red[threadIdx.x + 128] + s_red[threadIdx.x + 192]);
        __syncthreads();
    }
}

static long stats_fold_blocks(long B) { return max(1L, (B + kStatsFoldRows - 1) / kStatsFoldRows); }
size_t bn_stats_fold_bytes(long B, int F) { return ((size_t)B + stats_fold_blocks(B)) * 2 * F * sizeof(float); }    // partial rows | fold stage

// partial [B][2][F] (at ws) -> sums[0][F] = sum g, sums[1][F] = sum g * xhat
int bn_sums_from_partials(float* ws, long B, int F, float* sums, hipStream_t st) {
    float* stage = ws + (size_t)B * 2 * F;
    const long nb = stats_fold_blocks(B);
    stats_fold_kernel<<<(unsigned)nb, 256, 0, st>>>(ws, B, 2 * F, stage);
    KAGNN_LAUNCH_CHECK();
    bn_finish_kernel<1><<<cdiv(F, 32), 1024, 0, st>>>(stage, nb, F, 0, nullptr, 0.f, 0.f, sums, sums + F, nullptr, nullptr);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// few partial row pairs [B][2][F] (one per workgroup of a persistent kernel): the finish launch alone
int bn_finish_partials(const float* partial, long B, int F, float* sums, hipStream_t st) {
    bn_finish_kernel<1><<<cdiv(F, 32), 1024, 0, st>>>(partial, B, F, 0, nullptr, 0.f, 0.f, sums, sums + F, nullptr, nullptr);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// the table of bn_finish_table_kernel + the two gradient outputs in one launch (the sums are given: nothing else to do before the table)
__global__ void bn_bwd_table_given_kernel(const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                          const float* __restrict__ sums, long N, int F, float* __restrict__ tab, int ldt,
                                          float* __restrict__ g_gamma, float* __restrict__ g_beta) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= ldt) return;
    float A = 0.0f, B = 0.0f, C = 0.0f, m = 0.0f;
    if (f < F) {
        const float sg = sums[f], sgx = sums[F + f];
        m = mean[f];
        bn_bwd_consts(rstd[f], gamma ? gamma[f] : 1.0f, sg, sgx, 1.0f / (float)N, A, B, C);
        if (g_beta) g_beta[f] = sg;
        if (g_gamma) g_gamma[f] = sgx;
    }
    tab[f] = m; tab[ldt + f] = A; tab[2 * ldt + f] = B; tab[3 * ldt + f] = C;
}

// bn_bwd_stats with the two column sums given (sums[0] = sum g, sums[1] = sum g * xhat) instead of a pass over g and x
int bn_bwd_stats_given(const float* sums, long N, int F, const float* gamma, const float* save_mean, const float* save_rstd,
                       float* g_gamma, float* g_beta, float* tab, int ldt, hipStream_t st) {
    bn_bwd_table_given_kernel<<<cdiv(ldt, 256), 256, 0, st>>>(save_mean, save_rstd, gamma, sums, N, F, tab, ldt, g_gamma, g_beta);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

int bn_bwd(const float* x, long ldx, const float* gy, long ldgy, long N, int F, const float* gamma,
           const float* save_mean, const float* save_rstd, int training, float dropout_p,
           unsigned long long dropout_seed, float* gx, long ldgx, float* g_gamma,
           float* g_beta, void* ws, size_t ws_bytes, hipStream_t st) {
    if (ws_bytes < bn_ws_bytes(N, F)) return fail(KAGNN_ERR_ARG, "%s: workspace too small", "bn_bwd");
    const BnShape s = bn_shape(F);
    const BnPlan p = bn_plan(N, F);
    const DropArgs dr = drop_args(training ? dropout_p : 0.0f, dropout_seed);
    float* partial = static_cast<float*>(ws);
    float* sums = reinterpret_cast<float*>(static_cast<char*>(ws) + p.partial_bytes);     // [2][F] when the caller wants no g_gamma / g_beta
    float* sg = g_beta ? g_beta : sums;
    float* sgx = g_gamma ? g_gamma : sums + F;
    const size_t lds = (size_t)s.rs * 2 * 4 * s.cl * sizeof(float);
    bn_colsum_kernel<1><<<p.blocks, 256, lds, st>>>(x, ldx, gy, ldgy, N, F, save_mean, save_rstd, s.cl, s.rs, p.rpb, partial, dr);
    KAGNN_LAUNCH_CHECK();
    bn_finish_kernel<1><<<cdiv(F, 32), 1024, 0, st>>>(partial, p.blocks, F, N, nullptr, 0.f, 0.f, sg, sgx, nullptr, nullptr);
    KAGNN_LAUNCH_CHECK();
    if (gx) {
        const int grid = (int)min(4096L, max(1L, (long)cdiv(N, s.rs)));
        bn_bwd_apply_kernel<<<grid, 256, 0, st>>>(x, ldx, gy, ldgy, N, F, save_mean, save_rstd, gamma, sg, sgx, training, gx, ldgx, s.cl, s.rs, dr);
        KAGNN_LAUNCH_CHECK();
    }
    return KAGNN_OK;
}

}  // namespace kagnn
