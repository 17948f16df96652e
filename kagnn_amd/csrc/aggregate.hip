// Neighbour aggregation over a CSR (sum), the MI355X replacement for the
// index_select -> scatter_add_ pair inside torch_geometric's MessagePassing.propagate
// (reached from node_classification_clean/models.py:48-56 GIKANLayer/GINConv and :31-37
// KAGCNConv/GCNConv).  HBM-bound: a destination row is owned by a group of
// LPR lanes that each keep one float4 of the row in registers, walk the row's neighbour list
// with four independent 16-byte gathers in flight, and write the row once -- no [E,F] message
// tensor, no atomics.  Rows above `hub_threshold` edges are split into segments handled by one
// workgroup each (agg_hub_kernel) so a 10^4-degree hub does not serialise a single lane group.
#include "common.h"

namespace kagnn {


__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void fma4(float4& a, float w, const float4& v) {
    a.x = fmaf(w, v.x, a.x); a.y = fmaf(w, v.y, a.y); a.z = fmaf(w, v.z, a.z); a.w = fmaf(w, v.w, a.w);
}

// weight of edge e into row i (0 drops it)
__device__ __forceinline__ float edge_w(const AggArgs& a, int e, int j, long i) {
    float w = a.ew ? a.ew[e] : 1.0f;
    if (a.in_scale) w *= a.in_scale[j];
    if (a.skip_self && j == (int)i) w = 0.0f;
    return w;
}

// the statistics side of a stored row (AggArgs::st_*): o and o * xhat for this lane's four columns
__device__ __forceinline__ void stats_products(const AggArgs& a, long row, int c4, const float4& o, float4& pa, float4& pb) {
    const float4 y = ld4(a.st_y + row * a.st_ldy + c4), m = ld4(a.st_mean + c4), q = ld4(a.st_rstd + c4);
    pa = o;
    pb = make_float4(o.x * ((y.x - m.x) * q.x), o.y * ((y.y - m.y) * q.y), o.z * ((y.z - m.z) * q.z), o.w * ((y.w - m.w) * q.w));
}

// STATS: also leave the workgroup's column sums of the rows it stores (and of their products with xhat) in
// a.st_partial[row group][2][F] -- rows of one workgroup meet in LDS and are added in row order (deterministic); hub rows
// contribute nothing here (their final value exists only in agg_hub_merge_kernel, which adds their share)
template <int LPR, bool STATS = false>
__global__ __launch_bounds__(256) void agg_rows_v4_kernel(AggArgs a) {
    // XCD placement (speed only -- nothing depends on it): workgroups reach the 8 XCDs round-robin (block b runs on XCD b % 8:
    // observed, not promised).  Each XCD takes ONE contiguous eighth of the row groups instead of every 8th one, so the rowptr / col
    // lines two neighbouring row groups share land in one L2 instead of two: -0.012 ms of 0.868 per step at the headline graph
    // (profiles/r06_experiments.md 6).  The launch rounds the grid up to whole eighths; `wg` is the row group, as blockIdx.x was.
    const unsigned wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const long gid = (wg * 256L + threadIdx.x) / LPR;
    const int c4 = (threadIdx.x % LPR) * 4;
    const bool live = gid < a.N && c4 < a.F;
    if (!STATS && !live) return;
    float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa;
    if (live) {
        const int s = a.rowptr[gid], t = a.rowptr[gid + 1];
        const bool hub = (t - s) > a.hub_threshold;
        float4 acc = ld4(a.x + gid * a.ldx + c4);
        const float sw = a.self_scale * (a.in_scale ? a.in_scale[gid] : 1.0f);
        acc.x *= sw; acc.y *= sw; acc.z *= sw; acc.w *= sw;
        if (!hub) {
            int e = s;
            for (; e + 4 <= t; e += 4) {
                const int j0 = a.col[e], j1 = a.col[e + 1], j2 = a.col[e + 2], j3 = a.col[e + 3];
                const float4 v0 = ld4(a.x + (long)j0 * a.ldx + c4);
                const float4 v1 = ld4(a.x + (long)j1 * a.ldx + c4);
                const float4 v2 = ld4(a.x + (long)j2 * a.ldx + c4);
                const float4 v3 = ld4(a.x + (long)j3 * a.ldx + c4);
                fma4(acc, edge_w(a, e, j0, gid), v0);
                fma4(acc, edge_w(a, e + 1, j1, gid), v1);
                fma4(acc, edge_w(a, e + 2, j2, gid), v2);
                fma4(acc, edge_w(a, e + 3, j3, gid), v3);
            }
            for (; e < t; ++e) {
                const int j = a.col[e];
                fma4(acc, edge_w(a, e, j, gid), ld4(a.x + (long)j * a.ldx + c4));
            }
        }
        const float os = a.out_scale ? a.out_scale[gid] : 1.0f;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) b = ld4(a.bias + c4);
        float4 o = make_float4(fmaf(os, acc.x, b.x), fmaf(os, acc.y, b.y), fmaf(os, acc.z, b.z), fmaf(os, acc.w, b.w));
        if (a.col_scale) {                         // gathered rows = col_scale * x + col_shift (see AggArgs); unit edge weights only.
            // Hub rows: `o` holds the self term, the shift is added here for ALL t - s edges, the merge kernel scales the segment sums
            const float4 cs = ld4(a.col_scale + c4), ch = ld4(a.col_shift + c4);
            const float cnt = (float)(t - s) + a.self_scale;
            o.x = fmaf(cs.x, o.x, cnt * ch.x); o.y = fmaf(cs.y, o.y, cnt * ch.y);
            o.z = fmaf(cs.z, o.z, cnt * ch.z); o.w = fmaf(cs.w, o.w, cnt * ch.w);
        }
        if (a.addend && !hub) {                    // (hub rows: agg_hub_merge_kernel adds it after the segments -- the order of the separate sum)
            const float4 d = ld4(a.addend + gid * a.lda + c4);
            o.x += d.x; o.y += d.y; o.z += d.z; o.w += d.w;
        }
        *reinterpret_cast<float4*>(a.out + gid * a.ldo + c4) = o;
        if (STATS && !hub) stats_products(a, gid, c4, o, pa, pb);
    }
    if constexpr (STATS) {
        __shared__ float4 s_a[256], s_b[256];
        s_a[threadIdx.x] = pa; s_b[threadIdx.x] = pb;
        __syncthreads();
        if (threadIdx.x < LPR && c4 < a.F && wg * 256L < a.N * LPR) {      // (the grid's round-up row groups own no partial slot)
            constexpr int R = 256 / LPR;
#pragma unroll 4
            for (int k = 1; k < R; ++k) {          // rows of the workgroup, in order
                const float4 u = s_a[k * LPR + threadIdx.x], v = s_b[k * LPR + threadIdx.x];
                pa.x += u.x; pa.y += u.y; pa.z += u.z; pa.w += u.w;
                pb.x += v.x; pb.y += v.y; pb.z += v.z; pb.w += v.w;
            }
            *reinterpret_cast<float4*>(a.st_partial + (wg * 2L + 0) * a.F + c4) = pa;
            *reinterpret_cast<float4*>(a.st_partial + (wg * 2L + 1) * a.F + c4) = pb;
        }
    }
}

// Narrow rows (F <= 16): with LPR = F/4 lanes per row a wave would own 64/LPR rows and run as long as the
// longest of their neighbour lists -- on power-law graphs several times the mean.  Here a row still gets 16
// lanes: LPR column groups x EP = 16/LPR edge slots; slot k walks edges s+k, s+k+EP, ... and the EP partial sums
// are combined across lanes in a fixed butterfly order (deterministic).
template <int LPR>
__global__ __launch_bounds__(256) void agg_rows_ep_kernel(AggArgs a) {
    constexpr int EP = 16 / LPR;
    const long gid = (blockIdx.x * 256L + threadIdx.x) >> 4;          // 16 lanes per row
    const int l16 = threadIdx.x & 15, cg = l16 % LPR, eg = l16 / LPR;
    const int c4 = cg * 4;
    const bool live = gid < a.N && c4 < a.F;
    const long row = min(gid, a.N - 1);
    const int cc = live ? c4 : 0;
    const int s = a.rowptr[row], t = a.rowptr[row + 1];
    const bool hub = (t - s) > a.hub_threshold;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!hub) {
        int e = s + eg;
        for (; e + EP < t; e += 2 * EP) {               // two independent gathers in flight per lane
            const int j0 = a.col[e], j1 = a.col[e + EP];
            const float4 v0 = ld4(a.x + (long)j0 * a.ldx + cc);
            const float4 v1 = ld4(a.x + (long)j1 * a.ldx + cc);
            fma4(acc, edge_w(a, e, j0, row), v0);
            fma4(acc, edge_w(a, e + EP, j1, row), v1);
        }
        if (e < t) {
            const int j = a.col[e];
            fma4(acc, edge_w(a, e, j, row), ld4(a.x + (long)j * a.ldx + cc));
        }
    }
#pragma unroll
    for (int o = LPR; o < 16; o <<= 1) {                 // sum over the edge slots (lanes cg + LPR*k)
        acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o);
        acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
    }
    if (live && eg == 0) {
        const float sw = a.self_scale * (a.in_scale ? a.in_scale[row] : 1.0f);
        const float4 xs = ld4(a.x + row * a.ldx + c4);
        const float os = a.out_scale ? a.out_scale[row] : 1.0f;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) b = ld4(a.bias + c4);
        float4 o = make_float4(fmaf(os, fmaf(sw, xs.x, acc.x), b.x), fmaf(os, fmaf(sw, xs.y, acc.y), b.y),
                               fmaf(os, fmaf(sw, xs.z, acc.z), b.z), fmaf(os, fmaf(sw, xs.w, acc.w), b.w));
        if (a.col_scale) {
            const float4 cs = ld4(a.col_scale + c4), ch = ld4(a.col_shift + c4);
            const float cnt = (float)(t - s) + a.self_scale;
            o.x = fmaf(cs.x, o.x, cnt * ch.x); o.y = fmaf(cs.y, o.y, cnt * ch.y);
            o.z = fmaf(cs.z, o.z, cnt * ch.z); o.w = fmaf(cs.w, o.w, cnt * ch.w);
        }
        if (a.addend && !hub) {
            const float4 d = ld4(a.addend + row * a.lda + c4);
            o.x += d.x; o.y += d.y; o.z += d.z; o.w += d.w;
        }
        *reinterpret_cast<float4*>(a.out + row * a.ldo + c4) = o;
    }
}

// one workgroup per hub segment {row, e0, e1}: its partial sum (fixed order inside the segment) goes to
// part[segment][F]; agg_hub_merge_kernel then folds a row's segments into out[row] in segment order -- no atomics,
// so hub rows are as reproducible run to run as every other row (reference utils.py:25-28 seeds everything and
// expects repeatable runs).
template <int LPR>
__global__ __launch_bounds__(256) void agg_hub_v4_kernel(AggArgs a, const int* __restrict__ seg,
                                                         float* __restrict__ part, int ldp) {
    __shared__ float4 s_part[256];
    const int row = seg[3 * blockIdx.x], e0 = seg[3 * blockIdx.x + 1], e1 = seg[3 * blockIdx.x + 2];
    constexpr int G = 256 / LPR;
    const int g = threadIdx.x / LPR, lg = threadIdx.x % LPR, c4 = lg * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < a.F) {
        int e = e0 + g;
        for (; e + 3 * G < e1; e += 4 * G) {   // four gathers of a lane group in flight; the sums keep the edge order
            const int j0 = a.col[e], j1 = a.col[e + G], j2 = a.col[e + 2 * G], j3 = a.col[e + 3 * G];
            const float4 v0 = ld4(a.x + (long)j0 * a.ldx + c4);
            const float4 v1 = ld4(a.x + (long)j1 * a.ldx + c4);
            const float4 v2 = ld4(a.x + (long)j2 * a.ldx + c4);
            const float4 v3 = ld4(a.x + (long)j3 * a.ldx + c4);
            fma4(acc, edge_w(a, e, j0, row), v0);
            fma4(acc, edge_w(a, e + G, j1, row), v1);
            fma4(acc, edge_w(a, e + 2 * G, j2, row), v2);
            fma4(acc, edge_w(a, e + 3 * G, j3, row), v3);
        }
        for (; e < e1; e += G) {
            const int j = a.col[e];
            fma4(acc, edge_w(a, e, j, row), ld4(a.x + (long)j * a.ldx + c4));
        }
    }
    s_part[threadIdx.x] = acc;
    __syncthreads();
    if (g == 0 && c4 < a.F) {
#pragma unroll 8                               // (fully unrolled at G = 64 / 128 the loads of all partials are hoisted: 510 registers, scratch)
        for (int k = 1; k < G; ++k) {          // fixed order inside the segment
            const float4 p = s_part[k * LPR + lg];
            acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
        }
        *reinterpret_cast<float4*>(part + (long)blockIdx.x * ldp + c4) = acc;
    }
}

// One workgroup per segment; only the workgroup of a row's FIRST segment works: its 256 / LPR lane groups take the row's
// consecutive segments (csr.hip lists them in edge order) round-robin, then the partial sums meet in LDS in a fixed
// order and are added, scaled, onto what the row kernel wrote (self term + bias).  A fixed association order is all
// bit-reproducibility needs; walking the ~80 segments of a 10^4-degree hub with ONE lane group took 28 us per launch.
// STATS (AggArgs::st_*): the hub row's share of the column statistics goes to partial row `st_row0 + segment index` (the
// workgroups of a row's later segments leave zeros there)
template <int LPR, bool STATS = false>
__global__ __launch_bounds__(256) void agg_hub_merge_kernel(AggArgs a, const int* __restrict__ seg, long nseg,
                                                            const float* __restrict__ part, int ldp, long st_row0 = 0) {
    __shared__ float4 s_part[256];
    const long sidx = blockIdx.x;
    const int row = seg[3 * sidx];
    constexpr int G = 256 / LPR;
    const int g = threadIdx.x / LPR, lg = threadIdx.x % LPR, c4 = lg * 4;
    if (sidx > 0 && seg[3 * (sidx - 1)] == row) {                // workgroup-uniform
        if (STATS && g == 0 && c4 < a.F) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(a.st_partial + ((st_row0 + sidx) * 2 + 0) * a.F + c4) = z;
            *reinterpret_cast<float4*>(a.st_partial + ((st_row0 + sidx) * 2 + 1) * a.F + c4) = z;
        }
        return;
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < a.F) {
        for (long k = sidx + g; k < nseg && seg[3 * k] == row; k += G) {
            const float4 p = ld4(part + k * ldp + c4);
            acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
        }
    }
    s_part[threadIdx.x] = acc;
    __syncthreads();
    if (g == 0 && c4 < a.F) {
#pragma unroll 8                               // (fully unrolled at G = 64 / 128 the loads of all partials are hoisted: 510 registers, scratch)
        for (int k = 1; k < G; ++k) {
            const float4 p = s_part[k * LPR + lg];
            acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
        }
        const float os = a.out_scale ? a.out_scale[row] : 1.0f;
        float4* o = reinterpret_cast<float4*>(a.out + (long)row * a.ldo + c4);
        float4 v = *o;
        if (a.col_scale) {                     // (the row kernel added the shift for every edge of the row)
            const float4 cs = ld4(a.col_scale + c4);
            acc.x *= cs.x; acc.y *= cs.y; acc.z *= cs.z; acc.w *= cs.w;
        }
        v.x = fmaf(os, acc.x, v.x); v.y = fmaf(os, acc.y, v.y); v.z = fmaf(os, acc.z, v.z); v.w = fmaf(os, acc.w, v.w);
        if (a.addend) {
            const float4 d = ld4(a.addend + (long)row * a.lda + c4);
            v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
        }
        *o = v;
        if constexpr (STATS) {
            float4 pa, pb;
            stats_products(a, row, c4, v, pa, pb);
            *reinterpret_cast<float4*>(a.st_partial + ((st_row0 + sidx) * 2 + 0) * a.F + c4) = pa;
            *reinterpret_cast<float4*>(a.st_partial + ((st_row0 + sidx) * 2 + 1) * a.F + c4) = pb;
        }
    }
}

// any F / any alignment: grid (N, ceil(F/256)), one thread per feature, edges walked in order.
__global__ __launch_bounds__(256) void agg_rows_generic_kernel(AggArgs a) {
    const long i = blockIdx.x;
    const int f = blockIdx.y * 256 + threadIdx.x;
    if (f >= a.F) return;
    const int s = a.rowptr[i], t = a.rowptr[i + 1];
    float acc = a.self_scale * (a.in_scale ? a.in_scale[i] : 1.0f) * a.x[i * a.ldx + f];
    for (int e = s; e < t; ++e) {
        const int j = a.col[e];
        acc = fmaf(edge_w(a, e, j, i), a.x[(long)j * a.ldx + f], acc);
    }
    const float os = a.out_scale ? a.out_scale[i] : 1.0f;
    float o = fmaf(os, acc, a.bias ? a.bias[f] : 0.0f);
    if (a.col_scale) o = fmaf(a.col_scale[f], o, ((float)(t - s) + a.self_scale) * a.col_shift[f]);
    a.out[i * a.ldo + f] = a.addend ? o + a.addend[i * a.lda + f] : o;
}

__global__ void gcn_deg_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, long N,
                               float* __restrict__ dis) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= N) return;
    int d = 1;                                    // the one self loop gcn_norm guarantees
    for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) d += (col[e] != (int)i);
    dis[i] = rsqrtf((float)d);
}

// ------------------------------------------------------------------ GINE message + pooling
// out[i,:] = self_scale*x[i,:] + sum_e relu(x[col[e],:] + edge_attr[perm[e],:]); wave per row.
__global__ __launch_bounds__(256) void gine_fwd_kernel(const float* __restrict__ x, long ldx,
                                                       const float* __restrict__ ea, long lde,
                                                       float* __restrict__ out, long ldo,
                                                       const int* __restrict__ rowptr,
                                                       const int* __restrict__ col,
                                                       const int* __restrict__ perm, long N, int F,
                                                       float self_scale) {
    const long i = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (i >= N) return;
    const int lane = threadIdx.x & 63;
    const int s = rowptr[i], t = rowptr[i + 1];
    for (int f = lane; f < F; f += 64) {
        float acc = self_scale * x[i * ldx + f];
        for (int e = s; e < t; ++e)
            acc += fmaxf(x[(long)col[e] * ldx + f] + ea[(long)perm[e] * lde + f], 0.0f);
        out[i * ldo + f] = acc;
    }
}

// on the TRANSPOSED structure (rows = source nodes j): for e in row j with target i = col[e],
// original edge id pe = perm[e]:  g = gout[i,:] * (x[j,:] + ea[pe,:] > 0);
// g_ea[pe,:] = g ; gx[j,:] = self_scale*gout[j,:] + sum_e g.   Deterministic, no atomics.
__global__ __launch_bounds__(256) void gine_bwd_kernel(const float* __restrict__ x, long ldx,
                                                       const float* __restrict__ ea, long lde,
                                                       const float* __restrict__ gout, long ldg,
                                                       float* __restrict__ gx, long ldgx,
                                                       float* __restrict__ gea, long ldge,
                                                       const int* __restrict__ rowptr,
                                                       const int* __restrict__ col,
                                                       const int* __restrict__ perm, long N, int F,
                                                       float self_scale, int gea_accumulate) {
    const long j = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (j >= N) return;
    const int lane = threadIdx.x & 63;
    const int s = rowptr[j], t = rowptr[j + 1];
    for (int f = lane; f < F; f += 64) {
        const float xj = x[j * ldx + f];
        float acc = self_scale * gout[j * ldg + f];
        for (int e = s; e < t; ++e) {
            const long pe = perm[e];
            const float g = (xj + ea[pe * lde + f] > 0.0f) ? gout[(long)col[e] * ldg + f] : 0.0f;
            // (every edge sits in exactly one source row: a plain read-modify-write, no atomics; accumulate = the edge attributes
            // feed several convolutions and their gradients add up in place -- kagnn_gine_kan_stack_bwd)
            if (gea) gea[pe * ldge + f] = gea_accumulate ? gea[pe * ldge + f] + g : g;
            acc += g;
        }
        gx[j * ldgx + f] = acc;
    }
}

__global__ __launch_bounds__(256) void segment_pool_kernel(const float* __restrict__ x, long ldx,
                                                           float* __restrict__ out, long ldo,
                                                           const int* __restrict__ seg, long B, int F,
                                                           int mean) {
    const long b = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    const int s = seg[b], t = seg[b + 1];
    const float inv = mean ? 1.0f / (float)max(t - s, 1) : 1.0f;
    for (int f = lane; f < F; f += 64) {
        float acc = 0.0f;
        for (int i = s; i < t; ++i) acc += x[(long)i * ldx + f];
        out[b * ldo + f] = acc * inv;
    }
}

__global__ __launch_bounds__(256) void segment_bcast_kernel(const float* __restrict__ g, long ldg,
                                                            float* __restrict__ gx, long ldgx,
                                                            const int* __restrict__ seg, long B, int F,
                                                            int mean) {
    const long b = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    const int s = seg[b], t = seg[b + 1];
    const float inv = mean ? 1.0f / (float)max(t - s, 1) : 1.0f;
    for (int f = lane; f < F; f += 64) {
        const float v = g[b * ldg + f] * inv;
        for (int i = s; i < t; ++i) gx[(long)i * ldgx + f] = v;
    }
}

// ------------------------------------------------------------------ embedding-table encoders of the graph-level models
// (reference graph_regression/models.py:244-281: AtomEncoder / BondEncoder -- out = sum over the integer feature columns of one table
// lookup each).  One launch per column each way instead of aten's gather (forward) and its sort-based embedding_dense_backward
// (~12 launches per table and step: radix sort, segment offsets, partial sums, scatter) -- on a 256-molecule mini-batch the two
// tables' backward was ~25 launches and ~170 us of a ~1.1 ms step.  The tables are tiny (21 atom / 4 bond types for ZINC, <= 119
// rows for OGB molecules), so the backward is one workgroup per (table row, 64-column chunk) scanning the index column: the rows that
// hit it are added in row order by four waves (rows w, w + 4, ...), the four partial sums in wave order -- deterministic.
// An index outside [0, V) (aten: a device-side assert) gives a NaN row in the forward and is skipped in the backward.
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const int64_t* __restrict__ idx, long stride, long N,
                                                            const float* __restrict__ table, int V, int F,
                                                            float* __restrict__ out, long ldo, int accumulate) {
    const long i = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (i >= N) return;
    const int lane = threadIdx.x & 63;
    const int64_t v = idx[i * stride];
    const bool ok = v >= 0 && v < V;
    for (int f = lane; f < F; f += 64) {
        const float t = ok ? table[v * F + f] : __builtin_nanf("");
        out[i * ldo + f] = accumulate ? out[i * ldo + f] + t : t;
    }
}

// backward, phase 1: a workgroup of WAVES waves per (block of kEmbRows rows, 64-column chunk); every wave walks ITS kEmbRows / WAVES
// rows IN ORDER and adds each gradient row into its own LDS copy of the table row it hit (the index is wave-uniform: one LDS row
// per step, lane = column, no conflicts), then the waves' tables are added in wave order -> partial[block][V][F].  Phase 2 adds the
// blocks in block order: deterministic, no atomics.  The rows are fetched 16 at a time with UNCONDITIONAL loads (16 index + 16
// gradient loads in flight; the first form -- `if (ok) s_acc[v] += g[i]` per row -- compiled to a branch and a wait on the index
// load per row: 43 us for 12.7k edges x 64 columns, whatever the row count), and the adds are plain LDS read-modify-writes:
// `ds_add_f32` runs at ~150 ns per wave instruction on gfx950 (tools/experiments/embedding_bwd_scaling.py: 128 rows = 20 us, and
// four waves of a CU take twice that), slower than the round trip it was meant to avoid.
// (The very first version had one workgroup per table row scan the whole index column: 0.5 ms.)
constexpr int kEmbRows = 128;
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void embedding_bwd_partial_kernel(const int64_t* __restrict__ idx, long stride, long N,
                                                                           const float* __restrict__ g, long ldg, int V, int F,
                                                                           float* __restrict__ partial) {
    extern __shared__ float s_acc[];                    // [WAVES][V][64]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), f = blockIdx.y * 64 + lane;
    float* acc = s_acc + (size_t)wave * V * 64;
    for (int k = lane; k < V * 64; k += 64) acc[k] = 0.0f;
    __builtin_amdgcn_wave_barrier();
    constexpr int RPW = kEmbRows / WAVES;               // rows per wave: a workgroup covers kEmbRows rows either way
    const long r0 = ((long)blockIdx.x * WAVES + wave) * RPW, r1 = min(N, r0 + RPW);
    const bool col_ok = f < F;
    const int fc = min(f, F - 1);                       // (unconditional loads: a predicated one costs a branch and a wait per row)
    for (long i0 = r0; i0 < r1; i0 += 16) {
        int64_t iv[16];
        float gv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) iv[u] = idx[min(i0 + u, r1 - 1) * stride];      // (clamped: the tail repeats the last row, masked below)
#pragma unroll
        for (int u = 0; u < 16; ++u) gv[u] = g[min(i0 + u, r1 - 1) * ldg + fc];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const bool hit = i0 + u < r1 && iv[u] >= 0 && iv[u] < V;                   // wave-uniform
            const float val = col_ok ? gv[u] : 0.0f;
            if (hit) acc[(int)iv[u] * 64 + lane] += val;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < V * 64; k += 64 * WAVES) {
        const int v = k >> 6, ff = blockIdx.y * 64 + (k & 63);
        float a = s_acc[k];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) a += s_acc[(size_t)w * V * 64 + k];
        if (ff < F) partial[((long)blockIdx.x * V + v) * F + ff] = a;
    }
}

// phase 2: 32 table elements x 8 contiguous ranges of blocks per workgroup; a thread adds its range in block order (eight loads in
// flight), the eight range sums meet in LDS and are added in range order
__global__ __launch_bounds__(256) void embedding_bwd_reduce_kernel(const float* __restrict__ partial, int nb, int V, int F,
                                                                   float* __restrict__ g_table) {
    __shared__ float s_part[8][32];
    const int e = threadIdx.x & 31, sub = threadIdx.x >> 5;
    const long per = (long)V * F, k = blockIdx.x * 32L + e;
    const int chunk = (nb + 7) / 8, b0 = sub * chunk, b1 = min(nb, b0 + chunk);
    float a = 0.0f;
    if (k < per) {
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = partial[(b + u) * per + k];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += t[u];
        }
        for (; b < b1; ++b) a += partial[b * per + k];
    }
    s_part[sub][e] = a;
    __syncthreads();
    if (sub == 0 && k < per) {
        float t = s_part[0][e];
#pragma unroll
        for (int q = 1; q < 8; ++q) t += s_part[q][e];
        g_table[k] = t;
    }
}

int embedding_fwd(const int64_t* idx, long stride, long N, const float* table, int V, int F, float* out, long ldo, int accumulate,
                  hipStream_t st) {
    if (N == 0) return KAGNN_OK;
    embedding_fwd_kernel<<<cdiv(N, 4), 256, 0, st>>>(idx, stride, N, table, V, F, out, ldo, accumulate);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

size_t embedding_bwd_ws_bytes(long N, int V, int F) { return (size_t)cdiv(max(N, 1L), kEmbRows) * V * F * sizeof(float); }

int embedding_bwd(const int64_t* idx, long stride, long N, const float* g, long ldg, int V, int F, float* g_table, float* ws,
                  size_t ws_bytes, hipStream_t st) {
    if (V == 0) return KAGNN_OK;
    if (V > 512) return fail(KAGNN_ERR_UNSUPPORTED, "%s: tables of at most 512 rows (the categorical encoders of the graph-level models)", "embedding_bwd");
    if (ws_bytes < embedding_bwd_ws_bytes(N, V, F)) return fail(KAGNN_ERR_ARG, "%s: workspace too small", "embedding_bwd");
    // four waves per workgroup (four LDS tables, 32 rows each) while they fit, then two, then one.  Either way
    // cdiv(N, kEmbRows) partial tables (what embedding_bwd_ws_bytes sizes)
    const size_t tbl = (size_t)V * 64 * sizeof(float);
    const int waves = 4 * tbl <= 64 * 1024 ? 4 : 2 * tbl <= 128 * 1024 ? 2 : 1;
    const int nb = cdiv(max(N, 1L), (long)kEmbRows);
    const size_t lds = waves * tbl;
    if (lds > 64 * 1024) {
        static unsigned long long configured = 0;
        if (auto first_use_ = first_use_on_this_device(configured)) {
            KAGNN_HIP(hipFuncSetAttribute((const void*)embedding_bwd_partial_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024 + 1024));
            KAGNN_HIP(hipFuncSetAttribute((const void*)embedding_bwd_partial_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024 + 1024));
        }
    }
    const dim3 grid((unsigned)nb, (unsigned)cdiv(F, 64));
    if (waves == 4) embedding_bwd_partial_kernel<4><<<grid, 256, lds, st>>>(idx, stride, N, g, ldg, V, F, ws);
    else if (waves == 2) embedding_bwd_partial_kernel<2><<<grid, 128, lds, st>>>(idx, stride, N, g, ldg, V, F, ws);
    else embedding_bwd_partial_kernel<1><<<grid, 64, lds, st>>>(idx, stride, N, g, ldg, V, F, ws);
    KAGNN_LAUNCH_CHECK();
    embedding_bwd_reduce_kernel<<<cdiv((long)V * F, 32), 256, 0, st>>>(ws, nb, V, F, g_table);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// ------------------------------------------------------------------ launchers
static bool vec4_ok(const AggArgs& a) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return a.F % 4 == 0 && a.F <= 256 && a.ldx % 4 == 0 && a.ldo % 4 == 0 && al(a.x) && al(a.out) &&
           (!a.bias || al(a.bias)) && (!a.addend || (al(a.addend) && a.lda % 4 == 0)) &&
           (!a.col_scale || (al(a.col_scale) && al(a.col_shift)));
}

size_t aggregate_ws_bytes(long num_hub_seg, int F) { return (size_t)num_hub_seg * (size_t)((F + 3) & ~3) * sizeof(float); }

// only the hub part of aggregate_sum (segment partial sums + ordered fold ONTO a.out, which already holds the rows' self
// terms): for the forward kernel that produces the other rows itself (kan_sparse_fwd_agg)
int aggregate_hub_rows(const AggArgs& a, const int* hub_seg, long num_hub_seg, float* ws, size_t ws_bytes, hipStream_t st) {
    if (num_hub_seg <= 0) return KAGNN_OK;
    const int ldp = (a.F + 3) & ~3;
    if ((size_t)num_hub_seg * ldp * sizeof(float) > ws_bytes) return fail(KAGNN_ERR_ARG, "%s: workspace too small", "aggregate_hub_rows");
#define HUBS(LPR)                                                                                   \
    {                                                                                               \
        agg_hub_v4_kernel<LPR><<<(unsigned)num_hub_seg, 256, 0, st>>>(a, hub_seg, ws, ldp);         \
        KAGNN_LAUNCH_CHECK();                                                                       \
        agg_hub_merge_kernel<LPR><<<(unsigned)num_hub_seg, 256, 0, st>>>(a, hub_seg, num_hub_seg, ws, ldp); \
        KAGNN_LAUNCH_CHECK();                                                                       \
    }
    if (a.F <= 4) HUBS(1) else if (a.F <= 8) HUBS(2) else if (a.F <= 16) HUBS(4) else if (a.F <= 32) HUBS(8)
    else if (a.F <= 64) HUBS(16) else if (a.F <= 128) HUBS(32) else HUBS(64)
#undef HUBS
    return KAGNN_OK;
}

// column statistics of the result (AggArgs::st_*): which shapes the row kernels cover, and how many partial row pairs
// [2][F] they leave in st_partial (row workgroups first, then one slot per hub segment)
static int stats_lpr(int F) { return F <= 32 ? 8 : F <= 64 ? 16 : F <= 128 ? 32 : 64; }
bool aggregate_stats_ok(const AggArgs& a) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return vec4_ok(a) && a.F > 16 && a.st_y && a.st_mean && a.st_rstd && a.st_partial && a.st_ldy % 4 == 0 && al(a.st_y) &&
           al(a.st_mean) && al(a.st_rstd) && al(a.st_partial);
}
long aggregate_stats_rows(long N, int F, long num_hub_seg) { return (long)cdiv(N * stats_lpr(F), 256) + (num_hub_seg > 0 ? num_hub_seg : 0); }

int aggregate_sum(const AggArgs& a, const int* hub_seg, long num_hub_seg, float* ws, size_t ws_bytes, hipStream_t st) {
    if (a.N == 0) return KAGNN_OK;
    const bool stats = a.st_partial != nullptr;
    if (stats && !aggregate_stats_ok(a)) return fail(KAGNN_ERR_UNSUPPORTED, "%s: column statistics need 16-byte aligned fp32 rows of 17..256 columns", "aggregate_sum");
    if (!vec4_ok(a)) {
        AggArgs b = a;
        b.hub_threshold = 0x7fffffff;
        dim3 grid((unsigned)a.N, cdiv(a.F, 256));
        agg_rows_generic_kernel<<<grid, 256, 0, st>>>(b);
        KAGNN_LAUNCH_CHECK();
        return KAGNN_OK;
    }
    AggArgs b = a;
    if (num_hub_seg == 0 || hub_seg == nullptr) b.hub_threshold = 0x7fffffff;
    const int ldp = (a.F + 3) & ~3;
    if (b.hub_threshold != 0x7fffffff && (ws == nullptr || ws_bytes < aggregate_ws_bytes(num_hub_seg, a.F)))
        return fail(KAGNN_ERR_ARG, "%s: workspace too small for the hub segments (see kagnn_aggregate_workspace_bytes)", "aggregate_sum");
#define HUBS(LPR, ST)                                                                               \
    if (b.hub_threshold != 0x7fffffff) {                                                            \
        agg_hub_v4_kernel<LPR><<<(unsigned)num_hub_seg, 256, 0, st>>>(b, hub_seg, ws, ldp);         \
        KAGNN_LAUNCH_CHECK();                                                                       \
        agg_hub_merge_kernel<LPR, ST><<<(unsigned)num_hub_seg, 256, 0, st>>>(b, hub_seg, num_hub_seg, ws, ldp, (long)cdiv(a.N * LPR, 256)); \
        KAGNN_LAUNCH_CHECK();                                                                       \
    }
#define AGG_ROWS_GRID(threads) (cdiv(cdiv((threads), 256), 8) * 8)        // whole eighths: one per XCD (see the kernel)
#define ROWS(LPR)                                                                     \
    {                                                                                 \
        if (stats) {                                                                  \
            agg_rows_v4_kernel<LPR, true><<<AGG_ROWS_GRID(a.N * LPR), 256, 0, st>>>(b);   \
            KAGNN_LAUNCH_CHECK();                                                     \
            HUBS(LPR, true)                                                           \
        } else {                                                                      \
            agg_rows_v4_kernel<LPR><<<AGG_ROWS_GRID(a.N * LPR), 256, 0, st>>>(b);       \
            KAGNN_LAUNCH_CHECK();                                                     \
            HUBS(LPR, false)                                                          \
        }                                                                             \
    }
#define ROWS_EP(LPR)                                                                  \
    {                                                                                 \
        agg_rows_ep_kernel<LPR><<<cdiv(a.N * 16, 256), 256, 0, st>>>(b);              \
        KAGNN_LAUNCH_CHECK();                                                         \
        HUBS(LPR, false)                                                              \
    }
    // measured at N=1M / E=10M: edge-parallel wins for F <= 16 (0.19 vs 0.23 ms at F=8), loses slightly at F=32
    if (a.F <= 4) ROWS_EP(1) else if (a.F <= 8) ROWS_EP(2) else if (a.F <= 16) ROWS_EP(4) else if (a.F <= 32) ROWS(8)
    else if (a.F <= 64) ROWS(16)
    else if (a.F <= 128) ROWS(32) else ROWS(64)
#undef ROWS
#undef AGG_ROWS_GRID
#undef ROWS_EP
#undef HUBS
    return KAGNN_OK;
}

int gcn_deg_inv_sqrt(const int* rowptr, const int* col, long N, float* dis, hipStream_t st) {
    if (N == 0) return KAGNN_OK;
    gcn_deg_kernel<<<cdiv(N, 256), 256, 0, st>>>(rowptr, col, N, dis);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

int gine_fwd(const float* x, long ldx, const float* ea, long lde, float* out, long ldo,
             const int* rowptr, const int* col, const int* perm, long N, int F, float self_scale,
             hipStream_t st) {
    if (N == 0) return KAGNN_OK;
    gine_fwd_kernel<<<cdiv(N, 4), 256, 0, st>>>(x, ldx, ea, lde, out, ldo, rowptr, col, perm, N, F, self_scale);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

int gine_bwd(const float* x, long ldx, const float* ea, long lde, const float* gout, long ldg,
             float* gx, long ldgx, float* gea, long ldge, const int* rowptr, const int* col,
             const int* perm, long N, int F, float self_scale, hipStream_t st, int gea_accumulate) {
    if (N == 0) return KAGNN_OK;
    gine_bwd_kernel<<<cdiv(N, 4), 256, 0, st>>>(x, ldx, ea, lde, gout, ldg, gx, ldgx, gea, ldge, rowptr, col, perm, N, F, self_scale, gea_accumulate);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

int segment_pool(const float* x, long ldx, float* out, long ldo, const int* seg, long B, int F,
                 int mean, hipStream_t st) {
    if (B == 0) return KAGNN_OK;
    segment_pool_kernel<<<cdiv(B, 4), 256, 0, st>>>(x, ldx, out, ldo, seg, B, F, mean);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

int segment_bcast(const float* g, long ldg, float* gx, long ldgx, const int* seg, long B, int F,
                  int mean, hipStream_t st) {
    if (B == 0) return KAGNN_OK;
    segment_bcast_kernel<<<cdiv(B, 4), 256, 0, st>>>(g, ldg, gx, ldgx, seg, B, F, mean);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

}  // namespace kagnn
