// Attention aggregation of the KAGAT / FASTKAGAT convolutions (reference node_classification_clean/models.py:39-46,
// 76-83: torch_geometric 2.5.3 GATConv whose `lin` is a KAN layer).  xh = lin(x) viewed as [N, H, C]:
//
//   a_s[j,h] = <xh[j,h,:], att_src[h,:]>,  a_d[i,h] = <xh[i,h,:], att_dst[h,:]>
//   e_ij = leaky_relu(a_s[j,h] + a_d[i,h], 0.2)   over the edges j -> i, existing self loops removed, one self loop added
//   alpha_ij = softmax_j(e_ij);   out[i,h,:] = sum_j alpha_ij * xh[j,h,:]  (+ bias)
//
// One group of 16 lanes owns one (destination row, head) and walks the row's neighbour list ONCE with an online
// softmax (running maximum m and normaliser z, the partial sum rescaled when the maximum moves): no [E, H] attention
// tensor and no segment-max / segment-sum passes.  m and z are kept for the backward, which recomputes alpha_ij:
//   S_i = <g_out_i, out_i>;  g_pre_ij = alpha_ij (<g_out_i, xh_j> - S_i) * leaky'(.)          (by-destination pass)
//   g_xh_j = sum_i alpha_ij g_out_i + (sum_i g_pre_ij) att_src + (sum_j' g_pre_jj') att_dst   (by-source pass)
#include "common.h"

namespace kagnn {

constexpr float kSlope = 0.2f;
__device__ __forceinline__ float lrelu(float v) { return v > 0.0f ? v : kSlope * v; }
constexpr int kGatMaxK = 8;          // columns per lane: C <= 16 * kGatMaxK = 128 per head

__global__ __launch_bounds__(256) void gat_logits_kernel(const float* __restrict__ xh, long ld, long N, int H, int C,
                                                         const float* __restrict__ att_src,
                                                         const float* __restrict__ att_dst, float* __restrict__ a_s,
                                                         float* __restrict__ a_d) {
    const long gid = (blockIdx.x * 256L + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    if (gid >= N * H) return;
    const long i = gid / H; const int h = gid % H;
    const float* row = xh + i * ld + (long)h * C;
    float s = 0.0f, d = 0.0f;
    for (int c = l; c < C; c += 16) { const float v = row[c]; s = fmaf(v, att_src[h * C + c], s); d = fmaf(v, att_dst[h * C + c], d); }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) { s += __shfl_xor(s, o); d += __shfl_xor(d, o); }
    if (l == 0) { a_s[gid] = s; a_d[gid] = d; }
}

__global__ __launch_bounds__(256) void gat_fwd_kernel(const float* __restrict__ xh, long ld, const float* __restrict__ a_s,
                                                      const float* __restrict__ a_d, const int* __restrict__ rowptr,
                                                      const int* __restrict__ col, long N, int H, int C,
                                                      const float* __restrict__ bias, float* __restrict__ out, long ldo,
                                                      float* __restrict__ m_out, float* __restrict__ z_out, int hub_threshold) {
    // one WAVE per (row, head): its four 16-lane groups take every 4th edge each (a 500-edge row is 125 dependent
    // steps instead of 500 -- the kernel's run time is its longest row) and merge their softmax states with two
    // butterfly steps across the groups, always in the same order
    const long gid = (blockIdx.x * 256L + threadIdx.x) >> 6;
    const int l = threadIdx.x & 15, grp = (threadIdx.x >> 4) & 3;
    if (gid >= N * H) return;
    const long i = gid / H; const int h = gid % H;
    if (rowptr[i + 1] - rowptr[i] > hub_threshold) return;      // long rows: gat_fwd_hub_kernel
    const float ad = a_d[gid];
    float acc[kGatMaxK];
    float m = -3.0e38f, z = 0.0f;
#pragma unroll
    for (int k = 0; k < kGatMaxK; ++k) acc[k] = 0.0f;
    if (grp == 0) {                                       // the added self loop
        m = lrelu(a_s[gid] + ad); z = 1.0f;
#pragma unroll
        for (int k = 0; k < kGatMaxK; ++k) { const int c = l + 16 * k; acc[k] = c < C ? xh[i * ld + (long)h * C + c] : 0.0f; }
    }
    for (int e = rowptr[i] + grp; e < rowptr[i + 1]; e += 4) {
        const int j = col[e];
        if (j == (int)i) continue;                       // existing self loops are removed
        const float v = lrelu(a_s[(long)j * H + h] + ad);
        if (v > m) {                                     // group-uniform
            const float sc = __expf(m - v);
            z *= sc;
#pragma unroll
            for (int k = 0; k < kGatMaxK; ++k) acc[k] *= sc;
            m = v;
        }
        const float p = __expf(v - m);
        z += p;
        const float* xj = xh + (long)j * ld + (long)h * C;
#pragma unroll
        for (int k = 0; k < kGatMaxK; ++k) { const int c = l + 16 * k; if (c < C) acc[k] = fmaf(p, xj[c], acc[k]); }
    }
    // merge the four groups: common maximum, rescale, butterfly sums (lanes l, l+16, l+32, l+48)
    const float M = fmaxf(fmaxf(m, __shfl_xor(m, 16)), fmaxf(__shfl_xor(m, 32), __shfl_xor(m, 48)));
    const float sc = __expf(m - M);
    z *= sc;
    z += __shfl_xor(z, 16); z += __shfl_xor(z, 32);
    const float inv = 1.0f / z;
#pragma unroll
    for (int k = 0; k < kGatMaxK; ++k) {
        if (16 * k < C) {                                 // wave-uniform
            float a = acc[k] * sc;
            a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
            const int c = l + 16 * k;
            if (grp == 0 && c < C) out[i * ldo + (long)h * C + c] = fmaf(a, inv, bias ? bias[h * C + c] : 0.0f);
        }
    }
    if (threadIdx.x % 64 == 0) { m_out[gid] = M; z_out[gid] = z; }
}

// Rows with more than hub_threshold edges (power-law hubs): one workgroup per (row, head); its 16 lane groups each
// walk every 16th edge with their own online softmax, the 16 partial (m, z, acc) states are merged through LDS in
// a fixed order.  Launched over the hub SEGMENT list of the CSR build; only a row's first segment does the work.
__global__ __launch_bounds__(256) void gat_fwd_hub_kernel(const float* __restrict__ xh, long ld, const float* __restrict__ a_s,
                                                          const float* __restrict__ a_d, const int* __restrict__ rowptr,
                                                          const int* __restrict__ col, const int* __restrict__ seg, int H,
                                                          int C, const float* __restrict__ bias, float* __restrict__ out,
                                                          long ldo, float* __restrict__ m_out, float* __restrict__ z_out) {
    __shared__ float s_m[16], s_z[16], s_acc[16][16 * kGatMaxK];
    const int sgi = blockIdx.x / H, h = blockIdx.x % H;
    const long i = seg[3 * sgi];
    if (seg[3 * sgi + 1] != rowptr[i]) return;            // not the first segment of its row
    const int grp = threadIdx.x >> 4, l = threadIdx.x & 15;
    const long gid = i * H + h;
    const float ad = a_d[gid];
    float acc[kGatMaxK];
    float m = -3.0e38f, z = 0.0f;
#pragma unroll
    for (int k = 0; k < kGatMaxK; ++k) acc[k] = 0.0f;
    if (grp == 0) {                                       // the added self loop
        m = lrelu(a_s[gid] + ad); z = 1.0f;
#pragma unroll
        for (int k = 0; k < kGatMaxK; ++k) { const int c = l + 16 * k; acc[k] = c < C ? xh[i * ld + (long)h * C + c] : 0.0f; }
    }
    const int e_end = rowptr[i + 1];
    for (int e0 = rowptr[i] + grp; e0 < e_end; e0 += 64) {   // four edges (stride 16) per step: gathers issued together
        int j[4]; float v[4]; float xv[4][kGatMaxK];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            j[q] = col[min(e0 + 16 * q, e_end - 1)];
            v[q] = lrelu(a_s[(long)j[q] * H + h] + ad);
            const float* xj = xh + (long)j[q] * ld + (long)h * C;
#pragma unroll
            for (int k = 0; k < kGatMaxK; ++k) { const int c = l + 16 * k; xv[q][k] = c < C ? xj[c] : 0.0f; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (e0 + 16 * q >= e_end) break;                // group-uniform
            if (j[q] == (int)i) continue;
            if (v[q] > m) {
                const float sc = __expf(m - v[q]);
                z *= sc;
#pragma unroll
                for (int k = 0; k < kGatMaxK; ++k) acc[k] *= sc;
                m = v[q];
            }
            const float p = __expf(v[q] - m);
            z += p;
#pragma unroll
            for (int k = 0; k < kGatMaxK; ++k) acc[k] = fmaf(p, xv[q][k], acc[k]);
        }
    }
    if (l == 0) { s_m[grp] = m; s_z[grp] = z; }
#pragma unroll
    for (int k = 0; k < kGatMaxK; ++k) s_acc[grp][l + 16 * k] = acc[k];
    __syncthreads();
    if (grp == 0) {
        float M = s_m[0];
        for (int q = 1; q < 16; ++q) M = fmaxf(M, s_m[q]);
        float Z = 0.0f;
#pragma unroll
        for (int k = 0; k < kGatMaxK; ++k) acc[k] = 0.0f;
        for (int q = 0; q < 16; ++q) {                    // fixed order
            const float sc = __expf(s_m[q] - M);
            Z = fmaf(s_z[q], sc, Z);
#pragma unroll
            for (int k = 0; k < kGatMaxK; ++k) acc[k] = fmaf(s_acc[q][l + 16 * k], sc, acc[k]);
        }
        const float inv = 1.0f / Z;
#pragma unroll
        for (int k = 0; k < kGatMaxK; ++k) {
            const int c = l + 16 * k;
            if (c < C) out[i * ldo + (long)h * C + c] = fmaf(acc[k], inv, bias ? bias[h * C + c] : 0.0f);
        }
        if (l == 0) { m_out[gid] = M; z_out[gid] = Z; }
    }
}

// by-destination pass of the backward: g_pre per edge (indexed by the ORIGINAL edge id through perm) and per self
// loop, and g_d[i,h] = sum_j g_pre_ij
__global__ __launch_bounds__(256) void gat_bwd_dst_kernel(const float* __restrict__ xh, long ld, const float* __restrict__ gout,
                                                          long ldg, const float* __restrict__ y, long ldy,
                                                          const float* __restrict__ bias, const float* __restrict__ a_s,
                                                          const float* __restrict__ a_d, const float* __restrict__ m_in,
                                                          const float* __restrict__ z_in, const int* __restrict__ rowptr,
                                                          const int* __restrict__ col, const int* __restrict__ perm, long N,
                                                          int H, int C, float* __restrict__ gpre, float* __restrict__ gpre_self,
                                                          float* __restrict__ g_d, int hub_threshold,
                                                          const int* __restrict__ seg /* non-null: hub launch, one workgroup per (segment, head) */) {
    // row launch: 16 (row, head) groups per workgroup, each walks its whole neighbour list, hub rows skipped;
    // hub launch: the workgroup of a hub row's FIRST segment spreads the row's edges over its 16 groups
    __shared__ float s_gd[16];
    long gid; int first, stride;
    const int l = threadIdx.x & 15, grp = threadIdx.x >> 4;
    if (seg) {
        const int sgi = blockIdx.x / H;
        const long row = seg[3 * sgi];
        if (seg[3 * sgi + 1] != rowptr[row]) return;
        gid = row * H + blockIdx.x % H; first = grp; stride = 16;
    } else {
        gid = (blockIdx.x * 256L + threadIdx.x) >> 6; first = grp & 3; stride = 4;    // one wave per (row, head)
        if (gid >= N * H) return;
        const long row = gid / H;
        if (rowptr[row + 1] - rowptr[row] > hub_threshold) return;
    }
    const long i = gid / H; const int h = gid % H;
    const float ad = a_d[gid], m = m_in[gid], inv = 1.0f / z_in[gid];
    float g[kGatMaxK];
    float S = 0.0f, gself = 0.0f;
#pragma unroll
    for (int k = 0; k < kGatMaxK; ++k) {
        const int c = l + 16 * k;
        g[k] = c < C ? gout[i * ldg + (long)h * C + c] : 0.0f;
        if (c < C) {
            S = fmaf(g[k], y[i * ldy + (long)h * C + c] - (bias ? bias[h * C + c] : 0.0f), S);
            gself = fmaf(g[k], xh[i * ld + (long)h * C + c], gself);
        }
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) { S += __shfl_xor(S, o); gself += __shfl_xor(gself, o); }
    float gd = 0.0f;
    if (first == 0) {
        const float pre = a_s[gid] + ad;
        const float alpha = __expf(lrelu(pre) - m) * inv;
        const float gp = alpha * (gself - S) * (pre > 0.0f ? 1.0f : kSlope);
        gd = gp;
        if (l == 0) gpre_self[gid] = gp;
    }
    const int e_end = rowptr[i + 1];
    for (int e0 = rowptr[i] + first; e0 < e_end; e0 += 4 * stride) {   // four edges per step: gathers issued together
        int j[4], eid[4]; float asj[4]; float xv[4][kGatMaxK];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = min(e0 + q * stride, e_end - 1);
            j[q] = col[e]; eid[q] = perm[e];
            asj[q] = a_s[(long)j[q] * H + h];
            const float* xj = xh + (long)j[q] * ld + (long)h * C;
#pragma unroll
            for (int k = 0; k < kGatMaxK; ++k) { const int c = l + 16 * k; xv[q][k] = c < C ? xj[c] : 0.0f; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (e0 + q * stride >= e_end) break;            // group-uniform
            if (j[q] == (int)i) { if (l == 0) gpre[(long)eid[q] * H + h] = 0.0f; continue; }
            float ga = 0.0f;
#pragma unroll
            for (int k = 0; k < kGatMaxK; ++k) ga = fmaf(g[k], xv[q][k], ga);
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) ga += __shfl_xor(ga, o);
            const float pre = asj[q] + ad;
            const float alpha = __expf(lrelu(pre) - m) * inv;
            const float gp = alpha * (ga - S) * (pre > 0.0f ? 1.0f : kSlope);
            gd += gp;
            if (l == 0) gpre[(long)eid[q] * H + h] = gp;
        }
    }
    if (!seg) {                                           // row launch: sum over the wave's four groups
        gd += __shfl_xor(gd, 16); gd += __shfl_xor(gd, 32);
        if (threadIdx.x % 64 == 0) g_d[gid] = gd;
    } else {                                              // hub launch: fixed-order sum over the 16 groups
        if (l == 0) s_gd[grp] = gd;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.0f;
            for (int q = 0; q < 16; ++q) t += s_gd[q];
            g_d[gid] = t;
        }
    }
}

// by-source pass: gradient w.r.t. xh through the aggregation and through both logits; g_s[j,h] for the att_src gradient
__global__ __launch_bounds__(256) void gat_bwd_src_kernel(const float* __restrict__ gout, long ldg, const float* __restrict__ a_s,
                                                          const float* __restrict__ a_d, const float* __restrict__ m_in,
                                                          const float* __restrict__ z_in, const int* __restrict__ rowptr_t,
                                                          const int* __restrict__ col_t, const int* __restrict__ perm_t,
                                                          const float* __restrict__ gpre, const float* __restrict__ gpre_self,
                                                          const float* __restrict__ g_d, const float* __restrict__ att_src,
                                                          const float* __restrict__ att_dst, long N, int H, int C,
                                                          float* __restrict__ gx, long ldgx, float* __restrict__ g_s) {
    const long gid = (blockIdx.x * 256L + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    if (gid >= N * H) return;
    const long j = gid / H; const int h = gid % H;
    const float as = a_s[gid];
    float acc[kGatMaxK];
    float gs = gpre_self[gid];
    {
        const float alpha = __expf(lrelu(as + a_d[gid]) - m_in[gid]) / z_in[gid];
#pragma unroll
        for (int k = 0; k < kGatMaxK; ++k) { const int c = l + 16 * k; acc[k] = c < C ? alpha * gout[j * ldg + (long)h * C + c] : 0.0f; }
    }
    for (int e = rowptr_t[j]; e < rowptr_t[j + 1]; ++e) {
        const int i = col_t[e];
        if (i == (int)j) continue;
        const long gi = (long)i * H + h;
        const float alpha = __expf(lrelu(as + a_d[gi]) - m_in[gi]) / z_in[gi];
        gs += gpre[(long)perm_t[e] * H + h];
        const float* go = gout + (long)i * ldg + (long)h * C;
#pragma unroll
        for (int k = 0; k < kGatMaxK; ++k) { const int c = l + 16 * k; if (c < C) acc[k] = fmaf(alpha, go[c], acc[k]); }
    }
    const float gd = g_d[gid];
#pragma unroll
    for (int k = 0; k < kGatMaxK; ++k) {
        const int c = l + 16 * k;
        if (c < C) gx[j * ldgx + (long)h * C + c] = fmaf(gs, att_src[h * C + c], fmaf(gd, att_dst[h * C + c], acc[k]));
    }
    if (l == 0) g_s[gid] = gs;
}


// ====================================================================== packed-heads fast path
// C in {4, 8, 16, 32, 64}, H*C <= 128, 16-byte aligned rows: ONE 16-lane group walks a row's neighbour list for ALL
// heads at once (the per-head kernels above issue H times as many small gathers).  Lane l owns the float4 column
// groups c4 = 4l + 64k (k = 0, 1); head(l, k) = c4 / C, a head's channels sit in C/4 consecutive lanes, so the
// per-head reductions of the backward are butterflies over those lanes.  Softmax state per (lane, k), replicated
// inside a head.  Hub rows are left to the per-head hub kernels (same outputs).
constexpr int kPk = 2;
__device__ __forceinline__ float4 f4ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void f4fma(float4& a, float w, const float4& v) {
    a.x = fmaf(w, v.x, a.x); a.y = fmaf(w, v.y, a.y); a.z = fmaf(w, v.z, a.z); a.w = fmaf(w, v.w, a.w);
}
__device__ __forceinline__ float f4dot(const float4& a, const float4& b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }
// sum over the lanes of one head (L = C/4 consecutive lanes, a power of two <= 16)
// DPP swaps inside the 16-lane row (no LDS round trips: this sits on the per-edge dependency chain of the backward)
__device__ __forceinline__ float dpp_add(float v, int ctrl_sel) {
    const unsigned u = __float_as_uint(v);
    unsigned t;
    switch (ctrl_sel) {
        case 0: t = __builtin_amdgcn_update_dpp(0, u, 0xB1, 0xf, 0xf, true); break;    // quad_perm [1,0,3,2]
        case 1: t = __builtin_amdgcn_update_dpp(0, u, 0x4E, 0xf, 0xf, true); break;    // quad_perm [2,3,0,1]
        case 2: t = __builtin_amdgcn_update_dpp(0, u, 0x141, 0xf, 0xf, true); break;   // row_half_mirror
        default: t = __builtin_amdgcn_update_dpp(0, u, 0x140, 0xf, 0xf, true); break;  // row_mirror
    }
    return v + __uint_as_float(t);
}
__device__ __forceinline__ float head_sum(float v, int L) {
    if (L >= 2) v = dpp_add(v, 0);
    if (L >= 4) v = dpp_add(v, 1);
    if (L >= 8) v = dpp_add(v, 2);
    if (L >= 16) v = dpp_add(v, 3);
    return v;
}

__global__ __launch_bounds__(256) void gat_fwd_rows_kernel(const float* __restrict__ xh, long ld, const float* __restrict__ a_s,
                                                           const float* __restrict__ a_d, const int* __restrict__ rowptr,
                                                           const int* __restrict__ col, long N, int H, int C,
                                                           const float* __restrict__ bias, float* __restrict__ out, long ldo,
                                                           float* __restrict__ m_out, float* __restrict__ z_out, int hub_threshold) {
    const long i = (blockIdx.x * 256L + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15, HC = H * C;
    if (i >= N) return;
    if (rowptr[i + 1] - rowptr[i] > hub_threshold) return;
    int hd[kPk]; float ad[kPk], m[kPk], z[kPk]; float4 acc[kPk]; bool on[kPk];
#pragma unroll
    for (int k = 0; k < kPk; ++k) {
        const int c4 = 4 * l + 64 * k;
        on[k] = c4 < HC;
        hd[k] = on[k] ? c4 / C : 0;
        ad[k] = a_d[i * H + hd[k]];
        m[k] = lrelu(a_s[i * H + hd[k]] + ad[k]); z[k] = 1.0f;              // the added self loop
        acc[k] = on[k] ? f4ld(xh + i * ld + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int e_end = rowptr[i + 1];
    for (int e0 = rowptr[i]; e0 < e_end; e0 += 4) {        // four edges per step: independent gathers issued together
        int j[4]; float4 xj[4][kPk]; float asj[4][kPk];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            j[q] = col[min(e0 + q, e_end - 1)];
#pragma unroll
            for (int k = 0; k < kPk; ++k) {
                xj[q][k] = f4ld(xh + (long)j[q] * ld + (on[k] ? 4 * l + 64 * k : 0));
                asj[q][k] = a_s[(long)j[q] * H + hd[k]];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (e0 + q >= e_end) break;                     // group-uniform
            if (j[q] == (int)i) continue;                   // existing self loops are removed
#pragma unroll
            for (int k = 0; k < kPk; ++k) {
                if (!on[k]) continue;
                const float v = lrelu(asj[q][k] + ad[k]);
                const float mn = fmaxf(m[k], v);
                const float sc = __expf(m[k] - mn), p = __expf(v - mn);
                z[k] = fmaf(z[k], sc, p);
                acc[k].x = fmaf(acc[k].x, sc, p * xj[q][k].x); acc[k].y = fmaf(acc[k].y, sc, p * xj[q][k].y);
                acc[k].z = fmaf(acc[k].z, sc, p * xj[q][k].z); acc[k].w = fmaf(acc[k].w, sc, p * xj[q][k].w);
                m[k] = mn;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kPk; ++k) {
        if (!on[k]) continue;
        const int c4 = 4 * l + 64 * k;
        const float inv = 1.0f / z[k];
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) b = f4ld(bias + c4);
        *reinterpret_cast<float4*>(out + i * ldo + c4) = make_float4(fmaf(acc[k].x, inv, b.x), fmaf(acc[k].y, inv, b.y),
                                                                      fmaf(acc[k].z, inv, b.z), fmaf(acc[k].w, inv, b.w));
        if (c4 % C == 0) { m_out[i * H + hd[k]] = m[k]; z_out[i * H + hd[k]] = z[k]; }
    }
}

__global__ __launch_bounds__(256) void gat_bwd_dst_rows_kernel(const float* __restrict__ xh, long ld, const float* __restrict__ gout,
                                                               long ldg, const float* __restrict__ y, long ldy,
                                                               const float* __restrict__ bias, const float* __restrict__ a_s,
                                                               const float* __restrict__ a_d, const float* __restrict__ m_in,
                                                               const float* __restrict__ z_in, const int* __restrict__ rowptr,
                                                               const int* __restrict__ col, const int* __restrict__ perm, long N,
                                                               int H, int C, float* __restrict__ gpre, float* __restrict__ gpre_self,
                                                               float* __restrict__ g_d, int hub_threshold) {
    const long i = (blockIdx.x * 256L + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15, HC = H * C, L = C / 4;
    if (i >= N) return;                                   // (whole 16-lane groups leave together)
    if (rowptr[i + 1] - rowptr[i] > hub_threshold) return;
    int hd[kPk]; float ad[kPk], m[kPk], inv[kPk], S[kPk], gd[kPk]; float4 g[kPk]; bool on[kPk], first[kPk];
#pragma unroll
    for (int k = 0; k < kPk; ++k) {
        const int c4 = 4 * l + 64 * k;
        on[k] = c4 < HC;
        const int cc = on[k] ? c4 : 0;
        hd[k] = cc / C;
        first[k] = on[k] && (cc % C == 0);
        const long gi = i * H + hd[k];
        ad[k] = a_d[gi]; m[k] = m_in[gi]; inv[k] = 1.0f / z_in[gi];
        g[k] = on[k] ? f4ld(gout + i * ldg + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 yo = f4ld(y + i * ldy + cc);
        if (bias) { const float4 b = f4ld(bias + cc); yo.x -= b.x; yo.y -= b.y; yo.z -= b.z; yo.w -= b.w; }
        S[k] = head_sum(on[k] ? f4dot(g[k], yo) : 0.0f, L);
        const float gself = head_sum(on[k] ? f4dot(g[k], f4ld(xh + i * ld + cc)) : 0.0f, L);
        const float pre = a_s[gi] + ad[k];
        const float alpha = __expf(lrelu(pre) - m[k]) * inv[k];
        gd[k] = alpha * (gself - S[k]) * (pre > 0.0f ? 1.0f : kSlope);
        if (first[k]) gpre_self[gi] = gd[k];
    }
    // four edges per step: their gathers are independent, so issue them all before the (DPP) reductions
    const int e_end = rowptr[i + 1];
    for (int e0 = rowptr[i]; e0 < e_end; e0 += 4) {
        int j[4]; long eo[4]; float4 xj[4][kPk]; float asj[4][kPk];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = min(e0 + q, e_end - 1);
            j[q] = col[e];
            eo[q] = (long)perm[e] * H;
#pragma unroll
            for (int k = 0; k < kPk; ++k) {
                const int cc = on[k] ? 4 * l + 64 * k : 0;
                xj[q][k] = f4ld(xh + (long)j[q] * ld + cc);
                asj[q][k] = a_s[(long)j[q] * H + hd[k]];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (e0 + q >= e_end) break;                     // group-uniform
#pragma unroll
            for (int k = 0; k < kPk; ++k) {
                if (j[q] == (int)i) { if (first[k]) gpre[eo[q] + hd[k]] = 0.0f; continue; }
                const float ga = head_sum(on[k] ? f4dot(g[k], xj[q][k]) : 0.0f, L);
                const float pre = asj[q][k] + ad[k];
                const float alpha = __expf(lrelu(pre) - m[k]) * inv[k];
                const float gp = alpha * (ga - S[k]) * (pre > 0.0f ? 1.0f : kSlope);
                gd[k] += gp;
                if (first[k]) gpre[eo[q] + hd[k]] = gp;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kPk; ++k)
        if (first[k]) g_d[i * H + hd[k]] = gd[k];
}

__global__ __launch_bounds__(256) void gat_bwd_src_rows_kernel(const float* __restrict__ gout, long ldg, const float* __restrict__ a_s,
                                                               const float* __restrict__ a_d, const float* __restrict__ m_in,
                                                               const float* __restrict__ z_in, const int* __restrict__ rowptr_t,
                                                               const int* __restrict__ col_t, const int* __restrict__ perm_t,
                                                               const float* __restrict__ gpre, const float* __restrict__ gpre_self,
                                                               const float* __restrict__ g_d, const float* __restrict__ att_src,
                                                               const float* __restrict__ att_dst, long N, int H, int C,
                                                               float* __restrict__ gx, long ldgx, float* __restrict__ g_s) {
    const long j = (blockIdx.x * 256L + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15, HC = H * C;
    if (j >= N) return;
    int hd[kPk]; float as[kPk], gs[kPk]; float4 acc[kPk]; bool on[kPk];
#pragma unroll
    for (int k = 0; k < kPk; ++k) {
        const int c4 = 4 * l + 64 * k;
        on[k] = c4 < HC;
        const int cc = on[k] ? c4 : 0;
        hd[k] = cc / C;
        const long gj = j * H + hd[k];
        as[k] = a_s[gj];
        gs[k] = gpre_self[gj];
        const float alpha = __expf(lrelu(as[k] + a_d[gj]) - m_in[gj]) / z_in[gj];
        const float4 go = f4ld(gout + j * ldg + cc);
        acc[k] = make_float4(alpha * go.x, alpha * go.y, alpha * go.z, alpha * go.w);
    }
    for (int e = rowptr_t[j]; e < rowptr_t[j + 1]; ++e) {
        const int i = col_t[e];
        if (i == (int)j) continue;
        const long eo = (long)perm_t[e] * H;
#pragma unroll
        for (int k = 0; k < kPk; ++k) {
            if (!on[k]) continue;
            const long gi = (long)i * H + hd[k];
            const float alpha = __expf(lrelu(as[k] + a_d[gi]) - m_in[gi]) / z_in[gi];
            gs[k] += gpre[eo + hd[k]];
            f4fma(acc[k], alpha, f4ld(gout + (long)i * ldg + 4 * l + 64 * k));
        }
    }
#pragma unroll
    for (int k = 0; k < kPk; ++k) {
        if (!on[k]) continue;
        const int c4 = 4 * l + 64 * k;
        const float gd = g_d[j * H + hd[k]];
        const float4 s4 = f4ld(att_src + c4), d4 = f4ld(att_dst + c4);       // att_* are [H, C] = flat H*C
        *reinterpret_cast<float4*>(gx + j * ldgx + c4) =
            make_float4(fmaf(gs[k], s4.x, fmaf(gd, d4.x, acc[k].x)), fmaf(gs[k], s4.y, fmaf(gd, d4.y, acc[k].y)),
                        fmaf(gs[k], s4.z, fmaf(gd, d4.z, acc[k].z)), fmaf(gs[k], s4.w, fmaf(gd, d4.w, acc[k].w)));
        if (c4 % C == 0) g_s[j * H + hd[k]] = gs[k];
    }
}

static bool gat_packed_ok(int H, int C, long ld0, long ld1, long ld2, long ld3, const void* p0, const void* p1, const void* p2,
                          const void* p3) {
    const bool pow2 = C == 4 || C == 8 || C == 16 || C == 32 || C == 64;
    auto al = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return pow2 && H * C <= 64 * kPk && ((ld0 | ld1 | ld2 | ld3) & 3) == 0 && al(p0) && al(p1) && al(p2) && al(p3);
}

// ------------------------------------------------------------------ host side
int gat_logits(const float* xh, long ld, long N, int H, int C, const float* att_src, const float* att_dst, float* a_s,
               float* a_d, hipStream_t st) {
    if (N == 0) return KAGNN_OK;
    gat_logits_kernel<<<cdiv(N * H * 16, 256), 256, 0, st>>>(xh, ld, N, H, C, att_src, att_dst, a_s, a_d);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

int gat_fwd(const float* xh, long ld, const float* a_s, const float* a_d, const int* rowptr, const int* col, long N,
            int H, int C, const float* bias, float* out, long ldo, float* m, float* z, const int* hub_seg,
            long num_hub_seg, int hub_threshold, hipStream_t st) {
    if (N == 0) return KAGNN_OK;
    if (C > 16 * kGatMaxK) return fail(KAGNN_ERR_UNSUPPORTED, "%s: more than 128 channels per head", "gat_fwd");
    const int thr = (hub_seg && num_hub_seg > 0) ? hub_threshold : 0x7fffffff;
    if (gat_packed_ok(H, C, ld, ldo, 0, 0, xh, out, bias, nullptr))
        gat_fwd_rows_kernel<<<cdiv(N * 16, 256), 256, 0, st>>>(xh, ld, a_s, a_d, rowptr, col, N, H, C, bias, out, ldo, m, z, thr);
    else
        gat_fwd_kernel<<<cdiv(N * H * 64, 256), 256, 0, st>>>(xh, ld, a_s, a_d, rowptr, col, N, H, C, bias, out, ldo, m, z, thr);
    KAGNN_LAUNCH_CHECK();
    if (thr != 0x7fffffff) {
        gat_fwd_hub_kernel<<<(unsigned)(num_hub_seg * H), 256, 0, st>>>(xh, ld, a_s, a_d, rowptr, col, hub_seg, H, C, bias, out,
                                                                       ldo, m, z);
        KAGNN_LAUNCH_CHECK();
    }
    return KAGNN_OK;
}

int gat_bwd(const float* xh, long ld, const float* gout, long ldg, const float* y, long ldy, const float* bias,
            const float* a_s, const float* a_d, const float* m, const float* z, const int* rowptr, const int* col,
            const int* perm, const int* rowptr_t, const int* col_t, const int* perm_t, const float* att_src,
            const float* att_dst, long N, int H, int C, float* gpre, float* gpre_self, float* g_d, float* g_s,
            float* gx, long ldgx, const int* hub_seg, long num_hub_seg, int hub_threshold, hipStream_t st) {
    if (N == 0) return KAGNN_OK;
    if (C > 16 * kGatMaxK) return fail(KAGNN_ERR_UNSUPPORTED, "%s: more than 128 channels per head", "gat_bwd");
    const int grid = cdiv(N * H * 16, 256);
    const int thr = (hub_seg && num_hub_seg > 0) ? hub_threshold : 0x7fffffff;
    const bool packed = gat_packed_ok(H, C, ld, ldg, ldy, ldgx, xh, gout, y, gx) &&
                        (bias == nullptr || (reinterpret_cast<uintptr_t>(bias) & 15) == 0) &&
                        (reinterpret_cast<uintptr_t>(att_src) & 15) == 0 && (reinterpret_cast<uintptr_t>(att_dst) & 15) == 0;
    if (packed)
        gat_bwd_dst_rows_kernel<<<cdiv(N * 16, 256), 256, 0, st>>>(xh, ld, gout, ldg, y, ldy, bias, a_s, a_d, m, z, rowptr, col, perm,
                                                                   N, H, C, gpre, gpre_self, g_d, thr);
    else
        gat_bwd_dst_kernel<<<cdiv(N * H * 64, 256), 256, 0, st>>>(xh, ld, gout, ldg, y, ldy, bias, a_s, a_d, m, z, rowptr, col, perm, N,
                                                                  H, C, gpre, gpre_self, g_d, thr, nullptr);
    KAGNN_LAUNCH_CHECK();
    if (thr != 0x7fffffff) {
        gat_bwd_dst_kernel<<<(unsigned)(num_hub_seg * H), 256, 0, st>>>(xh, ld, gout, ldg, y, ldy, bias, a_s, a_d, m, z, rowptr, col,
                                                                       perm, N, H, C, gpre, gpre_self, g_d, thr, hub_seg);
        KAGNN_LAUNCH_CHECK();
    }
    if (packed)
        gat_bwd_src_rows_kernel<<<cdiv(N * 16, 256), 256, 0, st>>>(gout, ldg, a_s, a_d, m, z, rowptr_t, col_t, perm_t, gpre, gpre_self,
                                                                   g_d, att_src, att_dst, N, H, C, gx, ldgx, g_s);
    else
        gat_bwd_src_kernel<<<grid, 256, 0, st>>>(gout, ldg, a_s, a_d, m, z, rowptr_t, col_t, perm_t, gpre, gpre_self, g_d,
                                                  att_src, att_dst, N, H, C, gx, ldgx, g_s);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// ------------------------------------------------------------------ gradients of the attention vectors
// g_att_src[h][c] = sum_n g_src[n][h] * xh[n][h*C + c]  (and the same with g_dst): two skinny [N,H]^T x [N,H,C]
// contractions -- as torch einsums they ran as two 130 us GEMMs on a 170k-node graph; one pass over xh here,
// per-workgroup partial sums combined in a fixed order (deterministic).  Also the bias gradient (column sums of
// gout) when asked for.
template <int KC>             // columns per thread: H*C <= 64 * KC
__global__ __launch_bounds__(256) void gat_att_grad_partial_kernel(const float* __restrict__ xh, long ld,
                                                                   const float* __restrict__ g_src,
                                                                   const float* __restrict__ g_dst, long N, int H, int C,
                                                                   long rows_per_block, float* __restrict__ partial) {
    __shared__ float s_red[4][2][64];
    const int HC = H * C, cl = threadIdx.x & 63, rq = threadIdx.x >> 6;
    const long r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
    float as[KC], ad[KC];
    int hk[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) { as[k] = 0.0f; ad[k] = 0.0f; hk[k] = min(cl + 64 * k, HC - 1) / C; }
    for (long n0 = r0 + rq; n0 < r1; n0 += 16) {          // four rows (stride 4) per trip: their loads go out together
        float v[4][KC], ws[4][KC], wd[4][KC];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long n = min(n0 + 4 * q, N - 1);
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                v[q][k] = xh[n * ld + min(cl + 64 * k, HC - 1)];
                ws[q][k] = g_src[n * H + hk[k]]; wd[q][k] = g_dst[n * H + hk[k]];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (n0 + 4 * q < r1) {
#pragma unroll
                for (int k = 0; k < KC; ++k) { as[k] = fmaf(ws[q][k], v[q][k], as[k]); ad[k] = fmaf(wd[q][k], v[q][k], ad[k]); }
            }
        }
    }
    float* out = partial + (long)blockIdx.x * 2 * HC;
#pragma unroll
    for (int k = 0; k < KC; ++k) {                         // combine the four row lanes in a fixed order, 64 columns at a time
        s_red[rq][0][cl] = as[k]; s_red[rq][1][cl] = ad[k];
        __syncthreads();
        const int c = cl + 64 * k;
        if (rq < 2 && c < HC) out[rq * HC + c] = (s_red[0][rq][cl] + s_red[1][rq][cl]) + (s_red[2][rq][cl] + s_red[3][rq][cl]);
        __syncthreads();
    }
}

// g_att[which][c] = sum over workgroups of partial[b][which][c]: 64 columns x 4 interleaved partial chains per
// workgroup, chains combined in a fixed order
__global__ __launch_bounds__(256) void gat_att_grad_finish_kernel(const float* __restrict__ partial, int nb, int HC,
                                                                  float* __restrict__ g_att_src, float* __restrict__ g_att_dst) {
    __shared__ float s_red[4][64];
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6, i = blockIdx.x * 64 + cl;
    float a = 0.0f;
    if (i < 2 * HC)
        for (int b = q; b < nb; b += 4) a += partial[(long)b * 2 * HC + i];
    s_red[q][cl] = a;
    __syncthreads();
    if (q == 0 && i < 2 * HC)
        (i < HC ? g_att_src : g_att_dst)[i < HC ? i : i - HC] = (s_red[0][cl] + s_red[1][cl]) + (s_red[2][cl] + s_red[3][cl]);
}

static void att_plan(long N, int* nb, long* rpb) {
    int b = (int)max(1L, min((long)cdiv(N, 256), 256L));
    long r = (N + b - 1) / b;
    *nb = (int)max(1L, (long)cdiv(N, max(r, 1L))); *rpb = max(r, 1L);
}

size_t gat_att_grad_ws_bytes(long N, int H, int C) {
    int nb; long rpb;
    att_plan(N, &nb, &rpb);
    return (size_t)nb * 2 * H * C * sizeof(float);
}

int gat_att_grad(const float* xh, long ld, const float* g_src, const float* g_dst, long N, int H, int C, float* g_att_src,
                 float* g_att_dst, void* ws, size_t ws_bytes, hipStream_t st) {
    if (H * C > 1024) return fail(KAGNN_ERR_UNSUPPORTED, "%s: heads * channels > 1024", "gat_att_grad");
    if (ws_bytes < gat_att_grad_ws_bytes(N, H, C)) return fail(KAGNN_ERR_ARG, "%s: workspace too small", "gat_att_grad");
    int nb; long rpb;
    att_plan(N, &nb, &rpb);
    if (N == 0) nb = 0;
    if (nb) {
        const int kc = cdiv(H * C, 64);
#define L(KK) gat_att_grad_partial_kernel<KK><<<nb, 256, 0, st>>>(xh, ld, g_src, g_dst, N, H, C, rpb, (float*)ws)
        if (kc <= 1) L(1); else if (kc <= 2) L(2); else if (kc <= 4) L(4); else if (kc <= 8) L(8); else L(16);
#undef L
        KAGNN_LAUNCH_CHECK();
    }
    gat_att_grad_finish_kernel<<<cdiv(2 * H * C, 64), 256, 0, st>>>((const float*)ws, nb, H * C, g_att_src, g_att_dst);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

}  // namespace kagnn
