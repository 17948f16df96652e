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
    for (int e = rowptr[i] + grp; e < rowptr[i + 1]; e += 16) {
        const int j = col[e];
        if (j == (int)i) continue;
        const float v = lrelu(a_s[(long)j * H + h] + ad);
        if (v > m) {
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
    for (int e = rowptr[i] + first; e < rowptr[i + 1]; e += stride) {
        const int j = col[e];
        if (j == (int)i) { if (l == 0) gpre[(long)perm[e] * H + h] = 0.0f; continue; }
        const float* xj = xh + (long)j * ld + (long)h * C;
        float ga = 0.0f;
#pragma unroll
        for (int k = 0; k < kGatMaxK; ++k) { const int c = l + 16 * k; if (c < C) ga = fmaf(g[k], xj[c], ga); }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) ga += __shfl_xor(ga, o);
        const float pre = a_s[(long)j * H + h] + ad;
        const float alpha = __expf(lrelu(pre) - m) * inv;
        const float gp = alpha * (ga - S) * (pre > 0.0f ? 1.0f : kSlope);
        gd += gp;
        if (l == 0) gpre[(long)perm[e] * H + h] = gp;
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
    gat_bwd_dst_kernel<<<cdiv(N * H * 64, 256), 256, 0, st>>>(xh, ld, gout, ldg, y, ldy, bias, a_s, a_d, m, z, rowptr, col, perm, N, H, C,
                                              gpre, gpre_self, g_d, thr, nullptr);
    KAGNN_LAUNCH_CHECK();
    if (thr != 0x7fffffff) {
        gat_bwd_dst_kernel<<<(unsigned)(num_hub_seg * H), 256, 0, st>>>(xh, ld, gout, ldg, y, ldy, bias, a_s, a_d, m, z, rowptr, col,
                                                                       perm, N, H, C, gpre, gpre_self, g_d, thr, hub_seg);
        KAGNN_LAUNCH_CHECK();
    }
    gat_bwd_src_kernel<<<grid, 256, 0, st>>>(gout, ldg, a_s, a_d, m, z, rowptr_t, col_t, perm_t, gpre, gpre_self, g_d,
                                              att_src, att_dst, N, H, C, gx, ldgx, g_s);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

}  // namespace kagnn
