// Neighbour aggregation with bf16 GATHER OPERANDS (KAGNN_ACT=bf16, BASELINE.json config 2 "... bf16 on 1 MI355X").
//
// The reference has no reduced-precision path ("bf16 storage variant is a build-defined mode").  What the
// mode buys is the term of the layer's traffic that scales with the number of EDGES: every edge gathers one
// source row, 4F bytes in fp32, 2F in bf16 -- at the ogbn-arxiv / headline shapes that is ~85 % of an aggregation
// launch, and the gathered matrix (N x F x 2 bytes) then fits the Infinity Cache twice over.  Accumulation is fp32; the
// result is written as fp32 (forward: it feeds the KAN layer, which stays fp32 / split precision) or bf16 (the input
// gradient of a bf16 activation).  Same structure as aggregate.hip: a destination row is owned by LPR lanes holding
// 8 features each (one 16-byte load per gathered row and lane), four gathers in flight, no atomics; rows above the hub
// threshold are summed per segment into a workspace and finished by one lane group in segment order (deterministic).
//
// Reference behaviour replaced: MessagePassing.propagate of torch_geometric as called from
// node_classification_clean/models.py:31-37,48-56, at reduced storage precision.
#include "common.h"

namespace kagnn {

struct f8 { float v[8]; };

__device__ __forceinline__ f8 ld8_bf16(const unsigned short* p) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    f8 r;
    r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xffff0000u);
    r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xffff0000u);
    r.v[4] = __uint_as_float(u.z << 16); r.v[5] = __uint_as_float(u.z & 0xffff0000u);
    r.v[6] = __uint_as_float(u.w << 16); r.v[7] = __uint_as_float(u.w & 0xffff0000u);
    return r;
}
__device__ __forceinline__ void fma8(f8& a, float w, const f8& b) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a.v[i] = fmaf(w, b.v[i], a.v[i]);
}
// round-to-nearest-even fp32 -> bf16 (NaN stays NaN), two values per dword
__device__ __forceinline__ unsigned bf16_rne(float f) {
    const unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) { return bf16_rne(lo) | (bf16_rne(hi) << 16); }

template <bool OUT16>
__device__ __forceinline__ void store8(void* out, long ldo, long row, int c8, const f8& o) {
    if constexpr (OUT16) {
        uint4 u;
        u.x = pack_bf16(o.v[0], o.v[1]); u.y = pack_bf16(o.v[2], o.v[3]);
        u.z = pack_bf16(o.v[4], o.v[5]); u.w = pack_bf16(o.v[6], o.v[7]);
        *reinterpret_cast<uint4*>(static_cast<unsigned short*>(out) + row * ldo + c8) = u;
    } else {
        float* p = static_cast<float*>(out) + row * ldo + c8;
        *reinterpret_cast<float4*>(p) = make_float4(o.v[0], o.v[1], o.v[2], o.v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(o.v[4], o.v[5], o.v[6], o.v[7]);
    }
}

struct Agg16Args {
    const unsigned short* x; long ldx;      // bf16 rows
    void* out; long ldo;                    // fp32 or bf16 rows
    const int* rowptr; const int* col; const float* ew;
    long N; int F;
    float self_scale;
    const float* in_scale; const float* out_scale; const float* bias;
    int skip_self; int hub_threshold;
};

__device__ __forceinline__ float edge_w16(const Agg16Args& a, int e, int j, long i) {
    float w = a.ew ? a.ew[e] : 1.0f;
    if (a.in_scale) w *= a.in_scale[j];
    if (a.skip_self && j == (int)i) w = 0.0f;
    return w;
}

template <bool OUT16>
__device__ __forceinline__ void finish_row(const Agg16Args& a, long row, int c8, const f8& acc) {
    const float os = a.out_scale ? a.out_scale[row] : 1.0f;
    f8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o.v[i] = fmaf(os, acc.v[i], a.bias ? a.bias[c8 + i] : 0.0f);
    store8<OUT16>(a.out, a.ldo, row, c8, o);
}

template <int LPR, bool OUT16>
__global__ __launch_bounds__(256) void agg16_rows_kernel(Agg16Args a) {
    // one contiguous eighth of the row groups per XCD, as agg_rows_v4_kernel (aggregate.hip); the grid is whole eighths
    const unsigned wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const long gid = (wg * 256L + threadIdx.x) / LPR;
    const int c8 = (threadIdx.x % LPR) * 8;
    if (gid >= a.N || c8 >= a.F) return;
    const int s = a.rowptr[gid], t = a.rowptr[gid + 1];
    if ((t - s) > a.hub_threshold) return;                      // hub rows: agg16_hub_* write them completely
    f8 acc = ld8_bf16(a.x + gid * a.ldx + c8);
    const float sw = a.self_scale * (a.in_scale ? a.in_scale[gid] : 1.0f);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc.v[i] *= sw;
    int e = s;
    for (; e + 4 <= t; e += 4) {
        const int j0 = a.col[e], j1 = a.col[e + 1], j2 = a.col[e + 2], j3 = a.col[e + 3];
        const f8 v0 = ld8_bf16(a.x + (long)j0 * a.ldx + c8);
        const f8 v1 = ld8_bf16(a.x + (long)j1 * a.ldx + c8);
        const f8 v2 = ld8_bf16(a.x + (long)j2 * a.ldx + c8);
        const f8 v3 = ld8_bf16(a.x + (long)j3 * a.ldx + c8);
        fma8(acc, edge_w16(a, e, j0, gid), v0);
        fma8(acc, edge_w16(a, e + 1, j1, gid), v1);
        fma8(acc, edge_w16(a, e + 2, j2, gid), v2);
        fma8(acc, edge_w16(a, e + 3, j3, gid), v3);
    }
    for (; e < t; ++e) {
        const int j = a.col[e];
        fma8(acc, edge_w16(a, e, j, gid), ld8_bf16(a.x + (long)j * a.ldx + c8));
    }
    finish_row<OUT16>(a, gid, c8, acc);
}

// one workgroup per hub segment {row, e0, e1}: fp32 partial sums (fixed order) to part[segment][F]
template <int LPR>
__global__ __launch_bounds__(256) void agg16_hub_kernel(Agg16Args a, const int* __restrict__ seg, float* __restrict__ part, int ldp) {
    __shared__ f8 s_part[256];
    const int row = seg[3 * blockIdx.x], e0 = seg[3 * blockIdx.x + 1], e1 = seg[3 * blockIdx.x + 2];
    constexpr int G = 256 / LPR;
    const int g = threadIdx.x / LPR, lg = threadIdx.x % LPR, c8 = lg * 8;
    f8 acc;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc.v[i] = 0.0f;
    if (c8 < a.F) {
        for (int e = e0 + g; e < e1; e += G) {
            const int j = a.col[e];
            fma8(acc, edge_w16(a, e, j, row), ld8_bf16(a.x + (long)j * a.ldx + c8));
        }
    }
    s_part[threadIdx.x] = acc;
    __syncthreads();
    if (g == 0 && c8 < a.F) {
#pragma unroll 8                               // (fully unrolled at G = 64 / 128 the loads of all partials are hoisted: 510 registers, scratch)
        for (int k = 1; k < G; ++k) {
            const f8 p = s_part[k * LPR + lg];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc.v[i] += p.v[i];
        }
        float* o = part + (long)blockIdx.x * ldp + c8;
        *reinterpret_cast<float4*>(o) = make_float4(acc.v[0], acc.v[1], acc.v[2], acc.v[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(acc.v[4], acc.v[5], acc.v[6], acc.v[7]);
    }
}

// the workgroup of a row's FIRST segment: its lane groups take the row's segments round-robin, the partial sums meet in
// LDS in a fixed order; self term + sum, ONE rounding into out
template <int LPR, bool OUT16>
__global__ __launch_bounds__(256) void agg16_hub_merge_kernel(Agg16Args a, const int* __restrict__ seg, long nseg,
                                                              const float* __restrict__ part, int ldp) {
    __shared__ f8 s_part[256];
    const long sidx = blockIdx.x;
    const int row = seg[3 * sidx];
    if (sidx > 0 && seg[3 * (sidx - 1)] == row) return;          // workgroup-uniform
    constexpr int G = 256 / LPR;
    const int g = threadIdx.x / LPR, lg = threadIdx.x % LPR, c8 = lg * 8;
    f8 acc;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc.v[i] = 0.0f;
    if (c8 < a.F) {
        for (long k = sidx + g; k < nseg && seg[3 * k] == row; k += G) {
            const float* p = part + k * ldp + c8;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc.v[i] += p[i];
        }
    }
    s_part[threadIdx.x] = acc;
    __syncthreads();
    if (g == 0 && c8 < a.F) {
#pragma unroll 8                               // (fully unrolled at G = 64 / 128 the loads of all partials are hoisted: 510 registers, scratch)
        for (int k = 1; k < G; ++k) {
            const f8 p = s_part[k * LPR + lg];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc.v[i] += p.v[i];
        }
        f8 self = ld8_bf16(a.x + (long)row * a.ldx + c8);
        const float sw = a.self_scale * (a.in_scale ? a.in_scale[row] : 1.0f);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc.v[i] = fmaf(sw, self.v[i], acc.v[i]);
        finish_row<OUT16>(a, row, c8, acc);
    }
}

size_t aggregate_bf16_ws_bytes(long num_hub_seg, int F) { return (size_t)num_hub_seg * (size_t)((F + 7) & ~7) * sizeof(float); }

bool aggregate_bf16_ok(const void* x, long ldx, const void* out, long ldo, int out_bf16, int F, const float* bias) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return F % 8 == 0 && F <= 512 && ldx % 8 == 0 && ldo % (out_bf16 ? 8 : 4) == 0 && al(x) && al(out) && (!bias || al(bias));
}

template <bool OUT16>
static int run16(const Agg16Args& a, const int* hub_seg, long num_hub_seg, float* ws, hipStream_t st) {
    Agg16Args b = a;
    const bool hubs = num_hub_seg > 0 && hub_seg != nullptr;
    if (!hubs) b.hub_threshold = 0x7fffffff;
    const int ldp = (a.F + 7) & ~7;
#define GO(LPR)                                                                                                   \
    {                                                                                                             \
        agg16_rows_kernel<LPR, OUT16><<<cdiv(cdiv(a.N * LPR, 256), 8) * 8, 256, 0, st>>>(b);                       \
        KAGNN_LAUNCH_CHECK();                                                                                     \
        if (hubs) {                                                                                               \
            agg16_hub_kernel<LPR><<<(unsigned)num_hub_seg, 256, 0, st>>>(b, hub_seg, ws, ldp);                     \
            KAGNN_LAUNCH_CHECK();                                                                                 \
            agg16_hub_merge_kernel<LPR, OUT16><<<(unsigned)num_hub_seg, 256, 0, st>>>(b, hub_seg, num_hub_seg, ws, ldp); \
            KAGNN_LAUNCH_CHECK();                                                                                 \
        }                                                                                                         \
    }
    if (a.F <= 8) GO(1) else if (a.F <= 16) GO(2) else if (a.F <= 32) GO(4) else if (a.F <= 64) GO(8)
    else if (a.F <= 128) GO(16) else if (a.F <= 256) GO(32) else GO(64)
#undef GO
    return KAGNN_OK;
}

int aggregate_sum_bf16(const void* x, long ldx, void* out, long ldo, int out_bf16, const int* rowptr, const int* col,
                       const float* ew, long N, int F, float self_scale, const float* in_scale, const float* out_scale,
                       const float* bias, int skip_self, const int* hub_seg, long num_hub_seg, int hub_threshold,
                       float* ws, size_t ws_bytes, hipStream_t st) {
    if (N == 0) return KAGNN_OK;
    if (num_hub_seg > 0 && hub_seg && (ws == nullptr || ws_bytes < aggregate_bf16_ws_bytes(num_hub_seg, F)))
        return fail(KAGNN_ERR_ARG, "%s: workspace too small for the hub segments", "aggregate_sum_bf16");
    Agg16Args a{static_cast<const unsigned short*>(x), ldx, out, ldo, rowptr, col, ew, N, F, self_scale, in_scale, out_scale,
                bias, skip_self, hub_threshold > 0 ? hub_threshold : 0x7fffffff};
    return out_bf16 ? run16<true>(a, hub_seg, num_hub_seg, ws, st) : run16<false>(a, hub_seg, num_hub_seg, ws, st);
}

// fp32 -> bf16 rows (round to nearest even), 8 values per thread; for activations that arrive in fp32
__global__ __launch_bounds__(256) void to_bf16_kernel(const float* __restrict__ x, long ldx, unsigned short* __restrict__ y,
                                                      long ldy, long N, int F8) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= N * F8) return;
    const long row = i / F8; const int c8 = (int)(i % F8) * 8;
    const float4 a = *reinterpret_cast<const float4*>(x + row * ldx + c8);
    const float4 b = *reinterpret_cast<const float4*>(x + row * ldx + c8 + 4);
    uint4 u;
    u.x = pack_bf16(a.x, a.y); u.y = pack_bf16(a.z, a.w); u.z = pack_bf16(b.x, b.y); u.w = pack_bf16(b.z, b.w);
    *reinterpret_cast<uint4*>(y + row * ldy + c8) = u;
}

int rows_to_bf16(const float* x, long ldx, void* y, long ldy, long N, int F, hipStream_t st) {
    if (N == 0) return KAGNN_OK;
    if (F % 8 || ldx % 4 || ldy % 8 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15))
        return fail(KAGNN_ERR_UNSUPPORTED, "%s: needs num_feat % 8 == 0 and 16-byte aligned rows", "rows_to_bf16");
    to_bf16_kernel<<<cdiv(N * (F / 8), 256), 256, 0, st>>>(x, ldx, static_cast<unsigned short*>(y), ldy, N, F / 8);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

}  // namespace kagnn
