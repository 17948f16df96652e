// efficient-KAN layer, exact-fp32 mode (KAGNN_PREC_FP32): v_mfma_f32_32x32x2_f32 everywhere.
// Numerics are bit-for-bit an ordered fp32 fma chain, so this mode doubles as the on-device
// reference for the split-precision fast path in kan_split.hip.
//
// Reference behaviour replaced: node_classification_clean/ekan.py:79-112 (b_splines),
// :146-152 (scaled_spline_weight), :154-162 (forward) and the autograd backward of those.
//
// Mapping (fwd): one wave owns 32 consecutive rows.  For v_mfma_f32_32x32x2_f32 lane l feeds
// A[row = l&31][k = l>>5]; we let the two k-lanes of an instruction be two different INPUT
// FEATURES (f = p + (l>>5)*P) and issue one MFMA per spline coefficient c (plus one for the
// SiLU base branch), so each lane evaluates the K+1 non-zero bases of exactly one scalar
// x[row,f] in registers and feeds them straight into the matrix core -- the [N,in,G+k] basis
// tensor of the reference never exists.
#include "common.h"
#include "split_common.h"

namespace kagnn {

// ------------------------------------------------------------------ weight packing
// pack_fwd[p][c][ot][lane] = Wcat[o = 32*ot + (lane&31)][f = p + (lane>>5)*P][c]
// pack_dx [ft][c][q][lane] = Wcat[o = q + (lane>>5)*Q][f = 32*ft + (lane&31)][c]
// Wcat[o][f][c] = spline_weight[o][f][c]*scaler[o][f] for c < C, base_weight[o][f] for c == C.
__device__ __forceinline__ float wcat(const float* bw, const float* sw, const float* sc, int in,
                                      int out, int C, int o, int f, int c) {
    if (o >= out || f >= in) return 0.0f;
    if (c == C) return bw ? bw[(long)o * in + f] : 0.0f;
    float w = sw[((long)o * in + f) * C + c];
    return sc ? w * sc[(long)o * in + f] : w;
}

__global__ void kan_pack_f32_kernel(const float* __restrict__ bw, const float* __restrict__ sw,
                                    const float* __restrict__ sc, int in, int out, int C,
                                    float* __restrict__ pf, float* __restrict__ pd) {
    const int CT = C + 1, P = (in + 1) / 2, OT = cdiv(out, 32), FT = cdiv(in, 32), Q = 16 * OT;
    const long nf = (long)P * CT * OT * 64, nd = (long)FT * CT * Q * 64;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nf + nd;
         i += (long)gridDim.x * blockDim.x) {
        if (i < nf) {
            int lane = i & 63; long r = i >> 6;
            int ot = r % OT; r /= OT;
            int c = r % CT; int p = r / CT;
            if (pf) pf[i] = wcat(bw, sw, sc, in, out, C, 32 * ot + (lane & 31), p + (lane >> 5) * P, c);
        } else {
            long j = i - nf;
            int lane = j & 63; long r = j >> 6;
            int q = r % Q; r /= Q;
            int c = r % CT; int ft = r / CT;
            if (pd) pd[j] = wcat(bw, sw, sc, in, out, C, q + (lane >> 5) * Q, 32 * ft + (lane & 31), c);
        }
    }
}

// ------------------------------------------------------------------ forward
template <int K, int OT, bool PF>
__global__ __launch_bounds__(256) void kan_fwd_f32_kernel(
    const float* __restrict__ x, long ldx, long N, int in, int C, const float* __restrict__ knots_g,
    int nknots, const float* __restrict__ pack, int ot0, int OT_total,
    float* __restrict__ y, long ldy, int out) {
    __shared__ float s_knots[kMaxKnots];
    if (threadIdx.x < nknots) s_knots[threadIdx.x] = knots_g[threadIdx.x];
    __syncthreads();
    const SplineGeom geom = geom_from_knots(s_knots, nknots);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + wave) * 32;
    if (row0 >= N) return;
    const int r = lane & 31, kh = lane >> 5;
    const long row = row0 + r;
    const bool rv = row < N;
    const int P = (in + 1) / 2, CT = C + 1;
    const float* xr = x + (rv ? row : 0) * ldx;

    f32x16 acc[OT];
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

    for (int p = 0; p < P; ++p) {
        const int f = p + kh * P;
        const bool fv = rv && f < in;
        const float xv = xr[min(f, in - 1)];          // unconditional clamped load (masked below): no per-load branch
        float Nv[K + 1], dummy[K + 1];
        int m = eval_basis<K, false, PF>(xv, s_knots, geom, knots_g, min(f, in - 1), Nv, dummy);
        float sl = siluf(xv);
        if (!fv) {
            sl = 0.0f;
#pragma unroll
            for (int i = 0; i <= K; ++i) Nv[i] = 0.0f;
        }
        const float* wp = pack + ((long)p * CT * OT_total + ot0) * 64 + lane;
        for (int c = 0; c < C; ++c) {
            const float a = pick_basis<K>(Nv, m, c);
#pragma unroll
            for (int t = 0; t < OT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wp[((long)c * OT_total + t) * 64], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < OT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(sl, wp[((long)C * OT_total + t) * 64], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < OT; ++t) {
        const int col = 32 * (ot0 + t) + r;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const long rr = row0 + mfma32_row(i, kh);
            if (rr < N && col < out) y[rr * ldy + col] = acc[t][i];
        }
    }
}

// ------------------------------------------------------------------ input gradient
// D_c[n][f] = sum_o gy[n][o] * Wcat[o][f][c]  (one 32x32 accumulator per coefficient c), then
// gx[n][f] = sum_c D_c * dB_c/dx(x[n][f]) + D_C * silu'(x[n][f]) -- the lane that owns D[.][f]
// owns all c for that (n,f), so the contraction over c is register-local.
constexpr int kDxGroup = 3;   // accumulators held at once (48 registers; 9 -- a single pass for C+1 <= 9 -- left 112 for everything else at two waves per SIMD: up to 53 spilled)

// STAGE: the W fragments of one (feature tile, coefficient group) -- [kDxGroup][Q][64 lanes] floats, 72 KB at
// out = 64 -- are copied to LDS once per workgroup and read from there by all its waves (8 = two per SIMD, so one
// wave's MFMAs cover the other's post-processing); without it every MFMA waits on its own global load of the B
// operand (that version ran at 10 % of the fp32 MFMA peak).  Falls back to the unstaged form when the tile does
// not fit beside the gy tiles (out > 96).  x is read and gx written through buffer descriptors opened at the
// workgroup's first row (32-bit offsets, the row step in an SGPR): no 64-bit address registers per element.
template <int K, bool PF, bool STAGE>
__global__ __launch_bounds__(512) void kan_dx_f32_kernel(
    const float* __restrict__ x, long ldx, const float* __restrict__ gy, long ldgy, long N, int in,
    int out, int C, const float* __restrict__ knots_g, int nknots,
    const float* __restrict__ pack, int OT_total, float* __restrict__ gx, long ldgx) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_knots = smem;                       // kMaxKnots
    const int Q = 16 * OT_total, outP = 2 * Q, ldt = outP + 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    float* s_gy = smem + kMaxKnots + (long)wave * 32 * ldt;
    float* s_w = smem + kMaxKnots + (long)nw * 32 * ldt;       // STAGE only
    if (threadIdx.x < nknots) s_knots[threadIdx.x] = knots_g[threadIdx.x];
    const long blk0 = (long)blockIdx.x * nw * 32, row0 = blk0 + wave * 32;
    // stage this wave's gy tile [32][outP] (zero padded)
    for (int i = lane; i < 32 * outP; i += 64) {
        const int rr = i / outP, o = i - rr * outP;
        const long row = row0 + rr;
        const float gv = gy[min(row, N - 1) * ldgy + min(o, out - 1)];      // unconditional clamped load
        s_gy[rr * ldt + o] = (row < N && o < out) ? gv : 0.0f;
    }
    __syncthreads();
    if (!STAGE && row0 >= N) return;             // STAGE: every wave keeps walking (barriers below); rows >= N are never stored
    const SplineGeom geom = geom_from_knots(s_knots, nknots);
    const int r = lane & 31, kh = lane >> 5;
    const int CT = C + 1, FT = cdiv(in, 32);
    const float* arow = s_gy + r * ldt + kh * Q;
    const GBuf xb = gbuf_at(x, N, ldx, in, blk0), gxb = gbuf_at(gx, N, ldgx, in, blk0);
    const unsigned ldx4 = (unsigned)ldx * 4u, ldgx4 = (unsigned)ldgx * 4u;
    const unsigned x_rb = (unsigned)(wave * 32 + 4 * kh) * ldx4, gx_rb = (unsigned)(wave * 32 + 4 * kh) * ldgx4;

    for (int ft = 0; ft < FT; ++ft) {
        const int f = 32 * ft + r;
        const unsigned fcol = (unsigned)min(f, in - 1) * 4u;
        float xq[16];                            // this lane's 16 x values of the tile: they land under the MFMAs
#pragma unroll
        for (int i = 0; i < 16; ++i) xq[i] = gld_s(xb, x_rb + fcol, (unsigned)((i & 3) + 8 * (i >> 2)) * ldx4);
        float gacc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) gacc[i] = 0.0f;
        for (int c0 = 0; c0 < CT; c0 += kDxGroup) {
            f32x16 D[kDxGroup];
#pragma unroll
            for (int j = 0; j < kDxGroup; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) D[j][i] = 0.0f;
            const float* gsrc = pack + ((long)ft * CT + c0) * Q * 64;
            if (STAGE) {
                const int n4 = min(kDxGroup, CT - c0) * Q * 16;           // float4 items
                __syncthreads();                                           // the previous tile's readers are done
                for (int i = threadIdx.x; i < n4; i += blockDim.x)
                    reinterpret_cast<float4*>(s_w)[i] = reinterpret_cast<const float4*>(gsrc)[i];
                __syncthreads();
            }
            const float* wp = (STAGE ? s_w : gsrc) + lane;
            for (int q = 0; q < Q; ++q) {
                const float a = arow[q];
#pragma unroll
                for (int j = 0; j < kDxGroup; ++j)
                    if (c0 + j < CT)
                        D[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wp[((long)j * Q + q) * 64], D[j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float xv = xq[i];
                float Nv[K + 1], dN[K + 1];
                const int m = eval_basis<K, true, PF>(xv, s_knots, geom, knots_g, min(f, in - 1), Nv, dN);
                const float sg = silu_gradf(xv);
                float s = 0.0f;
#pragma unroll
                for (int j = 0; j < kDxGroup; ++j) {
                    const int c = c0 + j;
                    if (c < CT) {
                        const float coef = (c == C) ? sg : pick_basis<K>(dN, m, c);
                        s = fmaf(D[j][i], coef, s);
                    }
                }
                gacc[i] += s;
                // one value at a time: left free, the scheduler interleaves the 16 basis evaluations of this unrolled loop and
                // the kernel spilled up to 53 VGPRs beside its 144 accumulators (exact-fp32 mode is the documented fallback)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (f < in) {
#pragma unroll
            for (int i = 0; i < 16; ++i)           // rows >= N fall past the descriptor: dropped
                gst_s(gxb, gx_rb + fcol, (unsigned)((i & 3) + 8 * (i >> 2)) * ldgx4, gacc[i]);
        }
    }
}

// ------------------------------------------------------------------ weight gradient
// D_c[f][o] += sum_n B_c(x[n][f]) * gy[n][o]: the contraction runs over rows, two per MFMA.
// grid = (NBx, FT*OT): block.y picks the (f-tile, o-tile) role, every wave walks its own row
// range and writes one partial slab; kan_dw_reduce sums the slabs in a fixed order.
constexpr int kDwGroup = 9;
constexpr int kDwAhead = 4;   // row pairs loaded ahead per trip of the weight-gradient loop

template <int K, bool PF>
__global__ __launch_bounds__(256, 2) void kan_dw_f32_kernel(
    const float* __restrict__ x, long ldx, const float* __restrict__ gy, long ldgy, long N, int in,
    int out, int C, const float* __restrict__ knots_g, int nknots, int OT, long rows_per_wave,
    float* __restrict__ slab) {
    __shared__ float s_knots[kMaxKnots];
    if (threadIdx.x < nknots) s_knots[threadIdx.x] = knots_g[threadIdx.x];
    __syncthreads();
    const SplineGeom geom = geom_from_knots(s_knots, nknots);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, kh = lane >> 5;
    const int ft = blockIdx.y / OT, ot = blockIdx.y % OT;
    const int FT = gridDim.y / OT;
    const long s = (long)blockIdx.x * 4 + wave;           // slab index
    const long rbeg = s * rows_per_wave;
    const long rend = min(N, rbeg + rows_per_wave);
    const int CT = C + 1;
    const int f = 32 * ft + r, o = 32 * ot + r;
    const bool fv = f < in, ov = o < out;
    const long inP = 32L * FT, outP = 32L * OT;

    for (int c0 = 0; c0 < CT; c0 += kDwGroup) {
        f32x16 D[kDwGroup];
#pragma unroll
        for (int j = 0; j < kDwGroup; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) D[j][i] = 0.0f;
        // kDwAhead row pairs per trip: their x / gy loads are all in flight before the first basis is evaluated (one
        // pair per trip left every trip waiting on its own two loads)
        for (long n0 = rbeg; n0 < rend; n0 += 2 * kDwAhead) {
            float xs[kDwAhead], bs[kDwAhead];
#pragma unroll
            for (int u = 0; u < kDwAhead; ++u) {
                const long nc = min(n0 + 2 * u + kh, N - 1);   // clamped unconditional loads; `live` masks the products
                xs[u] = x[nc * ldx + min(f, in - 1)];
                bs[u] = gy[nc * ldgy + min(o, out - 1)];
            }
#pragma unroll
            for (int u = 0; u < kDwAhead; ++u) {
                const bool nv = n0 + 2 * u + kh < rend;
                const float xv = xs[u], b = bs[u];
                float Nv[K + 1], dummy[K + 1];
                const int m = eval_basis<K, false, PF>(xv, s_knots, geom, knots_g, min(f, in - 1), Nv, dummy);
                const float sl = siluf(xv);
                const bool live = nv && fv;
#pragma unroll
                for (int j = 0; j < kDwGroup; ++j) {
                    const int c = c0 + j;
                    if (c < CT) {
                        float a = (c == C) ? sl : pick_basis<K>(Nv, m, c);
                        a = live ? a : 0.0f;
                        D[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, D[j], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kDwGroup; ++j) {
            const int c = c0 + j;
            if (c < CT) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int fl = 32 * ft + mfma32_row(i, kh);
                    slab[((s * CT + c) * inP + fl) * outP + o] = D[j][i];
                }
            }
        }
    }
}

// gcat[c][f][o] = sum_s slab[s][c][f][o]   (fixed order => deterministic)
__global__ void kan_dw_reduce_kernel(const float* __restrict__ slab, long NS, long per_slab,
                                     float* __restrict__ gcat) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= per_slab) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    long s = 0;
    for (; s + 4 <= NS; s += 4) {
        a0 += slab[(s + 0) * per_slab + i];
        a1 += slab[(s + 1) * per_slab + i];
        a2 += slab[(s + 2) * per_slab + i];
        a3 += slab[(s + 3) * per_slab + i];
    }
    for (; s < NS; ++s) a0 += slab[s * per_slab + i];
    gcat[i] = (a0 + a1) + (a2 + a3);
}

int kan_dw_reduce(const float* slab, long NS, long per_slab, float* gcat, hipStream_t st) {
    kan_dw_reduce_kernel<<<cdiv(per_slab, 256), 256, 0, st>>>(slab, NS, per_slab, gcat);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// chain rule through scaled_spline_weight (ekan.py:146-152):
//   g_spline_weight = gW * scaler ; g_scaler = sum_c gW * spline_weight ; g_base = gcat[C]
__global__ void kan_dw_unpack_kernel(const float* __restrict__ gcat, int in, int out, int C,
                                     long inP, long outP, const float* __restrict__ sw,
                                     const float* __restrict__ sc, float* __restrict__ g_bw,
                                     float* __restrict__ g_sw, float* __restrict__ g_sc) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)in * out) return;
    const int o = i % out, f = i / out;       // consecutive threads -> consecutive o (coalesced gcat reads)
    const long of = (long)o * in + f;
    float gs = 0.0f;
    const float scale = sc ? sc[of] : 1.0f;
    for (int c = 0; c < C; ++c) {
        const float g = gcat[((long)c * inP + f) * outP + o];
        g_sw[of * C + c] = g * scale;
        gs = fmaf(g, sw[of * C + c], gs);
    }
    if (g_sc) g_sc[of] = gs;
    if (g_bw) g_bw[of] = gcat[((long)C * inP + f) * outP + o];
}

int kan_dw_unpack(const float* gcat, int in, int out, int C, long inP, long outP, const float* sw,
                  const float* sc, float* g_bw, float* g_sw, float* g_sc, hipStream_t st) {
    kan_dw_unpack_kernel<<<cdiv((long)in * out, 256), 256, 0, st>>>(gcat, in, out, C, inP, outP, sw, sc, g_bw, g_sw, g_sc);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// ------------------------------------------------------------------ host launchers

size_t kan_f32_pack_fwd_bytes(int in, int out, int C) {
    return (size_t)((in + 1) / 2) * (C + 1) * cdiv(out, 32) * 64 * sizeof(float);
}
size_t kan_f32_pack_dx_bytes(int in, int out, int C) {
    return (size_t)cdiv(in, 32) * (C + 1) * 16 * cdiv(out, 32) * 64 * sizeof(float);
}

int kan_f32_pack(const float* bw, const float* sw, const float* sc, int in, int out, int C,
                 float* pf, float* pd, hipStream_t st) {
    long n = (long)(kan_f32_pack_fwd_bytes(in, out, C) + kan_f32_pack_dx_bytes(in, out, C)) / 4;
    int blocks = (int)min((n + 255) / 256, (long)4096);
    kan_pack_f32_kernel<<<blocks, 256, 0, st>>>(bw, sw, sc, in, out, C, pf, pd);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

template <int K, bool PF>
static int fwd_dispatch(const float* x, long ldx, long N, int in, int out, int C, const float* knots,
                        int g, const float* pack, float* y, long ldy, hipStream_t st) {
    const int OTt = cdiv(out, 32);
    dim3 grid(cdiv(N, 128));
    for (int ot0 = 0; ot0 < OTt; ot0 += 4) {
        const int n = min(4, OTt - ot0);
#define L(OTN) kan_fwd_f32_kernel<K, OTN, PF><<<grid, 256, 0, st>>>(x, ldx, N, in, C, knots, g, pack, ot0, OTt, y, ldy, out)
        if (n == 1) L(1); else if (n == 2) L(2); else if (n == 3) L(3); else L(4);
#undef L
        KAGNN_LAUNCH_CHECK();
    }
    return KAGNN_OK;
}

int kan_f32_fwd(const float* x, long ldx, long N, const float* knots, int in,
                int out, int G, int K, const float* pack, float* y, long ldy, bool pf, hipStream_t st) {
    const int g = G + 2 * K + 1;   // number of knots
    const int C = G + K;
    if (pf) switch (K) {
        case 1: return fwd_dispatch<1, true>(x, ldx, N, in, out, C, knots, g, pack, y, ldy, st);
        case 2: return fwd_dispatch<2, true>(x, ldx, N, in, out, C, knots, g, pack, y, ldy, st);
        case 3: return fwd_dispatch<3, true>(x, ldx, N, in, out, C, knots, g, pack, y, ldy, st);
        case 4: return fwd_dispatch<4, true>(x, ldx, N, in, out, C, knots, g, pack, y, ldy, st);
    }
    else switch (K) {
        case 1: return fwd_dispatch<1, false>(x, ldx, N, in, out, C, knots, g, pack, y, ldy, st);
        case 2: return fwd_dispatch<2, false>(x, ldx, N, in, out, C, knots, g, pack, y, ldy, st);
        case 3: return fwd_dispatch<3, false>(x, ldx, N, in, out, C, knots, g, pack, y, ldy, st);
        case 4: return fwd_dispatch<4, false>(x, ldx, N, in, out, C, knots, g, pack, y, ldy, st);
    }
    return fail(KAGNN_ERR_UNSUPPORTED, "%s: spline_order must be 1..4", "kan_f32_fwd");
}

int kan_f32_dx(const float* x, long ldx, const float* gy, long ldgy, long N, const float* knots,
               int in, int out, int G, int K, const float* pack, float* gx,
               long ldgx, bool pf, hipStream_t st) {
    const int g = G + 2 * K + 1;   // number of knots
    const int C = G + K, OTt = cdiv(out, 32);
    // waves per workgroup: 8 (two per SIMD) with the W tile staged in LDS when both fit, else as many as the gy tiles leave room for
    const size_t wtile = (size_t)kDxGroup * 16 * OTt * 64 * sizeof(float);
    auto gy_bytes = [&](int w) { return (kMaxKnots + (size_t)w * 32 * (32 * OTt + 1)) * sizeof(float); };
    int W = 8;
    while (W > 1 && gy_bytes(W) + wtile > 160 * 1024) W >>= 1;
    const bool stage = gy_bytes(W) + wtile <= 160 * 1024 && W >= 2;
    if (!stage) { W = 4; while (W > 1 && gy_bytes(W) > 160 * 1024) W >>= 1; }
    const size_t lds = gy_bytes(W) + (stage ? wtile : 0);
    if (lds > 160 * 1024) return fail(KAGNN_ERR_UNSUPPORTED, "%s: out_features too large for the fp32 dx kernel", "kan_f32_dx");
    if ((long)32 * W * max(ldx, ldgx) * 4 >= 0xF0000000L) return fail(KAGNN_ERR_UNSUPPORTED, "%s: leading dimension too large", "kan_f32_dx");
    dim3 grid(cdiv(N, 32 * W));
#define L2(KK, PF, ST)                                                                            \
    {                                                                                             \
        if (lds > 64 * 1024)                                                                      \
            KAGNN_HIP(hipFuncSetAttribute((const void*)kan_dx_f32_kernel<KK, PF, ST>,             \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        kan_dx_f32_kernel<KK, PF, ST><<<grid, 64 * W, lds, st>>>(x, ldx, gy, ldgy, N, in, out, C, knots, g, pack, OTt, gx, ldgx); \
    }
#define L1(KK, PF) { if (stage) L2(KK, PF, true) else L2(KK, PF, false) }
#define L(KK) { if (pf) L1(KK, true) else L1(KK, false) }
    switch (K) {
        case 1: L(1) break;
        case 2: L(2) break;
        case 3: L(3) break;
        case 4: L(4) break;
        default: return fail(KAGNN_ERR_UNSUPPORTED, "%s: spline_order must be 1..4", "kan_f32_dx");
    }
#undef L2
#undef L
#undef L1
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

void dw_plan(long N, int in, int out, int* NBx, long* rpw) {
    const int roles = cdiv(in, 32) * cdiv(out, 32);
    int nb = max(1, 512 / roles);                 // ~2 waves per SIMD across the chip
    long waves = (long)nb * 4;
    long r = (N + waves - 1) / waves;
    r = max(2L, (r + 1) & ~1L);
    nb = (int)max(1L, (cdiv(N, r) + 3) / 4);
    *NBx = nb;
    *rpw = r;
}

size_t kan_f32_dw_ws_bytes(long N, int in, int out, int C) {
    int nb; long rpw;
    dw_plan(N, in, out, &nb, &rpw);
    const size_t per = (size_t)(C + 1) * 32 * cdiv(in, 32) * 32 * cdiv(out, 32);
    return ((size_t)nb * 4 + 1) * per * sizeof(float);
}

int kan_f32_dw(const float* x, long ldx, const float* gy, long ldgy, long N, const float* knots,
               int in, int out, int G, int K, const float* sw, const float* sc,
               float* g_bw, float* g_sw, float* g_sc, float* ws, size_t ws_bytes, bool pf, hipStream_t st) {
    const int g = G + 2 * K + 1;   // number of knots
    const int C = G + K, FT = cdiv(in, 32), OT = cdiv(out, 32);
    if (ws_bytes < kan_f32_dw_ws_bytes(N, in, out, C)) return fail(KAGNN_ERR_ARG, "%s: workspace too small", "kan_f32_dw");
    int nb; long rpw;
    dw_plan(N, in, out, &nb, &rpw);
    const long NS = (long)nb * 4;
    const long per = (long)(C + 1) * 32 * FT * 32 * OT;
    float* gcat = ws;
    float* slab = ws + per;
    dim3 grid(nb, FT * OT);
#define L(KK) if (pf) kan_dw_f32_kernel<KK, true><<<grid, 256, 0, st>>>(x, ldx, gy, ldgy, N, in, out, C, knots, g, OT, rpw, slab); \
              else kan_dw_f32_kernel<KK, false><<<grid, 256, 0, st>>>(x, ldx, gy, ldgy, N, in, out, C, knots, g, OT, rpw, slab)
    switch (K) {
        case 1: L(1); break;
        case 2: L(2); break;
        case 3: L(3); break;
        case 4: L(4); break;
        default: return fail(KAGNN_ERR_UNSUPPORTED, "%s: spline_order must be 1..4", "kan_f32_dw");
    }
#undef L
    KAGNN_LAUNCH_CHECK();
    { int rc = kan_dw_reduce(slab, NS, per, gcat, st); if (rc) return rc; }
    kan_dw_unpack_kernel<<<cdiv((long)in * out, 256), 256, 0, st>>>(gcat, in, out, C, 32L * FT, 32L * OT, sw, sc, g_bw, g_sw, g_sc);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

}  // namespace kagnn
