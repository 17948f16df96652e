// Split-precision backward of the efficient-KAN layer (see kan_split.hip for the numeric scheme).
//
// Input gradient (kan_split_dx_kernel):  D_c[n][f] = sum_o gy[n][o] * W[o][f][c]  as 32x32x16 fp16
// MFMAs (3 per product), one accumulator per coefficient c and one for the base weight; the lane that
// owns D[.][f] then contracts over c in registers with the K+1 non-zero basis derivatives of x[n][f]
// (recomputed -- only the layer input was saved).  gy rows are scaled per row by an exact power of two
// into fp16 range; the packed W^T fragments stay resident in LDS.
//
// Weight gradient (kan_split_dw_kernel):  D_c[f][o] += sum_n B_c(x[n][f]) * gy[n][o]  as 16x16x32 fp16
// MFMAs over 32-row chunks; a lane evaluates the bases of 8 consecutive rows of ONE feature, the 8x8
// (row, coefficient) block is transposed in registers with v_perm_b32.  gy is scaled per wave with a
// running power-of-two maximum (accumulators are rescaled, exactly, when it grows).  The SiLU branch
// (unbounded inputs) uses v_mfma_f32_16x16x4_f32.  Partial sums go to per-wave slabs reduced in a
// fixed order (deterministic).
//
// Reference behaviour replaced: the autograd backward of node_classification_clean/ekan.py:154-162.
#include "split_common.h"

namespace kagnn {

int split_absmax(const float*, const float*, const float*, int, int, int, unsigned*, hipStream_t);
int kan_dw_reduce(const float* slab, long NS, long per_slab, float* gcat, hipStream_t st);
int kan_dw_unpack(const float* gcat, int in, int out, int C, long inP, long outP, const float* sw,
                  const float* sc, float* g_bw, float* g_sw, float* g_sc, hipStream_t st);

// ====================================================================== input gradient
constexpr int kCTmax = 9;     // C + 1 <= 9 accumulators (8 spline coefficients + base); unused slots carry zero weights
static inline int dx_q2(int out) { return out <= 32 ? 1 : (out <= 64 ? 2 : 4); }   // 32-wide k-steps

bool kan_split_dx_ok(int in, int out, int G, int K) { return K >= 1 && K <= 3 && G + K <= 8 && out <= 128; }

size_t kan_split_pack_dx_bytes(int in, int out, int C) {
    return kHdrBytes + (size_t)cdiv(in, 16) * kCTmax * dx_q2(out) * 2 * 1024;   // always 9 slots: branch-free MFMA loop
}

// pack_dx[ft16][c][q2][part][lane][8] : lane (f = lane&15, kg = lane>>4), j -> W'[o = 32*q2+8*kg+j][16*ft16+f][c]
__global__ void split_pack_dx_kernel(const float* __restrict__ bw, const float* __restrict__ sw,
                                     const float* __restrict__ sc, int in, int out, int C, int Q2,
                                     unsigned char* __restrict__ pack) {
    unsigned* hdr = reinterpret_cast<unsigned*>(pack);
    const int e = scale_exp_from_max(__uint_as_float(hdr[2]));
    const float wscale = ldexpf(1.0f, -e);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        reinterpret_cast<float*>(pack)[0] = ldexpf(1.0f, e - 10);
        reinterpret_cast<int*>(pack)[1] = e;
    }
    const int CT = kCTmax;
    const long total = (long)cdiv(in, 16) * CT * Q2 * 64;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int lane = i & 63; long r = i >> 6;
        const int q = r % Q2; r /= Q2;
        const int c = r % CT; const int ft = r / CT;
        const int f = 16 * ft + (lane & 15);
        _Float16* dh = reinterpret_cast<_Float16*>(pack + kHdrBytes + ((size_t)((ft * CT + c) * Q2 + q) * 2 + 0) * 1024 + lane * 16);
        _Float16* dl = reinterpret_cast<_Float16*>(pack + kHdrBytes + ((size_t)((ft * CT + c) * Q2 + q) * 2 + 1) * 1024 + lane * 16);
        for (int j = 0; j < 8; ++j) {
            const int o = 32 * q + 8 * (lane >> 4) + j;
            const float w = wcat_s(bw, sw, sc, in, out, C, o, f, c) * wscale;
            const _Float16 h = (_Float16)w;
            dh[j] = h;
            dl[j] = (_Float16)(w - (float)h);
        }
    }
}

int kan_split_pack_dx(const float* bw, const float* sw, const float* sc, int in, int out, int C,
                      void* pack_dx, hipStream_t st) {
    unsigned char* p = static_cast<unsigned char*>(pack_dx);
    KAGNN_HIP(hipMemsetAsync(p, 0, kHdrBytes, st));
    { int rc = split_absmax(bw, sw, sc, in, out, C, reinterpret_cast<unsigned*>(p), st); if (rc) return rc; }
    const int Q2 = dx_q2(out);
    const long items = (long)cdiv(in, 16) * kCTmax * Q2 * 64;
    split_pack_dx_kernel<<<(int)min((items + 255) / 256, 2048L), 256, 0, st>>>(bw, sw, sc, in, out, C, Q2, p);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// sum_r dN[r] * d[m - K + r]  (d[.] = 0 outside 0..7).  Two select stages over the register array:
// by m>>2 (window of K+4 consecutive entries), then by m&3.  Written as select CHAINS on purpose:
// `cond ? a[i] : a[j]` on one array gets folded into a dynamically indexed (scratch) access.
template <int K>
__device__ __forceinline__ float barrel_dot(const float (&d)[8], int m, const float (&dN)[K + 1]) {
    const int a = m >> 2, b = m & 3;
    float U[K + 4];
#pragma unroll
    for (int j = 0; j < K + 4; ++j) {
        float v = 0.0f;
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) {
            const int c = 4 * aa + j - K;          // a constant after unrolling
            if (c >= 0 && c < 8) v = (a == aa) ? d[c] : v;
        }
        U[j] = v;
    }
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r <= K; ++r) {
        float e = U[r];
#pragma unroll
        for (int bb = 1; bb < 4; ++bb) e = (b == bb) ? U[r + bb] : e;
        s = fmaf(e, dN[r], s);
    }
    return s;
}


// One wave = 32 rows (two 16-row MFMA tiles) x one 16-feature tile at a time.  v_mfma_f32_16x16x32_f16:
// A lane (row = l&15, kg = l>>4) holds gy[row][32*q2 + 8*kg + j]; B lane (f = l&15, kg) holds W^T;
// D lane (f = l&15) holds rows 4*kg + reg.
template <int K, int Q2>
__global__ __launch_bounds__(512) void kan_split_dx_kernel(
    const float* __restrict__ x, long ldx, const float* __restrict__ gy, long ldgy, long N, int in,
    int out, int C, const float* __restrict__ knots_g, int nknots,
    const unsigned char* __restrict__ pack, int resident, float* __restrict__ gx, long ldgx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_knots = reinterpret_cast<float*>(smem);
    unsigned char* s_w = smem + kLdsHdr;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid < nknots) s_knots[tid] = knots_g[tid];
    const int FT = cdiv(in, 16);
    constexpr int FT_BYTES = kCTmax * Q2 * 2 * 1024;
    const int e_w = reinterpret_cast<const int*>(pack)[1];
    const unsigned char* gw = pack + kHdrBytes;
    auto stage = [&](int ft0, int nft) {
        const uint4* src = reinterpret_cast<const uint4*>(gw + (size_t)ft0 * FT_BYTES);
        uint4* dst = reinterpret_cast<uint4*>(s_w);
        const int n16 = nft * FT_BYTES / 16;
        for (int i = tid; i < n16; i += 512) dst[i] = src[i];
    };
    if (resident) stage(0, FT);
    __syncthreads();
    const SplineGeom geom = geom_from_knots(s_knots, nknots);
    const FastGeom fgeo = fast_geom(s_knots, nknots);
    const int li = lane & 15, kg = lane >> 4;
    const bool al4 = ((ldgy & 3) == 0) && ((reinterpret_cast<uintptr_t>(gy) & 15) == 0);
    const GBuf xb = gbuf(x, N, ldx, in), gyb = gbuf(gy, N, ldgy, out), gxb = gbuf(gx, N, ldgx, in);
    const unsigned ldx4 = (unsigned)ldx * 4u, ldgy4 = (unsigned)ldgy * 4u, ldgx4 = (unsigned)ldgx * 4u;

    for (long tile = blockIdx.x; tile * 256 < N; tile += gridDim.x) {
        const long row0 = tile * 256 + wave * 32;
        // ---- A operand: gy rows scaled per row by 2^(10 - rexp), split into fp16 hi / lo
        u32x4 ahi[2][Q2], alo[2][Q2];
        float rinv[2][4];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            // buffer loads with 32-bit offsets: rows >= N read as 0 (never stored anyway); columns >= out are
            // clamped to the row's last value and meet zero weights in the pack
            const unsigned ro = (unsigned)(row0 + 16 * rt + li) * ldgy4;
            float raw[Q2][8];
            float mx = 0.0f;
#pragma unroll
            for (int q = 0; q < Q2; ++q) {
                const int o0 = 32 * q + 8 * kg;
                if (al4 && 32 * Q2 == out) {              // wave-uniform
                    gld4(gyb, ro + o0 * 4, raw[q]);
                    gld4(gyb, ro + o0 * 4 + 16, raw[q] + 4);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) raw[q][j] = gld(gyb, ro + min(o0 + j, out - 1) * 4);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(raw[q][j]));
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const int rexp = exp_for_max(mx);
            const float sc = ldexpf(1.0f, 10 - rexp);
#pragma unroll
            for (int q = 0; q < Q2; ++q) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = raw[q][j] * sc;
                split_f16x2(v, ahi[rt][q], alo[rt][q]);
            }
            const float mine = ldexpf(1.0f, e_w + rexp - 10);      // undo factor of this lane's row
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) rinv[rt][reg] = __shfl(mine, 4 * kg + reg);
        }

        for (int ft = 0; ft < FT; ++ft) {
            if (!resident) {
                __syncthreads();
                stage(ft, 1);
                __syncthreads();
            }
            const unsigned char* wft = s_w + (size_t)(resident ? ft : 0) * FT_BYTES + lane * 16;
            // this lane's 8 x values of the tile: issue the loads now, they land under the MFMAs
            const int f = 16 * ft + li;
            float xq[2][4];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const long rr = row0 + 16 * rt + 4 * kg + reg;
                    xq[rt][reg] = gld(xb, (unsigned)rr * ldx4 + min(f, in - 1) * 4);   // rows >= N -> 0, never stored
                }
            f32x4 D[kCTmax][2];
#pragma unroll
            for (int c = 0; c < kCTmax; ++c) { D[c][0] = f32x4{0.f, 0.f, 0.f, 0.f}; D[c][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int q = 0; q < Q2; ++q) {
#pragma unroll
                for (int c = 0; c < kCTmax; ++c) {          // all 9 slots, no branch: unused ones hold zero weights
                    const u32x4 bhi = *reinterpret_cast<const u32x4*>(wft + ((size_t)(c * Q2 + q) * 2 + 0) * 1024);
                    const u32x4 blo = *reinterpret_cast<const u32x4*>(wft + ((size_t)(c * Q2 + q) * 2 + 1) * 1024);
                    D[c][0] = mfma16_f16(ahi[0][q], bhi, D[c][0]);
                    D[c][1] = mfma16_f16(ahi[1][q], bhi, D[c][1]);
                    D[c][0] = mfma16_f16(ahi[0][q], blo, D[c][0]);
                    D[c][1] = mfma16_f16(ahi[1][q], blo, D[c][1]);
                    D[c][0] = mfma16_f16(alo[0][q], bhi, D[c][0]);
                    D[c][1] = mfma16_f16(alo[1][q], bhi, D[c][1]);
                }
            }
            // ---- contraction over c with the local basis derivatives (barrel shift by the span index)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const long rr = row0 + 16 * rt + 4 * kg + reg;
                    const float xv = xq[rt][reg];
                    float dN[K + 1];
                    int m;
                    if constexpr (K == 3) {
                        float u; bool inside;
                        fast_span(xv, fgeo, m, u, inside);
                        cubic_dbases(u, inside ? 0.5f * fgeo.inv_h : 0.0f, dN);
                    } else {
                        float Nv[K + 1];
                        m = bspline_local<K, true>(xv, s_knots, geom, Nv, dN);
                    }
                    float d[kCTmax - 1];                       // this element's per-coefficient sums
#pragma unroll
                    for (int c = 0; c < kCTmax - 1; ++c) d[c] = (c < C) ? D[c][rt][reg] : 0.0f;
                    float db = 0.0f;                          // base-weight accumulator sits at index C
#pragma unroll
                    for (int c = 0; c < kCTmax; ++c) db = (c == C) ? D[c][rt][reg] : db;
                    const float s = fmaf(db, silu_gradf(xv), barrel_dot<K>(d, m, dN));
                    if (f < in) gst(gxb, (unsigned)rr * ldgx4 + f * 4, s * rinv[rt][reg]);   // rows >= N: dropped by the descriptor
                }
            }
        }
    }
}

template <int K, int Q2>
static int launch_dx(const float* x, long ldx, const float* gy, long ldgy, long N, int in, int out, int C,
                     const float* knots, int nknots, const unsigned char* pack, float* gx, long ldgx,
                     hipStream_t st) {
    const int FT = cdiv(in, 16);
    const size_t ft_bytes = (size_t)kCTmax * Q2 * 2 * 1024;
    const size_t budget = 160 * 1024 - kLdsHdr;
    const bool resident = (size_t)FT * ft_bytes <= budget;
    const size_t lds = kLdsHdr + (resident ? FT : 1) * ft_bytes;
    static bool configured = false;
    if (!configured) {
        KAGNN_HIP(hipFuncSetAttribute((const void*)kan_split_dx_kernel<K, Q2>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        configured = true;
    }
    const int grid = (int)min((long)cdiv(N, 256), 256L);
    kan_split_dx_kernel<K, Q2><<<grid, 512, lds, st>>>(x, ldx, gy, ldgy, N, in, out, C, knots, nknots, pack,
                                                       resident ? 1 : 0, gx, ldgx);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

int kan_split_dx(const float* x, long ldx, const float* gy, long ldgy, long N, const float* knots, int in,
                 int out, int G, int K, const void* pack, float* gx, long ldgx, hipStream_t st) {
    const int C = G + K, nk = G + 2 * K + 1, Q2 = dx_q2(out);
    const unsigned char* p = static_cast<const unsigned char*>(pack);
#define GO(KK, QQ) return launch_dx<KK, QQ>(x, ldx, gy, ldgy, N, in, out, C, knots, nk, p, gx, ldgx, st)
#define BYQ(KK) switch (Q2) { case 1: GO(KK, 1); case 2: GO(KK, 2); case 4: GO(KK, 4); }
    switch (K) {
        case 1: BYQ(1) break;
        case 2: BYQ(2) break;
        case 3: BYQ(3) break;
    }
#undef BYQ
#undef GO
    return fail(KAGNN_ERR_UNSUPPORTED, "%s: shape not covered by the split path", "kan_split_dx");
}

// ====================================================================== weight gradient
bool kan_split_dw_ok(int in, int out, int G, int K) { return K >= 1 && K <= 3 && G + K <= 8; }

struct DwPlan { int nbx; long rpw; long NS; long per; int FG, OC; long inP, outP; };

static DwPlan split_dw_plan(long N, int in, int out, int C) {
    DwPlan p;
    p.FG = cdiv(in, 64); p.OC = cdiv(out, 64);
    const int roles = p.FG * p.OC;
    int nb = max(1, 256 / roles);                      // ~1 workgroup per CU
    long r = (N + nb - 1) / nb;
    r = max(32L, (r + 31) & ~31L);                     // whole 32-row chunks
    nb = (int)max(1L, (long)cdiv(N, r));
    p.nbx = nb; p.rpw = r; p.NS = nb;
    p.inP = 32L * cdiv(in, 32); p.outP = 32L * cdiv(out, 32);
    p.per = (long)(C + 1) * p.inP * p.outP;
    return p;
}

size_t kan_split_dw_ws_bytes(long N, int in, int out, int C) {
    const DwPlan p = split_dw_plan(N, in, out, C);
    return (size_t)(p.NS + 1) * p.per * sizeof(float);
}

// one workgroup = 4 waves = 64 features x 64 outputs over rows [rbeg, rend); wave w owns features
// 64*fg + 16*w .. +15.  slab[s][c][f][o].
//
// Software pipeline (1 wave per SIMD: the 160 accumulator registers leave no room for a second wave,
// and a lone wave issues at most one instruction per ~4 cycles): the MFMAs of chunk i are interleaved
// at source level with the VALU work that prepares chunk i+1, and the global loads of chunk i+2 are
// already in flight.
struct DwRaw { float x[8]; float g[4][8]; };
struct DwFrag {
    u32x4 rh[8], rl[8];          // per row: 8-slot windows (hi / lo) of this lane's feature
    u32x4 bhi[4], blo[4];        // gy * 2^(10-T), per 16-wide output tile
};

template <int K>
__global__ __launch_bounds__(256) void kan_split_dw_kernel(
    const float* __restrict__ x, long ldx, const float* __restrict__ gy, long ldgy, long N, int in,
    int out, int C, const float* __restrict__ knots_g, int nknots, int OC, long rows_per_block,
    long inP, long outP, float* __restrict__ slab) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[kLdsHdr];
    float* s_knots = reinterpret_cast<float*>(smem);
    unsigned* s_tbl = reinterpret_cast<unsigned*>(smem + 256);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid < nknots) s_knots[tid] = knots_g[tid];
    build_perm_table(s_tbl, tid);
    __syncthreads();
    const SplineGeom geom = geom_from_knots(s_knots, nknots);
    const FastGeom fgeo = fast_geom(s_knots, nknots);
    const int fg = blockIdx.y / OC, oc = blockIdx.y % OC;
    const int li = lane & 15, kg = lane >> 4;
    const int f = 64 * fg + 16 * wave + li;            // A side: this lane's feature
    const long s = blockIdx.x;
    const long rbeg = s * rows_per_block, rend = min(N, rbeg + rows_per_block);

    f32x4 D[kCTmax - 1][4];        // spline coefficients x 4 o-tiles (scaled by 2^(20 - T))
    f32x4 Db[4];                   // base weight (plain fp32)
#pragma unroll
    for (int c = 0; c < kCTmax - 1; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) D[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) Db[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const GBuf xb = gbuf(x, N, ldx, in), gyb = gbuf(gy, N, ldgy, out);
    const unsigned ldx4 = (unsigned)ldx * 4u, ldgy4 = (unsigned)ldgy * 4u;
    const unsigned fo = (unsigned)min(f, in - 1) * 4u;
    unsigned go[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) go[t] = (unsigned)min(64 * oc + 16 * t + li, out - 1) * 4u;
    auto load_raw = [&](long n0, DwRaw& r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // unconditional buffer loads, 32-bit offsets.  Rows >= N read as 0 through the descriptor; the chunk
            // prefetched past this block's range is never multiplied (spline) or is masked (base, `live`);
            // features >= in / outputs >= out are clamped and only reach slab entries nobody reads.
            const unsigned n = (unsigned)(n0 + 8 * kg + j);
            r.x[j] = gld(xb, n * ldx4 + fo);
#pragma unroll
            for (int t = 0; t < 4; ++t) r.g[t][j] = gld(gyb, n * ldgy4 + go[t]);
        }
    };
    // wave-uniform exponent of the chunk's largest |gy|
    auto chunk_exp = [&](const DwRaw& r) -> int {
        float mx = 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(r.g[t][j]));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        return exp_for_max(mx);
    };
    auto make_b = [&](const DwRaw& r, int t, float gs, DwFrag& fr) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = r.g[t][j] * gs;
        split_f16x2(v, fr.bhi[t], fr.blo[t]);
    };
    auto make_row = [&](const DwRaw& r, int j, DwFrag& fr) {
        spline_frag<K>(r.x[j], s_knots, s_tbl, geom, fgeo, fr.rh[j], fr.rl[j]);
    };
    // SiLU base branch of one chunk straight from the raw values: exact fp32 MFMA, 4 rows per
    // instruction (k-lane kg <-> row 8*kg + j); rows past the end have gy == 0
    auto base_row = [&](const DwRaw& r, int j, float live) {
        const float a = siluf(r.x[j]) * live;
#pragma unroll
        for (int t = 0; t < 4; ++t) Db[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, r.g[t][j], Db[t], 0, 0, 0);
    };

    int T = -1000;                 // running exponent: gy is fed as gy * 2^(10 - T)
    DwRaw r1, r2;
    DwFrag cur, nxt;
    load_raw(rbeg, r1);
    T = chunk_exp(r1);
    {
        const float gs = ldexpf(1.0f, 10 - T);
#pragma unroll
        for (int t = 0; t < 4; ++t) make_b(r1, t, gs, cur);
#pragma unroll
        for (int j = 0; j < 8; ++j) { make_row(r1, j, cur); base_row(r1, j, 1.0f); }
    }
    load_raw(rbeg + 32, r1);

    for (long n0 = rbeg; n0 < rend; n0 += 32) {
        load_raw(n0 + 64, r2);                           // two chunks ahead, lands during this iteration
        const float live_next = (n0 + 32 < rend) ? 1.0f : 0.0f;   // the next chunk may belong to another workgroup
        const int Tn = max(T, chunk_exp(r1));            // exponent the NEXT chunk's gy is scaled with
        const float gsn = ldexpf(1.0f, 10 - Tn);
        // ---- MFMAs of the current chunk, interleaved with the preparation of the next one
#pragma unroll
        for (int c = 0; c < kCTmax - 1; ++c) {              // all 8 slots, no branch: slots >= C are always zero
            {
                const int q = c >> 1;
                const unsigned sel = (c & 1) ? 0x07060302u : 0x05040100u;
                u32x4 ah, al;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    ah[p] = __builtin_amdgcn_perm(cur.rh[2 * p + 1][q], cur.rh[2 * p][q], sel);
                    al[p] = __builtin_amdgcn_perm(cur.rl[2 * p + 1][q], cur.rl[2 * p][q], sel);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) D[c][t] = mfma16_f16(ah, cur.bhi[t], D[c][t]);
#pragma unroll
                for (int t = 0; t < 4; ++t) D[c][t] = mfma16_f16(ah, cur.blo[t], D[c][t]);
#pragma unroll
                for (int t = 0; t < 4; ++t) D[c][t] = mfma16_f16(al, cur.bhi[t], D[c][t]);
            }
            make_row(r1, c, nxt);                        // independent VALU work: row c of the next chunk
            base_row(r1, c, live_next);                  // ... and its base-branch MFMAs (fp32, unscaled)
            if (c < 4) make_b(r1, c, gsn, nxt);
        }
        if (Tn > T) {                                    // wave-uniform: rescale what was accumulated so far
            const float dn = ldexpf(1.0f, T - Tn);
#pragma unroll
            for (int c = 0; c < kCTmax - 1; ++c)
#pragma unroll
                for (int t = 0; t < 4; ++t) D[c][t] *= dn;
            T = Tn;
        }
        cur = nxt;
        r1 = r2;
    }
    // ---- slab write: D rows <-> features 4*kg + reg, cols <-> outputs li
    const float undo = ldexpf(1.0f, T - 20);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const long o = 64 * oc + 16 * t + li;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const long fl = 64 * fg + 16 * wave + 4 * kg + reg;
            if (fl < inP && o < outP) {
#pragma unroll
                for (int c = 0; c < kCTmax - 1; ++c)
                    if (c < C) slab[((s * (C + 1) + c) * inP + fl) * outP + o] = D[c][t][reg] * undo;
                slab[((s * (C + 1) + C) * inP + fl) * outP + o] = Db[t][reg];
            }
        }
    }
}

int kan_split_dw(const float* x, long ldx, const float* gy, long ldgy, long N, const float* knots, int in,
                 int out, int G, int K, const float* sw, const float* sc, float* g_bw, float* g_sw,
                 float* g_sc, float* ws, size_t ws_bytes, hipStream_t st) {
    const int C = G + K, nk = G + 2 * K + 1;
    const DwPlan p = split_dw_plan(N, in, out, C);
    if (ws_bytes < (size_t)(p.NS + 1) * p.per * sizeof(float)) return fail(KAGNN_ERR_ARG, "%s: workspace too small", "kan_split_dw");
    float* gcat = ws;
    float* slab = ws + p.per;
    dim3 grid(p.nbx, p.FG * p.OC);
#define L(KK) kan_split_dw_kernel<KK><<<grid, 256, 0, st>>>(x, ldx, gy, ldgy, N, in, out, C, knots, nk, p.OC, p.rpw, p.inP, p.outP, slab)
    switch (K) {
        case 1: L(1); break;
        case 2: L(2); break;
        case 3: L(3); break;
        default: return fail(KAGNN_ERR_UNSUPPORTED, "%s: spline_order must be 1..3", "kan_split_dw");
    }
#undef L
    KAGNN_LAUNCH_CHECK();
    { int rc = kan_dw_reduce(slab, p.NS, p.per, gcat, st); if (rc) return rc; }
    return kan_dw_unpack(gcat, in, out, C, p.inP, p.outP, sw, sc, g_bw, g_sw, g_sc, st);
}

}  // namespace kagnn
