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
#include <type_traits>
#include "split_common.h"

namespace kagnn {

int kan_dw_reduce(const float* slab, long NS, long per_slab, float* gcat, hipStream_t st);

int kan_dw_unpack(const float* gcat, int in, int out, int C, long inP, long outP, const float* sw,
                  const float* sc, float* g_bw, float* g_sw, float* g_sc, hipStream_t st);

// ====================================================================== input gradient

bool kan_split_dx_ok(int in, int out, int G, int K) { return K >= 0 && K <= 4 && G + K <= 16; }   // K == 0: RBF basis
static inline int vshift(int C) { return C > 8 ? 1 : 0; }     // 9..16 coefficients: two 8-slot windows per feature (wcat_v)

// outputs (the contraction dimension here) go in blocks of <= 128, one pack / launch per block
static size_t dx_blk_bytes(int inv /* virtual features */, int ob) {
    return kHdrBytes + (size_t)cdiv(inv, 16) * kCTmax * dx_q2(ob) * 2 * 1024;   // always 9 slots: branch-free MFMA loop
}
// cubic layers with 9..16 coefficients and one output block: window-major tiles for kan_split_dx_w2_kernel
bool kan_dx_w2_ok(int in, int out, int C, int K) { return K == 3 && C > 8 && C <= 16 && out <= kOutBlk; }
static int dx_inv(int in, int out, int C, int K) { return kan_dx_w2_ok(in, out, C, K) ? 32 * cdiv(in, 16) : (in << vshift(C)); }

size_t kan_split_pack_dx_bytes(int in, int out, int C, int K) {
    return (size_t)cdiv(out, kOutBlk) * dx_blk_bytes(dx_inv(in, out, C, K), min(out, kOutBlk));
}

__global__ void split_pack_dx_kernel(const float* __restrict__ bw, const float* __restrict__ sw,
                                     const float* __restrict__ sc, int in, int out, int C, int Q2,
                                     unsigned char* __restrict__ pack, int self_scale, int w2) {
    unsigned* hdr = reinterpret_cast<unsigned*>(pack);
    __shared__ float s_m[17];
    const float wmax = self_scale == 2 ? header_absmax(pack) : self_scale ? block_absmax_w(bw, sw, sc, in, out, C, s_m) : __uint_as_float(hdr[2]);
    const int e = scale_exp_from_max(wmax);
    const float wscale = ldexpf(1.0f, -e);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        reinterpret_cast<float*>(pack)[0] = ldexpf(1.0f, e - 10);
        reinterpret_cast<int*>(pack)[1] = e;
    }
    pack_dx_items(bw, sw, sc, in, out, C, Q2, pack, wscale, blockIdx.x * (long)blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x, w2);
}

int kan_split_pack_dx_noscale(const float* bw, const float* sw, const float* sc, int in, int out, int C, int K,
                              void* pack_dx, hipStream_t st) {
    const int w2 = kan_dx_w2_ok(in, out, C, K) ? 1 : 0;
    const int inv = dx_inv(in, out, C, K);
    const size_t stride = dx_blk_bytes(inv, min(out, kOutBlk));
    for (int b = 0; b * kOutBlk < out; ++b) {
        const int ob = min(kOutBlk, out - b * kOutBlk), Q2 = dx_q2(ob);
        const long o0 = (long)b * kOutBlk;
        const long items = (long)cdiv(inv, 16) * kCTmax * Q2 * 64;
        const bool two = (long)in * ob * C >= kAbsmaxTwoLaunchMin;      // large blocks: partial maxima first (split_common.h)
        unsigned char* pd = static_cast<unsigned char*>(pack_dx) + b * stride;
        if (two) { int rc = launch_absmax_partials(bw ? bw + o0 * in : nullptr, sw + o0 * in * C, sc ? sc + o0 * in : nullptr, in, ob, C, pd, st); if (rc) return rc; }
        split_pack_dx_kernel<<<(int)min((items + 1023) / 1024, two ? 256L : 64L), 1024, 0, st>>>(
            bw ? bw + o0 * in : nullptr, sw + o0 * in * C, sc ? sc + o0 * in : nullptr, in, ob, C, Q2, pd, two ? 2 : 1, w2);
        KAGNN_LAUNCH_CHECK();
    }
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
// GEN == false is the lean instantiation of the common case (one output block, <= 8 coefficients): no
// accumulate / virtual-feature code at all.  GEN == true takes both as run-time (wave-uniform) flags.
//
// Schedule (PP).  A unit of work is (row tile, 16-feature tile): an M phase (9 slots x Q2 k-steps x 6 MFMAs, fed
// from the W^T fragments in LDS) and a V phase (per scalar: basis derivatives, barrel contraction over the slots,
// SiLU', store).  The two phases use different pipes but inside ONE wave they run back to back, and the two waves
// that share a SIMD drift into the same phase (PMC, round 1: VALU-active 61 % + MFMA-busy 44 % of the SIMD's
// time, 40 % of every wave's cycles parked in s_waitcnt -- the gy rows of a tile were loaded at its start).
//   PP >= 1: the gy rows of the NEXT row tile are requested a whole tile ahead and split into fragments at the end
//            of the current tile's last V phase (no exposed HBM latency at tile boundaries);
//   PP == 2: the workgroup's waves 4..7 (the SIMD partners of waves 0..3) run one phase behind, held there by one
//            s_barrier per phase: a SIMD always has one wave on the matrix pipe and one on the VALU
//            (MI355X_MICROARCH.md, "Two waves per SIMD").
//   BNB: `gy` is the gradient of the BatchNorm1d that follows the layer; the norm's backward (one affine expression per
//   element, split_common.h BnBack) is applied to the rows as they are loaded, and the transformed rows are stored for the
//   weight-gradient kernel -- the stand-alone normalisation-backward pass (read g, read y, write gy) disappears.
// XST (with XAFF; the read-out's gradient of the LAST convolution's output): also leave the column sums of the stored gradient rows
// and of their products with xhat (RbfArgs::st_*) -- the two sums the backward of the norm whose folded output this layer read
// starts from.  Per feature tile a lane adds its 8 rows, the four row groups of a wave meet by two shuffles, and the wave adds the
// tile's 16 columns into its own LDS slots behind the W tiles; the 8 waves are added in order at the end: one partial row pair
// per workgroup, deterministic.  Needs <= 4 feature tiles per workgroup (in <= 64, no split over blockIdx.y).
// HALF (KAGNN_PREC_HALF, split_common.h): gy rounded once per row scale, only the hi W^T fragments are staged and read: 2 MFMAs per
// (slot, k-step) instead of 6; the V phase (basis derivatives, contraction, SiLU') stays fp32.
template <int K, int Q2, bool GEN, int PP, bool GX16 = false, bool BNB = false, bool XAFF = false, bool XST = false, bool HALF = false>
__global__ __launch_bounds__(512) void kan_split_dx_kernel(
    const float* __restrict__ x, long ldx, const float* __restrict__ gy, long ldgy, long N, int in,
    int out, int C, const float* __restrict__ knots_g, int nknots,
    const unsigned char* __restrict__ pack, int resident, float* __restrict__ gx, long ldgx,
    RbfArgs rb, int sh_arg /* 1: virtual features, two 8-slot windows per input feature */, int acc_arg,
    int ft_per_block /* feature tiles per blockIdx.y: few-row inputs spread their feature tiles over the chip */, BnBack bnb) {
    static_assert(!BNB || (K == 3 && !GEN && PP == 0 && !GX16), "the fused normalisation backward serves the lean cubic instantiation");
    static_assert(!XAFF || (K == 3 && !GEN && !GX16 && !BNB), "the input affine serves the lean cubic instantiation (the read-out of the node models)");
    static_assert(!XST || XAFF, "the column statistics ride in the read-out instantiation");
    static_assert(!HALF || (K == 3 && !GEN), "single-product mode: the lean cubic instantiation");
    // GX16: gx rows are bf16 (a compile-time variant of the lean cubic instantiation -- as a run-time flag the 2-byte
    // store path cost every launch 14 %: round 2, profiles/r02_experiments.md)
    const int sh = GEN ? sh_arg : 0;
    const bool ACC = GEN && acc_arg;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_knots = reinterpret_cast<float*>(smem);
    unsigned char* s_w = smem + kLdsHdr;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (K > 0 && tid < nknots) s_knots[tid] = knots_g[tid];
    unsigned* s_btbl = reinterpret_cast<unsigned*>(smem + 256);      // barrel selectors (K == 3)
    constexpr unsigned gxes = GX16 ? 2u : 4u;                        // bytes per gx element
    if (K == 3) build_barrel_table(s_btbl, tid);
    const int inv = in << sh, FT = cdiv(inv, 16);
    constexpr int FT_BYTES = kCTmax * Q2 * 2 * 1024;
    const int e_w = reinterpret_cast<const int*>(pack)[1];
    const unsigned char* gw = pack + kHdrBytes;
    // global -> LDS by LDS-DMA (lds_dma_1k, split_common.h): one KiB per wave and instruction, no register round trip
    const unsigned lds_w = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)s_w);
    auto stage = [&](int ft0, int nft) {
        const unsigned char* src = gw + (size_t)ft0 * FT_BYTES;
        const int nblk = nft * (FT_BYTES / 1024);
        if constexpr (HALF) {                                // [c][q][hi|lo] KiB blocks: the even ones
            for (int blk = 2 * wave; blk < nblk; blk += 16)
                lds_dma_1k(src + (size_t)blk * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(lds_w + blk * 1024));
        } else
        for (int blk = wave; blk < nblk; blk += 8)
            lds_dma_1k(src + (size_t)blk * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(lds_w + blk * 1024));
        lds_dma_wait();
    };
    const int ft_begin = blockIdx.y * ft_per_block, ft_end = min(FT, ft_begin + ft_per_block);
    if (resident) stage(ft_begin, ft_end - ft_begin);
    __syncthreads();
    // the phase barriers need every wave of the workgroup to run the same number of phases: that holds when the
    // W^T fragments are resident (no staging barriers inside the loop); otherwise fall back to the plain schedule
    const bool pingpong = (PP == 2) && resident;
    SplineGeom geom{}; FastGeom fgeo{};
    const int li = lane & 15, kg = lane >> 4;
    const int win = li & sh;                                     // this lane's slot window (virtual feature parity)
    float ca[8] = {};
    if constexpr (K == 0) {
        float c0[8], c1[8];
        rbf_centers(rb, c0, 0); rbf_centers(rb, c1, sh);
#pragma unroll
        for (int g = 0; g < 8; ++g) ca[g] = win ? c1[g] : c0[g];
    }
    if constexpr (K > 0) { geom = geom_from_knots(s_knots, nknots); fgeo = fast_geom(s_knots, nknots); }
    const bool ln_on = (K == 0) && rb.ln_w != nullptr;           // wave-uniform
    const bool al4 = ((ldgy & 3) == 0) && ((reinterpret_cast<uintptr_t>(gy) & 15) == 0);
    const unsigned ldx4 = (unsigned)ldx * 4u, ldgy4 = (unsigned)ldgy * 4u, ldgx4 = (unsigned)ldgx * gxes;
    const unsigned gy_ro = (unsigned)(wave * 32 + li) * ldgy4;   // tile-relative byte offsets (descriptors open at the tile)
    const unsigned x_rb = (unsigned)(wave * 32 + 4 * kg) * ldx4;
    const unsigned gx_rb = (unsigned)(wave * 32 + 4 * kg) * ldgx4;
    const unsigned gz_rb = (unsigned)(wave * 32 + 4 * kg) * (unsigned)in * 4u;

    // ---- gy rows of one row tile: raw loads (buffer loads with 32-bit offsets: rows >= N read as 0 and are never
    // stored; columns >= out are clamped to the row's last value and meet zero weights in the pack) ...
    auto load_gy = [&](long tile, float (&raw)[2][Q2][8]) {
        const GBuf gyb = gbuf_at(gy, N, ldgy, out, tile * 256);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const unsigned ro = gy_ro + kg * 32;             // (row0 + li) * ldgy4 + 8*kg*4
            const unsigned so = (unsigned)(16 * rt) * ldgy4; // wave-uniform
#pragma unroll
            for (int q = 0; q < Q2; ++q) {
                if (al4 && 32 * Q2 == out) {              // wave-uniform
                    gld4_s(gyb, ro, so + 128 * q, raw[rt][q]);
                    gld4_s(gyb, ro, so + 128 * q + 16, raw[rt][q] + 4);
                } else if (al4 && (out & 7) == 0) {       // wave-uniform: out = 8, 16, 24, 40, 48, 56 (the 40-class read-out: eight 4-byte
                    // loads per group made its input gradient 313 us against 251 us at 64 outputs).  A group that lies beyond
                    // `out` re-reads the row's last whole group: finite values that meet zero weights, as the clamped scalars did
                    const unsigned rq = gy_ro + (unsigned)min(32 * q + 8 * kg, out - 8) * 4u;
                    gld4_s(gyb, rq, so, raw[rt][q]);
                    gld4_s(gyb, rq, so + 16, raw[rt][q] + 4);
                } else {
                    const int o0 = 32 * q + 8 * kg;
#pragma unroll
                    for (int j = 0; j < 8; ++j) raw[rt][q][j] = gld_s(gyb, gy_ro + min(o0 + j, out - 1) * 4, so);
                }
            }
        }
        if constexpr (BNB) {                                 // (the host only launches this with out == 32 * Q2, aligned rows)
            const GBuf yb = gbuf_at(bnb.y, N, bnb.ldy, out, tile * 256), ob = gbuf_at(bnb.gy_out, N, bnb.ldo, out, tile * 256);
            const unsigned ldy4 = (unsigned)bnb.ldy * 4u, ldo4 = (unsigned)bnb.ldo * 4u;
            const unsigned yro = (unsigned)(wave * 32 + li) * ldy4 + kg * 32, oro = (unsigned)(wave * 32 + li) * ldo4 + kg * 32;
#pragma unroll
            for (int q = 0; q < Q2; ++q) {
                float cm[8], cA[8], cB[8], cC[8];
                const float* t0 = bnb.tab + 32 * q + 8 * kg;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 a = *reinterpret_cast<const float4*>(t0 + 4 * h), b = *reinterpret_cast<const float4*>(t0 + bnb.ldt + 4 * h);
                    const float4 c = *reinterpret_cast<const float4*>(t0 + 2 * bnb.ldt + 4 * h), d = *reinterpret_cast<const float4*>(t0 + 3 * bnb.ldt + 4 * h);
                    cm[4 * h] = a.x; cm[4 * h + 1] = a.y; cm[4 * h + 2] = a.z; cm[4 * h + 3] = a.w;
                    cA[4 * h] = b.x; cA[4 * h + 1] = b.y; cA[4 * h + 2] = b.z; cA[4 * h + 3] = b.w;
                    cB[4 * h] = c.x; cB[4 * h + 1] = c.y; cB[4 * h + 2] = c.z; cB[4 * h + 3] = c.w;
                    cC[4 * h] = d.x; cC[4 * h + 1] = d.y; cC[4 * h + 2] = d.z; cC[4 * h + 3] = d.w;
                }
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    float yv[8];
                    gld4_s(yb, yro, (unsigned)(16 * rt) * ldy4 + 128 * q, yv);
                    gld4_s(yb, yro, (unsigned)(16 * rt) * ldy4 + 128 * q + 16, yv + 4);
#pragma unroll
                    for (int j = 0; j < 8; ++j) raw[rt][q][j] = bn_bwd_value(raw[rt][q][j], yv[j], cm[j], cA[j], cB[j], cC[j]);
                    if (blockIdx.y == 0) {                   // (rows >= N: past the descriptor, dropped)
                        gst4_s(ob, oro, (unsigned)(16 * rt) * ldo4 + 128 * q, raw[rt][q]);
                        gst4_s(ob, oro, (unsigned)(16 * rt) * ldo4 + 128 * q + 16, raw[rt][q] + 4);
                    }
                }
            }
        }
    };
    // ... and their A fragments: scaled per row by 2^(10 - rexp), split into fp16 hi / lo
    u32x4 ahi[2][Q2], alo[2][Q2];
    float rinv[2][4];
    auto split_gy = [&](const float (&raw)[2][Q2][8]) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            float mx = 0.0f;
#pragma unroll
            for (int q = 0; q < Q2; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(raw[rt][q][j]));
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const int rexp = exp_for_max(mx);
            const float sc = ldexpf(1.0f, 10 - rexp);
#pragma unroll
            for (int q = 0; q < Q2; ++q) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = raw[rt][q][j] * sc;
                if constexpr (HALF) round_f16x2(v, ahi[rt][q]);
                else split_f16x2_asm(v, ahi[rt][q], alo[rt][q]);
            }
            const float mine = ldexpf(1.0f, e_w + rexp - 10);      // undo factor of this lane's row
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) rinv[rt][reg] = __shfl(mine, 4 * kg + reg);
        }
    };
    float mu[2][4], rs[2][4];                          // layernorm statistics of this lane's 8 rows (RBF basis only)
    auto load_stats = [&](long tile) {
        if (ln_on) {
            const long row0 = tile * 256 + wave * 32;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const long rc = min(row0 + 16 * rt + 4 * kg + reg, N - 1);
                    mu[rt][reg] = rb.stats[2 * rc]; rs[rt][reg] = rb.stats[2 * rc + 1];
                }
        }
    };

    // XST: [wave 8][stat 2][feature 64] floats behind the (resident) W tiles
    float* s_stat = reinterpret_cast<float*>(s_w + (size_t)(ft_end - ft_begin) * FT_BYTES) + wave * 128;
    if constexpr (XST) { s_stat[lane] = 0.0f; s_stat[64 + lane] = 0.0f; }   // (each wave touches only its own slice until the end)
    float graw[2][Q2][8];
    long tile = blockIdx.x;
    if (!XST && tile * 256 >= N) return;                         // (workgroup-uniform; the grid never over-covers)
    if (!XST || tile * 256 < N) {
    load_gy(tile, graw);
    load_stats(tile);
    split_gy(graw);
    if (PP >= 1 && (tile + gridDim.x) * 256 < N) load_gy(tile + gridDim.x, graw);
    }
    if (pingpong && wave >= 4) __builtin_amdgcn_s_barrier();     // the partner half starts one phase late

    for (; tile * 256 < N; tile += gridDim.x) {
        const bool more = (tile + gridDim.x) * 256 < N;
        if (PP == 0 && tile != (long)blockIdx.x) {               // plain schedule: load + split at the tile's start
            load_gy(tile, graw);
            load_stats(tile);
            split_gy(graw);
        }
        // descriptors opened at the workgroup's tile: per-lane offsets are tile-relative and 32-bit for any N
        const GBuf gzb = gbuf_at(rb.gz, N, in, in, tile * 256);
        const GBuf xb = gbuf_at(x, N, ldx, in, tile * 256), gxb = gbuf_at_es(gx, N, ldgx, in, tile * 256, (int)gxes);

        for (int ft = ft_begin; ft < ft_end; ++ft) {
            if (!resident) {
                __syncthreads();
                stage(ft, 1);
                __syncthreads();
            }
            const unsigned char* wft = s_w + (size_t)(resident ? ft - ft_begin : 0) * FT_BYTES + lane * 16;
            // this lane's 8 x values of the tile: issue the loads now, they land under the MFMAs
            const int f = 16 * ft + li;                  // (virtual) feature of this lane; fr = the real one
            const int fr = f >> sh;
            const unsigned fcol = (unsigned)min(fr, in - 1) * 4u;
            const unsigned gx_ro = gx_rb + (unsigned)min(fr, in - 1) * gxes, gz_ro = gz_rb + fcol;
            float gam = 1.0f, bet = 0.0f;
            if (ln_on) { gam = rb.ln_w[min(fr, in - 1)]; bet = rb.ln_b[min(fr, in - 1)]; }
            float xa = 1.0f, xs = 0.0f;                  // XAFF: the input is  xa * x + xs  (a folded BatchNorm1d)
            if constexpr (XAFF) { xa = rb.x_affine[min(fr, in - 1)]; xs = rb.x_affine[in + min(fr, in - 1)]; }
            float st_m = 0.0f, st_q = 0.0f, st_g = 0.0f, st_gx = 0.0f;
            if constexpr (XST) { st_m = rb.st_mean[min(fr, in - 1)]; st_q = rb.st_rstd[min(fr, in - 1)]; }
            float xq[2][4];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    // rows >= N -> 0 (never stored); a partial last feature tile re-reads the clamped column
                    xq[rt][reg] = gld_s(xb, x_rb + fcol, (unsigned)(16 * rt + reg) * ldx4);
                }
            // ================= M phase
            f32x4 D[kCTmax][2];
#pragma unroll
            for (int c = 0; c < kCTmax; ++c) { D[c][0] = f32x4{0.f, 0.f, 0.f, 0.f}; D[c][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            // 9 slots x Q2 k-steps, no branch (unused slots hold zero weights).  The W fragments of group g+1 are
            // read from LDS BEFORE the MFMAs of group g are issued, so the LDS latency hides under them.
            {
                constexpr int NGRP = kCTmax * Q2;
#ifndef KAGNN_DX_LDS_DEPTH
#define KAGNN_DX_LDS_DEPTH 1      // (2 and 3 measured: no change -- the M phase does not wait on LDS; profiles/r03_experiments.md)
#endif
                constexpr int RD = KAGNN_DX_LDS_DEPTH;       // fragment reads in flight ahead of the MFMAs that use them
                u32x4 bh[RD + 1], bl[RD + 1];
#pragma unroll
                for (int g = 0; g < RD; ++g) {
                    bh[g] = *reinterpret_cast<const u32x4*>(wft + (size_t)(2 * g + 0) * 1024);
                    if constexpr (!HALF) bl[g] = *reinterpret_cast<const u32x4*>(wft + (size_t)(2 * g + 1) * 1024);
                }
#pragma unroll
                for (int g = 0; g < NGRP; ++g) {
                    const int c = g / Q2, q = g % Q2;      // LDS order is [c][q][hi|lo], i.e. group g at 2g KiB
                    if (g + RD < NGRP) {
                        bh[(g + RD) % (RD + 1)] = *reinterpret_cast<const u32x4*>(wft + (size_t)(2 * (g + RD) + 0) * 1024);
                        if constexpr (!HALF) bl[(g + RD) % (RD + 1)] = *reinterpret_cast<const u32x4*>(wft + (size_t)(2 * (g + RD) + 1) * 1024);
                    }
                    __builtin_amdgcn_sched_barrier(0);     // pin: hipcc otherwise sinks the reads back to their first use
                    const u32x4 bhi = bh[g % (RD + 1)];
                    D[c][0] = mfma16_f16(ahi[0][q], bhi, D[c][0]);
                    D[c][1] = mfma16_f16(ahi[1][q], bhi, D[c][1]);
                    if constexpr (!HALF) {
                        const u32x4 blo = bl[g % (RD + 1)];
                        D[c][0] = mfma16_f16(ahi[0][q], blo, D[c][0]);
                        D[c][1] = mfma16_f16(ahi[1][q], blo, D[c][1]);
                        D[c][0] = mfma16_f16(alo[0][q], bhi, D[c][0]);
                        D[c][1] = mfma16_f16(alo[1][q], bhi, D[c][1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (pingpong) __builtin_amdgcn_s_barrier();
            // ================= V phase: contraction over c with the local basis derivatives (barrel shift by the span index)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const float xv = XAFF ? fmaf(xq[rt][reg], xa, xs) : xq[rt][reg];
                    if constexpr (K == 0) {
                        // Gaussian RBF: gz = k2 * sum_g D_g * phi_g(z) * t_g   (gradient w.r.t. z), gb = D_base * silu'(x)
                        const float z = ln_on ? fmaf((xv - mu[rt][reg]) * rs[rt][reg], gam, bet) : xv;
                        const float t0 = z * rb.a;
                        float sgz = 0.0f;
#pragma unroll
                        for (int g = 0; g < kCTmax - 1; ++g) {
                            const float t = t0 - ca[g];
                            sgz = fmaf(D[g][rt][reg], __builtin_amdgcn_exp2f(-t * t) * t, sgz);
                        }
                        float vz = sgz * rb.k2 * rinv[rt][reg];
                        float vb = D[kCTmax - 1][rt][reg] * silu_gradf(xv) * rinv[rt][reg];
                        if (sh) vz += __shfl_xor(vz, 1);                 // the two windows of one input feature (wave-uniform branch)
                        const unsigned so_x = (unsigned)(16 * rt + reg) * ldgx4, so_z = (unsigned)(16 * rt + reg) * (unsigned)in * 4u;
                        if (f < inv && win == 0) {
                            if (ln_on) {
                                if (ACC) { vz += gld_s(gzb, gz_ro, so_z); vb += gld_s(gxb, gx_ro, so_x); }
                                gst_s(gzb, gz_ro, so_z, vz);
                                gst_s(gxb, gx_ro, so_x, vb);
                            } else {
                                float v = vz + vb;
                                if (ACC) v += gld_s(gxb, gx_ro, so_x);
                                gst_s(gxb, gx_ro, so_x, v);
                            }
                        }
                    } else {
                    float dN[K + 1];
                    int m;
                    if constexpr (K == 3) {
                        // span from arithmetic only, support folded into the barrel table: floor(t) outside [0, last span] selects an
                        // all-zero entry (negative -> 15 through the unsigned min; beyond the last span the window holds only slots
                        // >= C, whose sums are exact zeros), so no compare / select per scalar; the derivative scale 1/(2h) is applied
                        // once to the contracted sum instead of to the four pieces
#ifdef KAGNN_ABLATE_SHARED_EXPANSION
                        // TIMING-ONLY ablation (wrong results): what a dX that RECEIVED span and cubic pieces from elsewhere would
                        // still execute -- the upper bound of sharing the per-scalar expansion with dW (profiles/r04_experiments.md)
                        m = (int)(__float_as_uint(xv) & 7u);
                        dN[0] = xv; dN[1] = xv; dN[2] = xv; dN[3] = xv;
#else
                        const float t = fmaf(xv, fgeo.inv_h, fgeo.c0);
                        const float u = __builtin_amdgcn_fractf(t);
                        m = (int)floorf(t);
                        cubic_dbases(u, 1.0f, dN);           // (round 3: -9 instructions per scalar, dX -4.7 %)
#endif
                    } else {
                        float Nv[K + 1];
                        m = bspline_local<K, true>(xv, s_knots, geom, Nv, dN);
                    }
                    float d[kCTmax - 1];                       // per-coefficient sums (slots >= C are exact zeros)
#pragma unroll
                    for (int c = 0; c < kCTmax - 1; ++c) d[c] = D[c][rt][reg];
                    float bar;
                    if constexpr (K == 3) {
                        const int mm = m - 8 * win;
                        const unsigned* be = s_btbl + kBarrelDw * min((unsigned)mm, 15u);
                        const u32x4 sel = *reinterpret_cast<const u32x4*>(be);
                        const uint2 rot = *reinterpret_cast<const uint2*>(be + 4);
                        bar = barrel_dot3(d, sel, rot.x, rot.y, dN) * (0.5f * fgeo.inv_h);
                    } else {
                        bar = barrel_dot<K>(d, m - 8 * win, dN);
                    }
                    float s = fmaf(D[kCTmax - 1][rt][reg], silu_gradf(xv), bar) * rinv[rt][reg];
                    if (sh) s += __shfl_xor(s, 1);                       // the two windows of one input feature (wave-uniform branch)
                    if constexpr (XST) {                                 // (rows >= N: gy reads as 0 there, so s == 0)
                        st_g += s;
                        st_gx = fmaf(s, (xq[rt][reg] - st_m) * st_q, st_gx);
                    }
                    if (f < inv && win == 0) {
                        const unsigned so_x = (unsigned)(16 * rt + reg) * ldgx4;
                        if (ACC) s += gld_s(gxb, gx_ro, so_x);                          // second and later output blocks
                        if constexpr (GX16) gst16_s(gxb, gx_ro, so_x, s);               // bf16 rows
                        else gst_s(gxb, gx_ro, so_x, s);                                // rows >= N: dropped
                    }
                    }
                }
            }
            if constexpr (XST) {                           // this tile's 16 columns: the four row groups, then the wave's LDS slots
                st_g += __shfl_xor(st_g, 16); st_gx += __shfl_xor(st_gx, 16);
                st_g += __shfl_xor(st_g, 32); st_gx += __shfl_xor(st_gx, 32);
                if (kg == 0 && f < in) { s_stat[f] += st_g; s_stat[64 + f] += st_gx; }
            }
            if (PP >= 1 && ft + 1 == ft_end && more) {
                // fragments of the next row tile (its rows arrived long ago), and the request for the one after
                load_stats(tile + gridDim.x);
                split_gy(graw);
                if ((tile + 2 * (long)gridDim.x) * 256 < N) load_gy(tile + 2 * (long)gridDim.x, graw);
            }
            if (pingpong) __builtin_amdgcn_s_barrier();
        }
    }
    if (pingpong && wave < 4) __builtin_amdgcn_s_barrier();      // balance the partner half's initial barrier
    if constexpr (XST) {
        __syncthreads();
        if (tid < 128) {                                         // (stat, feature): the 8 waves in order
            const float* base = reinterpret_cast<const float*>(s_w + (size_t)(ft_end - ft_begin) * FT_BYTES);
            float a = 0.0f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) a += base[w8 * 128 + tid];
            const int stat = tid >> 6, col = tid & 63;
            if (col < in) rb.st_partial[((long)blockIdx.x * 2 + stat) * in + col] = a;
        }
    }
}

// Schedule: the gy rows of the next row tile are requested a tile ahead (PP = 1).  The plain schedule (PP = 0) serves the instantiations
// that cannot afford the prefetched rows' registers; the phase ping-pong of SIMD partners (PP = 2) measured 0.555 vs 0.561 ms per step in
// round 2 -- the schedule is not what limits this kernel -- and is no longer instantiated (its environment switch went in round 5).
template <int K, int Q2, bool GEN, int PP, bool GX16 = false, bool BNB = false, bool XAFF = false, bool XST = false, bool HALF = false>
static int launch_dx_pp(const float* x, long ldx, const float* gy, long ldgy, long N, int in, int out, int C,
                        const float* knots, int nknots, const unsigned char* pack, float* gx, long ldgx,
                        const RbfArgs& rb, int accumulate, hipStream_t st, const BnBack& bnb = BnBack{}) {
    if constexpr (!HALF && K == 3 && !GEN && Q2 <= 2 && PP != 2) {      // single-product mode (thread-local, set by the entry point)
        if (g_half_products)
            return launch_dx_pp<K, Q2, GEN, PP, GX16, BNB, XAFF, XST, true>(x, ldx, gy, ldgy, N, in, out, C, knots, nknots, pack, gx, ldgx, rb, accumulate, st, bnb);
    }
    const int sh = vshift(C), FT = cdiv(in << sh, 16);
    const size_t ft_bytes = (size_t)kCTmax * Q2 * 2 * 1024;
    const size_t budget = 160 * 1024 - kLdsHdr;
    // few rows, many features (Cora: 2708 x 1433): spread the feature tiles over blockIdx.y so the chip fills
    const long row_blocks = cdiv(N, 256);
    int splits = row_blocks >= 128 ? 1 : (int)min((long)FT, 256 / row_blocks);
    const int fpb = cdiv(FT, splits);
    splits = cdiv(FT, fpb);
    const bool resident = (size_t)fpb * ft_bytes <= budget;
    if (XST && !(resident && splits == 1 && in <= 64 && (size_t)fpb * ft_bytes + 8 * 128 * sizeof(float) <= budget))
        return fail(KAGNN_ERR_UNSUPPORTED, "%s: column statistics need <= 64 input features, resident weights and >= 32768 rows", "kan_split_dx");
    const size_t lds = kLdsHdr + (resident ? fpb : 1) * ft_bytes + (XST ? 8 * 128 * sizeof(float) : 0);
    static unsigned long long configured = 0;          // (per device: common.h)
    if (auto first_use_ = first_use_on_this_device(configured)) {
        KAGNN_HIP(hipFuncSetAttribute((const void*)kan_split_dx_kernel<K, Q2, GEN, PP, GX16, BNB, XAFF, XST, HALF>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
    }
    const dim3 grid((unsigned)min(row_blocks, 256L), (unsigned)splits);
    kan_split_dx_kernel<K, Q2, GEN, PP, GX16, BNB, XAFF, XST, HALF><<<grid, 512, lds, st>>>(x, ldx, gy, ldgy, N, in, out, C, knots, nknots, pack,
                                                                           resident ? 1 : 0, gx, ldgx, rb, sh, accumulate, fpb, bnb);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

template <int K, int Q2, bool GEN>
static int launch_dx(const float* x, long ldx, const float* gy, long ldgy, long N, int in, int out, int C,
                     const float* knots, int nknots, const unsigned char* pack, float* gx, long ldgx,
                     const RbfArgs& rb, int accumulate, int gx16, hipStream_t st) {
    // the schedule experiments only pay on the common cubic instantiation; the others keep the prefetch form
    if constexpr (K == 3 && !GEN && Q2 == 4) {
        // 128 outputs: the prefetched gy rows of the next tile are 64 more live registers -- the kernel spilled 27..36 VGPRs
        // with them (profiles/r03_kernel_resources.txt); plain schedule
        if (gx16) return launch_dx_pp<K, Q2, GEN, 0, true>(x, ldx, gy, ldgy, N, in, out, C, knots, nknots, pack, gx, ldgx, rb, accumulate, st);
        return launch_dx_pp<K, Q2, GEN, 0>(x, ldx, gy, ldgy, N, in, out, C, knots, nknots, pack, gx, ldgx, rb, accumulate, st);
    } else if constexpr (K == 3 && !GEN) {
        if (rb.x_affine) {                               // the input is a folded BatchNorm1d output (read-out of the node models)
            if (gx16) return fail(KAGNN_ERR_UNSUPPORTED, "%s: an input affine and bf16 gradient rows do not combine", "kan_split_dx");
            if (rb.st_partial) {
                if constexpr (Q2 <= 2)
                    return launch_dx_pp<K, Q2, GEN, 1, false, false, true, true>(x, ldx, gy, ldgy, N, in, out, C, knots, nknots, pack, gx, ldgx, rb, accumulate, st);
                return fail(KAGNN_ERR_UNSUPPORTED, "%s: column statistics need <= 64 outputs", "kan_split_dx");
            }
            return launch_dx_pp<K, Q2, GEN, 1, false, false, true>(x, ldx, gy, ldgy, N, in, out, C, knots, nknots, pack, gx, ldgx, rb, accumulate, st);
        }
        if (gx16) return launch_dx_pp<K, Q2, GEN, 1, true>(x, ldx, gy, ldgy, N, in, out, C, knots, nknots, pack, gx, ldgx, rb, accumulate, st);
        return launch_dx_pp<K, Q2, GEN, 1>(x, ldx, gy, ldgy, N, in, out, C, knots, nknots, pack, gx, ldgx, rb, accumulate, st);
    }
    if (gx16) return fail(KAGNN_ERR_UNSUPPORTED, "%s: bf16 gradient rows need a cubic layer with <= 8 coefficients", "kan_split_dx");
    if (rb.x_affine) return fail(KAGNN_ERR_UNSUPPORTED, "%s: an input affine needs a cubic layer with <= 8 coefficients and <= 64 outputs", "kan_split_dx");
    return launch_dx_pp<K, Q2, GEN, 0>(x, ldx, gy, ldgy, N, in, out, C, knots, nknots, pack, gx, ldgx, rb, accumulate, st);
}

// ---------------------------------------------------------------------------------------------------------------
// Cubic layers with 9..16 coefficients (grid 6..13; BASELINE config 3 is grid 8 => C = 11), one output block.
// The general kernel above runs them as 2*in "virtual" features whose two 8-slot windows sit on neighbouring LANES:
// the span / derivative / SiLU' arithmetic of a scalar runs twice and all 2 x 9 slots go through the matrix cores
// although only C + 1 carry weights.  Here a lane owns ONE input feature and walks its two windows in turn over
// window-major W^T tiles (wcat_v sh == 2): window 0 = 8 spline slots + base, window 1 = only its C - 8 live slots;
// the scalar's span and SiLU' are evaluated once, the second window costs one more barrel contraction.
// Per 32 rows x 16 input features: (9 + C - 8) * Q2 * 6 MFMAs and ~95 VALU per scalar instead of 18 * Q2 * 6 and 2 x 74.
// NS1: planes of the second window the kernel runs -- C - 8 when that is <= 4, else all 8 (slots >= C - 8 carry zero weights).
// A template parameter: behind a run-time `c < C - 8` the conditional MFMAs made the compiler carry the whole 64-register
// accumulator array through phi copies (21 spilled VGPRs at Q2 = 4).
template <int Q2, int NS1>
__global__ __launch_bounds__(512) void kan_split_dx_w2_kernel(
    const float* __restrict__ x, long ldx, const float* __restrict__ gy, long ldgy, long N, int in, int out, int C,
    const float* __restrict__ knots_g, int nknots, const unsigned char* __restrict__ pack,
    float* __restrict__ gx, long ldgx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_knots = reinterpret_cast<float*>(smem);
    unsigned* s_btbl = reinterpret_cast<unsigned*>(smem + 256);
    unsigned char* s_w = smem + kLdsHdr;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid < nknots) s_knots[tid] = knots_g[tid];
    build_barrel_table(s_btbl, tid);
    const int T = cdiv(in, 16);                                  // real feature tiles
    constexpr int ns1 = NS1;                                     // planes of the second window (stage() copies that prefix)
    constexpr int FT_BYTES = kCTmax * Q2 * 2 * 1024;
    const int e_w = reinterpret_cast<const int*>(pack)[1];
    const unsigned char* gw = pack + kHdrBytes;
    // global -> LDS by LDS-DMA (lds_dma_1k, split_common.h), one KiB per wave and instruction: copied through registers
    // (global_load / s_waitcnt / ds_write by 512 threads between the two barriers) the two windows of the 8 feature tiles
    // were 23 % of this kernel at 128 -> 128.  (A double-buffered, overlapped form -- as in kan_sparse_fwd.hip -- costs
    // ~50 more registers than this kernel has: profiles/r02_experiments.md.)
    const unsigned lds_w = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)s_w);
    // Round 5: the two windows live in TWO LDS regions (window 0: 9 slots at s_w, window 1: its NS1 live slots behind them), and the
    // copy of window 0 -- 72 of the 96 KB a feature tile stages at Q2 = 4 -- is issued one phase ahead: window 0 of the next
    // (feature tile, or row tile) streams in under the MFMAs and the V phase of window 1, so its phase starts with a wait for a
    // copy issued a whole section earlier and ONE barrier (it was barrier / copy + wait / barrier: the timing-only ablation of
    // profiles/r05_experiments.md section 2 put the exposed staging at 16 % of this kernel at 128 -> 128).  The copy is issued
    // AFTER the barrier that ends the last use of its region.  Window 1 keeps the exposed form: issuing ITS copy ahead (before or
    // after the M phase of window 0) makes hipcc spill 46-48 VGPRs at Q2 = 4 in every placement tried (round 2 saw the same).
    auto stage_issue = [&](int tile, int nslots, unsigned dst_off) {      // [c][q][hi|lo] order: the first nslots slots are a prefix
        const unsigned char* src = gw + (size_t)tile * FT_BYTES;
        const int nblk = nslots * Q2 * 2;
        for (int blk = wave; blk < nblk; blk += 8)
            lds_dma_1k(src + blk * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(lds_w + dst_off + blk * 1024));
    };
    // (Q2 = 4 with exactly four live slots in the second window -- C = 12 at 128 outputs -- spills one VGPR in the ahead form: exposed there)
    constexpr bool AHEAD = !(Q2 == 4 && NS1 == 4);
    __syncthreads();
    if (AHEAD && (long)blockIdx.x * 256 < N) stage_issue(0, kCTmax, 0u);
    const FastGeom fgeo = fast_geom(s_knots, nknots);
    const float wd = 0.5f * fgeo.inv_h;
    const int li = lane & 15, kg = lane >> 4;
    const bool al4 = ((ldgy & 3) == 0) && ((reinterpret_cast<uintptr_t>(gy) & 15) == 0) && (32 * Q2 == out);
    const unsigned ldx4 = (unsigned)ldx * 4u, ldgy4 = (unsigned)ldgy * 4u, ldgx4 = (unsigned)ldgx * 4u;
    const unsigned gy_ro = (unsigned)(wave * 32 + li) * ldgy4;
    const unsigned x_rb = (unsigned)(wave * 32 + 4 * kg) * ldx4;
    const unsigned gx_rb = (unsigned)(wave * 32 + 4 * kg) * ldgx4;
    const unsigned char* wl = s_w + lane * 16;
    const unsigned char* wl1 = wl + FT_BYTES;                     // window 1's region

    for (long tile = blockIdx.x; tile * 256 < N; tile += gridDim.x) {
        const GBuf gyb = gbuf_at(gy, N, ldgy, out, tile * 256);
        const GBuf xb = gbuf_at(x, N, ldx, in, tile * 256), gxb = gbuf_at(gx, N, ldgx, in, tile * 256);
        // ---- A operand: gy rows scaled per row by 2^(10 - rexp), split into fp16 hi / lo (as in the general kernel)
        u32x4 ahi[2][Q2], alo[2][Q2];
        float rinv[2][4];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const unsigned ro = gy_ro + kg * 32, so = (unsigned)(16 * rt) * ldgy4;
            float raw[Q2][8];
            float mx = 0.0f;
#pragma unroll
            for (int q = 0; q < Q2; ++q) {
                if (al4) {
                    gld4_s(gyb, ro, so + 128 * q, raw[q]);
                    gld4_s(gyb, ro, so + 128 * q + 16, raw[q] + 4);
                } else {
                    const int o0 = 32 * q + 8 * kg;
#pragma unroll
                    for (int j = 0; j < 8; ++j) raw[q][j] = gld_s(gyb, gy_ro + min(o0 + j, out - 1) * 4, so);
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
                split_f16x2_asm(v, ahi[rt][q], alo[rt][q]);
            }
            const float mine = ldexpf(1.0f, e_w + rexp - 10);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) rinv[rt][reg] = __shfl(mine, 4 * kg + reg);
        }

        for (int t = 0; t < T; ++t) {
            const int f = 16 * t + li;
            const unsigned fcol = (unsigned)min(f, in - 1) * 4u;
            float xq[2][4];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) xq[rt][reg] = gld_s(xb, x_rb + fcol, (unsigned)(16 * rt + reg) * ldx4);
            f32x4 D[kCTmax][2];
            // ================= window 0: slots 0..7 + base
            if constexpr (AHEAD) {
                lds_dma_wait();              // this wave's blocks of window 0 (issued a phase ago) have landed ...
                __syncthreads();             // ... so have everyone's; and everyone is done with window 1 of the previous tile
            } else {
                __syncthreads();
                stage_issue(2 * t, kCTmax, 0u); lds_dma_wait();
                __syncthreads();
            }
#pragma unroll
            for (int c = 0; c < kCTmax; ++c) { D[c][0] = f32x4{0.f, 0.f, 0.f, 0.f}; D[c][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            {
                constexpr int NGRP = kCTmax * Q2;
                u32x4 bh[2], bl[2];
                bh[0] = *reinterpret_cast<const u32x4*>(wl + 0 * 1024);
                bl[0] = *reinterpret_cast<const u32x4*>(wl + 1 * 1024);
#pragma unroll
                for (int g = 0; g < NGRP; ++g) {
                    const int c = g / Q2, q = g % Q2;
                    if (g + 1 < NGRP) {
                        bh[(g + 1) & 1] = *reinterpret_cast<const u32x4*>(wl + (size_t)(2 * (g + 1) + 0) * 1024);
                        bl[(g + 1) & 1] = *reinterpret_cast<const u32x4*>(wl + (size_t)(2 * (g + 1) + 1) * 1024);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const u32x4 bhi = bh[g & 1], blo = bl[g & 1];
                    D[c][0] = mfma16_f16(ahi[0][q], bhi, D[c][0]);
                    D[c][1] = mfma16_f16(ahi[1][q], bhi, D[c][1]);
                    D[c][0] = mfma16_f16(ahi[0][q], blo, D[c][0]);
                    D[c][1] = mfma16_f16(ahi[1][q], blo, D[c][1]);
                    D[c][0] = mfma16_f16(alo[0][q], bhi, D[c][0]);
                    D[c][1] = mfma16_f16(alo[1][q], bhi, D[c][1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // V0: SiLU' once per scalar; first window's barrel contraction (the span is cheap enough to redo for the second
            // window: keeping m / u / weight across its MFMAs costs 24 registers and spills at Q2 = 4)
            float s0[2][4];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                __builtin_amdgcn_sched_barrier(0);       // 4 scalars in flight, not 8: the barrel temporaries of 8 spill at Q2 = 4
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const float xv = xq[rt][reg];
                    // (span from arithmetic only, support folded into the barrel table, 1/(2h) applied to the contracted sum:
                    // see kan_split_dx_kernel)
                    const float tt = fmaf(xv, fgeo.inv_h, fgeo.c0);
                    const float u = __builtin_amdgcn_fractf(tt);
                    const int m = (int)floorf(tt);
                    float dN[4];
                    cubic_dbases(u, 1.0f, dN);
                    float d[kCTmax - 1];
#pragma unroll
                    for (int c = 0; c < kCTmax - 1; ++c) d[c] = D[c][rt][reg];
                    const unsigned* be = s_btbl + kBarrelDw * min((unsigned)m, 15u);
                    const u32x4 sel = *reinterpret_cast<const u32x4*>(be);
                    const uint2 rot = *reinterpret_cast<const uint2*>(be + 4);
                    s0[rt][reg] = fmaf(D[kCTmax - 1][rt][reg], silu_gradf(xv), barrel_dot3(d, sel, rot.x, rot.y, dN) * wd);
                }
            }
            // ================= window 1: its C - 8 live slots only
            __syncthreads();
            stage_issue(2 * t + 1, ns1, (unsigned)FT_BYTES); lds_dma_wait();
            __syncthreads();
            if constexpr (AHEAD) {
                if (t + 1 < T) stage_issue(2 * (t + 1), kCTmax, 0u);
                else if ((tile + gridDim.x) * 256 < N) stage_issue(0, kCTmax, 0u);
            }
            f32x4 D1[NS1][2];
#pragma unroll
            for (int c = 0; c < NS1; ++c) { D1[c][0] = f32x4{0.f, 0.f, 0.f, 0.f}; D1[c][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int c = 0; c < NS1; ++c) {
#pragma unroll
                for (int q = 0; q < Q2; ++q) {
                    const u32x4 bhi = *reinterpret_cast<const u32x4*>(wl1 + (size_t)(2 * (c * Q2 + q) + 0) * 1024);
                    const u32x4 blo = *reinterpret_cast<const u32x4*>(wl1 + (size_t)(2 * (c * Q2 + q) + 1) * 1024);
                    D1[c][0] = mfma16_f16(ahi[0][q], bhi, D1[c][0]);
                    D1[c][1] = mfma16_f16(ahi[1][q], bhi, D1[c][1]);
                    D1[c][0] = mfma16_f16(ahi[0][q], blo, D1[c][0]);
                    D1[c][1] = mfma16_f16(ahi[1][q], blo, D1[c][1]);
                    D1[c][0] = mfma16_f16(alo[0][q], bhi, D1[c][0]);
                    D1[c][1] = mfma16_f16(alo[1][q], bhi, D1[c][1]);
                    __builtin_amdgcn_sched_barrier(0);                // keep the fragment reads of later slots where they are
                }
            }
            const unsigned gx_ro = gx_rb + fcol;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const float tt = fmaf(xq[rt][reg], fgeo.inv_h, fgeo.c0);
                    const float u = __builtin_amdgcn_fractf(tt);
                    const int m = (int)floorf(tt);
                    float dN[4];
                    cubic_dbases(u, 1.0f, dN);
                    float d[kCTmax - 1];
#pragma unroll
                    for (int c = 0; c < kCTmax - 1; ++c) d[c] = c < NS1 ? D1[c < NS1 ? c : 0][rt][reg] : 0.0f;
                    const int m1 = m - 8;
                    const unsigned* be = s_btbl + kBarrelDw * min((unsigned)m1, 15u);
                    const u32x4 sel = *reinterpret_cast<const u32x4*>(be);
                    const uint2 rot = *reinterpret_cast<const uint2*>(be + 4);
                    const float s = fmaf(barrel_dot3(d, sel, rot.x, rot.y, dN), wd, s0[rt][reg]) * rinv[rt][reg];
                    if (f < in) gst_s(gxb, gx_ro, (unsigned)(16 * rt + reg) * ldgx4, s);      // rows >= N: dropped
                }
            }
        }
    }
}

template <int Q2, int NS1>
static int launch_dx_w2(const float* x, long ldx, const float* gy, long ldgy, long N, int in, int out, int C,
                        const float* knots, int nknots, const unsigned char* pack, float* gx, long ldgx, hipStream_t st) {
    const size_t lds = kLdsHdr + (size_t)(kCTmax + NS1) * Q2 * 2 * 1024;       // window 0's 9 slots + window 1's NS1 (two regions)
    static unsigned long long configured = 0;          // (per device: common.h)
    if (auto first_use_ = first_use_on_this_device(configured)) {
        KAGNN_HIP(hipFuncSetAttribute((const void*)kan_split_dx_w2_kernel<Q2, NS1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
    }
    kan_split_dx_w2_kernel<Q2, NS1><<<(unsigned)min((long)cdiv(N, 256), 256L), 512, lds, st>>>(x, ldx, gy, ldgy, N, in, out, C, knots, nknots,
                                                                                        pack, gx, ldgx);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

static int dx_block(const float* x, long ldx, const float* gy, long ldgy, long N, const float* knots, int in,
                    int out, int G, int K, const unsigned char* p, float* gx, long ldgx, const RbfArgs& rb,
                    int accumulate, int gx16, hipStream_t st) {
    const int C = G + K, nk = K ? G + 2 * K + 1 : 0, Q2 = dx_q2(out);
#define GO(KK, QQ) return (accumulate || C > 8) ? launch_dx<KK, QQ, true>(x, ldx, gy, ldgy, N, in, out, C, knots, nk, p, gx, ldgx, rb, accumulate, gx16, st) \
                                                 : launch_dx<KK, QQ, false>(x, ldx, gy, ldgy, N, in, out, C, knots, nk, p, gx, ldgx, rb, 0, gx16, st)
#define BYQ(KK) switch (Q2) { case 1: GO(KK, 1); case 2: GO(KK, 2); case 4: GO(KK, 4); }
    switch (K) {
        case 0: BYQ(0) break;
        case 1: BYQ(1) break;
        case 2: BYQ(2) break;
        case 3: BYQ(3) break;
        case 4: BYQ(4) break;
    }
#undef BYQ
#undef GO
    return fail(KAGNN_ERR_UNSUPPORTED, "%s: shape not covered by the split path", "kan_split_dx");
}

// K == 0: Gaussian RBF basis (with layernorm: rb.gz receives dL/dz, gx the base-branch part only)
int kan_split_dx_any(const float* x, long ldx, const float* gy, long ldgy, long N, const float* knots, int in,
                     int out, int G, int K, const void* pack, float* gx, long ldgx, const RbfArgs& rb,
                     hipStream_t st, int gx16) {
    if (gx16 && (K == 0 || out > kOutBlk))
        return fail(KAGNN_ERR_UNSUPPORTED, "%s: bf16 gradient rows need a B-spline layer with <= 128 outputs", "kan_split_dx");
    if (kan_dx_w2_ok(in, out, G + K, K)) {               // 9..16 coefficients: one lane per input feature, two windows in turn
        if (gx16) return fail(KAGNN_ERR_UNSUPPORTED, "%s: bf16 gradient rows need <= 8 coefficients", "kan_split_dx");
        const unsigned char* p = static_cast<const unsigned char*>(pack);
        const int nk = G + 2 * K + 1;
#define W2Q(QQ, NN) launch_dx_w2<QQ, NN>(x, ldx, gy, ldgy, N, in, out, G + K, knots, nk, p, gx, ldgx, st)
#define W2(QQ) switch (G + K - 8) { case 1: return W2Q(QQ, 1); case 2: return W2Q(QQ, 2); case 3: return W2Q(QQ, 3); case 4: return W2Q(QQ, 4); \
                                    default: return W2Q(QQ, 8); }
        switch (dx_q2(out)) {
            case 1: W2(1)
            case 2: W2(2)
            default:                        // (Q2 = 4 with ONE plane happens to spill 32 VGPRs; two planes -- the second on zero weights -- do not:
                                            // C = 9 runs as <4, 2> and <4, 1> is not instantiated)
                switch (G + K - 8) { case 1: case 2: return W2Q(4, 2); case 3: return W2Q(4, 3); case 4: return W2Q(4, 4); default: return W2Q(4, 8); }
        }
#undef W2
#undef W2Q
    }
    const size_t stride = dx_blk_bytes(in << vshift(G + K), min(out, kOutBlk));
    for (int b = 0; b * kOutBlk < out; ++b) {
        const int rc = dx_block(x, ldx, gy + b * kOutBlk, ldgy, N, knots, in, min(kOutBlk, out - b * kOutBlk), G, K,
                                static_cast<const unsigned char*>(pack) + b * stride, gx, ldgx, rb, b > 0, gx16, st);
        if (rc) return rc;
    }
    return KAGNN_OK;
}

// the input gradient of a layer whose output feeds a BatchNorm1d, given the gradient `g` of the norm's OUTPUT: the norm's
// backward is applied to the rows as they are loaded (BnBack: table from bn_bwd_table, y = the norm's input) and the
// transformed rows are left in bnb.gy_out for the weight-gradient kernel.  Covered: cubic layers of <= 8 coefficients with 32
// or 64 outputs, 16-byte aligned rows everywhere.
bool kan_split_dx_bn_ok(long ldg, int in, int out, int G, int K, const BnBack& b, const void* g) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return K == 3 && G + K <= 8 && (out == 32 || out == 64) && !kan_dx_w2_ok(in, out, G + K, K) && ldg % 4 == 0 && b.ldy % 4 == 0 &&
           b.ldo % 4 == 0 && b.ldt % 4 == 0 && b.ldt >= out && b.ldy <= 7680 && b.ldo <= 7680 && al(g) && al(b.y) && al(b.gy_out) && al(b.tab);
}
int kan_split_dx_bn(const float* x, long ldx, const float* g, long ldg, long N, const float* knots, int in, int out, int G, int K,
                    const void* pack, float* gx, long ldgx, const BnBack& bnb, hipStream_t st) {
    if (!kan_split_dx_bn_ok(ldg, in, out, G, K, bnb, g)) return fail(KAGNN_ERR_UNSUPPORTED, "%s: shape not covered", "kan_split_dx_bn");
    const unsigned char* p = static_cast<const unsigned char*>(pack);
    const int nk = G + 2 * K + 1;
    if (out == 32) return launch_dx_pp<3, 1, false, 0, false, true>(x, ldx, g, ldg, N, in, out, G + K, knots, nk, p, gx, ldgx, RbfArgs{}, 0, st, bnb);
    return launch_dx_pp<3, 2, false, 0, false, true>(x, ldx, g, ldg, N, in, out, G + K, knots, nk, p, gx, ldgx, RbfArgs{}, 0, st, bnb);
}

// (statistics variant: st_partial[min(cdiv(N, 256), 256)][2][in], see kan_split_dx_kernel<..., XST>)
int kan_split_dx_stats_blocks(long N) { return (int)min((long)cdiv(N, 256), 256L); }
bool kan_split_dx_stats_ok(long N, int in, int out, int G, int K) {
    return K == 3 && G + K <= 8 && in <= 64 && in % 4 == 0 && out <= 64 && cdiv(N, 256) >= 128;
}
int kan_split_dx_stats(const float* x, long ldx, const float* gy, long ldgy, long N, const float* knots, int in, int out, int G, int K,
                       const void* pack, float* gx, long ldgx, hipStream_t st, const float* x_affine, const float* st_mean,
                       const float* st_rstd, float* st_partial) {
    if (!kan_split_dx_stats_ok(N, in, out, G, K) || !x_affine)
        return fail(KAGNN_ERR_UNSUPPORTED, "%s: column statistics need a folded input, a cubic layer of <= 8 coefficients, <= 64 inputs / outputs and >= 32768 rows", "kan_split_dx");
    RbfArgs rb{};
    rb.x_affine = x_affine; rb.st_mean = st_mean; rb.st_rstd = st_rstd; rb.st_partial = st_partial;
    return kan_split_dx_any(x, ldx, gy, ldgy, N, knots, in, out, G, K, pack, gx, ldgx, rb, st, 0);
}

int kan_split_dx(const float* x, long ldx, const float* gy, long ldgy, long N, const float* knots, int in,
                 int out, int G, int K, const void* pack, float* gx, long ldgx, hipStream_t st, int gx16, const float* x_affine) {
    RbfArgs rb{};
    rb.x_affine = x_affine;
    if (x_affine && (kan_dx_w2_ok(in, out, G + K, K) || out > 64))
        return fail(KAGNN_ERR_UNSUPPORTED, "%s: an input affine needs a cubic layer with <= 8 coefficients and <= 64 outputs", "kan_split_dx");
    return kan_split_dx_any(x, ldx, gy, ldgy, N, knots, in, out, G, K, pack, gx, ldgx, rb, st, gx16);
}

// ====================================================================== weight gradient
bool kan_split_dw_ok(int in, int out, int G, int K) { return K >= 0 && K <= 4 && G + K <= 16; }   // K == 0: RBF basis

struct DwPlan { int nbx; long rpw; long NS; long per; int FG, OC; long inP, outP; int rs; };

// rows one dW workgroup may own: with leading dimensions up to kDwMaxLd floats its slice of x / gy spans < 4 GiB
constexpr long kDwMaxRowsPerBlock = 1L << 17;

static DwPlan split_dw_plan(long N, int in, int out, int C, int K) {
    DwPlan p;
    const bool virt = C > 8;
    if (C > 8) { in <<= 1; C = 8; }                    // virtual features: 2*in features of 8 slots (wcat_v)
    p.FG = cdiv(in, 64); p.OC = cdiv(out, 64);
    const int roles = p.FG * p.OC;
    int nb = max(1, 256 / roles);                      // ~1 workgroup per CU
    long r = (N + nb - 1) / nb;
    r = max(32L, (r + 31) & ~31L);                     // whole 32-row chunks
    r = min(r, kDwMaxRowsPerBlock);                    // a workgroup's rows stay inside one 4 GiB buffer window
    nb = (int)max(1L, (long)cdiv(N, r));
    // narrow layers (<= 32 features: the input slices of the feature-sharded layer, first layers on few features): a
    // workgroup's four waves own four 16-feature tiles, so with 1 / 2 live tiles 3 / 2 of them would work on padding --
    // they take row sub-ranges of the live tiles instead (rs per tile), each with a slab of its own
    p.rs = (virt || !(K == 3 || K == 0)) ? 1 : in <= 16 ? 4 : in <= 32 ? 2 : 1;   // (built for the cubic and RBF kernels)
    p.nbx = nb; p.rpw = r; p.NS = (long)nb * p.rs;
    p.inP = 32L * cdiv(in, 32); p.outP = 32L * cdiv(out, 32);
    p.per = (long)(C + 1) * p.inP * p.outP;
    return p;
}

// cubic layers with 9..12 coefficients: kan_split_dw_w2_kernel (one lane per input feature, both slot windows)
bool kan_dw_w2_ok(int in, int out, int C, int K) { return K == 3 && C > 8 && C <= 12; }

static DwPlan split_dw_plan_w2(long N, int in, int out, int C) {
    DwPlan p;
    p.FG = cdiv(in, 64); p.OC = cdiv(out, 64);
    const int roles = p.FG * p.OC;
    int nb = max(1, 256 / roles);
    long r = (N + nb - 1) / nb;
    r = max(32L, (r + 31) & ~31L);
    r = min(r, kDwMaxRowsPerBlock);
    nb = (int)max(1L, (long)cdiv(N, r));
    p.rs = in <= 16 ? 4 : in <= 32 ? 2 : 1;            // narrow layers: idle waves take row sub-ranges (split_dw_plan)
    p.nbx = nb; p.rpw = r; p.NS = (long)nb * p.rs;
    p.inP = 32L * cdiv(in, 32); p.outP = 32L * cdiv(out, 32);
    p.per = (long)(C + 1) * p.inP * p.outP;
    return p;
}

size_t kan_split_dw_ws_bytes(long N, int in, int out, int C, int K) {
    const DwPlan p = kan_dw_w2_ok(in, out, C, K) ? split_dw_plan_w2(N, in, out, C) : split_dw_plan(N, in, out, C, K);
    return (size_t)(p.NS + 1) * p.per * sizeof(float);
}

// one workgroup = 4 waves = 64 features x 64 outputs over rows [rbeg, rend); wave w owns features
// 64*fg + 16*w .. +15.  slab[s][c][f][o].
//
// One wave per SIMD (160 accumulator registers).  Measured on MI355X these kernels cost
// ~4 cycles per VALU instruction PLUS the MFMA cycles -- the two barely overlap -- so the loop is kept
// minimal instead of deeply pipelined: loads of chunk i+1 are issued, the MFMAs of chunk i run (and
// cover the load latency), then chunk i+1 is expanded in place.  Everything non-accumulator fits the
// 256 architectural VGPRs, so nothing shuttles through AGPRs.
template <int NTO> struct DwRaw { float x[8]; float g[NTO][8]; float mu[8], rs[8]; };   // mu / rs: layernorm statistics (RBF basis only)

// RS (1, 2, 4): row sub-ranges per feature tile -- narrow layers with 4 / RS live tiles, see split_dw_plan (a template
// parameter: as a runtime value it cost the 64-feature layer 45 %)
// NTO: 16-wide output tiles per wave.  4 = one wave per SIMD owns 16 features x 64 outputs (144 accumulators); 1..3 serve layers of
// <= 48 outputs (read-outs) with only the tiles that exist.  As a way to put TWO waves on a SIMD for 64-output layers, NTO = 2
// (16 x 32 outputs, 72 accumulators, 198 registers: TWO waves per SIMD) was measured in round 3 and is not instantiated: the
// basis expansion runs twice and the pair gains nothing from sharing the SIMD -- 0.769 vs 0.604 ms per step
// (profiles/r03_experiments.md)
// HALF (KAGNN_PREC_HALF, split_common.h): bases, SiLU and gy rounded once -- no lo fragments, no lo transposition, one MFMA
// per (slot plane, output tile) instead of three
template <int K, bool GEN, int RS = 1, int NTO = 4, bool XAFF = false, bool HALF = false>      // GEN == false: <= 8 coefficients, no virtual-feature code (see kan_split_dx_kernel)
                                                                             // XAFF: the input is rb.x_affine's scale * x + shift (a folded BatchNorm1d)
__global__ __launch_bounds__(256, NTO == 4 ? 1 : 2) void kan_split_dw_kernel(
    const float* __restrict__ x, long ldx, const float* __restrict__ gy, long ldgy, long N, int in,
    int out, int C, const float* __restrict__ knots_g, int nknots, int OC, long rows_per_block,
    long inP, long outP, float* __restrict__ slab, RbfArgs rb, int sh_arg /* 1: virtual features (wcat_v); then C == 8 */) {
    constexpr int rs = RS;
    const int sh = GEN ? sh_arg : 0;
    __shared__ __attribute__((aligned(16))) unsigned char smem[kLdsHdr];
    float* s_knots = reinterpret_cast<float*>(smem);
    unsigned* s_tbl = reinterpret_cast<unsigned*>(smem + 256);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (K > 0 && tid < nknots) s_knots[tid] = knots_g[tid];
    if (K > 0) build_perm_table(s_tbl, tid);
    if (K == 4) build_perm_fix_table(s_tbl, tid);
    __syncthreads();
    SplineGeom geom{}; FastGeom fgeo{};
    const int fg = blockIdx.y / OC, oc = blockIdx.y % OC;
    const int li = lane & 15, kg = lane >> 4;
    // this wave's 16-feature tile and row sub-range.  The wave index goes through readfirstlane: the row range feeds the
    // buffer descriptors, and a descriptor the compiler cannot prove wave-uniform turns every load into a waterfall loop
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int tile = RS == 1 ? wave : wave_u % (4 / rs), rsub = RS == 1 ? 0 : wave_u / (4 / rs);
    const int fv = 64 * fg + 16 * tile + li;           // A side: this lane's (virtual) feature
    const int f = fv >> sh, win = fv & sh;             // input feature and slot window
    const unsigned woff = win ? kWinBytes : 0u;
    float ca[8] = {};
    if constexpr (K == 0) {
        float c0[8], c1[8];
        rbf_centers(rb, c0, 0); rbf_centers(rb, c1, sh);
#pragma unroll
        for (int g = 0; g < 8; ++g) ca[g] = win ? c1[g] : c0[g];
    }
    if constexpr (K > 0) { geom = geom_from_knots(s_knots, nknots); fgeo = fast_geom(s_knots, nknots); }
    const bool ln_on = (K == 0) && rb.ln_w != nullptr;           // wave-uniform
    const long s = (long)blockIdx.x * rs + rsub;       // slab of this (row block, sub-range)
    const long sub_rows = ((rows_per_block / 32 + rs - 1) / rs) * 32;     // whole 32-row chunks per sub-range
    const long rbeg = blockIdx.x * rows_per_block + rsub * sub_rows;
    const long rend = min(min(N, (blockIdx.x + 1L) * rows_per_block), rbeg + sub_rows);

    f32x4 D[kCTmax][NTO];          // spline coefficients x NTO o-tiles, scaled by 2^(20 - T); plane kCTmax - 1: the base
                                   // weight, scaled by 2^(14 - T): fp16 hi/lo products, and -- for chunks whose silu
                                   // overflows fp16 -- exact fp32 MFMAs on gy brought to the same scale.  (ONE array: as a
                                   // separate `Dh[4]` the compiler kept the base plane in arch VGPRs and copied its 16
                                   // registers to the accumulation file and back around its MFMAs in every chunk; a third
                                   // array for the fp32 path cost 16 more registers this kernel does not have)
#define Dh D[kCTmax - 1]
#pragma unroll
    for (int c = 0; c < kCTmax; ++c)
#pragma unroll
        for (int t = 0; t < NTO; ++t) D[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // descriptors opened at this workgroup's first row and closed at its last: offsets are relative to rbeg (32-bit
    // for any N; the host keeps rows_per_block * ld * 4 below 4 GiB) and rows >= rend read as 0
    const GBuf xb = gbuf_at(x, rend, ldx, in, rbeg), gyb = gbuf_at(gy, rend, ldgy, out, rbeg);
    const GBuf stb = gbuf_at(rb.stats, ln_on ? rend : 0, 2, 2, rbeg);   // rows >= N: (0, 0) -> z = beta, finite; their gy is 0
    const float gam = ln_on ? rb.ln_w[min(f, in - 1)] : 1.0f, bet = ln_on ? rb.ln_b[min(f, in - 1)] : 0.0f;
    float xa = 1.0f, xs = 0.0f;
    if constexpr (XAFF) { xa = rb.x_affine[min(f, in - 1)]; xs = rb.x_affine[in + min(f, in - 1)]; }
    unsigned sto = (unsigned)(8 * kg) * 8u;
    const unsigned ldx4 = (unsigned)ldx * 4u, ldgy4 = (unsigned)ldgy * 4u;
    // per-lane byte offsets of the chunk being fetched; rows advance by 32 per call.  Unconditional buffer
    // loads: rows >= N read as 0 through the descriptor (rows_per_block is a multiple of 32, so a chunk
    // never straddles two workgroups); features >= in / outputs >= out are clamped and only reach slab
    // entries nobody reads.
    unsigned xo = (unsigned)(8 * kg) * ldx4 + (unsigned)min(f, in - 1) * 4u, gvo[NTO];
#pragma unroll
    for (int t = 0; t < NTO; ++t) gvo[t] = (unsigned)(8 * kg) * ldgy4 + (unsigned)min(16 * NTO * oc + 16 * t + li, out - 1) * 4u;
    auto load_raw = [&](DwRaw<NTO>& r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r.x[j] = gld_s(xb, xo, (unsigned)j * ldx4);
        if (ln_on) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { r.mu[j] = gld(stb, sto + 8u * j); r.rs[j] = gld(stb, sto + 8u * j + 4u); }
            sto += 32u * 8u;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int t = 0; t < NTO; ++t) r.g[t][j] = gld_s(gyb, gvo[t], (unsigned)j * ldgy4);
        xo += 32u * ldx4;
#pragma unroll
        for (int t = 0; t < NTO; ++t) gvo[t] += 32u * ldgy4;
    };

    // wave-uniform exponent of the chunk's largest |gy|
    auto chunk_exp = [&](const DwRaw<NTO>& r) -> int {
        float mx = 0.0f;
#pragma unroll
        for (int t = 0; t < NTO; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(r.g[t][j]));
        return exp_for_max(wave_max_nonneg(mx));
    };

    // The loop is ROTATED: an iteration expands the chunk whose raw rows were requested one iteration earlier, requests the
    // next chunk's rows (they land under the MFMAs) and runs the MFMAs.  The fragments are therefore defined and consumed
    // inside ONE iteration; only the raw rows -- direct load destinations -- cross the back edge.  With the expansion at the
    // bottom of the loop (round 2) the 104 fragment registers were loop-carried and the compiler closed every iteration with
    // ~100 v_mov copies into them (16 % of the loop's VALU instructions).
    // (RBF layers) column sums of this slab's gy rows = its part of the base bias gradient: the wave of feature tile 0 of
    // feature group 0 adds up the rows it loads anyway (32 adds per chunk) instead of a pass of its own over gy
    float bsum[NTO] = {};
    const bool bias_on = (K == 0) && rb.colpart != nullptr && fg == 0 && tile == 0;      // wave-uniform
    int T;                         // running exponent: gy is fed as gy * 2^(10 - T)
    // one chunk: expansion of `raw`, request of the chunk that will next live in `raw`, MFMAs.  false = the chunk needs a larger
    // scale (nothing was touched: rescale outside, then redo it)
    auto step = [&](DwRaw<NTO>& raw) -> bool {
            // raw chunk -> fragments.  x side first (bases of 8 rows of this lane's feature; cubic splines: two rows per
            // packed-fp32 evaluation; SiLU values), so that the gy loads have that arithmetic on top of the MFMA section
            // to arrive
            u32x4 rh[8], rl[8];            // per row: 8-slot windows (hi / lo) of this lane's feature
            u32x4 sah, sal;                // silu(x) * 2^4 over the 8 rows, hi / lo
            float xr[8];                   // this chunk's x values (XAFF: the folded BatchNorm1d applied; rows >= rend become the
                                           // shift -- their gy rows read as 0, the products vanish).  A copy: `raw` must stay as it
                                           // was loaded, the chunk is redone when its scale turns out too small
#pragma unroll
            for (int j = 0; j < 8; ++j) xr[j] = XAFF ? fmaf(raw.x[j], xa, xs) : raw.x[j];
            if constexpr (K == 3) {
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    if constexpr (HALF) make_spline_frag3_pair_h(xr[j], xr[j + 1], s_tbl, fgeo, rh[j], rh[j + 1], woff);
                    else make_spline_frag3_pair(xr[j], xr[j + 1], s_tbl, fgeo, rh[j], rl[j], rh[j + 1], rl[j + 1], woff);
                }
            } else
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (K == 0) {
                    const float z = ln_on ? fmaf((xr[j] - raw.mu[j]) * raw.rs[j], gam, bet) : xr[j];
                    make_rbf_frag(z, rb.a, ca, rh[j], rl[j]);
                } else {
                    spline_frag<K>(xr[j], s_knots, s_tbl, geom, fgeo, rh[j], rl[j], woff);
                }
            }
            // ---- SiLU branch: fp16 hi/lo at scale 2^4 (|silu| < 4094); larger values take the fp32 MFMA
            float sv[8];
            float smx = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {                 // packed fp32, two rows per instruction (bit-identical to siluf(x) * 16)
                const f32x2 pr = silu16_pair(f32x2{xr[j], xr[j + 1]});
                sv[j] = pr.x; sv[j + 1] = pr.y;
                smx = fmaxf(fmaxf(smx, fabsf(pr.x)), fabsf(pr.y));
            }
            const bool base32 = __any(!(smx < 60000.0f));   // wave-uniform; also catches NaN / Inf
            if constexpr (HALF) round_f16x2(sv, sah); else split_f16x2(sv, sah, sal);
            if (chunk_exp(raw) > T) return false;            // wave-uniform, rare: rescale outside, then redo this chunk
            if constexpr (K == 0) {
                if (bias_on) {
#pragma unroll
                    for (int t = 0; t < NTO; ++t)
#pragma unroll
                        for (int j = 0; j < 8; ++j) bsum[t] += raw.g[t][j];
                }
            }
            // gy side, scaled by 2^(10 - T)
            u32x4 bhi[NTO], blo[NTO];      // gy * 2^(10-T), per 16-wide output tile
            const float gs = ldexpf(1.0f, 10 - T);
#pragma unroll
            for (int t = 0; t < NTO; ++t) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j += 2) {             // (packed fp32: one v_pk_mul_f32 per two values)
                    const f32x2 pr = f32x2{raw.g[t][j], raw.g[t][j + 1]} * splat2(gs);
                    v[j] = pr.x; v[j + 1] = pr.y;
                }
                if constexpr (HALF) round_f16x2(v, bhi[t]); else split_f16x2(v, bhi[t], blo[t]);
            }
            if (base32) {                                    // rare: this chunk's base branch in exact fp32, at Dh's scale
                const float gs16 = gs * 16.0f;               // (4 rows per MFMA, k-lane kg <-> row 8*kg + j)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float sj = sv[j] * 0.0625f;
#pragma unroll
                    for (int t = 0; t < NTO; ++t)
                        Dh[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(sj, raw.g[t][j] * gs16, Dh[t], 0, 0, 0);
                }
            }
            load_raw(raw);                                   // next chunk: lands under the MFMAs below
#pragma unroll
            for (int c = 0; c < kCTmax - 1; ++c) {           // all 8 slots, no branch: slots >= C are always zero
                const int q = c >> 1;
                const unsigned sel = (c & 1) ? 0x07060302u : 0x05040100u;
                u32x4 ah, al;                                // transpose (row, coefficient) 8x8 blocks on the fly
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    ah[p] = __builtin_amdgcn_perm(rh[2 * p + 1][q], rh[2 * p][q], sel);
                    if constexpr (!HALF) al[p] = __builtin_amdgcn_perm(rl[2 * p + 1][q], rl[2 * p][q], sel);
                }
#pragma unroll
                for (int t = 0; t < NTO; ++t) D[c][t] = mfma16_f16(ah, bhi[t], D[c][t]);
                if constexpr (!HALF) {
#pragma unroll
                    for (int t = 0; t < NTO; ++t) D[c][t] = mfma16_f16(ah, blo[t], D[c][t]);
#pragma unroll
                    for (int t = 0; t < NTO; ++t) D[c][t] = mfma16_f16(al, bhi[t], D[c][t]);
                }
            }
            if (!base32) {
#pragma unroll
                for (int t = 0; t < NTO; ++t) Dh[t] = mfma16_f16(sah, bhi[t], Dh[t]);
                if constexpr (!HALF) {
#pragma unroll
                    for (int t = 0; t < NTO; ++t) Dh[t] = mfma16_f16(sah, blo[t], Dh[t]);
#pragma unroll
                    for (int t = 0; t < NTO; ++t) Dh[t] = mfma16_f16(sal, bhi[t], Dh[t]);
                }
            }
        return true;
    };
    auto rescale = [&](DwRaw<NTO>& raw) {                        // the pending chunk needs a larger scale: rescale once, exactly
        const int ex = chunk_exp(raw);
        const float dn = ldexpf(1.0f, T - ex);
#pragma unroll
        for (int c = 0; c < kCTmax; ++c)
#pragma unroll
            for (int t = 0; t < NTO; ++t) D[c][t] *= dn;
        T = ex;
    };
    long n0 = rbeg;
    DwRaw<NTO> raw;
    load_raw(raw);
    T = chunk_exp(raw);
    while (n0 < rend) {
        // ---- hot loop: the scale exponent T is FIXED in here, so the accumulators are only ever touched by MFMAs (a
        // conditional rescale inside the loop makes the compiler copy them around every iteration)
        for (; n0 < rend; n0 += 32)
            if (!step(raw)) break;
        if (n0 < rend) rescale(raw);
    }
    if constexpr (K == 0) {
        if (bias_on) {                                           // rows 8*kg + j of every chunk: fold the four k-lane groups
#pragma unroll
            for (int t = 0; t < NTO; ++t) {
                float v = bsum[t];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                const long o = 16 * NTO * oc + 16 * t + li;
                if (kg == 0 && o < outP) rb.colpart[s * outP + o] = v;
            }
        }
    }
    // ---- slab write: D rows <-> features 4*kg + reg, cols <-> outputs li
    const float undo = ldexpf(1.0f, T - 20), undo_b = ldexpf(1.0f, T - 14);
#pragma unroll
    for (int t = 0; t < NTO; ++t) {
        const long o = 16 * NTO * oc + 16 * t + li;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const long fl = 64 * fg + 16 * tile + 4 * kg + reg;
            if (fl < inP && o < outP) {
#pragma unroll
                for (int c = 0; c < kCTmax - 1; ++c)
                    if (c < C) slab[((s * (C + 1) + c) * inP + fl) * outP + o] = D[c][t][reg] * undo;
                slab[((s * (C + 1) + C) * inP + fl) * outP + o] = Dh[t][reg] * undo_b;
            }
        }
    }
}
#undef Dh

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient for WIDE layers (two or more 64-output chunks: FastKAN hidden 256 = BASELINE config 5, KAN layers of
// 128 / 256 outputs with <= 8 coefficients).  kan_split_dw_kernel gives every 64-output chunk a workgroup of its own, and
// each of them expands the bases of the same x rows again: at 256 outputs the expansion -- span / cubic pieces or
// LayerNorm + 8 exponentials, hi/lo split, (row, slot) transposition: ~3/4 of the kernel's VALU instructions -- ran four
// times per scalar (config 5 spent 23 % of its epoch here).  In this kernel the SH waves that own the SH output chunks of ONE
// 16-feature tile share the expansion through LDS: the rows go in groups of SH 32-row chunks; wave k of the team expands
// chunk k of the group ONCE, stores the ready A fragments (8 slot planes x hi/lo + the SiLU plane: 18 x 16 B per lane,
// already transposed, lane-contiguous: conflict-free ds_write_b128 / ds_read_b128), one workgroup barrier, and every
// wave runs the MFMAs of all SH chunks against its own gy columns.  Two fragment buffers alternate between groups, so one
// barrier per group is enough: a wave writes buffer b again only after the barrier of the group in between, which every
// wave enters after its reads of b.  The accumulators, the gy side, the running power-of-two scale, the slab layout and
// the summation order over rows are those of kan_split_dw_kernel: same slabs, bit for bit.
constexpr int kDwShVecs = 18;                                  // u32x4 per lane and chunk: 8 planes x (hi, lo) + silu (hi, lo)
constexpr int kDwShChunkBytes = kDwShVecs * 64 * 16;           // 18 KiB
constexpr int kDwShFlagOff = 2304;                             // free bytes of the LDS header: base32 flags [tile][buf][chunk]
constexpr size_t kDwShLds = kLdsHdr + 8 * (size_t)kDwShChunkBytes;   // (4/SH tiles) x 2 buffers x SH chunks = 8 chunk slots

template <int K, int SH>      // K: 0 (RBF) or 3 (cubic), <= 8 coefficients; SH: 2 or 4 output chunks per feature tile
__global__ __launch_bounds__(256, 1) void kan_split_dw_shared_kernel(
    const float* __restrict__ x, long ldx, const float* __restrict__ gy, long ldgy, long N, int in,
    int out, int C, const float* __restrict__ knots_g, int nknots, int OC, long rows_per_block,
    long inP, long outP, float* __restrict__ slab, RbfArgs rb) {
    constexpr int NTO = 4, TPW = 4 / SH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_knots = reinterpret_cast<float*>(smem);
    unsigned* s_tbl = reinterpret_cast<unsigned*>(smem + 256);
    int* s_flag = reinterpret_cast<int*>(smem + kDwShFlagOff);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (K > 0 && tid < nknots) s_knots[tid] = knots_g[tid];
    if (K > 0) build_perm_table(s_tbl, tid);
    __syncthreads();
    FastGeom fgeo{};
    const int li = lane & 15, kg = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int tl = wave_u / SH, ocl = wave_u % SH;               // team (feature tile of this workgroup) and role in it
    const int OCG = OC / SH;
    const int tile = (blockIdx.y / OCG) * TPW + tl;               // 16-feature tile
    const int oc = (blockIdx.y % OCG) * SH + ocl;                 // 64-output chunk
    const int f = 16 * tile + li;
    float ca[8] = {};
    if constexpr (K == 0) rbf_centers(rb, ca, 0);
    if constexpr (K > 0) fgeo = fast_geom(s_knots, nknots);
    const bool ln_on = (K == 0) && rb.ln_w != nullptr;           // wave-uniform
    const long s = blockIdx.x;
    const long rbeg = blockIdx.x * rows_per_block;
    const long rend = min(N, (blockIdx.x + 1L) * rows_per_block);
    const long nchunks = (rend - rbeg + 31) / 32, ngroups = (nchunks + SH - 1) / SH;

    f32x4 D[kCTmax][NTO];
#define Dh D[kCTmax - 1]
#pragma unroll
    for (int c = 0; c < kCTmax; ++c)
#pragma unroll
        for (int t = 0; t < NTO; ++t) D[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const GBuf xb = gbuf_at(x, rend, ldx, in, rbeg), gyb = gbuf_at(gy, rend, ldgy, out, rbeg);
    const GBuf stb = gbuf_at(rb.stats, ln_on ? rend : 0, 2, 2, rbeg);
    const float gam = ln_on ? rb.ln_w[min(f, in - 1)] : 1.0f, bet = ln_on ? rb.ln_b[min(f, in - 1)] : 0.0f;
    const unsigned ldx4 = (unsigned)ldx * 4u, ldgy4 = (unsigned)ldgy * 4u;
    // the x rows this wave EXPANDS: chunk ocl of every group; the gy rows it multiplies with: every chunk
    unsigned xo = (unsigned)(32 * ocl + 8 * kg) * ldx4 + (unsigned)min(f, in - 1) * 4u;
    unsigned sto = (unsigned)(32 * ocl + 8 * kg) * 8u;
    unsigned gvo[NTO];
#pragma unroll
    for (int t = 0; t < NTO; ++t) gvo[t] = (unsigned)(8 * kg) * ldgy4 + (unsigned)min(64 * oc + 16 * t + li, out - 1) * 4u;
    struct XRaw { float x[8], mu[8], rs[8]; };
    struct GRaw { float g[NTO][8]; };
    auto load_x = [&](XRaw& r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r.x[j] = gld_s(xb, xo, (unsigned)j * ldx4);
        if (ln_on) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { r.mu[j] = gld(stb, sto + 8u * j); r.rs[j] = gld(stb, sto + 8u * j + 4u); }
            sto += (unsigned)(32 * SH) * 8u;
        }
        xo += (unsigned)(32 * SH) * ldx4;
    };
    auto load_g = [&](GRaw& r) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int t = 0; t < NTO; ++t) r.g[t][j] = gld_s(gyb, gvo[t], (unsigned)j * ldgy4);
#pragma unroll
        for (int t = 0; t < NTO; ++t) gvo[t] += 32u * ldgy4;
    };
    auto chunk_exp = [&](const GRaw& r) -> int {
        float mx = 0.0f;
#pragma unroll
        for (int t = 0; t < NTO; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(r.g[t][j]));
        return exp_for_max(wave_max_nonneg(mx));
    };
    u32x4* s_frag = reinterpret_cast<u32x4*>(smem + kLdsHdr);
    auto slot = [&](int buf, int k) -> u32x4* { return s_frag + (size_t)((tl * 2 + buf) * SH + k) * (kDwShVecs * 64) + lane; };

    // ---- produce: this wave's chunk of the group -> ready A fragments in LDS.  Four rows at a time (two dwords of every
    // fragment vector, ds_write_b64): with all eight rows' windows live at once this phase set the kernel's register
    // high-water mark (the consumers' state -- next chunk's gy rows and split fragments -- stays live across it).
    auto produce = [&](const XRaw& raw, int buf) {
        unsigned char* dst = reinterpret_cast<unsigned char*>(slot(buf, ocl));
        float sv[8];
        float smx = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const f32x2 pr = silu16_pair(f32x2{raw.x[j], raw.x[j + 1]});
            sv[j] = pr.x; sv[j + 1] = pr.y;
            smx = fmaxf(fmaxf(smx, fabsf(pr.x)), fabsf(pr.y));
        }
        const bool base32 = __any(!(smx < 60000.0f));
        u32x4* dv = reinterpret_cast<u32x4*>(dst);
        if (base32) {             // rare: silu beyond fp16 range somewhere in the chunk -- hand over the fp32 values themselves
            dv[16 * 64] = u32x4{__float_as_uint(sv[0]), __float_as_uint(sv[1]), __float_as_uint(sv[2]), __float_as_uint(sv[3])};
            dv[17 * 64] = u32x4{__float_as_uint(sv[4]), __float_as_uint(sv[5]), __float_as_uint(sv[6]), __float_as_uint(sv[7])};
        } else {
            u32x4 sah, sal;
            split_f16x2(sv, sah, sal);
            dv[16 * 64] = sah;
            dv[17 * 64] = sal;
        }
        if (lane == 0) s_flag[(tl * 2 + buf) * SH + ocl] = base32 ? 1 : 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4 rh[4], rl[4];
            if constexpr (K == 3) {
#pragma unroll
                for (int j = 0; j < 4; j += 2)
                    make_spline_frag3_pair(raw.x[4 * h + j], raw.x[4 * h + j + 1], s_tbl, fgeo, rh[j], rl[j], rh[j + 1], rl[j + 1], 0u);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * h + j;
                    const float z = ln_on ? fmaf((raw.x[r] - raw.mu[r]) * raw.rs[r], gam, bet) : raw.x[r];
                    make_rbf_frag(z, rb.a, ca, rh[j], rl[j]);
                }
            }
#pragma unroll
            for (int c = 0; c < kCTmax - 1; ++c) {
                const int q = c >> 1;
                const unsigned sel = (c & 1) ? 0x07060302u : 0x05040100u;
                u32x2 ah, al;
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    ah[pp] = __builtin_amdgcn_perm(rh[2 * pp + 1][q], rh[2 * pp][q], sel);
                    al[pp] = __builtin_amdgcn_perm(rl[2 * pp + 1][q], rl[2 * pp][q], sel);
                }
                *reinterpret_cast<u32x2*>(dst + (size_t)(2 * c) * 1024 + 8 * h) = ah;
                *reinterpret_cast<u32x2*>(dst + (size_t)(2 * c + 1) * 1024 + 8 * h) = al;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    float bsum[NTO] = {};
    const bool bias_on = (K == 0) && rb.colpart != nullptr && tile == 0;      // wave-uniform: one wave per output chunk
    int T;
    // ---- consume: chunk k of the group against this wave's gy columns.  false = the chunk needs a larger scale
    auto consume = [&](GRaw& raw, int buf, int k) -> bool {
        // ALL 18 fragment vectors of the chunk are requested up front into registers of their own (72 of the ~130 this
        // one-wave-per-SIMD kernel has to spare): they land under the gy-side arithmetic below.  Left to itself the compiler
        // reads each slot plane into the same 8 registers right before its 12 MFMAs and waits out the LDS latency nine
        // times per chunk (740 -> us per launch at 169k x 256 -> 256, config 5).
        const u32x4* src = slot(buf, k);
        u32x4 fr[kDwShVecs];
#pragma unroll
        for (int v = 0; v < kDwShVecs; ++v) fr[v] = src[v * 64];
        __builtin_amdgcn_sched_barrier(0);
        if (chunk_exp(raw) > T) return false;
        if constexpr (K == 0) {
            if (bias_on) {
#pragma unroll
                for (int t = 0; t < NTO; ++t)
#pragma unroll
                    for (int j = 0; j < 8; ++j) bsum[t] += raw.g[t][j];
            }
        }
        u32x4 bhi[NTO], blo[NTO];
        const float gs = ldexpf(1.0f, 10 - T);
#pragma unroll
        for (int t = 0; t < NTO; ++t) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const f32x2 pr = f32x2{raw.g[t][j], raw.g[t][j + 1]} * splat2(gs);
                v[j] = pr.x; v[j + 1] = pr.y;
            }
            split_f16x2(v, bhi[t], blo[t]);
        }
        const bool base32 = __builtin_amdgcn_readfirstlane(s_flag[(tl * 2 + buf) * SH + k]) != 0;
        if (base32) {
            const float gs16 = gs * 16.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float sj = __uint_as_float(j < 4 ? fr[16][j & 3] : fr[17][j & 3]) * 0.0625f;
#pragma unroll
                for (int t = 0; t < NTO; ++t)
                    Dh[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(sj, raw.g[t][j] * gs16, Dh[t], 0, 0, 0);
            }
        }
        load_g(raw);                                     // next chunk's gy rows: they land under the MFMAs below
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < kCTmax - 1; ++c) {
#pragma unroll
            for (int t = 0; t < NTO; ++t) D[c][t] = mfma16_f16(fr[2 * c], bhi[t], D[c][t]);
#pragma unroll
            for (int t = 0; t < NTO; ++t) D[c][t] = mfma16_f16(fr[2 * c], blo[t], D[c][t]);
#pragma unroll
            for (int t = 0; t < NTO; ++t) D[c][t] = mfma16_f16(fr[2 * c + 1], bhi[t], D[c][t]);
        }
        if (!base32) {
#pragma unroll
            for (int t = 0; t < NTO; ++t) Dh[t] = mfma16_f16(fr[16], bhi[t], Dh[t]);
#pragma unroll
            for (int t = 0; t < NTO; ++t) Dh[t] = mfma16_f16(fr[16], blo[t], Dh[t]);
#pragma unroll
            for (int t = 0; t < NTO; ++t) Dh[t] = mfma16_f16(fr[17], bhi[t], Dh[t]);
        }
        return true;
    };
    auto rescale = [&](const GRaw& raw) {
        const int ex = chunk_exp(raw);
        const float dn = ldexpf(1.0f, T - ex);
#pragma unroll
        for (int c = 0; c < kCTmax; ++c)
#pragma unroll
            for (int t = 0; t < NTO; ++t) D[c][t] *= dn;
        T = ex;
    };

    XRaw xr;
    load_x(xr);
    GRaw gr;
    load_g(gr);
    T = chunk_exp(gr);
    long g = 0;
    int k = 0;
    bool fresh = true;             // the group's fragments have not been produced yet (wave-uniform, identical in every wave:
                                   // every wave passes exactly ONE barrier per group, whatever its rescales)
    while (g < ngroups) {
        bool ok = true;
        for (; g < ngroups; ++g) {                     // hot loop: T fixed, accumulators only touched by MFMAs
            const int buf = (int)(g & 1);
            if (fresh) {
                produce(xr, buf);
                load_x(xr);                            // next group's rows: in flight under this group's MFMAs
                __syncthreads();
                fresh = false;
            }
            for (; k < SH; ++k)
                if (!consume(gr, buf, k)) { ok = false; break; }
            if (!ok) break;
            k = 0;
            fresh = true;
        }
        if (!ok) rescale(gr);
    }
    if constexpr (K == 0) {
        if (bias_on) {
#pragma unroll
            for (int t = 0; t < NTO; ++t) {
                float v = bsum[t];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                const long o = 64 * oc + 16 * t + li;
                if (kg == 0 && o < outP) rb.colpart[s * outP + o] = v;
            }
        }
    }
    const float undo = ldexpf(1.0f, T - 20), undo_b = ldexpf(1.0f, T - 14);
#pragma unroll
    for (int t = 0; t < NTO; ++t) {
        const long o = 64 * oc + 16 * t + li;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const long fl = 16 * tile + 4 * kg + reg;
            if (fl < inP && o < outP) {
#pragma unroll
                for (int c = 0; c < kCTmax - 1; ++c)
                    if (c < C) slab[((s * (C + 1) + c) * inP + fl) * outP + o] = D[c][t][reg] * undo;
                slab[((s * (C + 1) + C) * inP + fl) * outP + o] = Dh[t][reg] * undo_b;
            }
        }
    }
}
#undef Dh

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient for cubic layers with 9..12 coefficients (grid 6..9; BASELINE config 3 is grid 8 => C = 11).
// kan_split_dw_kernel runs them as 2*in virtual features: every scalar's span / cubic pieces / hi-lo split / SiLU is
// evaluated twice (once per 8-slot window, on separate lanes) and 2 x 9 slot planes go through the matrix cores.  Here a
// lane owns ONE input feature: the 4-value payload of a row is computed once and dropped into window 0 (8 slots) AND
// into the low half of window 1 (its <= 4 live slots: two dwords per part); accumulators: 8 + NS1 spline planes + base
// = 224 registers for NS1 = 4.  Slabs are written in the plain [C+1][in][out] plane order, so the fused reduce / unpack
// kernel of the <= 8-coefficient path finishes the job.  Per 32 rows x 16 input features x 64 outputs:
// (8 + C - 8 + 1) * 12 MFMAs and ~850 VALU instead of 216 and ~1550.
template <int NS1, int RS = 1>      // RS: row sub-ranges per feature tile (narrow layers), as in kan_split_dw_kernel
__global__ __launch_bounds__(256) void kan_split_dw_w2_kernel(
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
    const FastGeom fgeo = fast_geom(s_knots, nknots);
    const int fg = blockIdx.y / OC, oc = blockIdx.y % OC;
    const int li = lane & 15, kg = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);        // uniform: the row range feeds the buffer descriptors
    const int tile = RS == 1 ? wave : wave_u % (4 / RS), rsub = RS == 1 ? 0 : wave_u / (4 / RS);
    const int f = 64 * fg + 16 * tile + li;
    constexpr int ns1 = NS1;                                     // live slots of the second window = C - 8, a template parameter:
                                                                 // behind a run-time `c < C - 8` the conditional MFMAs on D1 made
                                                                 // the compiler copy 64 accumulators around in every chunk
    const long s = (long)blockIdx.x * RS + rsub;       // slab of this (row block, sub-range)
    const long sub_rows = ((rows_per_block / 32 + RS - 1) / RS) * 32;
    const long rbeg = blockIdx.x * rows_per_block + rsub * sub_rows;
    const long rend = min(min(N, (blockIdx.x + 1L) * rows_per_block), rbeg + sub_rows);

    f32x4 D[kCTmax - 1][4];        // window 0: slots 0..7, scaled by 2^(20 - T)
    f32x4 D1[NS1 + 1][4];          // window 1: slots 8..8+NS1-1; plane NS1: the base weight (2^(14 - T)), fp16 hi/lo products and,
                                   // for chunks whose silu overflows fp16, exact fp32 MFMAs on gy at the same scale
    // (the base plane is a row of D1, not an array of its own: as `Dh[4]` the compiler kept it in arch VGPRs and copied all
    // 16 registers to the accumulation file and back around its MFMAs in every chunk)
#define Dh D1[NS1]
#pragma unroll
    for (int c = 0; c < kCTmax - 1; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) D[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NS1 + 1; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) D1[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const GBuf xb = gbuf_at(x, rend, ldx, in, rbeg), gyb = gbuf_at(gy, rend, ldgy, out, rbeg);
    const unsigned ldx4 = (unsigned)ldx * 4u, ldgy4 = (unsigned)ldgy * 4u;
    unsigned xo = (unsigned)(8 * kg) * ldx4 + (unsigned)min(f, in - 1) * 4u, gvo[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) gvo[t] = (unsigned)(8 * kg) * ldgy4 + (unsigned)min(64 * oc + 16 * t + li, out - 1) * 4u;
    struct Raw { float x[8]; float g[4][8]; };
    auto load_raw = [&](Raw& r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            r.x[j] = gld_s(xb, xo, (unsigned)j * ldx4);
#pragma unroll
            for (int t = 0; t < 4; ++t) r.g[t][j] = gld_s(gyb, gvo[t], (unsigned)j * ldgy4);
        }
        xo += 32u * ldx4;
#pragma unroll
        for (int t = 0; t < 4; ++t) gvo[t] += 32u * ldgy4;
    };
    auto chunk_exp = [&](const Raw& r) -> int {
        float mx = 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(r.g[t][j]));
        return exp_for_max(wave_max_nonneg(mx));
    };

    // rotated loop, as in kan_split_dw_kernel: the fragments live inside one iteration, only the raw rows cross the back
    // edge.  (With the expansion at the bottom of the loop this kernel -- 208 accumulators -- spilled 652 VGPRs: 1 452 bytes
    // of scratch per lane, ~470 scratch loads / stores in the loop; profiles/r03_kernel_resources.txt)
    Raw raw;
    load_raw(raw);
    int T = chunk_exp(raw);
    long n0 = rbeg;
    while (n0 < rend) {
        for (; n0 < rend; n0 += 32) {
            u32x4 rh[8], rl[8];            // per row: window 0, 8 slots (hi / lo)
            unsigned r1h[8][2], r1l[8][2]; // per row: window 1, slots 0..3
            u32x4 bhi[4], blo[4];
            u32x4 sah, sal;
            // ---- one evaluation of the cubic pieces per row, two placements
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int m; float u; bool inside;
                fast_span(raw.x[j], fgeo, m, u, inside);
                float Nv[4];
                cubic_bases(u, inside ? (kAScale / 6.0f) : 0.0f, Nv);
                const unsigned h0 = pk_f16_rtz(Nv[0], Nv[1]), h1 = pk_f16_rtz(Nv[2], Nv[3]);
                const unsigned l0 = pk_f16_rtz(sub_f16lo(Nv[0], h0), sub_f16hi(Nv[1], h0));
                const unsigned l1 = pk_f16_rtz(sub_f16lo(Nv[2], h1), sub_f16hi(Nv[3], h1));
                const unsigned char* te = reinterpret_cast<const unsigned char*>(s_tbl) + 16 * (m + 1);
                const u32x4 sel0 = *reinterpret_cast<const u32x4*>(te);
                frag3_place_fwd(sel0, h0, h1, l0, l1, rh[j], rl[j]);
                const uint2 sel1 = *reinterpret_cast<const uint2*>(te + kWinBytes);          // window 1: slots 0..3 only
                r1h[j][0] = __builtin_amdgcn_perm(h1, h0, sel1.x); r1h[j][1] = __builtin_amdgcn_perm(h1, h0, sel1.y);
                r1l[j][0] = __builtin_amdgcn_perm(l1, l0, sel1.x); r1l[j][1] = __builtin_amdgcn_perm(l1, l0, sel1.y);
                // fence the scheduler every two rows: left alone it hoists all sixteen selector reads and the eight rows'
                // cubic pieces to the top (it budgets 512 registers, but everything a VALU instruction writes must sit in
                // the 256 architectural ones next to 160..208 accumulators) and the kernel spills
                if (j & 1) __builtin_amdgcn_sched_barrier(0);
            }
            float sv[8];
            float smx = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {                 // packed fp32, two rows per instruction (bit-identical to siluf(x) * 16)
                const f32x2 pr = silu16_pair(f32x2{raw.x[j], raw.x[j + 1]});
                sv[j] = pr.x; sv[j + 1] = pr.y;
                smx = fmaxf(fmaxf(smx, fabsf(pr.x)), fabsf(pr.y));
            }
            const bool base32 = __any(!(smx < 60000.0f));
            split_f16x2(sv, sah, sal);
            if (chunk_exp(raw) > T) break;                   // wave-uniform, rare: rescale outside, then redo this chunk
            const float gs = ldexpf(1.0f, 10 - T);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = raw.g[t][j] * gs;
                split_f16x2(v, bhi[t], blo[t]);
            }
            if (base32) {
                const float gs16 = gs * 16.0f;
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        Dh[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(sv[j] * 0.0625f, raw.g[t][j] * gs16, Dh[t], 0, 0, 0);
            }
            load_raw(raw);                                   // next chunk: lands under the MFMAs below
#pragma unroll
            for (int c = 0; c < kCTmax - 1; ++c) {
                const int q = c >> 1;
                const unsigned sel = (c & 1) ? 0x07060302u : 0x05040100u;
                u32x4 ah, al;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    ah[p] = __builtin_amdgcn_perm(rh[2 * p + 1][q], rh[2 * p][q], sel);
                    al[p] = __builtin_amdgcn_perm(rl[2 * p + 1][q], rl[2 * p][q], sel);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) D[c][t] = mfma16_f16(ah, bhi[t], D[c][t]);
#pragma unroll
                for (int t = 0; t < 4; ++t) D[c][t] = mfma16_f16(ah, blo[t], D[c][t]);
#pragma unroll
                for (int t = 0; t < 4; ++t) D[c][t] = mfma16_f16(al, bhi[t], D[c][t]);
            }
#pragma unroll
            for (int c = 0; c < NS1; ++c) {
                if (c < ns1) {                                   // wave-uniform
                    const int q = c >> 1;
                    const unsigned sel = (c & 1) ? 0x07060302u : 0x05040100u;
                    u32x4 ah, al;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        ah[p] = __builtin_amdgcn_perm(r1h[2 * p + 1][q], r1h[2 * p][q], sel);
                        al[p] = __builtin_amdgcn_perm(r1l[2 * p + 1][q], r1l[2 * p][q], sel);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) D1[c][t] = mfma16_f16(ah, bhi[t], D1[c][t]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) D1[c][t] = mfma16_f16(ah, blo[t], D1[c][t]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) D1[c][t] = mfma16_f16(al, bhi[t], D1[c][t]);
                }
            }
            if (!base32) {
#pragma unroll
                for (int t = 0; t < 4; ++t) Dh[t] = mfma16_f16(sah, bhi[t], Dh[t]);
#pragma unroll
                for (int t = 0; t < 4; ++t) Dh[t] = mfma16_f16(sah, blo[t], Dh[t]);
#pragma unroll
                for (int t = 0; t < 4; ++t) Dh[t] = mfma16_f16(sal, bhi[t], Dh[t]);
            }
        }
        if (n0 < rend) {
            const int ex = chunk_exp(raw);
            const float dn = ldexpf(1.0f, T - ex);
#pragma unroll
            for (int c = 0; c < kCTmax - 1; ++c)
#pragma unroll
                for (int t = 0; t < 4; ++t) D[c][t] *= dn;
#pragma unroll
            for (int c = 0; c < NS1 + 1; ++c)
#pragma unroll
                for (int t = 0; t < 4; ++t) D1[c][t] *= dn;
            T = ex;
        }
    }
    const float undo = ldexpf(1.0f, T - 20), undo_b = ldexpf(1.0f, T - 14);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const long o = 64 * oc + 16 * t + li;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const long fl = 64 * fg + 16 * tile + 4 * kg + reg;
            if (fl < inP && o < outP) {
#pragma unroll
                for (int c = 0; c < kCTmax - 1; ++c)
                    slab[((s * (C + 1) + c) * inP + fl) * outP + o] = D[c][t][reg] * undo;
#pragma unroll
                for (int c = 0; c < NS1; ++c)
                    if (c < ns1) slab[((s * (C + 1) + 8 + c) * inP + fl) * outP + o] = D1[c][t][reg] * undo;
                slab[((s * (C + 1) + C) * inP + fl) * outP + o] = Dh[t][reg] * undo_b;
            }
        }
    }
}
#undef Dh

// slab reduction and unpack in one launch (the non-virtual layout): thread (o, plane c) of workgroup (o-tile, f)
// sums slab[.][c][f][o] in a fixed order, writes g_spline_weight (chain rule through spline_scaler), and the
// planes of one (f, o) meet in LDS for g_spline_scaler = sum_c gW * spline_weight (fixed order, deterministic)
__device__ __forceinline__ void dw_reduce_unpack_body(const float* __restrict__ slab, long NS, int in, int out, int C,
                                                      long inP, long outP, const float* __restrict__ sw,
                                                      const float* __restrict__ sc, float* __restrict__ g_bw,
                                                      float* __restrict__ g_sw, float* __restrict__ g_sc, int SG, int bx, int by) {
    // thread = (output ol of 32, plane c of C+1, slab group sg of SG): the slab sum is split over SG groups (shorter
    // dependent load chains), combined through LDS in group order -- still a fixed summation order
    extern __shared__ float s_red[];                 // [SG][C+1][32] partial sums, then [C][32] products
    const int ol = threadIdx.x & 31, pc = threadIdx.x >> 5;
    const int c = pc % (C + 1), sg = pc / (C + 1);
    const int f = by, o = bx * 32 + ol;
    const long per = (long)(C + 1) * inP * outP;
    const long i = ((long)c * inP + f) * outP + o;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    long s = sg;
    for (; s + 3 * SG < NS; s += 4 * SG) {
        a0 += slab[(s + 0 * SG) * per + i];
        a1 += slab[(s + 1 * SG) * per + i];
        a2 += slab[(s + 2 * SG) * per + i];
        a3 += slab[(s + 3 * SG) * per + i];
    }
    for (; s < NS; s += SG) a0 += slab[s * per + i];
    s_red[(sg * (C + 1) + c) * 32 + ol] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    float g = 0.0f;
    if (sg == 0)
        for (int q = 0; q < SG; ++q) g += s_red[(q * (C + 1) + c) * 32 + ol];
    __syncthreads();
    const bool live = o < out;
    const long of = (long)min(o, out - 1) * in + f;
    if (sg == 0) {
        if (c < C) {
            const float scale = sc ? sc[of] : 1.0f;
            if (live) g_sw[of * C + c] = g * scale;
            s_red[c * 32 + ol] = g * sw[of * C + c];
        } else if (live && g_bw) {
            g_bw[of] = g;
        }
    }
    __syncthreads();
    if (sg == 0 && c == 0 && live && g_sc) {
        float gs = 0.0f;
        for (int k = 0; k < C; ++k) gs += s_red[k * 32 + ol];
        g_sc[of] = gs;
    }
}

__global__ void kan_dw_reduce_unpack_kernel(const float* __restrict__ slab, long NS, int in, int out, int C,
                                            long inP, long outP, const float* __restrict__ sw,
                                            const float* __restrict__ sc, float* __restrict__ g_bw,
                                            float* __restrict__ g_sw, float* __restrict__ g_sc, int SG) {
    dw_reduce_unpack_body(slab, NS, in, out, C, inP, outP, sw, sc, g_bw, g_sw, g_sc, SG, blockIdx.x, blockIdx.y);
}

// the same for up to kDwDeferMax layers in one launch (DwDefer, common.h): blockIdx.z = the layer; all of them have the same
// C and SG (one block shape), the grid covers the largest (outP / 32, in) and a workgroup outside its layer's range leaves
struct DwReduceBatch { DwReduceItem item[kDwDeferMax]; };
__global__ void kan_dw_reduce_unpack_batch_kernel(const DwReduceBatch b) {
    const DwReduceItem& it = b.item[blockIdx.z];
    if ((long)blockIdx.x * 32 >= it.outP || (int)blockIdx.y >= it.in) return;
    dw_reduce_unpack_body(it.slab, it.NS, it.in, it.out, it.C, it.inP, it.outP, it.sw, it.sc, it.g_bw, it.g_sw, it.g_sc, it.SG, blockIdx.x,
                          blockIdx.y);
}

int dw_defer_flush(hipStream_t st) {
    DwDefer* d = g_dw_defer;
    if (d == nullptr || d->n == 0) return KAGNN_OK;
    DwReduceBatch b{};
    unsigned gx = 1, gy = 1;
    for (int k = 0; k < d->n; ++k) {
        b.item[k] = d->item[k];
        gx = max(gx, (unsigned)(d->item[k].outP / 32));
        gy = max(gy, (unsigned)d->item[k].in);
    }
    const int C = d->item[0].C, SG = d->item[0].SG;
    kan_dw_reduce_unpack_batch_kernel<<<dim3(gx, gy, (unsigned)d->n), 32 * (C + 1) * SG, (size_t)SG * (C + 1) * 32 * sizeof(float), st>>>(b);
    d->n = 0;
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// the slab reduction + unpack of one layer: launched now, or recorded when the slab lives in the caller's arena (DwDefer)
static int dw_reduce_unpack(const float* slab, long NS, int in, int out, int C, long inP, long outP, const float* sw, const float* sc,
                            float* g_bw, float* g_sw, float* g_sc, hipStream_t st) {
    const int SG = max(1, min(3, 1024 / (32 * (C + 1))));
    DwDefer* d = g_dw_defer;
    const unsigned char* sp = reinterpret_cast<const unsigned char*>(slab);
    if (d != nullptr && sp >= d->arena && sp < d->arena + d->arena_bytes) {
        if (d->n > 0 && (d->item[0].C != C || d->n == kDwDeferMax)) { int rc = dw_defer_flush(st); if (rc) return rc; }
        d->item[d->n++] = DwReduceItem{slab, NS, in, out, C, SG, inP, outP, sw, sc, g_bw, g_sw, g_sc};
        return KAGNN_OK;
    }
    kan_dw_reduce_unpack_kernel<<<dim3((unsigned)(outP / 32), (unsigned)in), 32 * (C + 1) * SG, (size_t)SG * (C + 1) * 32 * sizeof(float), st>>>(
        slab, NS, in, out, C, inP, outP, sw, sc, g_bw, g_sw, g_sc, SG);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// virtual-feature layout (sh == 1) back to the parameter layout, with the spline_scaler chain rule of
// kan_dw_unpack: coefficient c of input feature f lives in plane c & 7 of virtual feature 2f + (c >> 3)
__global__ void kan_dw_unpack_v_kernel(const float* __restrict__ gcat, int in, int out, int C, long inP,
                                       long outP, const float* __restrict__ sw, const float* __restrict__ sc,
                                       float* __restrict__ g_bw, float* __restrict__ g_sw,
                                       float* __restrict__ g_sc) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)in * out) return;
    const int o = i % out, f = i / out;
    const long of = (long)o * in + f;
    float gs = 0.0f;
    const float scale = sc ? sc[of] : 1.0f;
    for (int c = 0; c < C; ++c) {
        const float g = gcat[((long)(c & 7) * inP + 2 * f + (c >> 3)) * outP + o];
        g_sw[of * C + c] = g * scale;
        gs = fmaf(g, sw[of * C + c], gs);
    }
    if (g_sc) g_sc[of] = gs;
    if (g_bw) g_bw[of] = gcat[(8L * inP + 2 * f) * outP + o];
}

// row slabs and padded output width of the weight-gradient launch: the shape of RbfArgs::colpart ([slabs][outP])
void kan_split_dw_slabs(long N, int in, int out, int C, int K, long* slabs, long* outP) {
    const DwPlan p = split_dw_plan(N, in, out, C, K);
    *slabs = p.NS; *outP = p.outP;
}

// K == 0: Gaussian RBF basis with G = num_grids; sc == nullptr and g_sw laid out [out][in][G] either way
int kan_split_dw_any(const float* x, long ldx, const float* gy, long ldgy, long N, const float* knots, int in,
                     int out, int G, int K, const float* sw, const float* sc, float* g_bw, float* g_sw,
                     float* g_sc, float* ws, size_t ws_bytes, const RbfArgs& rb, hipStream_t st) {
    const int C = G + K, nk = K ? G + 2 * K + 1 : 0;
    if (kan_dw_w2_ok(in, out, C, K) && rb.centers == nullptr) {
        const DwPlan p = split_dw_plan_w2(N, in, out, C);
        if (ws_bytes < (size_t)(p.NS + 1) * p.per * sizeof(float)) return fail(KAGNN_ERR_ARG, "%s: workspace too small", "kan_split_dw");
        float* slab = ws + p.per;
#define W2(NN, RR) kan_split_dw_w2_kernel<NN, RR><<<dim3(p.nbx, p.FG * p.OC), 256, 0, st>>>(x, ldx, gy, ldgy, N, in, out, C, knots, nk, p.OC, \
                                                                                     p.rpw, p.inP, p.outP, slab)
#define W2N(NN) if (p.rs == 4) W2(NN, 4); else if (p.rs == 2) W2(NN, 2); else W2(NN, 1)
        switch (C - 8) { case 1: W2N(1); break; case 2: W2N(2); break; case 3: W2N(3); break; default: W2N(4); break; }
#undef W2N
#undef W2
        KAGNN_LAUNCH_CHECK();
        return dw_reduce_unpack(slab, p.NS, in, out, C, p.inP, p.outP, sw, sc, g_bw, g_sw, g_sc, st);
    }
    const int sh = C > 8 ? 1 : 0, Ck = sh ? 8 : C;          // slots per (virtual) feature the kernel stores
    const DwPlan p = split_dw_plan(N, in, out, C, K);
    if (ws_bytes < (size_t)(p.NS + 1) * p.per * sizeof(float)) return fail(KAGNN_ERR_ARG, "%s: workspace too small", "kan_split_dw");
    float* gcat = ws;
    float* slab = ws + p.per;
    dim3 grid(p.nbx, p.FG * p.OC);
#define ARGS x, ldx, gy, ldgy, N, in, out, Ck, knots, nk, p.OC, p.rpw, p.inP, p.outP, slab, rb
#define L(KK) if (sh) kan_split_dw_kernel<KK, true><<<grid, 256, 0, st>>>(ARGS, sh); \
              else kan_split_dw_kernel<KK, false><<<grid, 256, 0, st>>>(ARGS, 0)
    // narrow OUTPUTS (read-out layers: 40 classes, 1 regression target): only the 16-wide output tiles that exist -- at 64 the
    // wave split 64 - out columns of gy and ran their MFMAs for nothing (out = 40: 3 tiles instead of 4)
    const int nto = (!sh && out <= 48) ? cdiv(out, 16) : 4;
    // wide layers: the SH waves that own the output chunks of one feature tile share its basis expansion through LDS
    // (kan_split_dw_shared_kernel: same slabs, bit for bit; KAGNN_DW_SHARED=0 keeps one workgroup per output chunk for A/B)
    const char* dw_env = getenv("KAGNN_DW_SHARED");           // (read per call: the bitwise A/B test flips it inside one process)
    const bool dw_shared = dw_env == nullptr || atoi(dw_env) != 0;
    if (dw_shared && !(g_half_products && K == 3) && !rb.x_affine && !sh && (K == 0 || K == 3) && p.rs == 1 && p.OC >= 2 && p.OC % 2 == 0 && in > 32) {
        const int SHn = p.OC % 4 == 0 ? 4 : 2;
#define LS(KK, SS) do { \
            static unsigned long long seen_##KK##_##SS = 0; \
            if (auto first_use_ = first_use_on_this_device(seen_##KK##_##SS)) \
                KAGNN_HIP(hipFuncSetAttribute((const void*)kan_split_dw_shared_kernel<KK, SS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDwShLds)); \
            kan_split_dw_shared_kernel<KK, SS><<<grid, 256, kDwShLds, st>>>(x, ldx, gy, ldgy, N, in, out, Ck, knots, nk, p.OC, p.rpw, p.inP, p.outP, slab, rb); } while (0)
        if (K == 0) { if (SHn == 4) LS(0, 4); else LS(0, 2); }
        else        { if (SHn == 4) LS(3, 4); else LS(3, 2); }
#undef LS
        KAGNN_LAUNCH_CHECK();
        return dw_reduce_unpack(slab, p.NS, in, out, C, p.inP, p.outP, sw, sc, g_bw, g_sw, g_sc, st);
    }
    // single-product mode (thread-local, set by the entry point): the cubic, <= 8-coefficient instantiations
    const bool half = g_half_products && K == 3 && !sh;
#define LH(RR, NN, XX) kan_split_dw_kernel<3, false, RR, NN, XX, true><<<grid, 256, 0, st>>>(ARGS, 0)
#define LN(KK) if (p.rs == 4) kan_split_dw_kernel<KK, false, 4><<<grid, 256, 0, st>>>(ARGS, 0); \
               else if (p.rs == 2) kan_split_dw_kernel<KK, false, 2><<<grid, 256, 0, st>>>(ARGS, 0); \
               else if (nto == 3) kan_split_dw_kernel<KK, false, 1, 3><<<grid, 256, 0, st>>>(ARGS, 0); \
               else if (nto == 2) kan_split_dw_kernel<KK, false, 1, 2><<<grid, 256, 0, st>>>(ARGS, 0); \
               else if (nto == 1) kan_split_dw_kernel<KK, false, 1, 1><<<grid, 256, 0, st>>>(ARGS, 0); \
               else { L(KK); }
    if (rb.x_affine) {                                   // the input is a folded BatchNorm1d output (read-out of the node models)
        if (K != 3 || sh || p.rs != 1) return fail(KAGNN_ERR_UNSUPPORTED, "%s: an input affine needs a cubic layer with <= 8 coefficients and more than 32 inputs", "kan_split_dw");
        if (half) { if (nto == 3) LH(1, 3, true); else if (nto == 2) LH(1, 2, true); else if (nto == 1) LH(1, 1, true); else LH(1, 4, true); }
        else if (nto == 3) kan_split_dw_kernel<3, false, 1, 3, true><<<grid, 256, 0, st>>>(ARGS, 0);
        else if (nto == 2) kan_split_dw_kernel<3, false, 1, 2, true><<<grid, 256, 0, st>>>(ARGS, 0);
        else if (nto == 1) kan_split_dw_kernel<3, false, 1, 1, true><<<grid, 256, 0, st>>>(ARGS, 0);
        else kan_split_dw_kernel<3, false, 1, 4, true><<<grid, 256, 0, st>>>(ARGS, 0);
    } else if (half) {
        if (p.rs == 4) LH(4, 4, false); else if (p.rs == 2) LH(2, 4, false);
        else if (nto == 3) LH(1, 3, false); else if (nto == 2) LH(1, 2, false); else if (nto == 1) LH(1, 1, false); else LH(1, 4, false);
    } else
    switch (K) {
        case 0: LN(0); break;
        case 1: L(1); break;
        case 2: L(2); break;
        case 3: LN(3); break;
        case 4: L(4); break;
        default: return fail(KAGNN_ERR_UNSUPPORTED, "%s: spline_order must be 1..4", "kan_split_dw");
    }
#undef L
#undef LN
#undef LH
#undef ARGS
    KAGNN_LAUNCH_CHECK();
    if (!sh) {
        return dw_reduce_unpack(slab, p.NS, in, out, C, p.inP, p.outP, sw, sc, g_bw, g_sw, g_sc, st);
    }
    { int rc = kan_dw_reduce(slab, p.NS, p.per, gcat, st); if (rc) return rc; }
    if (sh) {
        kan_dw_unpack_v_kernel<<<cdiv((long)in * out, 256), 256, 0, st>>>(gcat, in, out, C, p.inP, p.outP, sw, sc, g_bw, g_sw, g_sc);
        KAGNN_LAUNCH_CHECK();
        return KAGNN_OK;
    }
    return kan_dw_unpack(gcat, in, out, C, p.inP, p.outP, sw, sc, g_bw, g_sw, g_sc, st);
}

int kan_split_dw(const float* x, long ldx, const float* gy, long ldgy, long N, const float* knots, int in,
                 int out, int G, int K, const float* sw, const float* sc, float* g_bw, float* g_sw,
                 float* g_sc, float* ws, size_t ws_bytes, hipStream_t st, const float* x_affine) {
    RbfArgs rb{};
    rb.x_affine = x_affine;
    if (x_affine && (kan_dw_w2_ok(in, out, G + K, K) || out > 64))
        return fail(KAGNN_ERR_UNSUPPORTED, "%s: an input affine needs a cubic layer with <= 8 coefficients and <= 64 outputs", "kan_split_dw");
    return kan_split_dw_any(x, ldx, gy, ldgy, N, knots, in, out, G, K, sw, sc, g_bw, g_sw, g_sc, ws, ws_bytes, rb, st);
}

}  // namespace kagnn
