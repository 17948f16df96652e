// efficient-KAN layer, split-precision mode (KAGNN_PREC_SPLIT): the dense contraction runs on
// the 16-bit matrix cores at fp32-equivalent accuracy.
//
//   a*w ~= a_hi*w_hi + a_hi*w_lo + a_lo*w_hi          (a_hi = fp16(a), a_lo = fp16(a - a_hi))
//
// three v_mfma_f32_32x32x16_f16 into ONE fp32 accumulator.  Both operands are pre-scaled by exact
// powers of two (bases by 2^10, weights so that max|W| lands in [2^9, 2^10)) so hi and lo stay in
// fp16's normal range; the dropped a_lo*w_lo term is ~2^-22 relative -- the same class as fp32
// rounding (measured against an fp64 oracle in tests/test_gpu_parity.py).  That is 3/16 of the
// cost of v_mfma_f32_32x32x2_f32 (kan_fp32.hip), which is what lets the layer approach the HBM
// roofline instead of the fp32 matrix peak.  The SiLU base branch has unbounded
// inputs, so it uses a 3-way bf16 split (6 v_mfma_f32_32x32x16_bf16, ~2^-24) in the same
// accumulator.
//
// Fragment mapping (fwd): lane l of a wave owns row (l&31) of a 32-row tile and, per MFMA, the 8
// consecutive k-slots 8*(l>>5)..+7 -- we make those 8 slots the (up to) 8 spline coefficients of
// ONE input feature, so a lane evaluates the K+1 non-zero B-spline bases of one scalar x[row,f]
// in registers, splits them, drops them into place with v_perm_b32 and feeds the matrix core.
// The packed weights live in LDS for the whole (persistent) workgroup when they fit (152 KB at
// in=out=64, G+k=8), otherwise they stream through LDS per 64-feature chunk.
//
// Reference behaviour replaced: node_classification_clean/ekan.py:79-112,146-162.
#include "split_common.h"

namespace kagnn {

__host__ __device__ inline int split_cf(int OT) { return OT <= 2 ? 64 : 32; }   // features per chunk
__host__ __device__ inline size_t split_fwd_chunk_bytes(int OT) {
    const int CF = split_cf(OT);
    return (size_t)(CF / 2) * OT * 2 * 1024 + (size_t)(CF / 16) * OT * 3 * 1024;
}

// K == 0 denotes the Gaussian RBF basis of FastKAN (G = num_grids slots)
bool kan_split_fwd_ok(int in, int out, int G, int K) {
    return K >= 0 && K <= 4 && G + K <= 16;
}
static inline int vshift(int C) { return C > 8 ? 1 : 0; }     // 9..16 coefficients: two 8-slot windows per feature

// Outputs are processed in blocks of <= 128 columns (4 accumulator tiles per wave); every block has its
// own pack (own power-of-two weight scale) at a fixed stride.
static size_t fwd_blk_bytes(int in, int ob, int C) {
    const int OT = cdiv(ob, 32), CF = split_cf(OT);
    return kHdrBytes + (size_t)cdiv(in << vshift(C), CF) * split_fwd_chunk_bytes(OT);
}
size_t kan_split_pack_fwd_bytes(int in, int out, int C) {
    return (size_t)cdiv(out, kOutBlk) * fwd_blk_bytes(in, min(out, kOutBlk), C);
}

// ------------------------------------------------------------------ packing
__global__ void split_pack_fwd_kernel(const float* __restrict__ bw, const float* __restrict__ sw,
                                      const float* __restrict__ sc, int in, int out, int C,
                                      unsigned char* __restrict__ pack, int self_scale) {
    const int sh = C > 8 ? 1 : 0, inv = in << sh;          // virtual features (see wcat_v)
    const int OT = cdiv(out, 32), CF = split_cf(OT), HF = CF / 2;
    const int SPC = CF / 2, BPC = CF / 16;
    unsigned* hdr = reinterpret_cast<unsigned*>(pack);
    __shared__ float s_m[17];
    const float wmax = self_scale == 2 ? header_absmax(pack) : self_scale ? block_absmax_w(bw, sw, sc, in, out, C, s_m) : __uint_as_float(hdr[2]);
    const int e = scale_exp_from_max(wmax);
    const float wscale = ldexpf(1.0f, -e);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        reinterpret_cast<float*>(pack)[0] = ldexpf(1.0f, e - 10);   // post scale: undo 2^10 and 2^-e
        reinterpret_cast<int*>(pack)[1] = e;
    }
    const size_t chunk_bytes = split_fwd_chunk_bytes(OT);
    const long spl_per_chunk = (long)SPC * OT * 64, base_per_chunk = (long)BPC * OT * 64;
    const long per_chunk = spl_per_chunk + base_per_chunk;
    const long total = (long)cdiv(inv, CF) * per_chunk;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = i / per_chunk; long r = i % per_chunk;
        unsigned char* cbase = pack + kHdrBytes + (size_t)ch * chunk_bytes;
        if (r < spl_per_chunk) {
            const int lane = r & 63; r >>= 6;
            const int ot = r % OT; const int s = r / OT;
            const int o = 32 * ot + (lane & 31), f = ch * CF + (lane >> 5) * HF + s;
            _Float16 hi[8], lo[8];
            for (int j = 0; j < 8; ++j) {
                const float w = wcat_v(bw, sw, sc, in, out, C, o, f, j, sh) * wscale;
                hi[j] = (_Float16)w;
                lo[j] = (_Float16)(w - (float)hi[j]);
            }
            _Float16* dh = reinterpret_cast<_Float16*>(cbase + ((size_t)(s * OT + ot) * 2 + 0) * 1024 + lane * 16);
            _Float16* dl = reinterpret_cast<_Float16*>(cbase + ((size_t)(s * OT + ot) * 2 + 1) * 1024 + lane * 16);
            for (int j = 0; j < 8; ++j) { dh[j] = hi[j]; dl[j] = lo[j]; }
        } else {
            r -= spl_per_chunk;
            const int lane = r & 63; r >>= 6;
            const int ot = r % OT; const int sb = r / OT;
            const int o = 32 * ot + (lane & 31);
            unsigned char* bb = cbase + (size_t)SPC * OT * 2 * 1024 + (size_t)(sb * OT + ot) * 3 * 1024 + lane * 16;
            for (int j = 0; j < 8; ++j) {
                const int f = ch * CF + (lane >> 5) * HF + 8 * sb + j;
                float w = wcat_v(bw, sw, sc, in, out, C, o, f, 8, sh) * wscale;
                for (int p = 0; p < 3; ++p) {              // truncating bf16 split: w = w1 + w2 + w3 (+2^-24)
                    const unsigned bits = __float_as_uint(w) & 0xffff0000u;
                    reinterpret_cast<unsigned short*>(bb + p * 1024)[j] = (unsigned short)(bits >> 16);
                    w -= __uint_as_float(bits);
                }
            }
        }
    }
}

// ------------------------------------------------------------------ forward
// Workgroup = 1024 threads = 16 waves (4 per SIMD, <= 128 VGPRs each) so that LDS / VALU latencies of
// one wave hide under the other three; one wave = 32 rows.  x is consumed in groups of 8 features per
// lane (two float4), the next group is prefetched while the current one is expanded.
// (cubic splines with <= 8 coefficients take kan_sparse_fwd.hip instead)
template <int K, int OT, int NT>
__global__ __launch_bounds__(NT) void kan_split_fwd_kernel(
    const float* __restrict__ x, long ldx, long N, int in, const float* __restrict__ knots_g,
    int nknots, const unsigned char* __restrict__ pack, int nchunks, float* __restrict__ y, long ldy,
    int out, RbfArgs rb, int sh /* 1: two 8-slot windows per input feature (virtual features, see wcat_v) */,
    int chunks_per_split /* split-K for few-row inputs: blockIdx.y owns this many chunks and writes a partial y */,
    long part_stride /* elements between the partial outputs of consecutive splits (0: no split) */) {
    const int split = (int)blockIdx.y;
    constexpr int CF = (OT <= 2) ? 64 : 32, HF = CF / 2, SPC = CF / 2, BPC = CF / 16;
    constexpr int CHUNK_BYTES = SPC * OT * 2 * 1024 + BPC * OT * 3 * 1024;
    constexpr int NG = HF / 8;                       // groups of 8 features per lane-half and chunk
    constexpr int ROWS = (NT / 64) * 32;             // rows per workgroup iteration
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_knots = reinterpret_cast<float*>(smem);
    unsigned* s_tbl = reinterpret_cast<unsigned*>(smem + 256);
    unsigned char* s_w = smem + kLdsHdr;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (K > 0 && tid < nknots) s_knots[tid] = knots_g[tid];
    if (K == 3) build_perm_table3(s_tbl, tid, nknots); else if (K > 0) build_perm_table(s_tbl, tid);
    if (K == 4) build_perm_fix_table(s_tbl, tid);
    const float post = reinterpret_cast<const float*>(pack)[0];
    const unsigned char* gw = pack + kHdrBytes;
    // Packed W lives in LDS as two HALVES of a chunk (NG / 2 feature groups each: their MFMA steps + SiLU fragments),
    // filled by LDS-DMA (lds_dma_1k, split_common.h).  One chunk: both halves are loaded once and stay.  More chunks: the
    // halves are a double buffer -- the next half (of this chunk, the next chunk, or the next row tile's first chunk)
    // streams in while the waves work through this one; one barrier per half (as in kan_sparse_fwd.hip).
    constexpr int GPH = NG / 2, HALF_SPL = (SPC / 2) * OT * 2 * 1024, HALF_BASE = (BPC / 2) * OT * 3 * 1024;
    constexpr int HALF_BYTES = HALF_SPL + HALF_BASE, SPL_BYTES = SPC * OT * 2 * 1024;
    static_assert(NG % 2 == 0 && 2 * HALF_BYTES == CHUNK_BYTES, "two halves of NG / 2 groups");
    const unsigned lds_w = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)s_w);
    auto dma_half = [&](int ch, int h) {
        const unsigned char* src = gw + (size_t)ch * CHUNK_BYTES;
        const unsigned dst = lds_w + h * HALF_BYTES;
        for (int blk = wave; blk < HALF_SPL / 1024; blk += NT / 64)
            lds_dma_1k(src + h * HALF_SPL + blk * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(dst + blk * 1024));
        for (int blk = wave; blk < HALF_BASE / 1024; blk += NT / 64)
            lds_dma_1k(src + SPL_BYTES + h * HALF_BASE + blk * 1024 + lane * 16,
                       __builtin_amdgcn_readfirstlane(dst + HALF_SPL + blk * 1024));
    };
    const int ch_begin = split * chunks_per_split, ch_end = min(nchunks, ch_begin + chunks_per_split);
    const bool resident = (ch_end - ch_begin) == 1;      // this workgroup's only chunk stays in LDS
    dma_half(ch_begin, 0);
    if (resident) { dma_half(ch_begin, 1); lds_dma_wait(); }
    y += (long)split * part_stride;
    __syncthreads();
    SplineGeom geom{}; Frag3Geom f3geo{};
    float ca[8] = {}, cao[8] = {};                       // RBF centres of the even / odd virtual features
    if constexpr (K == 0) { rbf_centers(rb, ca, 0); rbf_centers(rb, cao, sh); }
    if constexpr (K > 0) { geom = geom_from_knots(s_knots, nknots); f3geo = frag3_geom(s_knots, nknots); }
    const unsigned wodd = sh ? kWinBytes : 0u;           // selector-table offset of odd virtual features
    const int r = lane & 31, kg = lane >> 5;
    const bool al4 = ((ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);

    // 8 consecutive features (group g of chunk ch) of this lane's row.  Unconditional loads on clamped
    // addresses, never masked (a per-lane `cond ? load : const` makes hipcc branch around every load):
    // rows >= N are never stored, features >= in meet zero weights in the pack.
    const unsigned ldx4 = (unsigned)ldx * 4u, ldy4 = (unsigned)ldy * 4u;
    auto load8 = [&](long tile0 /* first row of the workgroup's tile: wave-uniform */, int ch, int g, float (&v)[8]) {
        const GBuf xb = gbuf_at(x, N, ldx, in, tile0);
        const unsigned ro = (unsigned)(wave * 32 + r) * ldx4 + kg * HF * 4;   // rows >= N: past the descriptor -> zeros
        const unsigned so = (unsigned)(ch * CF + 8 * g) * 4u;              // wave-uniform part of the offset
        if (al4 && sh == 0 && ch * CF + CF <= in) {       // wave-uniform
            gld4_s(xb, ro, so, v);
            gld4_s(xb, ro, so + 16, v + 4);
        } else {
            const int f0 = ch * CF + kg * HF + 8 * g;
            const unsigned rb = (unsigned)(wave * 32 + r) * ldx4;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = gld(xb, rb + min((f0 + j) >> sh, in - 1) * 4);
        }
    };

    // one-chunk layers narrower than a lane half's HF features: the last 8-feature groups carry only zero weights (as in
    // kan_sparse_fwd.hip)
    const int inv_live = in << sh;
    const int ng_live = (resident && inv_live < HF) ? max(1, (inv_live + 7) / 8) : NG;
    float xn[8];
    load8((long)blockIdx.x * ROWS, ch_begin, 0, xn);
    for (long tile = blockIdx.x; tile * ROWS < N; tile += gridDim.x) {
        const long row0 = tile * ROWS + wave * 32;
        f32x16 acc[OT];
#pragma unroll
        for (int t = 0; t < OT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
        float mean = 0.0f, rstd = 1.0f;                  // K == 0 with layernorm: this lane's row statistics
        if (K == 0 && rb.ln_w) {
            if (rb.stats_out) {
                // one-chunk layer (in <= CF): the row's features sit in this lane and its partner half (lane ^ 32) -- two-pass
                // mean / variance like torch, from loads that leave the row in L1 for the group pipeline below; the statistics
                // pass over x of its own (0.058 ms per layer at 1M x 64) is gone
                float xr[NG][8];
#pragma unroll
                for (int g = 0; g < NG; ++g) load8(tile * ROWS, ch_begin, g, xr[g]);
                float s = 0.0f;
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int j = 0; j < 8; ++j) s += (kg * HF + 8 * g + j < in) ? xr[g][j] : 0.0f;
                s += __shfl_xor(s, 32);
                mean = s / (float)in;
                float q = 0.0f;
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float d = (kg * HF + 8 * g + j < in) ? xr[g][j] - mean : 0.0f;
                        q = fmaf(d, d, q);
                    }
                q += __shfl_xor(q, 32);
                rstd = rsqrtf(q / (float)in + rb.ln_eps);
                if (kg == 0 && row0 + r < N) { rb.stats_out[2 * (row0 + r)] = mean; rb.stats_out[2 * (row0 + r) + 1] = rstd; }
            } else {
                const long rc = min(row0 + r, N - 1);
                mean = rb.stats[2 * rc]; rstd = rb.stats[2 * rc + 1];
            }
        }

        for (int ch = ch_begin; ch < ch_end; ++ch) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g >= ng_live) continue;                                    // narrow layer: only zero weights left
                const unsigned char* hb = s_w + (g / GPH) * HALF_BYTES;       // this group's half buffer
                const int gl = g % GPH;                                        // group inside the half
                if (!resident && gl == 0) {
                    // my pieces of this half have landed; after the barrier so have everyone's, and everyone is done with
                    // the other buffer -- which the half after this one now streams into
                    lds_dma_wait();
                    __syncthreads();
                    if (g == 0) dma_half(ch, 1);
                    else if (ch + 1 < ch_end) dma_half(ch + 1, 0);
                    else if ((tile + gridDim.x) * ROWS < N) dma_half(ch_begin, 0);
                }
                float xv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] = xn[j];
                // prefetch the next group: same chunk, next chunk, or the first group of this wave's next tile
                if (g + 1 < ng_live) load8(tile * ROWS, ch, g + 1, xn);
                else if (ch + 1 < ch_end) load8(tile * ROWS, ch + 1, 0, xn);
                else load8((tile + gridDim.x) * ROWS, ch_begin, 0, xn);

                if constexpr (K == 3) {
                    // ---- software pipeline inside the group: while the 6*OT/2 MFMAs of feature j execute, the
                    // VALU expands feature j+1 (or prepares the SiLU fragments after the last feature) and the
                    // LDS reads of its weights / selectors are already in flight.
                    // With virtual features (sh: two 8-slot windows per input feature) the odd one has the SAME x as
                    // the even one before it: its cubic pieces and hi/lo payload are reused, only the placement
                    // (selector row of the second window) differs.
                    u32x4 ahi, alo, bw[2 * OT];
                    float u; unsigned off_even, h0, h1, l0, l1;
                    {
                        frag3_index<false>(xv[0], f3geo, u, off_even);
                        const u32x4 sel = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(s_tbl) + off_even);
                        const unsigned char* wp = hb + (size_t)((8 * gl) * OT) * 2 * 1024 + lane * 16;
#pragma unroll
                        for (int i = 0; i < 2 * OT; ++i) bw[i] = *reinterpret_cast<const u32x4*>(wp + i * 1024);
                        frag3_payload(u, h0, h1, l0, l1);
                        frag3_place(sel, h0, h1, l0, l1, ahi, alo);
                    }
                    u32x4 a1, a2, a3;                      // SiLU fragments, prepared under the last feature's MFMAs
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        u32x4 nhi, nlo, nbw[2 * OT], sel;
                        const bool reuse = sh && ((j + 1) & 1);   // wave-uniform; j is a compile-time constant
                        if (j < 7) {                      // issue the next feature's LDS reads first
                            unsigned off;
                            if (reuse) off = off_even + wodd;
                            else { frag3_index<false>(xv[j + 1], f3geo, u, off_even); off = off_even; }
                            sel = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(s_tbl) + off);
                            const unsigned char* wp = hb + (size_t)((8 * gl + j + 1) * OT) * 2 * 1024 + lane * 16;
#pragma unroll
                            for (int i = 0; i < 2 * OT; ++i) nbw[i] = *reinterpret_cast<const u32x4*>(wp + i * 1024);
                        }
#pragma unroll
                        for (int t = 0; t < OT; ++t) acc[t] = mfma_f16(ahi, bw[2 * t], acc[t]);
#pragma unroll
                        for (int t = 0; t < OT; ++t) acc[t] = mfma_f16(ahi, bw[2 * t + 1], acc[t]);
#pragma unroll
                        for (int t = 0; t < OT; ++t) acc[t] = mfma_f16(alo, bw[2 * t], acc[t]);
                        if (j < 7) {
                            if (!reuse) frag3_payload(u, h0, h1, l0, l1);
                            frag3_place(sel, h0, h1, l0, l1, nhi, nlo);
                            ahi = nhi; alo = nlo;
#pragma unroll
                            for (int i = 0; i < 2 * OT; ++i) bw[i] = nbw[i];
                        } else {
                            float sv[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) sv[i] = (siluf(xv[i]) + (xv[i] - xv[i])) * kAScale;   // +-Inf -> NaN like the reference
                            split_bf16x3(sv, a1, a2, a3);
                        }
                    }
                    {
                        const unsigned char* wp = hb + HALF_SPL + (size_t)(gl * OT) * 3 * 1024 + lane * 16;
#pragma unroll
                        for (int t = 0; t < OT; ++t) {
                            const u32x4 w1 = *reinterpret_cast<const u32x4*>(wp + (t * 3 + 0) * 1024);
                            const u32x4 w2 = *reinterpret_cast<const u32x4*>(wp + (t * 3 + 1) * 1024);
                            const u32x4 w3 = *reinterpret_cast<const u32x4*>(wp + (t * 3 + 2) * 1024);
                            acc[t] = mfma_bf16(a3, w1, acc[t]);
                            acc[t] = mfma_bf16(a2, w2, acc[t]);
                            acc[t] = mfma_bf16(a1, w3, acc[t]);
                            acc[t] = mfma_bf16(a2, w1, acc[t]);
                            acc[t] = mfma_bf16(a1, w2, acc[t]);
                            acc[t] = mfma_bf16(a1, w1, acc[t]);
                        }
                    }
                } else {
                // ---- generic orders (K = 1, 2: exact knot comparisons) and the RBF basis (K == 0): one MFMA
                // step per feature
                float gam[8] = {}, bet[8] = {};
                if (K == 0 && rb.ln_w) {                   // wave-uniform
                    const int f0 = ch * CF + kg * HF + 8 * g;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const int fc = min((f0 + j) >> sh, in - 1); gam[j] = rb.ln_w[fc]; bet[j] = rb.ln_b[fc]; }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int s = 8 * gl + j;                   // step inside the half
                    u32x4 ahi, alo;
                    if constexpr (K == 0) {
                        const float z = rb.ln_w ? fmaf((xv[j] - mean) * rstd, gam[j], bet[j]) : xv[j];
                        make_rbf_frag(z, rb.a, (j & 1) ? cao : ca, ahi, alo);
                    } else {
                        make_spline_frag<K>(xv[j], s_knots, s_tbl, geom, ahi, alo, (j & 1) ? wodd : 0u);
                    }
                    const unsigned char* wp = hb + (size_t)(s * OT) * 2 * 1024 + lane * 16;
#pragma unroll
                    for (int t = 0; t < OT; ++t) {
                        const u32x4 bhi = *reinterpret_cast<const u32x4*>(wp + (t * 2 + 0) * 1024);
                        const u32x4 blo = *reinterpret_cast<const u32x4*>(wp + (t * 2 + 1) * 1024);
                        acc[t] = mfma_f16(ahi, bhi, acc[t]);
                        acc[t] = mfma_f16(ahi, blo, acc[t]);
                        acc[t] = mfma_f16(alo, bhi, acc[t]);
                    }
                }
                {
                    float sv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) sv[j] = (siluf(xv[j]) + (xv[j] - xv[j])) * kAScale;
                    u32x4 a1, a2, a3;
                    split_bf16x3(sv, a1, a2, a3);
                    const unsigned char* wp = hb + HALF_SPL + (size_t)(gl * OT) * 3 * 1024 + lane * 16;
#pragma unroll
                    for (int t = 0; t < OT; ++t) {
                        const u32x4 w1 = *reinterpret_cast<const u32x4*>(wp + (t * 3 + 0) * 1024);
                        const u32x4 w2 = *reinterpret_cast<const u32x4*>(wp + (t * 3 + 1) * 1024);
                        const u32x4 w3 = *reinterpret_cast<const u32x4*>(wp + (t * 3 + 2) * 1024);
                        acc[t] = mfma_bf16(a3, w1, acc[t]);
                        acc[t] = mfma_bf16(a2, w2, acc[t]);
                        acc[t] = mfma_bf16(a1, w3, acc[t]);
                        acc[t] = mfma_bf16(a2, w1, acc[t]);
                        acc[t] = mfma_bf16(a1, w2, acc[t]);
                        acc[t] = mfma_bf16(a1, w1, acc[t]);
                    }
                }
                }
            }
        }
        const GBuf yb = gbuf_at(y, N, ldy, out, tile * ROWS);
#pragma unroll
        for (int t = 0; t < OT; ++t) {
            const int col = 32 * t + r;
            const unsigned base = (unsigned)(wave * 32 + 4 * kg) * ldy4 + col * 4;
            if (col < out) {
                const float bb = (K == 0 && rb.bias && split == 0) ? rb.bias[col] : 0.0f;
#pragma unroll
                for (int i = 0; i < 16; ++i)               // rows >= N fall past the descriptor: dropped
                    gst_s(yb, base, (unsigned)((i & 3) + 8 * (i >> 2)) * ldy4, fmaf(acc[t][i], post, bb));
            }
        }
    }
}

// ------------------------------------------------------------------ host side
int kan_split_pack_fwd_noscale(const float* bw, const float* sw, const float* sc, int in, int out, int C,
                               void* pack_fwd, hipStream_t st) {
    const size_t stride = fwd_blk_bytes(in, min(out, kOutBlk), C);
    for (int b = 0; b * kOutBlk < out; ++b) {
        const int ob = min(kOutBlk, out - b * kOutBlk);
        const long o0 = (long)b * kOutBlk;
        unsigned char* pf = static_cast<unsigned char*>(pack_fwd) + b * stride;
        const long items = (long)(fwd_blk_bytes(in, ob, C) - kHdrBytes) / 16;
        const bool two = (long)in * ob * C >= kAbsmaxTwoLaunchMin;      // large blocks: partial maxima first (split_common.h)
        if (two) { int rc = launch_absmax_partials(bw ? bw + o0 * in : nullptr, sw + o0 * in * C, sc ? sc + o0 * in : nullptr, in, ob, C, pf, st); if (rc) return rc; }
        split_pack_fwd_kernel<<<(int)min((items + 1023) / 1024, two ? 256L : 64L), 1024, 0, st>>>(
            bw ? bw + o0 * in : nullptr, sw + o0 * in * C, sc ? sc + o0 * in : nullptr, in, ob, C, pf, two ? 2 : 1);
        KAGNN_LAUNCH_CHECK();
    }
    return KAGNN_OK;
}

// split-K plan for few-row inputs (Cora: 2708 rows x 1433 features): with fewer than ~128 row blocks the
// chunk loop is spread over blockIdx.y; every split writes a partial y, summed in a fixed order afterwards
struct FwdSplit { int splits, cps; };
static FwdSplit fwd_split_plan(long N, int nchunks, int rows_per_block) {
    FwdSplit p{1, nchunks};
    const long row_blocks = cdiv(N, rows_per_block);
    if (row_blocks >= 128 || nchunks < 2) return p;
    const int want = (int)min((long)nchunks, 256 / row_blocks);
    p.cps = cdiv(nchunks, max(want, 1));
    p.splits = cdiv(nchunks, p.cps);
    return p;
}
static inline int fwd_rows_per_block(int OT) { (void)OT; return 256; }       // 512 threads = 8 waves x 32 rows

size_t kan_split_fwd_ws_bytes(long N, int in, int out, int C) {
    size_t worst = 0;
    for (int b = 0; b * kOutBlk < out; ++b) {
        const int ob = min(kOutBlk, out - b * kOutBlk), OT = cdiv(ob, 32);
        const FwdSplit p = fwd_split_plan(N, cdiv(in << vshift(C), split_cf(OT)), fwd_rows_per_block(OT));
        if (p.splits > 1) worst = max(worst, (size_t)p.splits * N * ob * sizeof(float));
    }
    return worst;
}

__global__ void fwd_sum_splits_kernel(const float* __restrict__ part, int splits, long N, int out,
                                      float* __restrict__ y, long ldy) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= N * out) return;
    float a = 0.0f;
    for (int s = 0; s < splits; ++s) a += part[(long)s * N * out + i];
    y[(i / out) * ldy + (i % out)] = a;
}

template <int K, int OT, int NT>
static int launch_fwd(const float* x, long ldx, long N, int in, const float* knots, int nknots,
                      const unsigned char* pack, int nchunks, float* y, long ldy, int out, const RbfArgs& rb,
                      int sh, float* ws, size_t ws_bytes, hipStream_t st) {
    const size_t lds = kLdsHdr + split_fwd_chunk_bytes(OT);
    static unsigned long long configured = 0;          // (per device: common.h)
    if (auto first_use_ = first_use_on_this_device(configured)) {
        KAGNN_HIP(hipFuncSetAttribute((const void*)kan_split_fwd_kernel<K, OT, NT>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int gx = (int)min((long)cdiv(N, NT / 2), 256L);
    const FwdSplit p = fwd_split_plan(N, nchunks, NT / 2);
    if (p.splits > 1) {
        if (!ws || ws_bytes < (size_t)p.splits * N * out * sizeof(float))
            return fail(KAGNN_ERR_ARG, "%s: workspace too small (see kagnn_kan_fwd_workspace_bytes)", "kan_split_fwd");
        kan_split_fwd_kernel<K, OT, NT><<<dim3(gx, p.splits), NT, lds, st>>>(x, ldx, N, in, knots, nknots, pack, nchunks,
                                                                               ws, out, out, rb, sh, p.cps, N * (long)out);
        KAGNN_LAUNCH_CHECK();
        fwd_sum_splits_kernel<<<cdiv(N * out, 256), 256, 0, st>>>(ws, p.splits, N, out, y, ldy);
        KAGNN_LAUNCH_CHECK();
        return KAGNN_OK;
    }
    // (cubic splines with <= 8 coefficients never get here: kan_sparse_fwd.hip serves them)
    kan_split_fwd_kernel<K, OT, NT><<<gx, NT, lds, st>>>(x, ldx, N, in, knots, nknots, pack, nchunks, y, ldy, out,
                                                               rb, sh, nchunks, 0L);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// one <= 128-column output block
static int fwd_block(const float* x, long ldx, long N, const float* knots, int in, int out, int G, int K,
                     const unsigned char* p, float* y, long ldy, const RbfArgs& rb, float* ws, size_t ws_bytes,
                     hipStream_t st) {
    const int sh = vshift(G + K);
    const int OT = cdiv(out, 32), nk = K ? G + 2 * K + 1 : 0, nch = cdiv(in << sh, split_cf(OT));
    // (The kernels with <= 128 VGPRs could run 4 waves per SIMD, 1024 threads.  Measured in round 1 on three boxes that made
    // the forward itself 3 % faster and the whole layer step 1.3 % SLOWER -- the chip is power-limited in these kernels and the
    // denser forward costs the following kernels more clock than it gains -- so only the 2-waves-per-SIMD launch is built.)
#define GO(KK, TT) return launch_fwd<KK, TT, 512>(x, ldx, N, in, knots, nk, p, nch, y, ldy, out, rb, sh, ws, ws_bytes, st)
#define BYOT(KK) switch (OT) { case 1: GO(KK, 1); case 2: GO(KK, 2); case 3: GO(KK, 3); case 4: GO(KK, 4); }
    switch (K) {
        case 0: BYOT(0) break;
        case 1: BYOT(1) break;
        case 2: BYOT(2) break;
        case 3: BYOT(3) break;
        case 4: BYOT(4) break;
    }
#undef BYOT
#undef GO
    return fail(KAGNN_ERR_UNSUPPORTED, "%s: shape not covered by the split path", "kan_split_fwd");
}

// K == 0: Gaussian RBF basis with G = num_grids (rb holds centres / layernorm / bias), else B-splines
int kan_split_fwd_any(const float* x, long ldx, long N, const float* knots, int in, int out, int G, int K,
                      const void* pack, float* y, long ldy, const RbfArgs& rb, void* ws, size_t ws_bytes,
                      hipStream_t st) {
    const size_t stride = fwd_blk_bytes(in, min(out, kOutBlk), G + K);
    for (int b = 0; b * kOutBlk < out; ++b) {
        RbfArgs rbb = rb;
        if (rbb.bias) rbb.bias += b * kOutBlk;
        const int rc = fwd_block(x, ldx, N, knots, in, min(kOutBlk, out - b * kOutBlk), G, K,
                                 static_cast<const unsigned char*>(pack) + b * stride, y + b * kOutBlk, ldy, rbb,
                                 static_cast<float*>(ws), ws_bytes, st);
        if (rc) return rc;
    }
    return KAGNN_OK;
}

int kan_split_fwd(const float* x, long ldx, long N, const float* knots, int in, int out, int G, int K,
                  const void* pack, float* y, long ldy, void* ws, size_t ws_bytes, hipStream_t st) {
    return kan_split_fwd_any(x, ldx, N, knots, in, out, G, K, pack, y, ldy, RbfArgs{}, ws, ws_bytes, st);
}

// ---- input-gradient / weight-gradient split kernels: see kan_split_bwd.hip
}  // namespace kagnn
