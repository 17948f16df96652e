// KANLinear forward for cubic splines with <= 16 coefficients per feature (9..16: two 8-slot windows per input
// feature, the second one reusing the first one's cubic pieces) on the 2:4-SPARSE matrix cores
// (v_smfmac_f32_32x32x32_f16).  Same numerics as kan_split.hip (fp16 hi/lo split operands, three products per
// fp32 product, fp32 accumulate) -- the sparse instruction multiplies exactly the stored values, so results are
// bit-identical to the dense formulation -- at half the matrix-core work:
//
//   Of a feature's 8 coefficient slots only the 4 CONSECUTIVE ones c0..c0+3 (c0 = span - 3) are non-zero.  With
//   the slots laid along K in the order [0,4,1,5 | 2,6,3,7] every window of 4 consecutive slots puts exactly two
//   non-zeros into each group of four K positions: the 2:4 pattern the sparse MFMA wants.  A lane therefore feeds
//   4 fp16 values + 8 index bits per feature instead of an 8-slot fragment with 4 zeros, and one instruction
//   covers 4 features (K = 32) in the time the dense one covers 2.  The placement also shrinks from 8 to 4
//   v_perm_b32 per scalar.
//
// Operand layout of v_smfmac_f32_32x32x32_f16, measured with tools/probes/smfmac_probe.hip (the ISA text is not
// available offline): lane (m = l&31, kg = l>>5) stores pair i (values 2i, 2i+1) of K group 4*kg + i, index bits
// [4i+1:4i] / [4i+3:4i+2] = their positions inside the group (low 16 bits of the index VGPR, ABID 0); the dense
// operand's lane (n = l&31, kg) holds, for e = 0..15, K = 8*kg + (e&7) + 16*(e>>3).
//
// Reference behaviour replaced: node_classification_clean/ekan.py:79-112,146-162 (as kan_split.hip).
#include "split_common.h"

namespace kagnn {

typedef _Float16 f16x16 __attribute__((ext_vector_type(16)));

constexpr int kSpCF = 64;                 // features per LDS chunk
constexpr int kSpOutBlk = 64;             // output columns per launch (2 accumulator tiles: the 64-feature chunk of packed W is 152 KB)
constexpr int kSpOutBlkWide = 128;        // ... of the WIDE instantiation (OT = 4, round 5: see kan_sparse_fwd_kernel)
constexpr int kSpSteps = kSpCF / 4;       // sparse MFMA steps per chunk: 2 features per lane half and step
__host__ __device__ inline size_t sparse_fwd_chunk_bytes(int OT) {
    return (size_t)kSpSteps * OT * 2 * 2048 + (size_t)(kSpCF / 16) * OT * 2 * 1024;
}

bool kan_sparse_fwd_ok(int in, int out, int G, int K) { return K == 3 && G + K <= 16; }
// 9..16 coefficients: 2*in virtual features of 8 slots (wcat_v), as in kan_split.hip
__host__ __device__ inline int sp_sh(int C) { return C > 8 ? 1 : 0; }

// one output block of the pack: header, chunks, then an fp32 copy of the block's base weights [ob][in] for the exact
// SiLU path (so the forward entry point needs nothing but the pack)
// `inv` = number of (virtual) features = in << sp_sh(C)
static size_t sp_chunks_bytes(int inv, int ob) { return (size_t)cdiv(inv, kSpCF) * sparse_fwd_chunk_bytes(cdiv(ob, 32)); }
static size_t sp_blk_bytes(int in, int inv, int ob) { return kHdrBytes + sp_chunks_bytes(inv, ob) + (((size_t)ob * in * 4 + 255) & ~(size_t)255); }
// Output block width of a layer.  Layers whose output count is a multiple of 128 (on more than 32 virtual features: BASELINE config 3's
// 128 -> 128 layers) run the WIDE instantiation: one launch covers 128 outputs, so every scalar is expanded once per 128 outputs
// instead of once per 64 -- the expansion (span, cubic pieces, hi/lo split, placement), not the matrix cores, bounds this kernel.
// KAGNN_FWD_WIDE=0 keeps the 64-wide blocks (A/B timing).  The pack layout follows the block width: read once per process.
static bool sp_wide_enabled() {
    static const bool on = [] { const char* e = getenv("KAGNN_FWD_WIDE"); return !(e && e[0] == '0'); }();
    return on;
}
static int sp_out_blk(int inv /* virtual features */, int out) {
    return (sp_wide_enabled() && out % kSpOutBlkWide == 0 && inv > 32) ? kSpOutBlkWide : kSpOutBlk;
}
size_t kan_sparse_pack_fwd_bytes(int in, int out, int C) {
    const int inv = in << sp_sh(C), blk = sp_out_blk(inv, out);
    return (size_t)cdiv(out, blk) * sp_blk_bytes(in, inv, min(out, blk));
}

// A lane half of the sparse MFMA's K owns `hf` consecutive (virtual) features of a chunk, 8 per group.  Layers of <= 32
// features use BOTH halves all the same: hf = 16 (two groups) up to 32 features, 8 (one group) up to 16 -- with hf fixed
// at 32 such a layer ran four groups with the second lane half on zero weights (the per-rank slices of the feature-sharded
// layer; first layers on narrow inputs).
__host__ __device__ inline int sp_hf(int inv) { return inv <= 16 ? 8 : inv <= 32 ? 16 : kSpCF / 2; }

// K position p of a feature's 8-slot block holds coefficient slot slot_at(p): order [0,4,1,5,2,6,3,7]
__host__ __device__ inline int slot_at(int p) { return (p >> 1) + 4 * (p & 1); }

// chunk = [step t 16][out tile][hi|lo][half 2][lane 64][8 halfs]  +  base fragments [group 4][out tile][hi|lo][lane 64][8 halfs]
__device__ __forceinline__ void pack_sparse_items(const float* __restrict__ bw, const float* __restrict__ sw,
                                                  const float* __restrict__ sc, int in, int out, int C,
                                                  unsigned char* __restrict__ pack, float wscale, long first, long step) {
    const int OT = cdiv(out, 32), BPC = kSpCF / 16;
    const int HF = sp_hf(in << sp_sh(C));               // features per lane half (groups beyond it: zero weights, never read)
    const size_t chunk_bytes = sparse_fwd_chunk_bytes(OT);
    const long spl_per_chunk = (long)kSpSteps * OT * 128, base_per_chunk = (long)BPC * OT * 64;   // spline items: (lane, half)
    const long per_chunk = spl_per_chunk + base_per_chunk;
    const int sh = sp_sh(C), inv = in << sh;
    const long total = (long)cdiv(inv, kSpCF) * per_chunk;
    float* bcopy = reinterpret_cast<float*>(pack + kHdrBytes + (size_t)cdiv(inv, kSpCF) * chunk_bytes);
    for (long i = first; i < (long)out * in; i += step) bcopy[i] = bw ? bw[i] : 0.0f;      // unscaled fp32 base weights
    for (long i = first; i < total; i += step) {
        const int ch = i / per_chunk; long r = i % per_chunk;
        unsigned char* cbase = pack + kHdrBytes + (size_t)ch * chunk_bytes;
        if (r < spl_per_chunk) {
            const int h = r & 1; r >>= 1;                 // which 8 of the lane's 16 halfs
            const int lane = r & 63; r >>= 6;
            const int ot = r % OT; const int t = r / OT;
            const int o = 32 * ot + (lane & 31), kg = lane >> 5;
            // the two 16-byte halves of a lane's 32 bytes sit in separate KiB: each ds_read_b128 then walks the lanes at a
            // 16-byte stride (at 32 bytes per lane half the LDS banks idle: 42 % of the forward's LDS cycles were conflicts)
            _Float16* dh = reinterpret_cast<_Float16*>(cbase + ((size_t)(t * OT + ot) * 2 + 0) * 2048 + h * 1024 + lane * 16);
            _Float16* dl = reinterpret_cast<_Float16*>(cbase + ((size_t)(t * OT + ot) * 2 + 1) * 2048 + h * 1024 + lane * 16);
            // element el = 8h + p holds K = 8*kg + p + 16*h: K block kg + 2h = feature #kg of A's lane half h
            const int f = ch * kSpCF + h * HF + 2 * t + kg;
            const bool dead = 2 * t + kg >= HF;           // a step beyond the lane half's features (narrow layers)
            for (int p = 0; p < 8; ++p) {
                const int slot = slot_at(p);
                const float w = dead ? 0.0f : wcat_v(bw, sw, sc, in, out, C, o, f, slot, sh) * wscale;     // f: (virtual) feature
                const _Float16 hv = (_Float16)w;
                dh[p] = hv;
                dl[p] = (_Float16)(w - (float)hv);
            }
        } else {
            r -= spl_per_chunk;
            const int lane = r & 63; r >>= 6;
            const int ot = r % OT; const int sb = r / OT;
            const int o = 32 * ot + (lane & 31);
            _Float16* bh = reinterpret_cast<_Float16*>(cbase + (size_t)kSpSteps * OT * 2 * 2048 + ((size_t)(sb * OT + ot) * 2 + 0) * 1024 + lane * 16);
            _Float16* bl = reinterpret_cast<_Float16*>(cbase + (size_t)kSpSteps * OT * 2 * 2048 + ((size_t)(sb * OT + ot) * 2 + 1) * 1024 + lane * 16);
            for (int j = 0; j < 8; ++j) {                 // base weight of feature j of the group, fp16 hi / lo
                const int f = ch * kSpCF + (lane >> 5) * HF + 8 * sb + j;
                const float w = 8 * sb >= HF ? 0.0f : wcat_v(bw, sw, sc, in, out, C, o, f, 8, sh) * wscale;
                const _Float16 hv = (_Float16)w;
                bh[j] = hv;
                bl[j] = (_Float16)(w - (float)hv);
            }
        }
    }
}

__device__ __forceinline__ float pack_header(const float* bw, const float* sw, const float* sc, int in, int out, int C,
                                             unsigned char* pack, bool writer, float* s_m) {
    const float wmax = block_absmax_w(bw, sw, sc, in, out, C, s_m);
    const int e = scale_exp_from_max(wmax);
    if (writer) {
        reinterpret_cast<float*>(pack)[0] = ldexpf(1.0f, e - 10);   // post scale: undo 2^10 and 2^-e
        reinterpret_cast<int*>(pack)[1] = e;
    }
    return ldexpf(1.0f, -e);
}

__global__ void sparse_pack_fwd_kernel(const float* __restrict__ bw, const float* __restrict__ sw,
                                       const float* __restrict__ sc, int in, int out, int C,
                                       unsigned char* __restrict__ pack) {
    __shared__ float s_m[17];
    const float wscale = pack_header(bw, sw, sc, in, out, C, pack, blockIdx.x == 0 && threadIdx.x == 0, s_m);
    pack_sparse_items(bw, sw, sc, in, out, C, pack, wscale, blockIdx.x * (long)blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// one launch for BOTH layouts of a layer (out <= 64: one block of each): workgroups [0, nbf) write the sparse
// forward pack, the rest the input-gradient pack.  Two launches of ~15 us each were 2.5 % of the layer step.
__global__ void fused_pack_kernel(const float* __restrict__ bw, const float* __restrict__ sw, const float* __restrict__ sc,
                                  int in, int out, int C, unsigned char* __restrict__ pack_fwd,
                                  unsigned char* __restrict__ pack_dx, int nbf) {
    __shared__ float s_m[17];
    const bool fwd = (int)blockIdx.x < nbf;
    const int bid = fwd ? blockIdx.x : blockIdx.x - nbf, nb = fwd ? nbf : gridDim.x - nbf;
    unsigned char* pack = fwd ? pack_fwd : pack_dx;
    const float wscale = pack_header(bw, sw, sc, in, out, C, pack, bid == 0 && threadIdx.x == 0, s_m);
    const long first = bid * (long)blockDim.x + threadIdx.x, step = (long)nb * blockDim.x;
    if (fwd) pack_sparse_items(bw, sw, sc, in, out, C, pack, wscale, first, step);
    else pack_dx_items(bw, sw, sc, in, out, C, dx_q2(out), pack, wscale, first, step);
}

// per span index i = floor((x-g0)/h) + 1 (clamped to [0,31]): {selector of group 0, selector of group 1, index byte}
// -- which payload halves (r = slot - c0, c0 = i - 4) land in the two stored values of each K group, and where.
// Second window (9..16 coefficients, slots 8..15 of the feature): the same table shifted by 8 slots, 512 bytes on.
__device__ __forceinline__ void build_sparse_table(unsigned* tbl /* LDS, 2*32*4 */, int tid, int nknots) {
    if (tid < 64) {
        const int i = tid & 31, c0 = i - 4 - 8 * (tid >> 5);
        const bool live = (i >= 1) && (i <= nknots - 1);
        unsigned sel[2] = {0x0c0c0c0cu, 0x0c0c0c0cu}, ib = 0;
        for (int g = 0; g < 2; ++g) {
            int pos[2] = {-1, -1}, pay[2] = {-1, -1}, n = 0;
            for (int p = 0; p < 4 && live; ++p) {
                const int r = slot_at(4 * g + p) - c0;
                if (r >= 0 && r <= 3 && n < 2) { pos[n] = p; pay[n] = r; ++n; }
            }
            for (int p = 0; p < 4 && n < 2; ++p) {      // pad with zero values at unused positions (ascending order kept)
                if (p != pos[0]) { pos[n] = p; pay[n] = -1; ++n; }
            }
            if (pos[0] > pos[1]) { int t = pos[0]; pos[0] = pos[1]; pos[1] = t; t = pay[0]; pay[0] = pay[1]; pay[1] = t; }
            unsigned s = 0;
            for (int k = 0; k < 2; ++k) {
                const unsigned b = pay[k] >= 0 ? (unsigned)((2 * pay[k]) | ((2 * pay[k] + 1) << 8)) : 0x0c0cu;
                s |= b << (16 * k);
            }
            sel[g] = s;
            ib |= (unsigned)(pos[0] | (pos[1] << 2)) << (4 * g);
        }
        tbl[4 * tid + 0] = sel[0]; tbl[4 * tid + 1] = sel[1]; tbl[4 * tid + 2] = ib; tbl[4 * tid + 3] = 0;
    }
}

__device__ __forceinline__ f32x16 smfmac(const u32x4& a, const u32x4& b0, const u32x4& b1, const f32x16& c, int idx) {
    typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
    const u32x8 b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    return __builtin_amdgcn_smfmac_f32_32x32x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x16, b), c, idx, 0, 0);
}

// 512 threads = 8 waves (2 per SIMD), one wave = 32 rows; persistent workgroups, split-K over blockIdx.y for few-row
// inputs (see kan_split.hip).  The packed W of a 64-feature chunk (152 KB for 64 outputs) lives in LDS as two HALVES
// (8 sparse steps + 2 SiLU groups each).  One chunk (in <= 64): both halves are loaded once and stay.  More chunks: the
// halves are a double buffer -- while the waves work through one half, the next one (of this chunk, the next chunk, or
// the next row tile's first chunk) streams in by LDS-DMA; one barrier per half.  (Staging a whole chunk through
// registers between two barriers, as before, was 37 % of the forward at 128 -> 128, grid 8: 2.4 GB of L2 -> LDS traffic
// per 64-output launch that nothing overlapped.)
// MOM: the epilogue also accumulates the COLUMN MOMENTS of y (count, mean, sum of squared deviations; per wave over its
// row tiles, merged pairwise in a fixed order -- Chan et al., no cancellation) and leaves one (mean, M2, count) row per
// workgroup in mom_partial[gridDim.x][3][out]: the BatchNorm1d that follows the convolution (reference
// models.py:198-200) then needs no statistics pass over y (bn.hip: moments_finish, bn_apply_from_moments_kernel).
// NARROW: a layer of <= 32 (virtual) features, laid over both lane halves (sp_hf); a template flag because the lane-half width
// as a runtime value cost the 64-feature forward 1.7 %
// AGG (>= 0, narrow layers only): the kernel's input is NOT read from memory but produced in place -- the GIN neighbour
// aggregation h0[i] = self_scale * x[i] + sum_{j -> i} x[j] of the convolution (reference models.py:48-56, GINConv around the KAN),
// gathered by the lane that then expands it; h0 is also stored (the backward's saved input).  This is the
// producer -> consumer fusion BASELINE.json's north_star names, in the form where it pays: layers of <= 32 input features
// (the per-rank slices of the feature-sharded layer, first layers on narrow inputs) run the forward at 160..250 registers.
// Measured (profiles/r03_experiments.md): not faster than two launches, hence opt-in (KAGNN_FUSE_AGG=1, api.hip).
// The summation ORDER is that of the stand-alone aggregation
// kernels, so h0 and y are bit-identical to the two-launch form: AGG == 0 -> agg_rows_v4_kernel (16 < in <= 32: self term
// first, edges in CSR order); AGG == 4 / 8 -> agg_rows_ep_kernel (in <= 16 / in <= 8: AGG edge slots walked in parallel,
// combined pairwise, self term last).  Rows above the hub threshold get their self term only, exactly like the row kernels;
// the hub kernels complete h0 and the caller recomputes y for those few rows (kan_sparse_fwd_agg).
struct SpAgg {
    const int* rowptr; const int* col; float self_scale; int hub_threshold;
    float* h0; long ldh;
};
__device__ __forceinline__ float4 spagg_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// PARTS: the input is the column concatenation of up to 8 row-major blocks that live in different buffers (the skip-concat
// read-out of the node models: [x | h1 | h2 | ...] is never built).  Every block is a whole number of 64-feature chunks wide,
// so chunk ch reads from p[ch] with leading dimension ld[ch] (the host resolves the blocks to per-chunk entries).
struct SpParts { const float* p[8]; int ld[8];
                 const float* sc[8]; const float* sh[8]; };     // per chunk: NULL, or the 64 column scales / shifts of a block that
                                                                // exists only as scale * x + shift (a folded BatchNorm1d, round 4)

// HALF (KAGNN_PREC_HALF, split_common.h): one fp16 product per fp32 product -- bases and SiLU rounded once (RNE), only the hi
// fragments of the packed weights are read: one sparse MFMA per step and tile instead of three, no lo payloads / placements
template <int OT, bool SH, bool MOM, bool NARROW = false, int AGG = -1, bool PARTS = false, bool HALF = false>      // SH: 9..16 coefficients as 2*in virtual features (two 8-slot windows per input feature)
__global__ __launch_bounds__(512) void kan_sparse_fwd_kernel(
    const float* __restrict__ x, long ldx, long N, int in, const float* __restrict__ knots_g, int nknots,
    const unsigned char* __restrict__ pack, int nchunks, float* __restrict__ y, long ldy, int out,
    int chunks_per_split, long part_stride, float* __restrict__ mom_partial, SpAgg ag, SpParts xp) {
    static_assert(AGG < 0 || (NARROW && !SH && !MOM), "the fused aggregation serves plain narrow layers");
    static_assert(!PARTS || (!NARROW && !SH && !MOM && AGG < 0), "column blocks: plain wide layers");
    static_assert(!HALF || (!SH && AGG < 0), "single-product mode: layers of <= 8 coefficients, no fused aggregation");
    static_assert(OT != 4 || (!NARROW && AGG < 0 && !PARTS && !HALF), "the wide instantiation serves plain layers");
    // OT == 4 (WIDE, round 5): 128 outputs per launch at the SAME two waves per SIMD and 256-row tiles as the 64-wide blocks --
    // round 4's OT = 4 ran one wave per SIMD on 128-row tiles and lost to the doubled weight stream (r04_experiments.md 8).  What
    // makes it fit 256 registers: (i) MERGED -- the SiLU branch is fed at the bases' scale 2^10 (not 2^4) and accumulates into
    // the spline accumulator, no second tile set (|silu| < 58 instead of < 3660 before a group takes the exact-fp32 fallback,
    // which accumulates at the same scale); (ii) the weight fragments of a step are read where they are used instead of one
    // step ahead into a second register set.  The packed chunk (288 KB) streams through the two LDS buffers in FOUR pieces.
    constexpr bool MERGED = OT == 4;
    constexpr int NT = 512, NW = NT / 64, CF = kSpCF, BPC = CF / 16, NG = CF / 16, ROWS = (NT / 64) * 32;
    const int HF = NARROW ? sp_hf(in << (SH ? 1 : 0)) : CF / 2;    // features per lane half: CF / 2, or 16 / 8 in narrow layers
    constexpr int CHUNK_BYTES = kSpSteps * OT * 2 * 2048 + BPC * OT * 2 * 1024;
    constexpr int SPL_BYTES = kSpSteps * OT * 2 * 2048;
    // a chunk streams through the two LDS buffers in PPC pieces of GPP feature groups (4 sparse steps + 1 SiLU group per group)
    constexpr int PPC = OT == 4 ? 4 : 2, GPP = NG / PPC;
    constexpr int HALF_SPL = SPL_BYTES / PPC, HALF_BASE = (BPC / PPC) * OT * 2 * 1024, HALF_BYTES = HALF_SPL + HALF_BASE;
    constexpr int BUF_BYTES = 2 * HALF_BYTES;             // both LDS buffers (== CHUNK_BYTES unless wide)
    static_assert(PPC * HALF_BYTES == CHUNK_BYTES && NG == 4 && kSpSteps == 16, "PPC pieces of GPP groups x 4 steps");
    static_assert((HALF_SPL / 1024) % NW == 0, "spline blocks of a piece divide over the waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_knots = reinterpret_cast<float*>(smem);
    unsigned* s_tbl = reinterpret_cast<unsigned*>(smem + 256);
    unsigned char* s_w = smem + kLdsHdr;
    // wave id: an SGPR in the column-moments instantiation (its per-wave LDS slice and the final merge then keep no VGPR alive
    // across the MFMA loop); the plain one keeps the VGPR form it was tuned with (scalar: 2 % slower, same-box A/B)
    const int tid = threadIdx.x, wave = MOM ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6), lane = tid & 63;
    if (tid < nknots) s_knots[tid] = knots_g[tid];
    build_sparse_table(s_tbl, tid, nknots);
    const float post = reinterpret_cast<const float*>(pack)[0];
    const float post_b = post * 64.0f;                  // the SiLU branch is fed at scale 2^4 instead of 2^10 (not MERGED)
    // exact-fp32 SiLU groups (values beyond fp16 range) accumulate into the SAME acc_b: their base weights are scaled by
    // 2^(4 - e) so the products sit at acc_b's scale (an accumulator tile of their own cost 16 OT registers the kernel does
    // not have: the column-moments instantiation spilled 53 VGPRs around its MFMA loop -- profiles/r03_kernel_resources.txt)
    const float wsc16 = ldexpf(1.0f, (MERGED ? 10 : 4) - reinterpret_cast<const int*>(pack)[1]);
    const float* base_w = reinterpret_cast<const float*>(pack + kHdrBytes + (size_t)nchunks * CHUNK_BYTES);   // [out][in] fp32
    const unsigned char* gw = pack + kHdrBytes;
    // half h (0 / 1) of chunk ch -> LDS buffer h: 32 OT one-KiB pieces of sparse-step fragments + 4 OT of SiLU fragments
    const unsigned lds_w = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)s_w);
    // piece h (0 .. PPC-1) of chunk ch -> LDS buffer h & 1: one-KiB blocks of sparse-step fragments, then of SiLU fragments
    auto dma_half = [&](int ch, int h) {
        const unsigned char* src = gw + (size_t)ch * CHUNK_BYTES;
        const unsigned dst = lds_w + (h & 1) * HALF_BYTES;
        constexpr int SPB = HALF_SPL / 1024 / NW, BB = HALF_BASE / 1024;      // spline blocks per wave; SiLU blocks of the piece
#pragma unroll
        for (int k = 0; k < SPB; ++k) {
            const int blk = wave * SPB + k;
            lds_dma_1k(src + h * HALF_SPL + blk * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(dst + blk * 1024));
        }
#pragma unroll
        for (int k = 0; k < (BB + NW - 1) / NW; ++k) {
            const int blk = wave + k * NW;
            if (blk < BB)
                lds_dma_1k(src + SPL_BYTES + h * HALF_BASE + blk * 1024 + lane * 16,
                           __builtin_amdgcn_readfirstlane(dst + HALF_SPL + blk * 1024));
        }
    };
    const int ch_begin = blockIdx.y * chunks_per_split, ch_end = min(nchunks, ch_begin + chunks_per_split);
    const bool resident = PPC == 2 && (ch_end - ch_begin) == 1;      // (a wide chunk never fits: it always streams)
    dma_half(ch_begin, 0);
    if (resident) { if (AGG < 0) dma_half(ch_begin, 1); lds_dma_wait(); }      // (AGG: the second half buffer is the gather's staging area)
    y += (long)blockIdx.y * part_stride;
    __syncthreads();
    const Frag3Geom f3geo = frag3_geom(s_knots, nknots);
    const int r = lane & 31, kg = lane >> 5;
    const bool al4 = ((ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    const unsigned ldx4 = (unsigned)ldx * 4u, ldy4 = (unsigned)ldy * 4u;
    auto load8 = [&](long tile0 /* first row of the workgroup's tile: wave-uniform */, int ch, int g, float (&v)[8]) {
        if constexpr (PARTS) {                            // this chunk's block (scalar selects: ch is wave-uniform)
            const float* xc = xp.p[0];
            int ldc = xp.ld[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) { xc = ch == i ? xp.p[i] : xc; ldc = ch == i ? xp.ld[i] : ldc; }
            const GBuf xb = gbuf_at(xc, N, ldc, CF, tile0);
            const unsigned ro = (unsigned)(wave * 32 + r) * ((unsigned)ldc * 4u) + kg * HF * 4;
            gld4_s(xb, ro, (unsigned)(8 * g) * 4u, v);
            gld4_s(xb, ro, (unsigned)(8 * g) * 4u + 16, v + 4);
            const float* as = xp.sc[0];
            const float* ah = xp.sh[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) { as = ch == i ? xp.sc[i] : as; ah = ch == i ? xp.sh[i] : ah; }
            if (as != nullptr) {                          // wave-uniform: this block is  scale * x + shift  (a folded BatchNorm1d; rows >= N
                                                          // are never stored).  Applied here: carried with the prefetched rows to the
                                                          // point of use instead (16 more live registers) the kernel was 3 % slower
                const float4 s0 = *reinterpret_cast<const float4*>(as + kg * HF + 8 * g), s1 = *reinterpret_cast<const float4*>(as + kg * HF + 8 * g + 4);
                const float4 h0 = *reinterpret_cast<const float4*>(ah + kg * HF + 8 * g), h1 = *reinterpret_cast<const float4*>(ah + kg * HF + 8 * g + 4);
                v[0] = fmaf(v[0], s0.x, h0.x); v[1] = fmaf(v[1], s0.y, h0.y); v[2] = fmaf(v[2], s0.z, h0.z); v[3] = fmaf(v[3], s0.w, h0.w);
                v[4] = fmaf(v[4], s1.x, h1.x); v[5] = fmaf(v[5], s1.y, h1.y); v[6] = fmaf(v[6], s1.z, h1.z); v[7] = fmaf(v[7], s1.w, h1.w);
            }
            return;
        }
        const GBuf xb = gbuf_at(x, N, ldx, in, tile0);
        const unsigned ro = (unsigned)(wave * 32 + r) * ldx4 + kg * HF * 4;   // rows >= N: past the descriptor -> zeros
        const unsigned so = (unsigned)(ch * CF + 8 * g) * 4u;
        if constexpr (SH) {                               // 4 input features, each feeding its two windows
            const int f0 = (ch * CF + kg * HF + 8 * g) >> 1;
            const unsigned rb = (unsigned)(wave * 32 + r) * ldx4;
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[2 * i] = gld(xb, rb + min(f0 + i, in - 1) * 4); v[2 * i + 1] = v[2 * i]; }
        } else if (al4 && ch * CF + CF <= in) {           // wave-uniform
            gld4_s(xb, ro, so, v);
            gld4_s(xb, ro, so + 16, v + 4);
        } else {
            const int f0 = ch * CF + kg * HF + 8 * g;
            const unsigned rb = (unsigned)(wave * 32 + r) * ldx4;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = gld(xb, rb + min(f0 + j, in - 1) * 4);   // features >= in meet zero weights
        }
    };
    // AGG: the aggregated rows of one 32-row wave tile, all groups of this lane (xa[g][8]).  Cooperative and edge-parallel:
    // the tile's rows own ONE contiguous CSR edge range; the wave's 64 lanes load it item by item (item = one float4 of one
    // neighbour row: coalesced index loads, every gather independent of the others) into this wave's staging area in LDS --
    // the second half buffer of the weight chunk, which narrow layers never read -- and then each (row, feature group) lane
    // sums ITS edges from LDS in the stand-alone kernels' order.  (A lane walking its own neighbour list in global memory --
    // the first form of this kernel -- ran at the pace of the longest of 32 lists, two dependent round trips per step:
    // 0.83 vs 0.56 ms for the layer forward, profiles/r03_experiments.md.)  Edge ranges of hub rows are skipped.
    const int ng_live_c = HF / 8;
    constexpr int AGRP = AGG == 0 ? 2 : 1;                 // groups per lane half: in <= 16 -> 1, in <= 32 -> 2
    constexpr int AEP = AGG > 0 ? AGG : 1;
    constexpr int STAGE_BYTES = (HALF_BYTES / 8) & ~15;    // per wave
    auto gather_tile = [&](long tile0, float (&xa)[AGRP][8]) {
        const int wave_s = __builtin_amdgcn_readfirstlane(wave);
        float4* stage = reinterpret_cast<float4*>(s_w + HALF_BYTES + wave_s * STAGE_BYTES);
        const int Q = in >> 2;                             // float4 per row
        const int cap = STAGE_BYTES / (16 * Q);            // edges per staging round
        const float rq = 1.0f / (float)Q;
        const long row0 = tile0 + wave_s * 32;
        const long row = row0 + r;
        const bool live_row = row < N;
        const int es = live_row ? ag.rowptr[row] : 0, et = live_row ? ag.rowptr[row + 1] : 0;
        const bool hub = (et - es) > ag.hub_threshold;
        const float sw = ag.self_scale;
        float4 acc[AGRP][2][AEP];
        float4 self[AGRP][2];
        bool l0[AGRP], l1[AGRP];
#pragma unroll
        for (int g = 0; g < AGRP; ++g) {
            const int f0 = kg * HF + 8 * g;
            l0[g] = live_row && f0 < in && g < ng_live_c; l1[g] = l0[g] && f0 + 4 < in;
            self[g][0] = l0[g] ? spagg_ld4(x + row * ldx + f0) : make_float4(0.f, 0.f, 0.f, 0.f);
            self[g][1] = l1[g] ? spagg_ld4(x + row * ldx + f0 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k = 0; k < AEP; ++k)
                    acc[g][h][k] = (AGG == 0) ? make_float4(self[g][h].x * sw, self[g][h].y * sw, self[g][h].z * sw, self[g][h].w * sw)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // the tile's edge range (wave-uniform) and its hub rows
        const int E0 = __builtin_amdgcn_readfirstlane(row0 < N ? ag.rowptr[row0] : 0);
        const int E1 = __builtin_amdgcn_readfirstlane(row0 < N ? ag.rowptr[min(row0 + 32, N)] : 0);
        const unsigned long long hubs = __builtin_amdgcn_ballot_w64(hub && kg == 0);
        int cb = E0;
        while (cb < E1) {
            int ce = min(cb + cap, E1);
            for (unsigned long long m = hubs; m; m &= m - 1) {        // (scalar loop; usually no hub in the tile)
                const int bpos = __builtin_ctzll(m);
                const int hs = __builtin_amdgcn_readlane(es, bpos), he = __builtin_amdgcn_readlane(et, bpos);
                if (cb >= hs && cb < he) { cb = he; ce = min(cb + cap, E1); }
                else if (hs > cb && hs < ce) ce = hs;
            }
            if (cb >= E1) break;
            const int nitems = (ce - cb) * Q;
            for (int it0 = 0; it0 < nitems; it0 += 256) {             // 4 items per lane in flight
                int idx[4], quad[4], jj[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    idx[u] = it0 + 64 * u + lane;
                    ok[u] = idx[u] < nitems;
                    const int edge = (int)(((float)idx[u] + 0.5f) * rq);     // idx / Q (exact: idx < 2^10)
                    quad[u] = idx[u] - edge * Q;
                    jj[u] = ok[u] ? ag.col[cb + edge] : 0;
                }
                float4 vv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    vv[u] = ok[u] ? spagg_ld4(x + (long)jj[u] * ldx + 4 * quad[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ok[u]) stage[idx[u]] = vv[u];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // this lane's edges inside [cb, ce)
            const int lo = max(es, cb), hi = hub ? lo : min(et, ce);
            if constexpr (AGG == 0) {
                for (int e = lo; e < hi; ++e) {
                    const float4* p = stage + (e - cb) * Q + ((kg * HF) >> 2);
#pragma unroll
                    for (int g = 0; g < AGRP; ++g) {
                        if (l0[g]) { const float4 v = p[2 * g]; acc[g][0][0].x += v.x; acc[g][0][0].y += v.y; acc[g][0][0].z += v.z; acc[g][0][0].w += v.w; }
                        if (l1[g]) { const float4 v = p[2 * g + 1]; acc[g][1][0].x += v.x; acc[g][1][0].y += v.y; acc[g][1][0].z += v.z; acc[g][1][0].w += v.w; }
                    }
                }
            } else {
                // edge number k of the row goes to slot k % AEP: walk k in aligned groups of AEP so that the slot index is static
                for (int kk = ((lo - es) / AEP) * AEP; es + kk < hi; kk += AEP) {
#pragma unroll
                    for (int sl = 0; sl < AEP; ++sl) {
                        const int e = es + kk + sl;
                        if (e >= lo && e < hi) {
                            const float4* p = stage + (e - cb) * Q + ((kg * HF) >> 2);
                            if (l0[0]) { const float4 v = p[0]; acc[0][0][sl].x += v.x; acc[0][0][sl].y += v.y; acc[0][0][sl].z += v.z; acc[0][0][sl].w += v.w; }
                            if (l1[0]) { const float4 v = p[1]; acc[0][1][sl].x += v.x; acc[0][1][sl].y += v.y; acc[0][1][sl].z += v.z; acc[0][1][sl].w += v.w; }
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            cb = ce;
        }
#pragma unroll
        for (int g = 0; g < AGRP; ++g) {
            float4 o[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if constexpr (AGG == 0) o[h] = acc[g][h][0];
                else {
#pragma unroll
                    for (int st = 1; st < AEP; st <<= 1)     // the butterfly of the stand-alone kernel, seen from slot 0
#pragma unroll
                        for (int k = 0; k < AEP; k += 2 * st) {
                            acc[g][h][k].x += acc[g][h][k + st].x; acc[g][h][k].y += acc[g][h][k + st].y;
                            acc[g][h][k].z += acc[g][h][k + st].z; acc[g][h][k].w += acc[g][h][k + st].w;
                        }
                    o[h] = make_float4(fmaf(sw, self[g][h].x, acc[g][h][0].x), fmaf(sw, self[g][h].y, acc[g][h][0].y),
                                       fmaf(sw, self[g][h].z, acc[g][h][0].z), fmaf(sw, self[g][h].w, acc[g][h][0].w));
                }
            }
            const int f0 = kg * HF + 8 * g;
            if (l0[g]) *reinterpret_cast<float4*>(ag.h0 + row * ag.ldh + f0) = o[0];
            if (l1[g]) *reinterpret_cast<float4*>(ag.h0 + row * ag.ldh + f0 + 4) = o[1];
            if (!l0[g]) o[0] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!l1[g]) o[1] = make_float4(0.f, 0.f, 0.f, 0.f);
            xa[g][0] = o[0].x; xa[g][1] = o[0].y; xa[g][2] = o[0].z; xa[g][3] = o[0].w;
            xa[g][4] = o[1].x; xa[g][5] = o[1].y; xa[g][6] = o[1].z; xa[g][7] = o[1].w;
        }
    };
    // one scalar -> its two stored dwords (hi and lo parts) and its index byte
    auto place1 = [&](const u32x4& e, unsigned h0, unsigned h1, unsigned l0, unsigned l1, unsigned& hi0, unsigned& hi1,
                      unsigned& lo0, unsigned& lo1) {
        hi0 = __builtin_amdgcn_perm(h1, h0, e[0]); hi1 = __builtin_amdgcn_perm(h1, h0, e[1]);
        lo0 = __builtin_amdgcn_perm(l1, l0, e[0]); lo1 = __builtin_amdgcn_perm(l1, l0, e[1]);
    };

    const int ng_live = HF / 8;                          // groups a lane half really has (4 unless the layer is narrow)
    (void)ng_live;
    // this wave's rows so far: count (wave-uniform), column mean and column M2.  The per-column pairs live in LDS behind the
    // weight chunk ([wave][t][mean | M2][32 columns], 4 KiB at OT = 2), not in registers: four more live VGPRs through the
    // MFMA loop were what tipped the 256-register instantiation into scratch
    float mom_n = 0.0f;
    float* s_momw = reinterpret_cast<float*>(s_w + BUF_BYTES) + wave * (OT * 64);
    if constexpr (MOM) {
#pragma unroll
        for (int t = 0; t < OT; ++t) s_momw[64 * t + lane] = 0.0f;      // (each wave touches only its own slice: no barrier)
    }
    float xn[8];
    float xa[AGG >= 0 ? AGRP : 1][8];
    if constexpr (AGG >= 0) {
        gather_tile((long)blockIdx.x * ROWS, xa);
#pragma unroll
        for (int j = 0; j < 8; ++j) xn[j] = xa[0][j];
    } else load8((long)blockIdx.x * ROWS, ch_begin, 0, xn);
    for (long tile = blockIdx.x; tile * ROWS < N; tile += gridDim.x) {
        const long row0 = tile * ROWS + wave * 32;
        // acc: spline part (bases * 2^10); acc_b: SiLU branch through fp16 hi/lo at scale 2^4 (|silu| < 4094) -- and, for the
        // rare groups whose values do not fit that, exact fp32 MFMAs on the fp32 weights brought to the same scale
        // (round 2 kept a third tile `acc_f` for them: 16 OT more registers; with the packed-fp32 payloads the leaner kernel is as
        // fast -- 0.393 vs 0.394 ms per step, same-box A/B -- and nothing spills)
        f32x16 acc[OT], acc_b[MERGED ? 1 : OT];
#pragma unroll
        for (int t = 0; t < OT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[t][i] = 0.0f; if constexpr (!MERGED) acc_b[t][i] = 0.0f; }

        for (int ch = ch_begin; ch < ch_end; ++ch) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g >= ng_live) continue;                                // narrow layer: nothing but zero weights left
                const int q = g / GPP, gq = g % GPP;                       // piece of the chunk, group inside the piece
                const unsigned char* hb = s_w + (q & 1) * HALF_BYTES;      // this piece's buffer
                if (!resident && gq == 0) {
                    // my blocks of this piece have landed; after the barrier so have everyone's, and everyone is done with
                    // the other buffer -- which the piece after this one now streams into
                    lds_dma_wait();
                    __syncthreads();
#ifndef KAGNN_ABLATE_FWD_NO_REFILL        // TIMING-ONLY ablation (wrong results): what the streamed forward costs without its L2 -> LDS refills
                    if (q + 1 < PPC) dma_half(ch, q + 1);
                    else if (ch + 1 < ch_end) dma_half(ch + 1, 0);
                    else if ((tile + gridDim.x) * ROWS < N) dma_half(ch_begin, 0);
#endif
                }
                float xv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] = xn[j];
                if constexpr (AGG >= 0) {                                  // (narrow: one chunk, one or two groups)
                    if (g + 1 < ng_live) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) xn[j] = xa[AGRP - 1][j];
                    } else {
                        gather_tile((tile + gridDim.x) * ROWS, xa);
#pragma unroll
                        for (int j = 0; j < 8; ++j) xn[j] = xa[0][j];
                    }
                } else if (g + 1 < ng_live) load8(tile * ROWS, ch, g + 1, xn);
                else if (ch + 1 < ch_end) load8(tile * ROWS, ch + 1, 0, xn);
                else load8((tile + gridDim.x) * ROWS, ch_begin, 0, xn);

                // ---- 4 sparse steps per group (features 2s, 2s+1 of the group): while the 3*OT sparse MFMAs of step
                // s execute, the VALU expands the two scalars of step s+1 and their weights / table entries are in flight
                u32x4 ahi, alo, bw[4 * OT];
                int aidx;
                auto prep_reads = [&](int s, u32x4& e0, u32x4& e1, float& u0, float& u1, u32x4 (&w)[4 * OT]) {
                    unsigned o0, o1;
                    frag3_index<false>(xv[2 * s], f3geo, u0, o0);
                    if constexpr (SH) { u1 = u0; o1 = o0 + 512u; }      // same x, second window's table
                    else frag3_index<false>(xv[2 * s + 1], f3geo, u1, o1);
                    e0 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(s_tbl) + o0);
                    e1 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(s_tbl) + o1);
                    if constexpr (MERGED) return;           // (wide: the weight fragments are read where they are used -- read_w)
                    const unsigned char* wp = hb + (size_t)((4 * gq + s) * OT) * 2 * 2048 + lane * 16;
#pragma unroll
                    for (int i = 0; i < 2 * OT; ++i) {      // [ot][hi|lo] x two 16-byte halves, a KiB apart
                        if (HALF && (i & 1)) continue;      // (single-product mode: the hi fragments only)
                        w[2 * i] = *reinterpret_cast<const u32x4*>(wp + i * 2048);
                        w[2 * i + 1] = *reinterpret_cast<const u32x4*>(wp + i * 2048 + 1024);
                    }
                };
                auto read_w = [&](int s, u32x4 (&w)[4 * OT]) {
                    const unsigned char* wp = hb + (size_t)((4 * gq + s) * OT) * 2 * 2048 + lane * 16;
#pragma unroll
                    for (int i = 0; i < 2 * OT; ++i) {
                        w[2 * i] = *reinterpret_cast<const u32x4*>(wp + i * 2048);
                        w[2 * i + 1] = *reinterpret_cast<const u32x4*>(wp + i * 2048 + 1024);
                    }
                };
                auto build = [&](int s, const u32x4& e0, const u32x4& e1, float u0, float u1, u32x4& hi, u32x4& lo, int& idx) {
                    unsigned a0, a1, a2, a3, b0 = 0, b1 = 0, b2 = 0, b3 = 0, h0, h1, l0, l1;
#ifdef KAGNN_ABLATE_FWD_NO_EXPAND          // TIMING-ONLY ablation (wrong results): the A operand "received from elsewhere" -- what is left is the
                                           // weight-fragment reads and the matrix-core work: the share of a launch that a 128-output forward would
                                           // NOT repeat per 64-output block
                    hi = u32x4{__float_as_uint(u0), __float_as_uint(u1), e0[0], e1[0]};
                    lo = u32x4{e0[1], e1[1], __float_as_uint(u1), __float_as_uint(u0)};
                    idx = (int)(e0[2] | (e1[2] << 8));
                    return;
#endif
                    if constexpr (HALF) {            // rounded once: hi payloads and placements only
                        unsigned ph[2][2];
                        frag3_payload_pair_h(u0, u1, ph);
                        a0 = __builtin_amdgcn_perm(ph[0][1], ph[0][0], e0[0]); a1 = __builtin_amdgcn_perm(ph[0][1], ph[0][0], e0[1]);
                        a2 = __builtin_amdgcn_perm(ph[1][1], ph[1][0], e1[0]); a3 = __builtin_amdgcn_perm(ph[1][1], ph[1][0], e1[1]);
                    } else
                    if constexpr (!SH) {             // the step's two scalars on packed fp32 (forward 0.399 -> 0.388 ms per step)
                        unsigned ph[2][2], pl[2][2];
                        frag3_payload_pair(u0, u1, ph, pl);
                        place1(e0, ph[0][0], ph[0][1], pl[0][0], pl[0][1], a0, a1, b0, b1);
                        place1(e1, ph[1][0], ph[1][1], pl[1][0], pl[1][1], a2, a3, b2, b3);
                    } else {                         // SH: the pair shares x -- one payload, two placements
                        frag3_payload(u0, h0, h1, l0, l1);
                        place1(e0, h0, h1, l0, l1, a0, a1, b0, b1);
                        place1(e1, h0, h1, l0, l1, a2, a3, b2, b3);
                    }
                    hi = u32x4{a0, a1, a2, a3};
                    lo = u32x4{b0, b1, b2, b3};
                    idx = (int)(e0[2] | (e1[2] << 8));
                };
                u32x4 sh_hi, sh_lo;                        // SiLU fragments, prepared under the last step's MFMAs
                float sv[8];
                bool big = false;
                auto silu_prep = [&]() {
                    float smx = 0.0f;
#pragma unroll
                    for (int i = 0; i < 8; i += 2) {     // packed fp32, two scalars per instruction
                        const f32x2 xp = {xv[i], xv[i + 1]};
                        f32x2 pr = silu16_pair(xp) + (xp - xp) * splat2(16.0f);      // +-Inf -> NaN like the reference
                        if constexpr (MERGED) pr = pr * splat2(64.0f);                      // the bases' scale: 2^10
                        sv[i] = pr.x; sv[i + 1] = pr.y;
                        smx = fmaxf(fmaxf(smx, fabsf(pr.x)), fabsf(pr.y));
                    }
                    big = __any(!(smx < 60000.0f) || sv[0] != sv[0] || sv[1] != sv[1] || sv[2] != sv[2] || sv[3] != sv[3] ||
                                sv[4] != sv[4] || sv[5] != sv[5] || sv[6] != sv[6] || sv[7] != sv[7]);   // wave-uniform
                    if constexpr (HALF) { round_f16x2(sv, sh_hi); sh_lo = sh_hi; }
                    else split_f16x2(sv, sh_hi, sh_lo);
                };
                auto base_branch = [&]() {
                if (!big) {

                    const unsigned char* wp = hb + HALF_SPL + (size_t)(gq * OT) * 2 * 1024 + lane * 16;
#pragma unroll
                    for (int t = 0; t < OT; ++t) {
                        f32x16& ab = MERGED ? acc[t] : acc_b[MERGED ? 0 : t];
                        const u32x4 wh = *reinterpret_cast<const u32x4*>(wp + (t * 2 + 0) * 1024);
                        ab = mfma_f16(sh_hi, wh, ab);
                        if constexpr (!HALF) {
                            const u32x4 wl = *reinterpret_cast<const u32x4*>(wp + (t * 2 + 1) * 1024);
                            ab = mfma_f16(sh_hi, wl, ab);
                            ab = mfma_f16(sh_lo, wh, ab);
                        }
                    }
                } else {
                    // values beyond fp16 range (|x| > ~3700) or non-finite: this group's SiLU branch in exact fp32,
                    // v_mfma_f32_32x32x2_f32 with k = lane half <-> this lane's own feature, weights straight from HBM
                    const int f0 = ch * CF + kg * HF + 8 * g;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
#pragma unroll
                        for (int t = 0; t < OT; ++t) {
                            const int o = 32 * t + r;
                            const int fr = SH ? (f0 + j) >> 1 : f0 + j;          // SH: only the first window carries the base weight
                            const float w = (o < out && fr < in && !(SH && (j & 1))) ? base_w[(long)o * in + fr] : 0.0f;
                            f32x16& ab = MERGED ? acc[t] : acc_b[MERGED ? 0 : t];
                            ab = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[j] * (MERGED ? 0.0009765625f : 0.0625f), w * wsc16, ab, 0, 0, 0);
                        }
                    }
                }
                };
                // wide (MERGED): the SiLU branch FIRST -- its fragments are dead before the sparse steps start and the x values die step by
                // step, instead of x, 16 SiLU values / fragments and the step's operands all being live under the last step (the
                // instantiation spilled 2-15 VGPRs that way); same sums, another order inside the accumulator
                if constexpr (MERGED) { silu_prep(); base_branch(); }
                {
                    u32x4 e0, e1; float u0, u1;
                    prep_reads(0, e0, e1, u0, u1, bw);
                    build(0, e0, e1, u0, u1, ahi, alo, aidx);
                    if constexpr (MERGED) read_w(0, bw);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    u32x4 e0, e1, nbw[MERGED ? 1 : 4 * OT]; float u0, u1;
                    if constexpr (MERGED) { if (s < 3) prep_reads(s + 1, e0, e1, u0, u1, bw); }      // (index + table reads only)
                    else if (s < 3) prep_reads(s + 1, e0, e1, u0, u1, reinterpret_cast<u32x4 (&)[4 * OT]>(nbw));
#ifdef KAGNN_ABLATE_FWD_12SLOT               // TIMING-ONLY ablation (wrong results; profiles/r06_experiments.md 4): the two-window layers with three
                    if (!(SH && s == 3)) {           // of every four sparse steps -- the matrix-core work a 12-slot 2:4 layout would leave, with the
#endif                                               // expansion, weight reads and everything else as they are: the upper bound of that layout's gain
#pragma unroll
                    for (int t = 0; t < OT; ++t) acc[t] = smfmac(ahi, bw[4 * t], bw[4 * t + 1], acc[t], aidx);
                    if constexpr (!HALF) {
#pragma unroll
                        for (int t = 0; t < OT; ++t) acc[t] = smfmac(ahi, bw[4 * t + 2], bw[4 * t + 3], acc[t], aidx);
#pragma unroll
                        for (int t = 0; t < OT; ++t) acc[t] = smfmac(alo, bw[4 * t], bw[4 * t + 1], acc[t], aidx);
                    }
#ifdef KAGNN_ABLATE_FWD_12SLOT
                    }
#endif
                    if (s < 3) {
                        u32x4 nhi, nlo; int nidx;
                        build(s + 1, e0, e1, u0, u1, nhi, nlo, nidx);
                        ahi = nhi; alo = nlo; aidx = nidx;
                        if constexpr (MERGED) read_w(s + 1, bw);
                        else {
#pragma unroll
                            for (int i = 0; i < 4 * OT; ++i) { if (HALF && ((i >> 1) & 1)) continue; bw[i] = nbw[i]; }
                        }
                    } else if constexpr (!MERGED) {        // (inline, not through the lambda: the 64-wide instantiations keep the code they were tuned with)
                        float smx = 0.0f;
    #pragma unroll
                        for (int i = 0; i < 8; i += 2) {     // packed fp32, two scalars per instruction
                            const f32x2 xp = {xv[i], xv[i + 1]};
                            f32x2 pr = silu16_pair(xp) + (xp - xp) * splat2(16.0f);      // +-Inf -> NaN like the reference
                            if constexpr (MERGED) pr = pr * splat2(64.0f);                      // the bases' scale: 2^10
                            sv[i] = pr.x; sv[i + 1] = pr.y;
                            smx = fmaxf(fmaxf(smx, fabsf(pr.x)), fabsf(pr.y));
                        }
                        big = __any(!(smx < 60000.0f) || sv[0] != sv[0] || sv[1] != sv[1] || sv[2] != sv[2] || sv[3] != sv[3] ||
                                    sv[4] != sv[4] || sv[5] != sv[5] || sv[6] != sv[6] || sv[7] != sv[7]);   // wave-uniform
                        if constexpr (HALF) { round_f16x2(sv, sh_hi); sh_lo = sh_hi; }
                        else split_f16x2(sv, sh_hi, sh_lo);
                    }
                }
                if constexpr (!MERGED) {
                if (!big) {

                    const unsigned char* wp = hb + HALF_SPL + (size_t)(gq * OT) * 2 * 1024 + lane * 16;
#pragma unroll
                    for (int t = 0; t < OT; ++t) {
                        f32x16& ab = MERGED ? acc[t] : acc_b[MERGED ? 0 : t];
                        const u32x4 wh = *reinterpret_cast<const u32x4*>(wp + (t * 2 + 0) * 1024);
                        ab = mfma_f16(sh_hi, wh, ab);
                        if constexpr (!HALF) {
                            const u32x4 wl = *reinterpret_cast<const u32x4*>(wp + (t * 2 + 1) * 1024);
                            ab = mfma_f16(sh_hi, wl, ab);
                            ab = mfma_f16(sh_lo, wh, ab);
                        }
                    }
                } else {
                    // values beyond fp16 range (|x| > ~3700) or non-finite: this group's SiLU branch in exact fp32,
                    // v_mfma_f32_32x32x2_f32 with k = lane half <-> this lane's own feature, weights straight from HBM
                    const int f0 = ch * CF + kg * HF + 8 * g;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
#pragma unroll
                        for (int t = 0; t < OT; ++t) {
                            const int o = 32 * t + r;
                            const int fr = SH ? (f0 + j) >> 1 : f0 + j;          // SH: only the first window carries the base weight
                            const float w = (o < out && fr < in && !(SH && (j & 1))) ? base_w[(long)o * in + fr] : 0.0f;
                            f32x16& ab = MERGED ? acc[t] : acc_b[MERGED ? 0 : t];
                            ab = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[j] * (MERGED ? 0.0009765625f : 0.0625f), w * wsc16, ab, 0, 0, 0);
                        }
                    }
                }
                }
            }
        }
        const GBuf yb = gbuf_at(y, N, ldy, out, tile * ROWS);
        const long rows_here = min(32L, N - row0);         // wave-uniform; <= 0 for the waves past the last row
        // (small integers: the reciprocals are the same instruction in every wave, so the merge order stays fixed)
        const float mom_nt = (float)max(rows_here, 1L), mom_rnt = __builtin_amdgcn_rcpf(mom_nt);
        const float mom_w = mom_nt * __builtin_amdgcn_rcpf(mom_n + mom_nt), mom_nw = mom_n * mom_w;
#pragma unroll
        for (int t = 0; t < OT; ++t) {
            const int col = 32 * t + r;
            const unsigned base = (unsigned)(wave * 32 + 4 * kg) * ldy4 + col * 4;
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = MERGED ? acc[t][i] * post : fmaf(acc[t][i], post, acc_b[MERGED ? 0 : t][i] * post_b);
            if (col < out) {
#pragma unroll
                for (int i = 0; i < 16; ++i)               // rows >= N fall past the descriptor: dropped
                    gst_s(yb, base, (unsigned)((i & 3) + 8 * (i >> 2)) * ldy4, v[i]);
            }
            if constexpr (MOM) {
                // the 32-row tile's own (mean, M2), then merged into the wave's.  Full tiles (all but the wave's last one)
                // take the unmasked form: the 32 row-validity selects per column were most of this epilogue's VALU and,
                // live next to the second accumulator tile, pushed the kernel into scratch
                float sm = 0.0f, q = 0.0f, mt;
                if (rows_here >= 32) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) sm += v[i];
                    sm += __shfl_xor(sm, 32);
                    mt = sm * mom_rnt;
#pragma unroll
                    for (int i = 0; i < 16; ++i) { const float d = v[i] - mt; q = fmaf(d, d, q); }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) sm += (4 * kg + (i & 3) + 8 * (i >> 2) < rows_here) ? v[i] : 0.0f;
                    sm += __shfl_xor(sm, 32);
                    mt = sm * mom_rnt;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float d = (4 * kg + (i & 3) + 8 * (i >> 2) < rows_here) ? v[i] - mt : 0.0f;
                        q = fmaf(d, d, q);
                    }
                }
                if (rows_here > 0) {
                    q += __shfl_xor(q, 32);
                    const float m_old = s_momw[64 * t + r], q_old = s_momw[64 * t + 32 + r];
                    const float d = mt - m_old;
                    if (kg == 0) {
                        s_momw[64 * t + r] = fmaf(d, mom_w, m_old);
                        s_momw[64 * t + 32 + r] = q_old + fmaf(d * d, mom_nw, q);
                    }
                }
            }
        }
        if constexpr (MOM) { if (rows_here > 0) mom_n += (float)rows_here; }
    }
    if constexpr (MOM) {
        // the 8 waves' moments already sit in LDS; merged in wave order
        __syncthreads();
        const float* s_mom = reinterpret_cast<const float*>(s_w + BUF_BYTES);     // [wave 8][t][mean | M2][32]
        float* s_cnt = reinterpret_cast<float*>(s_w);                              // (the weight chunk is dead)
        // (lane / wave ids taken afresh: the prologue's copies would otherwise stay live -- in scratch -- across the MFMA loop)
        const int lane2 = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (lane2 == 0) s_cnt[wave] = mom_n;
        __syncthreads();
        if (wave == 0)
        for (int tid = lane2; tid < OT * 32 && tid < out; tid += 64) {        // (wide blocks: 128 columns, two per lane)
            float n = 0.0f, m = 0.0f, q = 0.0f;
            for (int w8 = 0; w8 < NT / 64; ++w8) {
                const float nb = s_cnt[w8];
                if (nb > 0.0f) {
                    const float mb = s_mom[w8 * (OT * 64) + 64 * (tid >> 5) + (tid & 31)], qb = s_mom[w8 * (OT * 64) + 64 * (tid >> 5) + 32 + (tid & 31)];
                    const float nn = n + nb, d = mb - m, w = nb / nn;
                    m = fmaf(d, w, m);
                    q += qb + d * d * n * w;
                    n = nn;
                }
            }
            mom_partial[((long)blockIdx.x * 3 + 0) * out + tid] = m;
            mom_partial[((long)blockIdx.x * 3 + 1) * out + tid] = q;
            mom_partial[((long)blockIdx.x * 3 + 2) * out + tid] = n;
        }
    }
}

// ------------------------------------------------------------------ host side
int kan_sparse_pack_fwd(const float* bw, const float* sw, const float* sc, int in, int out, int C, void* pack_fwd,
                        hipStream_t st) {
    const int inv = in << sp_sh(C), blk = sp_out_blk(inv, out);
    const size_t stride = sp_blk_bytes(in, inv, min(out, blk));
    for (int b = 0; b * blk < out; ++b) {
        const int ob = min(blk, out - b * blk);
        const long o0 = (long)b * blk;
        const long items = (long)sp_chunks_bytes(inv, ob) / 16;
        sparse_pack_fwd_kernel<<<(int)min((items + 1023) / 1024, 64L), 1024, 0, st>>>(
            bw ? bw + o0 * in : nullptr, sw + o0 * in * C, sc ? sc + o0 * in : nullptr, in, ob, C,
            static_cast<unsigned char*>(pack_fwd) + b * stride);
        KAGNN_LAUNCH_CHECK();
    }
    return KAGNN_OK;
}

// both packs of one layer in one launch; only for out <= 64 (one output block in either layout), C <= 8
bool kan_fused_pack_ok(int in, int out, int C) { return out <= kSpOutBlk && C <= 8; }
int kan_fused_pack(const float* bw, const float* sw, const float* sc, int in, int out, int C, void* pack_fwd,
                   void* pack_dx, hipStream_t st) {
    const long items_f = (long)sp_chunks_bytes(in, out) / 16;
    const long items_d = (long)cdiv(in, 16) * kCTmax * dx_q2(out) * 64;
    const int nbf = (int)min((items_f + 1023) / 1024, 48L), nbd = (int)min((items_d + 1023) / 1024, 48L);
    fused_pack_kernel<<<nbf + nbd, 1024, 0, st>>>(bw, sw, sc, in, out, C, static_cast<unsigned char*>(pack_fwd),
                                                  static_cast<unsigned char*>(pack_dx), nbf);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// the same for up to kPackBatch layers of a chain in ONE launch (each ~22 us pack launch is latency, not work:
// a two-layer chain saves one of them per step)
constexpr int kPackBatch = 16;          // (round 5: the whole GINE stack of a graph-level model packs in one launch: 4 convs x 2 layers + ...)
struct PackBatch {
    const float* bw[kPackBatch]; const float* sw[kPackBatch]; const float* sc[kPackBatch];
    unsigned char* pf[kPackBatch]; unsigned char* pd[kPackBatch];
    int in[kPackBatch], out[kPackBatch], nbf[kPackBatch], blk0[kPackBatch + 1];
    int n, C;
};

__global__ void fused_pack_batch_kernel(PackBatch b) {
    __shared__ float s_m[17];
    int l = 0;
    while (l + 1 < b.n && (int)blockIdx.x >= b.blk0[l + 1]) ++l;                  // block-uniform
    const int local = blockIdx.x - b.blk0[l], nbl = b.blk0[l + 1] - b.blk0[l];
    const bool fwd = local < b.nbf[l];
    const int bid = fwd ? local : local - b.nbf[l], nb = fwd ? b.nbf[l] : nbl - b.nbf[l];
    const float *bw = b.bw[l], *sw = b.sw[l], *sc = b.sc[l];
    const int in = b.in[l], out = b.out[l], C = b.C;
    unsigned char* pack = fwd ? b.pf[l] : b.pd[l];
    const float wscale = pack_header(bw, sw, sc, in, out, C, pack, bid == 0 && threadIdx.x == 0, s_m);
    const long first = bid * (long)blockDim.x + threadIdx.x, step = (long)nb * blockDim.x;
    if (fwd) pack_sparse_items(bw, sw, sc, in, out, C, pack, wscale, first, step);
    else pack_dx_items(bw, sw, sc, in, out, C, dx_q2(out), pack, wscale, first, step);
}

int kan_fused_pack_batch(int n, const float* const* bw, const float* const* sw, const float* const* sc, const int* in,
                         const int* out, int C, void* const* pack_fwd, void* const* pack_dx, hipStream_t st) {
    if (n < 1 || n > kPackBatch) return fail(KAGNN_ERR_UNSUPPORTED, "%s: 1..16 layers per batch", "kan_fused_pack_batch");
    PackBatch b{};
    b.n = n; b.C = C; b.blk0[0] = 0;
    for (int l = 0; l < n; ++l) {
        if (!kan_fused_pack_ok(in[l], out[l], C)) return fail(KAGNN_ERR_UNSUPPORTED, "%s: layer shape not covered", "kan_fused_pack_batch");
        const long items_f = (long)sp_chunks_bytes(in[l], out[l]) / 16;
        const long items_d = (long)cdiv(in[l], 16) * kCTmax * dx_q2(out[l]) * 64;
        const int nbf = (int)min((items_f + 1023) / 1024, 48L), nbd = (int)min((items_d + 1023) / 1024, 48L);
        b.bw[l] = bw[l]; b.sw[l] = sw[l]; b.sc[l] = sc ? sc[l] : nullptr;
        b.pf[l] = static_cast<unsigned char*>(pack_fwd[l]); b.pd[l] = static_cast<unsigned char*>(pack_dx[l]);
        b.in[l] = in[l]; b.out[l] = out[l]; b.nbf[l] = nbf; b.blk0[l + 1] = b.blk0[l] + nbf + nbd;
    }
    fused_pack_batch_kernel<<<b.blk0[n], 1024, 0, st>>>(b);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

struct SpSplit { int splits, cps; };
static SpSplit sp_split_plan(long N, int nchunks) {        // same policy as kan_split.hip
    SpSplit p{1, nchunks};
    const long row_blocks = cdiv(N, 256);
    if (row_blocks >= 128 || nchunks < 2) return p;
    const int want = (int)min((long)nchunks, 256 / row_blocks);
    p.cps = cdiv(nchunks, max(want, 1));
    p.splits = cdiv(nchunks, p.cps);
    return p;
}

size_t kan_sparse_fwd_ws_bytes(long N, int in, int out, int C) {
    const SpSplit p = sp_split_plan(N, cdiv(in << sp_sh(C), kSpCF));
    return p.splits > 1 ? (size_t)p.splits * N * min(out, sp_out_blk(in << sp_sh(C), out)) * sizeof(float) : 0;
}

__global__ void sparse_sum_splits_kernel(const float* __restrict__ part, int splits, long N, int out,
                                         float* __restrict__ y, long ldy) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= N * out) return;
    float a = 0.0f;
    for (int s = 0; s < splits; ++s) a += part[(long)s * N * out + i];
    y[(i / out) * ldy + (i % out)] = a;
}

int moments_finish(const float* partial, int P, int F, float* col_mean, float* col_m2, hipStream_t st);   // bn.hip

static int sp_grid(long N) { return (int)min((long)cdiv(N, 256), 256L); }
// column moments in the epilogue: whenever the launch is not split over the chunks (few-row inputs)
bool kan_sparse_fwd_moments_ok(long N, int in, int out, int G, int K) {
    // (wide 128-output blocks of layers with <= 8 coefficients: the column-moments instantiation <4, false, true> spills 7 VGPRs --
    // eight distinct x values per group where the two-window layers have four; those layers take the stand-alone moments pass)
    if (sp_out_blk(in << sp_sh(G + K), out) == kSpOutBlkWide && !sp_sh(G + K)) return false;
    return kan_sparse_fwd_ok(in, out, G, K) && sp_split_plan(N, cdiv(in << sp_sh(G + K), kSpCF)).splits == 1;
}
size_t kan_sparse_fwd_moments_ws_bytes(long N, int out) { return (size_t)sp_grid(N) * 3 * min(out, kSpOutBlkWide) * sizeof(float); }   // (either block width)

template <int OT, bool SH, bool MOM, bool NARROW, bool HALF = false>
static int launch_sparse(const float* x, long ldx, long N, int in, const float* knots, int nknots,
                         const unsigned char* pack, float* y, long ldy, int out, float* ws, size_t ws_bytes,
                         float* col_mean, float* col_m2, hipStream_t st) {
    if constexpr (!HALF && !SH && !NARROW && OT != 4) {          // single-product mode (thread-local, set by the entry point): its instantiation
        if (g_half_products)
            return launch_sparse<OT, SH, MOM, NARROW, true>(x, ldx, N, in, knots, nknots, pack, y, ldy, out, ws, ws_bytes, col_mean, col_m2, st);
    }
    // (wide, OT = 4: two LDS buffers of a QUARTER chunk each)
    const size_t lds = kLdsHdr + sparse_fwd_chunk_bytes(OT) / (OT == 4 ? 2 : 1) + (MOM ? 8 * OT * 64 * sizeof(float) : 0);
    static unsigned long long configured = 0;          // (per device: common.h)
    if (auto first_use_ = first_use_on_this_device(configured)) {
        KAGNN_HIP(hipFuncSetAttribute((const void*)kan_sparse_fwd_kernel<OT, SH, MOM, NARROW, -1, false, HALF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int nchunks = cdiv(in << (SH ? 1 : 0), kSpCF);
    const int gx = sp_grid(N);
    const SpSplit p = sp_split_plan(N, nchunks);
    if constexpr (MOM) {
        if (p.splits > 1) return fail(KAGNN_ERR_UNSUPPORTED, "%s: no column moments from a launch split over the chunks", "kan_sparse_fwd");
        if (!ws || ws_bytes < (size_t)gx * 3 * out * sizeof(float))
            return fail(KAGNN_ERR_ARG, "%s: workspace too small for the column moments", "kan_sparse_fwd");
        kan_sparse_fwd_kernel<OT, SH, true, NARROW, -1, false, HALF><<<gx, 512, lds, st>>>(x, ldx, N, in, knots, nknots, pack, nchunks, y, ldy, out, nchunks, 0L, ws, SpAgg{}, SpParts{});
        KAGNN_LAUNCH_CHECK();
        if (g_mom_defer && gx <= kMomDeferMaxP) {        // the consumer (the norm's apply kernel) folds the partial rows itself
            g_mom_defer->partial = ws; g_mom_defer->P = gx;
            return KAGNN_OK;
        }
        return moments_finish(ws, gx, out, col_mean, col_m2, st);
    }
    if (p.splits > 1) {
        if (!ws || ws_bytes < (size_t)p.splits * N * out * sizeof(float))
            return fail(KAGNN_ERR_ARG, "%s: workspace too small (see kagnn_kan_fwd_workspace_bytes)", "kan_sparse_fwd");
        kan_sparse_fwd_kernel<OT, SH, false, NARROW, -1, false, HALF><<<dim3(gx, p.splits), 512, lds, st>>>(x, ldx, N, in, knots, nknots, pack, nchunks, ws, out, out,
                                                                                p.cps, N * (long)out, nullptr, SpAgg{}, SpParts{});
        KAGNN_LAUNCH_CHECK();
        sparse_sum_splits_kernel<<<cdiv(N * out, 256), 256, 0, st>>>(ws, p.splits, N, out, y, ldy);
        KAGNN_LAUNCH_CHECK();
        return KAGNN_OK;
    }
    kan_sparse_fwd_kernel<OT, SH, false, NARROW, -1, false, HALF><<<gx, 512, lds, st>>>(x, ldx, N, in, knots, nknots, pack, nchunks, y, ldy, out, nchunks, 0L, nullptr, SpAgg{}, SpParts{});
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// ------------------------------------------------------------------ aggregation fused into the first KANLinear (narrow layers)
bool kan_sparse_fwd_agg_ok(const float* x, long ldx, long N, int in, int out, int G, int K) {
    return K == 3 && G + K <= 8 && in >= 8 && in <= 32 && in % 4 == 0 && out <= kSpOutBlk && ldx % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(x) & 15) == 0 && sp_split_plan(N, 1).splits == 1;
}

// rows of the FIRST segment of every hub row: h0 rows -> compact rows (one per segment slot), and y rows back
__global__ void hub_rows_gather_kernel(const float* __restrict__ h0, long ldh, const int* __restrict__ seg, long nseg, int F,
                                       float* __restrict__ tmp) {
    const long sidx = blockIdx.x;
    const int row = seg[3 * sidx];
    const bool first = sidx == 0 || seg[3 * (sidx - 1)] != row;
    for (int f = threadIdx.x; f < F; f += blockDim.x) tmp[sidx * F + f] = first ? h0[(long)row * ldh + f] : 0.0f;
}
__global__ void hub_rows_scatter_kernel(const float* __restrict__ ytmp, const int* __restrict__ seg, long nseg, int F,
                                        float* __restrict__ y, long ldy) {
    const long sidx = blockIdx.x;
    const int row = seg[3 * sidx];
    if (sidx > 0 && seg[3 * (sidx - 1)] == row) return;
    for (int f = threadIdx.x; f < F; f += blockDim.x) y[(long)row * ldy + f] = ytmp[sidx * F + f];
}

int aggregate_hub_rows(const AggArgs& a, const int* hub_seg, long num_hub_seg, float* ws, size_t ws_bytes, hipStream_t st);   // aggregate.hip
int kan_sparse_fwd(const float* x, long ldx, long N, const float* knots, int in, int out, int G, int K, const void* pack, float* y,
                   long ldy, void* ws, size_t ws_bytes, float* col_mean, float* col_m2, hipStream_t st);

size_t kan_sparse_fwd_agg_ws_bytes(long num_hub_seg, int in, int out) {
    return (size_t)num_hub_seg * (size_t)(((in + 3) & ~3) * 2 + out) * sizeof(float);       // hub partials | compact h0 rows | their y rows
}

template <int OT, int AGG>
static int launch_sparse_agg(const float* x, long ldx, long N, int in, const float* knots, int nknots, const unsigned char* pack,
                             float* y, long ldy, int out, const SpAgg& ag, hipStream_t st) {
    const size_t lds = kLdsHdr + sparse_fwd_chunk_bytes(OT);
    static unsigned long long configured = 0;          // (per device: common.h)
    if (auto first_use_ = first_use_on_this_device(configured)) {
        KAGNN_HIP(hipFuncSetAttribute((const void*)kan_sparse_fwd_kernel<OT, false, false, true, AGG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    kan_sparse_fwd_kernel<OT, false, false, true, AGG><<<sp_grid(N), 512, lds, st>>>(x, ldx, N, in, knots, nknots, pack, 1, y, ldy, out, 1, 0L,
                                                                                      nullptr, ag, SpParts{});
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// h0 = self_scale * x + (sum over in-neighbours), y = KANLinear(h0): ONE kernel for every row below the hub threshold; the hub
// rows' h0 is completed by the aggregation's hub kernels and their y recomputed on a compact copy (rows of a KANLinear are
// independent, so those rows are bit-identical to a whole-matrix forward too)
int kan_sparse_fwd_agg(const float* x, long ldx, long N, const int* rowptr, const int* col, const int* hub_seg, long num_hub_seg,
                       int hub_threshold, float self_scale, const float* knots, int in, int out, int G, int K, const void* pack,
                       float* h0, long ldh, float* y, long ldy, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!kan_sparse_fwd_agg_ok(x, ldx, N, in, out, G, K)) return fail(KAGNN_ERR_UNSUPPORTED, "%s: shape not covered", "kan_sparse_fwd_agg");
    if (num_hub_seg > 0 && ws_bytes < kan_sparse_fwd_agg_ws_bytes(num_hub_seg, in, out))
        return fail(KAGNN_ERR_ARG, "%s: workspace too small", "kan_sparse_fwd_agg");
    const int nk = G + 2 * K + 1, OT = cdiv(out, 32);
    const unsigned char* p = static_cast<const unsigned char*>(pack);
    SpAgg ag{rowptr, col, self_scale, (num_hub_seg == 0 || hub_seg == nullptr) ? 0x7fffffff : hub_threshold, h0, ldh};
    int rc;
#define LA(AA) (OT == 1 ? launch_sparse_agg<1, AA>(x, ldx, N, in, knots, nk, p, y, ldy, out, ag, st) \
                        : launch_sparse_agg<2, AA>(x, ldx, N, in, knots, nk, p, y, ldy, out, ag, st))
    rc = in <= 8 ? LA(8) : in <= 16 ? LA(4) : LA(0);
#undef LA
    if (rc || num_hub_seg == 0 || hub_seg == nullptr) return rc;
    // hub rows: partial sums per segment + ordered fold onto h0 (the stand-alone kernels), then y for those rows
    AggArgs a{};
    a.x = x; a.ldx = ldx; a.out = h0; a.ldo = ldh; a.rowptr = rowptr; a.col = col; a.N = N; a.F = in;
    a.self_scale = self_scale; a.hub_threshold = hub_threshold;
    float* wsf = static_cast<float*>(ws);
    const size_t part = (size_t)num_hub_seg * ((in + 3) & ~3);
    rc = aggregate_hub_rows(a, hub_seg, num_hub_seg, wsf, part * sizeof(float), st);
    if (rc) return rc;
    float* tmp = wsf + part;
    float* ytmp = tmp + (size_t)num_hub_seg * in;
    hub_rows_gather_kernel<<<(unsigned)num_hub_seg, 64, 0, st>>>(h0, ldh, hub_seg, num_hub_seg, in, tmp);
    KAGNN_LAUNCH_CHECK();
    rc = kan_sparse_fwd(tmp, in, num_hub_seg, knots, in, out, G, K, pack, ytmp, out, nullptr, 0, nullptr, nullptr, st);
    if (rc) return rc;
    hub_rows_scatter_kernel<<<(unsigned)num_hub_seg, 64, 0, st>>>(ytmp, hub_seg, num_hub_seg, out, y, ldy);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// col_mean / col_m2 (both or neither; [out]): also leave the column moments of y there (kan_sparse_fwd_moments_ok)
int kan_sparse_fwd(const float* x, long ldx, long N, const float* knots, int in, int out, int G, int K,
                   const void* pack, float* y, long ldy, void* ws, size_t ws_bytes, float* col_mean, float* col_m2,
                   hipStream_t st) {
    const int nk = G + 2 * K + 1, sh = sp_sh(G + K), blk = sp_out_blk(in << sh, out);
    const size_t stride = sp_blk_bytes(in, in << sh, min(out, blk));
    for (int b = 0; b * blk < out; ++b) {
        const int ob = min(blk, out - b * blk), OT = cdiv(ob, 32);
        const unsigned char* p = static_cast<const unsigned char*>(pack) + b * stride;
        float* yb = y + b * blk;
        int rc;
        float* cm = col_mean ? col_mean + b * blk : nullptr;
        float* cq = col_mean ? col_m2 + b * blk : nullptr;
        const bool narrow = (in << sh) <= 32;
        if (OT == 4) {                                     // wide block (sp_out_blk: whole 128-output blocks, never narrow)
#define LW(SS, MM) launch_sparse<4, SS, MM, false>(x, ldx, N, in, knots, nk, p, yb, ldy, ob, static_cast<float*>(ws), ws_bytes, cm, cq, st)
            if (col_mean && !sh) return fail(KAGNN_ERR_UNSUPPORTED, "%s: no column moments from the wide block of a <= 8-coefficient layer (kan_sparse_fwd_moments_ok)", "kan_sparse_fwd");
            rc = col_mean ? LW(true, true) : (sh ? LW(true, false) : LW(false, false));
#undef LW
            if (rc) return rc;
            continue;
        }
#define LL(OO, SS, MM, NN) launch_sparse<OO, SS, MM, NN>(x, ldx, N, in, knots, nk, p, yb, ldy, ob, static_cast<float*>(ws), ws_bytes, cm, cq, st)
#define L(OO, SS, MM) (narrow ? LL(OO, SS, MM, true) : LL(OO, SS, MM, false))
        if (col_mean) {
            if (sh) rc = OT == 1 ? L(1, true, true) : L(2, true, true);
            else rc = OT == 1 ? L(1, false, true) : L(2, false, true);
        } else {
            if (sh) rc = OT == 1 ? L(1, true, false) : L(2, true, false);
            else rc = OT == 1 ? L(1, false, false) : L(2, false, false);
        }
#undef LL
#undef L
        if (rc) return rc;
    }
    return KAGNN_OK;
}

// ------------------------------------------------------------------ input given as column blocks (skip-concat read-out)
// blocks: cubic layer of <= 8 coefficients, every block a whole number of 64-feature chunks, at most 8 chunks in all,
// 16-byte aligned rows (leading dimensions: multiples of 4, <= 7680)
bool kan_sparse_fwd_parts_ok(const int* widths, int nparts, int in, int out, int G, int K) {
    if (!kan_sparse_fwd_ok(in, out, G, K) || sp_sh(G + K) || nparts < 1 || nparts > 8 || in > 8 * kSpCF) return false;
    if (sp_out_blk(in, out) != kSpOutBlk) return false;     // (a wide layer's pack has 128-output blocks; read-outs are narrow)
    int sum = 0;
    for (int i = 0; i < nparts; ++i) {
        if (widths[i] <= 0 || widths[i] % kSpCF) return false;
        sum += widths[i];
    }
    return sum == in;
}

template <int OT, bool HALF = false>
static int launch_sparse_parts(const SpParts& xp, long N, int in, const float* knots, int nknots, const unsigned char* pack,
                               float* y, long ldy, int out, float* ws, size_t ws_bytes, hipStream_t st) {
    if constexpr (!HALF) {
        if (g_half_products) return launch_sparse_parts<OT, true>(xp, N, in, knots, nknots, pack, y, ldy, out, ws, ws_bytes, st);
    }
    const size_t lds = kLdsHdr + sparse_fwd_chunk_bytes(OT);
    static unsigned long long configured = 0;          // (per device: common.h)
    if (auto first_use_ = first_use_on_this_device(configured)) {
        KAGNN_HIP(hipFuncSetAttribute((const void*)kan_sparse_fwd_kernel<OT, false, false, false, -1, true, HALF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int nchunks = in / kSpCF, gx = sp_grid(N);
    const SpSplit p = sp_split_plan(N, nchunks);
    if (p.splits > 1) {
        if (!ws || ws_bytes < (size_t)p.splits * N * out * sizeof(float))
            return fail(KAGNN_ERR_ARG, "%s: workspace too small (see kagnn_kan_fwd_workspace_bytes)", "kan_sparse_fwd_parts");
        kan_sparse_fwd_kernel<OT, false, false, false, -1, true, HALF><<<dim3(gx, p.splits), 512, lds, st>>>(
            xp.p[0], xp.ld[0], N, in, knots, nknots, pack, nchunks, ws, out, out, p.cps, N * (long)out, nullptr, SpAgg{}, xp);
        KAGNN_LAUNCH_CHECK();
        sparse_sum_splits_kernel<<<cdiv(N * out, 256), 256, 0, st>>>(ws, p.splits, N, out, y, ldy);
        KAGNN_LAUNCH_CHECK();
        return KAGNN_OK;
    }
    kan_sparse_fwd_kernel<OT, false, false, false, -1, true, HALF><<<gx, 512, lds, st>>>(xp.p[0], xp.ld[0], N, in, knots, nknots, pack, nchunks, y, ldy,
                                                                                    out, nchunks, 0L, nullptr, SpAgg{}, xp);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

int kan_sparse_fwd_parts(const float* const* parts, const int* widths, const long* lds, int nparts, long N, const float* knots, int in, int out,
                         int G, int K, const void* pack, float* y, long ldy, void* ws, size_t ws_bytes, hipStream_t st,
                         const float* const* part_affine /* NULL, or per block NULL / [2][width]: scales then shifts */) {
    if (!kan_sparse_fwd_parts_ok(widths, nparts, in, out, G, K))
        return fail(KAGNN_ERR_UNSUPPORTED, "%s: blocks not covered (kagnn_kan_fwd_parts_ok)", "kan_sparse_fwd_parts");
    SpParts xp{};
    int c = 0;
    for (int i = 0; i < nparts; ++i) {
        if (!parts[i] || (reinterpret_cast<uintptr_t>(parts[i]) & 15) || lds[i] % 4 || lds[i] < widths[i] || lds[i] > 7680)
            return fail(KAGNN_ERR_ARG, "%s: block %d is null, not 16-byte aligned, or its leading dimension is not a multiple of 4 in [width, 7680]", "kan_sparse_fwd_parts", i);
        const float* af = part_affine ? part_affine[i] : nullptr;
        if (af && (reinterpret_cast<uintptr_t>(af) & 15))
            return fail(KAGNN_ERR_ARG, "%s: the affine of block %d is not 16-byte aligned", "kan_sparse_fwd_parts", i);
        for (int k = 0; k < widths[i] / kSpCF; ++k) {
            xp.p[c] = parts[i] + (long)k * kSpCF; xp.ld[c] = (int)lds[i];
            xp.sc[c] = af ? af + (long)k * kSpCF : nullptr;
            xp.sh[c] = af ? af + widths[i] + (long)k * kSpCF : nullptr;
            ++c;
        }
    }
    for (; c < 8; ++c) { xp.p[c] = xp.p[0]; xp.ld[c] = xp.ld[0]; xp.sc[c] = nullptr; xp.sh[c] = nullptr; }
    const int nk = G + 2 * K + 1;
    const size_t stride = sp_blk_bytes(in, in, min(out, kSpOutBlk));
    for (int b = 0; b * kSpOutBlk < out; ++b) {
        const int ob = min(kSpOutBlk, out - b * kSpOutBlk), OT = cdiv(ob, 32);
        const unsigned char* p = static_cast<const unsigned char*>(pack) + b * stride;
        const int rc = OT == 1 ? launch_sparse_parts<1>(xp, N, in, knots, nk, p, y + b * kSpOutBlk, ldy, ob, static_cast<float*>(ws), ws_bytes, st)
                               : launch_sparse_parts<2>(xp, N, in, knots, nk, p, y + b * kSpOutBlk, ldy, ob, static_cast<float*>(ws), ws_bytes, st);
        if (rc) return rc;
    }
    return KAGNN_OK;
}

}  // namespace kagnn
