// helpers shared by the split-precision KAN kernels (kan_split.hip, kan_split_bwd.hip)
#pragma once
#include "common.h"

namespace kagnn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }

constexpr int kHdrBytes = 256;           // pack header: [0] float 2^(e-10), [1] int e, [2] absmax bits
constexpr int kLdsHdr = 2560;            // LDS: knots (48 f32) @0, perm tables (2 windows x 32 x 16 B) @256, order-4 fix-up tables @1280
constexpr unsigned kWinBytes = 512;      // byte distance between the selector tables of window 0 and window 1
constexpr float kAScale = 1024.0f;       // bases / silu pre-scale (2^10)
constexpr int kOutBlk = 128;             // output columns per launch of the fwd / input-gradient kernels

__device__ __forceinline__ float wcat_s(const float* bw, const float* sw, const float* sc, int in,
                                        int out, int C, int o, int f, int c) {
    if (o >= out || f >= in || c > C) return 0.0f;
    if (c == C) return bw ? bw[(long)o * in + f] : 0.0f;
    float w = sw[((long)o * in + f) * C + c];
    return sc ? w * sc[(long)o * in + f] : w;
}

// Layers with 9..16 coefficients per input feature run as 2*in "virtual" features of 8 slots each (sh = 1):
// virtual feature v = (input feature v >> 1, slot window v & 1); window 1 holds coefficients 8..15 and no
// base weight.  slot 0..7 = coefficient inside the window, slot 8 = the base (SiLU) weight.
// sh == 2: the same two windows, but WINDOW-MAJOR inside blocks of 32 virtual features -- tile 2t holds window 0 of
// input features 16t .. 16t+15 and tile 2t+1 their window 1 (kan_split_dx_w2_kernel walks a lane's two windows in turn).
__device__ __forceinline__ float wcat_v(const float* bw, const float* sw, const float* sc, int in,
                                        int out, int C, int o, int v, int slot, int sh) {
    const int f = (sh == 2) ? (((v >> 5) << 4) | (v & 15)) : (v >> sh), w = (sh == 2) ? ((v >> 4) & 1) : (v & sh);
    if (slot == 8) return w == 0 ? wcat_s(bw, sw, sc, in, out, C, o, f, C) : 0.0f;
    const int c = slot + 8 * w;
    return c < C ? wcat_s(bw, sw, sc, in, out, C, o, f, c) : 0.0f;
}

// ---- packed W^T fragments for the input-gradient kernel (kan_split_bwd.hip); the item loop lives here so that
// the fused pack launch of kan_sparse_fwd.hip can run it next to the forward layout's
constexpr int kCTmax = 9;     // C + 1 <= 9 accumulators (8 spline coefficients + base); unused slots carry zero weights
__host__ __device__ inline int dx_q2(int out) { return out <= 32 ? 1 : (out <= 64 ? 2 : 4); }   // 32-wide k-steps

// pack_dx[ft16][c][q2][part][lane][8] : lane (f = lane&15, kg = lane>>4), j -> W'[o = 32*q2+8*kg+j][16*ft16+f][c]
__device__ __forceinline__ void pack_dx_items(const float* __restrict__ bw, const float* __restrict__ sw,
                                              const float* __restrict__ sc, int in, int out, int C, int Q2,
                                              unsigned char* __restrict__ pack, float wscale, long first, long step,
                                              int w2 = 0 /* window-major tiles (wcat_v sh == 2) */) {
    const int CT = kCTmax;
    const int sh = w2 ? 2 : (C > 8 ? 1 : 0), inv = w2 ? 32 * ((in + 15) / 16) : (in << sh);
    const long total = (long)((inv + 15) / 16) * CT * Q2 * 64;
    for (long i = first; i < total; i += step) {
        const int lane = i & 63; long r = i >> 6;
        const int q = r % Q2; r /= Q2;
        const int c = r % CT; const int ft = r / CT;
        const int f = 16 * ft + (lane & 15);
        _Float16* dh = reinterpret_cast<_Float16*>(pack + kHdrBytes + ((size_t)((ft * CT + c) * Q2 + q) * 2 + 0) * 1024 + lane * 16);
        _Float16* dl = reinterpret_cast<_Float16*>(pack + kHdrBytes + ((size_t)((ft * CT + c) * Q2 + q) * 2 + 1) * 1024 + lane * 16);
        for (int j = 0; j < 8; ++j) {
            const int o = 32 * q + 8 * (lane >> 4) + j;
            // slot 8 = base weight, slots 0..7 = spline coefficients of this (virtual) feature's window
            const float w = wcat_v(bw, sw, sc, in, out, C, o, f, c, sh) * wscale;
            const _Float16 h = (_Float16)w;
            dh[j] = h;
            dl[j] = (_Float16)(w - (float)h);
        }
    }
}

// ---- global memory through buffer descriptors: 32-bit byte offsets (no 64-bit VALU address math) and
// hardware bounds checking -- a load past `bytes` returns 0, a store past it is dropped, so rows >= N
// need neither clamping nor predication.
struct GBuf { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ GBuf gbuf(const void* p, long rows, long ld, int width) {
    const long bytes = rows > 0 ? ((rows - 1) * ld + width) * 4 : 0;
    GBuf b;
    b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (unsigned)bytes, 0x00020000);
    return b;
}
// the same window opened at row `row0` (wave-uniform): 64-bit base arithmetic happens once, on the scalar unit,
// and the per-lane offsets stay 32-bit however large the tensor is -- a kernel only ever needs the rows of its own
// tile (plus the next one it prefetches) inside the 4 GiB a descriptor can span.
__device__ __forceinline__ GBuf gbuf_at(const void* p, long rows, long ld, int width, long row0) {
    const long left = rows - row0;
    long bytes = left > 0 ? ((left - 1) * ld + width) * 4 : 0;
    if (bytes > 0xFFFFFFF0L) bytes = 0xFFFFFFF0L;
    GBuf b;
    b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(p)) + row0 * ld * 4, 0,
                                            (unsigned)bytes, 0x00020000);
    return b;
}
// the same with `es`-byte elements (bf16 output rows: es = 2)
__device__ __forceinline__ GBuf gbuf_at_es(const void* p, long rows, long ld, int width, long row0, int es) {
    const long left = rows - row0;
    long bytes = left > 0 ? ((left - 1) * ld + width) * es : 0;
    if (bytes > 0xFFFFFFF0L) bytes = 0xFFFFFFF0L;
    GBuf b;
    b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(p)) + row0 * ld * es, 0,
                                            (unsigned)bytes, 0x00020000);
    return b;
}
// fp32 -> bf16, round to nearest even (NaN stays NaN); 2-byte store with a wave-uniform offset on top of the lane's
__device__ __forceinline__ unsigned short bf16_bits(float f) {
    const unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ void gst16_s(const GBuf& b, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b16((short)bf16_bits(v), b.r, voff, soff, 0);
}
__device__ __forceinline__ float gld(const GBuf& b, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, off, 0, 0));
}
__device__ __forceinline__ void gld4(const GBuf& b, unsigned off, float* v) {
    // NB: __builtin_bit_cast(float, t[i]) on the vector elements silently reads element 0 for every i
    // (hipcc 7.2); go through the named components instead.
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(b.r, off, 0, 0);
    v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y); v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
}
__device__ __forceinline__ void gst(const GBuf& b, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), b.r, off, 0, 0);
}
// variants with a wave-uniform (SGPR) byte offset on top of the per-lane one: row strides and tile steps
// cost no VALU address arithmetic at all
__device__ __forceinline__ float gld_s(const GBuf& b, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, voff, soff, 0));
}
__device__ __forceinline__ void gld4_s(const GBuf& b, unsigned voff, unsigned soff, float* v) {
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(b.r, voff, soff, 0);
    v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y); v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
}
__device__ __forceinline__ void gst_s(const GBuf& b, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), b.r, voff, soff, 0);
}

__device__ __forceinline__ void gst4_s(const GBuf& b, unsigned voff, unsigned soff, const float* v) {
    u32x4 t;
    t.x = __float_as_uint(v[0]); t.y = __float_as_uint(v[1]); t.z = __float_as_uint(v[2]); t.w = __float_as_uint(v[3]);
    __builtin_amdgcn_raw_buffer_store_b128(t, b.r, voff, soff, 0);
}

// exponent e with max|W| * 2^-e in [2^9, 2^10)
__device__ __forceinline__ int scale_exp_from_max(float m) {
    if (!(m > 0.0f)) return 0;
    int ex;
    frexpf(m, &ex);                 // m = frac * 2^ex, frac in [0.5,1)  =>  m < 2^ex
    return ex - 10;
}

// One launch packs BOTH layouts: every workgroup first computes max|Wcat| over the whole (small, L2-resident)
// parameter set by itself -- cheaper than a separate reduction launch + memset -- then packs its share.
__device__ __forceinline__ float block_absmax_w(const float* __restrict__ bw, const float* __restrict__ sw,
                                const float* __restrict__ sc, int in, int out, int C, float* s_m /* LDS[17] */) {
    // flat, coalesced sweeps (a maximum does not depend on the order it is taken in: same bits as any other traversal).  The
    // (o, f)-outer / c-inner form this replaces issued C strided 4-byte loads per thread and iteration and made the pack
    // launches of the wide FastKAN layers 53 us each (256 x 256 x 4: config 5 packs 26 times per epoch).
    float m = 0.0f;
    const unsigned nof = (unsigned)out * (unsigned)in, nsw = nof * (unsigned)C;
    const unsigned tid = threadIdx.x, nt = blockDim.x;
    auto take = [&](float v) { v = fabsf(v); m = fmaxf(m, (v <= 3.0e38f) ? v : 0.0f); };
    if (bw) for (unsigned i = tid; i < nof; i += nt) take(bw[i]);
    if (!sc) {
        if ((reinterpret_cast<uintptr_t>(sw) & 15) == 0) {
            const unsigned n4 = nsw >> 2;
            const float4* s4 = reinterpret_cast<const float4*>(sw);
            for (unsigned i = tid; i < n4; i += nt) { const float4 v = s4[i]; take(v.x); take(v.y); take(v.z); take(v.w); }
            for (unsigned i = (n4 << 2) + tid; i < nsw; i += nt) take(sw[i]);
        } else {
            for (unsigned i = tid; i < nsw; i += nt) take(sw[i]);
        }
    } else {
        // with a spline_scaler: one (o, f) pair per thread and trip, C strided loads that all hit L2 -- cheaper than a flat sweep
        // with an integer division per element (measured both ways: fused_pack_batch_kernel at 64 x 64 x 8 19.5 vs 24.6 us,
        // fused_pack_kernel at 40 x 256 x 8 31 vs 53 us)
        for (unsigned of = tid; of < nof; of += nt) {
            const float scale = sc[of];
            for (int c = 0; c < C; ++c) take(sw[of * (unsigned)C + c] * scale);
        }
    }
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, s_m[i]);
        s_m[16] = m;
    }
    __syncthreads();
    return s_m[16];
}

// Large layers (the 256 x 256 layers of BASELINE config 5: ~0.5M weights): with every pack workgroup sweeping the WHOLE parameter
// set for the maximum, 256 workgroups read 0.5 GB out of L2 before the first pack item (47 us per pack launch).  Instead a
// launch of <= kAbsmaxBlocks workgroups leaves one partial maximum each in the pack header (floats [8, 8 + blocks)), and the pack
// workgroups fold those few values (self_scale == 2).  Same maximum, hence the same scale and the same packs, bit for bit.
constexpr int kAbsmaxBlocks = 56;                 // header = 64 floats; [0..2] taken, [7] = number of partials
constexpr long kAbsmaxTwoLaunchMin = 1L << 16;    // weights from which the two-launch form pays

template <int Unused = 0>     // (a template: one definition across the translation units that include this header)
__global__ __launch_bounds__(1024) void absmax_partials_kernel(const float* __restrict__ bw, const float* __restrict__ sw,
                                                               const float* __restrict__ sc, int in, int out, int C,
                                                               unsigned char* __restrict__ pack) {
    __shared__ float s_m[17];
    float m = 0.0f;
    const unsigned nof = (unsigned)out * (unsigned)in, nsw = nof * (unsigned)C;
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    auto take = [&](float v) { v = fabsf(v); m = fmaxf(m, (v <= 3.0e38f) ? v : 0.0f); };
    if (bw) for (unsigned i = tid; i < nof; i += nt) take(bw[i]);
    if (!sc) for (unsigned i = tid; i < nsw; i += nt) take(sw[i]);
    else for (unsigned of = tid; of < nof; of += nt) {
        const float scale = sc[of];
        for (int c = 0; c < C; ++c) take(sw[of * (unsigned)C + c] * scale);
    }
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, s_m[i]);
        reinterpret_cast<float*>(pack)[8 + blockIdx.x] = m;
        if (blockIdx.x == 0) reinterpret_cast<int*>(pack)[7] = (int)gridDim.x;
    }
}
__device__ __forceinline__ float header_absmax(const unsigned char* pack) {
    const int n = reinterpret_cast<const int*>(pack)[7];
    float m = 0.0f;
    for (int i = 0; i < n; ++i) m = fmaxf(m, reinterpret_cast<const float*>(pack)[8 + i]);
    return m;
}
inline int launch_absmax_partials(const float* bw, const float* sw, const float* sc, int in, int out, int C, void* pack, hipStream_t st) {
    const long n = (long)in * out * C;
    const int blocks = (int)max(1L, min((long)kAbsmaxBlocks, n / 4096));
    absmax_partials_kernel<0><<<blocks, 1024, 0, st>>>(bw, sw, sc, in, out, C, static_cast<unsigned char*>(pack));
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// selector tables for v_perm_b32: entry t of window w (shift sh = t - 4 - 8w halfs) holds 4 selectors; output
// half s of the 8-slot window takes payload half s-sh (payload = 4 halfs in {p1:p0}), zero when out of range.
// Layout: [window 2][t 32][q 4] dwords; all callers run >= 256 threads.
__device__ __forceinline__ void build_perm_table(unsigned* tbl /* LDS, 2*32*4 */, int tid) {
    if (tid < 256) {
        const int w = tid >> 7, t = (tid >> 2) & 31, q = tid & 3, sh = t - 4 - 8 * w;
        unsigned sel = 0;
        for (int hh = 0; hh < 2; ++hh) {
            const int r = 2 * q + hh - sh;
            const unsigned b = (r >= 0 && r <= 3) ? (unsigned)((2 * r) | ((2 * r + 1) << 8)) : 0x0c0cu;
            sel |= b << (16 * hh);
        }
        tbl[tid] = sel;
    }
}

// order-4 splines have FIVE non-zero bases: the fifth payload half lives in a third dword and is merged with a
// second v_perm_b32 whose selectors keep the bytes placed so far and overwrite the slot that takes payload half 4.
// Same [window][t][q] layout, 1 KiB after the main tables.
constexpr unsigned kFixBytes = 1024;
__device__ __forceinline__ void build_perm_fix_table(unsigned* tbl /* LDS, main tables */, int tid) {
    if (tid < 256) {
        const int w = tid >> 7, t = (tid >> 2) & 31, q = tid & 3, sh = t - 4 - 8 * w;
        unsigned sel = 0;
        for (int hh = 0; hh < 2; ++hh) {
            const int r = 2 * q + hh - sh;
            const unsigned b = (r == 4) ? 0x0504u : (unsigned)((2 * hh) | ((2 * hh + 1) << 8));
            sel |= b << (16 * hh);
        }
        tbl[256 + tid] = sel;
    }
}

__device__ __forceinline__ unsigned pk_f16_rtz(float a, float b) {
    auto v = __builtin_amdgcn_cvt_pkrtz(a, b);     // two fp16, round toward zero
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float f16lo_to_f32(unsigned p) {
    return (float)__builtin_bit_cast(_Float16, (unsigned short)(p & 0xffffu));
}
__device__ __forceinline__ float f16hi_to_f32(unsigned p) {
    return (float)__builtin_bit_cast(_Float16, (unsigned short)(p >> 16));
}
// v - (float)half of p, in ONE instruction: v_fma_mix_f32 reads the fp16 half directly (op_sel picks it).  Written as
// fma(half, -1, v) hipcc folds the constant and emits v_cvt_f32_f16 + v_sub_f32; with a -1.0 it cannot see through (an SGPR
// set by a side-effect-free asm: merged and hoisted like any other pure value) it forms the mixed-precision fma itself at
// every site -- schedulable like any other instruction, and without the s_nop it pads every inline-asm statement with
// (84 of them in the forward kernel when these were asm v_fma_mix_f32 statements).
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float opaque_minus_one() {
    float m;
    asm("s_mov_b32 %0, 0xbf800000" : "=s"(m));
    return m;
}
__device__ __forceinline__ float sub_f16lo(float v, unsigned p) {
    return __builtin_fmaf((float)__builtin_bit_cast(f16x2_t, p)[0], opaque_minus_one(), v);
}
__device__ __forceinline__ float sub_f16hi(float v, unsigned p) {
    return __builtin_fmaf((float)__builtin_bit_cast(f16x2_t, p)[1], opaque_minus_one(), v);
}

// 8 fp32 values (already scaled into fp16 range) -> hi and lo fp16 fragments
__device__ __forceinline__ void split_f16x2(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned h = pk_f16_rtz(v[2 * q], v[2 * q + 1]);
        hi[q] = h;
        lo[q] = pk_f16_rtz(sub_f16lo(v[2 * q], h), sub_f16hi(v[2 * q + 1], h));
    }
}
// The same with the residual as an inline-asm v_fma_mix_f32 statement: the input-gradient kernels keep this form (their gy
// prologue measured 2.5 % faster with the statements pinned where they are written than with the compiler-scheduled form
// above; the forward and the weight-gradient kernels the other way round, profiles/r02_experiments.md).
__device__ __forceinline__ void split_f16x2_asm(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned h = pk_f16_rtz(v[2 * q], v[2 * q + 1]);
        hi[q] = h;
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(v[2 * q]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(v[2 * q + 1]));
        lo[q] = pk_f16_rtz(r0, r1);
    }
}

// ---- KAGNN_PREC_HALF (round 5; build-defined reduced-precision mode for BASELINE config 2): ONE fp16 product per fp32 product.
// Every operand is evaluated in fp32 and rounded ONCE, to nearest even (v_cvt_pk_f16_f32: gfx950 has the packed RNE
// conversion), after the same exact power-of-two pre-scale as the hi part of the split mode -- no `lo` operands, so a third of
// the matrix-core work and about half of the conversion / placement VALU work.  fp32 accumulation as before.  The HALF
// instantiations of the three KAN kernels read only the `hi` fragments of the split mode's weight packs (the packs' hi parts
// are already RNE roundings of w * 2^-e).  The thread-local flag is set by the C entry points (api.hip: ModeScope) for the
// duration of a call with mode == KAGNN_PREC_HALF; the launchers below pick the HALF instantiation where one exists and the
// three-product kernels (more accurate, never less) elsewhere.
extern thread_local bool g_half_products;
__device__ __forceinline__ unsigned pk_f16_rne(float a, float b) {
    const f16x2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, v);
}
// 8 fp32 values (already scaled into fp16 range) -> ONE fp16 fragment, round to nearest even
__device__ __forceinline__ void round_f16x2(const float (&v)[8], u32x4& hi) {
#pragma unroll
    for (int q = 0; q < 4; ++q) hi[q] = pk_f16_rne(v[2 * q], v[2 * q + 1]);
}

// drop a 4-half payload {p1:p0} (hi and lo parts) into an 8-slot window with the selectors of one table entry
__device__ __forceinline__ void frag3_place_fwd(const u32x4& sel, unsigned h0, unsigned h1, unsigned l0,
                                                unsigned l1, u32x4& ahi, u32x4& alo) {
    ahi[0] = __builtin_amdgcn_perm(h1, h0, sel[0]); ahi[1] = __builtin_amdgcn_perm(h1, h0, sel[1]);
    ahi[2] = __builtin_amdgcn_perm(h1, h0, sel[2]); ahi[3] = __builtin_amdgcn_perm(h1, h0, sel[3]);
    alo[0] = __builtin_amdgcn_perm(l1, l0, sel[0]); alo[1] = __builtin_amdgcn_perm(l1, l0, sel[1]);
    alo[2] = __builtin_amdgcn_perm(l1, l0, sel[2]); alo[3] = __builtin_amdgcn_perm(l1, l0, sel[3]);
}

// ---- uniform-grid fast path (K == 3, the order every KAGNN config uses) -------------------------
// span index from arithmetic only: m = clamp(floor((x-g0)/h)), u = (x-g0)/h - m.  Right at a knot
// the arithmetic may pick the neighbouring span with u ~ 1 or ~ 0; cubic pieces join C2 so values
// and first derivatives agree to rounding.  `inside` uses the stored first/last knots, i.e. the
// reference's half-open support [knots[0], knots[last]).  Non-finite x gives u = NaN/Inf and the
// polynomials below turn that into NaN (0 * Inf = NaN), matching the reference.
struct FastGeom {
    float inv_h, c0;          // t = x*inv_h + c0
    float k_first, k_last;
    float last_span;          // (float)(nknots-2)
};
__device__ __forceinline__ FastGeom fast_geom(const float* knots, int nknots) {
    FastGeom g;
    g.inv_h = (float)(nknots - 1) / (knots[nknots - 1] - knots[0]);
    g.c0 = -knots[0] * g.inv_h;
    g.k_first = knots[0];
    g.k_last = knots[nknots - 1];
    g.last_span = (float)(nknots - 2);
    return g;
}
__device__ __forceinline__ void fast_span(float x, const FastGeom& g, int& m, float& u, bool& inside) {
    const float t = fmaf(x, g.inv_h, g.c0);
    const float tf = fminf(fmaxf(floorf(t), 0.0f), g.last_span);
    m = (int)tf;
    u = t - tf;
    inside = (x >= g.k_first) && (x < g.k_last);
}
// cubic pieces times w6 (= scale/6 inside the support, 0 outside): N[r] = B_{m-3+r}(x) * scale
__device__ __forceinline__ void cubic_bases(float u, float w6, float (&N)[4]) {
    const float u2 = u * u, om = 1.0f - u, uw = u * w6, ow = om * w6;
    N[0] = ow * (om * om);
    N[3] = uw * u2;
    N[1] = fmaf(uw, fmaf(u, 3.0f, -6.0f) * u, 4.0f * w6);              // (3u^3 - 6u^2 + 4) w6
    N[2] = fmaf(uw, fmaf(fmaf(u, -3.0f, 3.0f), u, 3.0f), w6);          // (-3u^3 + 3u^2 + 3u + 1) w6
}
// d/dx of the same pieces times wd (= inv_h/2 inside, 0 outside)
__device__ __forceinline__ void cubic_dbases(float u, float wd, float (&dN)[4]) {
    const float om = 1.0f - u, u2 = u * u;
    dN[0] = -(om * om) * wd;
    dN[3] = u2 * wd;
    dN[1] = fmaf(u2, 3.0f, u * -4.0f) * wd;
    dN[2] = fmaf(u2, -3.0f, fmaf(u, 2.0f, 1.0f)) * wd;
}

// bases (scaled by 2^10) of one scalar -> hi / lo A fragments (8 fp16 each) for window slots 0..7
template <int K>
__device__ __forceinline__ void make_spline_frag(float x, const float* __restrict__ knots,
                                                 const unsigned* __restrict__ tbl,
                                                 const SplineGeom& g, u32x4& ahi, u32x4& alo,
                                                 unsigned woff = 0) {
    float N[K + 1], dummy[K + 1];
    const int m = bspline_local<K, false>(x, knots, g, N, dummy);
    float n0 = N[0] * kAScale, n1 = N[1] * kAScale;
    float n2 = (K >= 2) ? N[K >= 2 ? 2 : 0] * kAScale : 0.0f;
    float n3 = (K >= 3) ? N[K >= 3 ? 3 : 0] * kAScale : 0.0f;
    const unsigned h0 = pk_f16_rtz(n0, n1), h1 = pk_f16_rtz(n2, n3);
    const unsigned l0 = pk_f16_rtz(sub_f16lo(n0, h0), sub_f16hi(n1, h0));
    const unsigned l1 = pk_f16_rtz(sub_f16lo(n2, h1), sub_f16hi(n3, h1));
    int t = m - K + 4;
    t = t < 0 ? 0 : (t > 31 ? 31 : t);
    const u32x4 sel = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(tbl) + woff + 16 * t);
    ahi[0] = __builtin_amdgcn_perm(h1, h0, sel[0]); ahi[1] = __builtin_amdgcn_perm(h1, h0, sel[1]);
    ahi[2] = __builtin_amdgcn_perm(h1, h0, sel[2]); ahi[3] = __builtin_amdgcn_perm(h1, h0, sel[3]);
    alo[0] = __builtin_amdgcn_perm(l1, l0, sel[0]); alo[1] = __builtin_amdgcn_perm(l1, l0, sel[1]);
    alo[2] = __builtin_amdgcn_perm(l1, l0, sel[2]); alo[3] = __builtin_amdgcn_perm(l1, l0, sel[3]);
    if constexpr (K >= 4) {                          // fifth basis (see build_perm_fix_table)
        const float n4 = N[K >= 4 ? 4 : 0] * kAScale;
        const unsigned h2 = pk_f16_rtz(n4, 0.0f);
        const unsigned l2 = pk_f16_rtz(sub_f16lo(n4, h2), 0.0f);
        const u32x4 fix = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(tbl) + kFixBytes + woff + 16 * t);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ahi[q] = __builtin_amdgcn_perm(h2, ahi[q], fix[q]);
            alo[q] = __builtin_amdgcn_perm(l2, alo[q], fix[q]);
        }
    }
}

// K == 3 fast path: arithmetic span, closed-form cubic pieces, no knot lookups
__device__ __forceinline__ void make_spline_frag3(float x, const unsigned* __restrict__ tbl,
                                                  const FastGeom& g, u32x4& ahi, u32x4& alo,
                                                  unsigned woff = 0) {
    int m; float u; bool inside;
    fast_span(x, g, m, u, inside);
    float N[4];
    cubic_bases(u, inside ? (kAScale / 6.0f) : 0.0f, N);
    const unsigned h0 = pk_f16_rtz(N[0], N[1]), h1 = pk_f16_rtz(N[2], N[3]);
    const unsigned l0 = pk_f16_rtz(sub_f16lo(N[0], h0), sub_f16hi(N[1], h0));
    const unsigned l1 = pk_f16_rtz(sub_f16lo(N[2], h1), sub_f16hi(N[3], h1));
    const u32x4 sel = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(tbl) + woff + 16 * (m + 1));   // m - 3 + 4, m in [0, 30]
    ahi[0] = __builtin_amdgcn_perm(h1, h0, sel[0]); ahi[1] = __builtin_amdgcn_perm(h1, h0, sel[1]);
    ahi[2] = __builtin_amdgcn_perm(h1, h0, sel[2]); ahi[3] = __builtin_amdgcn_perm(h1, h0, sel[3]);
    alo[0] = __builtin_amdgcn_perm(l1, l0, sel[0]); alo[1] = __builtin_amdgcn_perm(l1, l0, sel[1]);
    alo[2] = __builtin_amdgcn_perm(l1, l0, sel[2]); alo[3] = __builtin_amdgcn_perm(l1, l0, sel[3]);
}

// two scalars at once: the span arithmetic and the cubic pieces run on packed fp32 (v_pk_fma_f32 / v_pk_mul_f32
// process both elements in one issue slot), conversions and placement per element as above
__device__ __forceinline__ void make_spline_frag3_pair(float x0, float x1, const unsigned* __restrict__ tbl,
                                                       const FastGeom& g, u32x4& ahi0, u32x4& alo0,
                                                       u32x4& ahi1, u32x4& alo1, unsigned woff = 0) {
    const f32x2 x = {x0, x1};
#ifdef KAGNN_ABLATE_SHARED_EXPANSION
    // TIMING-ONLY ablation (wrong results): span and cubic pieces "received from elsewhere" -- see kan_split_dx_kernel
    const int m0 = (int)(__float_as_uint(x0) & 7u), m1 = (int)(__float_as_uint(x1) & 7u);
    const f32x2 N0 = x, N1 = x, N2 = x, N3 = x;
    (void)g;
#else
    const f32x2 t = fma2(x, splat2(g.inv_h), splat2(g.c0));
    const f32x2 tf = {fminf(fmaxf(floorf(t.x), 0.0f), g.last_span), fminf(fmaxf(floorf(t.y), 0.0f), g.last_span)};
    const int m0 = (int)tf.x, m1 = (int)tf.y;
    const f32x2 u = t - tf;
    const f32x2 w6 = {(x0 >= g.k_first && x0 < g.k_last) ? kAScale / 6.0f : 0.0f,
                      (x1 >= g.k_first && x1 < g.k_last) ? kAScale / 6.0f : 0.0f};
    const f32x2 u2 = u * u, om = splat2(1.0f) - u, uw = u * w6, ow = om * w6;
    const f32x2 N0 = ow * (om * om);
    const f32x2 N3 = uw * u2;
    const f32x2 N1 = fma2(uw, fma2(u, splat2(3.0f), splat2(-6.0f)) * u, splat2(4.0f) * w6);
    const f32x2 N2 = fma2(uw, fma2(fma2(u, splat2(-3.0f), splat2(3.0f)), u, splat2(3.0f)), w6);
#endif
    {
        const unsigned h0 = pk_f16_rtz(N0.x, N1.x), h1 = pk_f16_rtz(N2.x, N3.x);
        const unsigned l0 = pk_f16_rtz(sub_f16lo(N0.x, h0), sub_f16hi(N1.x, h0));
        const unsigned l1 = pk_f16_rtz(sub_f16lo(N2.x, h1), sub_f16hi(N3.x, h1));
        const u32x4 sel = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(tbl) + woff + 16 * (m0 + 1));
        frag3_place_fwd(sel, h0, h1, l0, l1, ahi0, alo0);
    }
    {
        const unsigned h0 = pk_f16_rtz(N0.y, N1.y), h1 = pk_f16_rtz(N2.y, N3.y);
        const unsigned l0 = pk_f16_rtz(sub_f16lo(N0.y, h0), sub_f16hi(N1.y, h0));
        const unsigned l1 = pk_f16_rtz(sub_f16lo(N2.y, h1), sub_f16hi(N3.y, h1));
        const u32x4 sel = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(tbl) + woff + 16 * (m1 + 1));
        frag3_place_fwd(sel, h0, h1, l0, l1, ahi1, alo1);
    }
}

// HALF: the same two scalars, rounded once -- hi fragments only
__device__ __forceinline__ void make_spline_frag3_pair_h(float x0, float x1, const unsigned* __restrict__ tbl,
                                                         const FastGeom& g, u32x4& ahi0, u32x4& ahi1, unsigned woff = 0) {
    const f32x2 x = {x0, x1};
    const f32x2 t = fma2(x, splat2(g.inv_h), splat2(g.c0));
    const f32x2 tf = {fminf(fmaxf(floorf(t.x), 0.0f), g.last_span), fminf(fmaxf(floorf(t.y), 0.0f), g.last_span)};
    const int m0 = (int)tf.x, m1 = (int)tf.y;
    const f32x2 u = t - tf;
    const f32x2 w6 = {(x0 >= g.k_first && x0 < g.k_last) ? kAScale / 6.0f : 0.0f,
                      (x1 >= g.k_first && x1 < g.k_last) ? kAScale / 6.0f : 0.0f};
    const f32x2 u2 = u * u, om = splat2(1.0f) - u, uw = u * w6, ow = om * w6;
    const f32x2 N0 = ow * (om * om);
    const f32x2 N3 = uw * u2;
    const f32x2 N1 = fma2(uw, fma2(u, splat2(3.0f), splat2(-6.0f)) * u, splat2(4.0f) * w6);
    const f32x2 N2 = fma2(uw, fma2(fma2(u, splat2(-3.0f), splat2(3.0f)), u, splat2(3.0f)), w6);
    {
        const unsigned h0 = pk_f16_rne(N0.x, N1.x), h1 = pk_f16_rne(N2.x, N3.x);
        const u32x4 sel = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(tbl) + woff + 16 * (m0 + 1));
        ahi0[0] = __builtin_amdgcn_perm(h1, h0, sel[0]); ahi0[1] = __builtin_amdgcn_perm(h1, h0, sel[1]);
        ahi0[2] = __builtin_amdgcn_perm(h1, h0, sel[2]); ahi0[3] = __builtin_amdgcn_perm(h1, h0, sel[3]);
    }
    {
        const unsigned h0 = pk_f16_rne(N0.y, N1.y), h1 = pk_f16_rne(N2.y, N3.y);
        const u32x4 sel = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(tbl) + woff + 16 * (m1 + 1));
        ahi1[0] = __builtin_amdgcn_perm(h1, h0, sel[0]); ahi1[1] = __builtin_amdgcn_perm(h1, h0, sel[1]);
        ahi1[2] = __builtin_amdgcn_perm(h1, h0, sel[2]); ahi1[3] = __builtin_amdgcn_perm(h1, h0, sel[3]);
    }
}

// ---- the same K == 3 expansion in three pieces, so a caller can software-pipeline it under MFMAs:
//   frag3_index   : table index of the span (needs only t)           -> issue the LDS table read early
//   frag3_payload : cubic pieces, fp16 hi/lo payloads (pure VALU)     -> overlaps the MFMAs in flight
//   frag3_place   : 8 x v_perm_b32 with the selectors read from LDS
// Span / support handling is folded into the table: idx = floor(t) + 1 clamped to [0, 31]; entries
// outside [1, nspans] hold all-zero selectors (x outside [knots[0], knots[last]) -> all bases 0; at the
// two boundary knots every existing basis is 0 anyway, so an ulp of disagreement with the reference's
// half-open test is harmless).  NaN -> idx 1 with NaN payload -> NaN, +-Inf -> Inf*0 = NaN payload
// guarded below.
struct Frag3Geom { float inv_h, c1; };            // t' = x*inv_h + c1  with c1 = 1 - knots[0]*inv_h
__device__ __forceinline__ Frag3Geom frag3_geom(const float* knots, int nknots) {
    Frag3Geom g;
    g.inv_h = (float)(nknots - 1) / (knots[nknots - 1] - knots[0]);
    g.c1 = 1.0f - knots[0] * g.inv_h;
    return g;
}
// table with support folded in: entry i (1 <= i <= nspans) = shift (i-1) - 3; everything else zero
__device__ __forceinline__ void build_perm_table3(unsigned* tbl /* LDS, 2*32*4 */, int tid, int nknots) {
    if (tid < 256) {
        const int w = tid >> 7, t = (tid >> 2) & 31, q = tid & 3, sh = t - 4 - 8 * w;
        unsigned sel = 0;
        for (int hh = 0; hh < 2; ++hh) {
            const int r = 2 * q + hh - sh;
            const bool live = (t >= 1) && (t <= nknots - 1) && (r >= 0) && (r <= 3);
            const unsigned b = live ? (unsigned)((2 * r) | ((2 * r + 1) << 8)) : 0x0c0cu;
            sel |= b << (16 * hh);
        }
        tbl[tid] = sel;
    }
}
// NANSAFE: non-finite x selects a live table entry and a NaN payload (the reference's bases are NaN
// there).  The forward kernel does not need it: its SiLU branch already turns such rows into NaN.
template <bool NANSAFE>
__device__ __forceinline__ void frag3_index(float x, const Frag3Geom& g, float& u, unsigned& byte_off,
                                            unsigned woff = 0) {
    const float t = fmaf(x, g.inv_h, g.c1);
    u = __builtin_amdgcn_fractf(t);
    const int i = (int)floorf(t);                  // v_cvt_flr_i32_f32; NaN -> 0
    byte_off = (min((unsigned)i, 31u) << 4) + woff;   // negative -> huge unsigned -> 31 (a zero entry)
    if (NANSAFE) {
        const bool fin = fabsf(x) < __builtin_inff();
        byte_off = fin ? byte_off : 16u + woff;
        u = fin ? u : __builtin_nanf("");
    }
}
__device__ __forceinline__ void frag3_payload(float u, unsigned& h0, unsigned& h1, unsigned& l0,
                                              unsigned& l1) {
    float N[4];
    cubic_bases(u, kAScale / 6.0f, N);
    h0 = pk_f16_rtz(N[0], N[1]); h1 = pk_f16_rtz(N[2], N[3]);
    l0 = pk_f16_rtz(sub_f16lo(N[0], h0), sub_f16hi(N[1], h0));
    l1 = pk_f16_rtz(sub_f16lo(N[2], h1), sub_f16hi(N[3], h1));
}
// 16 * silu(x) for two scalars on packed fp32: x * rcp((1 + exp(-x)) / 16) -- bit-identical to (x * rcp(1 + exp(-x))) * 16
// (scaling by a power of two commutes with every rounding here), 3.5 issue slots per scalar instead of 6
__device__ __forceinline__ f32x2 silu16_pair(f32x2 x) {
    const f32x2 a = x * splat2(-1.4426950408889634f);
    const f32x2 e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    const f32x2 d = fma2(e, splat2(0.0625f), splat2(0.0625f));
    const f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    return x * r;
}

// two scalars' payloads at once: the cubic pieces on packed fp32 (one v_pk_mul_f32 / v_pk_fma_f32 per TWO scalars: these
// kernels are bound by instruction count, profiles/r03_experiments.md), conversions per scalar as above
__device__ __forceinline__ void frag3_payload_pair(float u0, float u1, unsigned (&h)[2][2], unsigned (&l)[2][2]) {
    const f32x2 u = {u0, u1}, w6 = splat2(kAScale / 6.0f);
    const f32x2 u2 = u * u, om = splat2(1.0f) - u, uw = u * w6, ow = om * w6;
    const f32x2 N0 = ow * (om * om);
    const f32x2 N3 = uw * u2;
    const f32x2 N1 = fma2(uw, fma2(u, splat2(3.0f), splat2(-6.0f)) * u, splat2(4.0f) * w6);
    const f32x2 N2 = fma2(uw, fma2(fma2(u, splat2(-3.0f), splat2(3.0f)), u, splat2(3.0f)), w6);
    h[0][0] = pk_f16_rtz(N0.x, N1.x); h[0][1] = pk_f16_rtz(N2.x, N3.x);
    l[0][0] = pk_f16_rtz(sub_f16lo(N0.x, h[0][0]), sub_f16hi(N1.x, h[0][0]));
    l[0][1] = pk_f16_rtz(sub_f16lo(N2.x, h[0][1]), sub_f16hi(N3.x, h[0][1]));
    h[1][0] = pk_f16_rtz(N0.y, N1.y); h[1][1] = pk_f16_rtz(N2.y, N3.y);
    l[1][0] = pk_f16_rtz(sub_f16lo(N0.y, h[1][0]), sub_f16hi(N1.y, h[1][0]));
    l[1][1] = pk_f16_rtz(sub_f16lo(N2.y, h[1][1]), sub_f16hi(N3.y, h[1][1]));
}
// HALF: the payloads of two scalars, rounded once (no lo parts)
__device__ __forceinline__ void frag3_payload_pair_h(float u0, float u1, unsigned (&h)[2][2]) {
    const f32x2 u = {u0, u1}, w6 = splat2(kAScale / 6.0f);
    const f32x2 u2 = u * u, om = splat2(1.0f) - u, uw = u * w6, ow = om * w6;
    const f32x2 N0 = ow * (om * om);
    const f32x2 N3 = uw * u2;
    const f32x2 N1 = fma2(uw, fma2(u, splat2(3.0f), splat2(-6.0f)) * u, splat2(4.0f) * w6);
    const f32x2 N2 = fma2(uw, fma2(fma2(u, splat2(-3.0f), splat2(3.0f)), u, splat2(3.0f)), w6);
    h[0][0] = pk_f16_rne(N0.x, N1.x); h[0][1] = pk_f16_rne(N2.x, N3.x);
    h[1][0] = pk_f16_rne(N0.y, N1.y); h[1][1] = pk_f16_rne(N2.y, N3.y);
}
__device__ __forceinline__ void frag3_place(const u32x4& sel, unsigned h0, unsigned h1, unsigned l0,
                                            unsigned l1, u32x4& ahi, u32x4& alo) {
    ahi[0] = __builtin_amdgcn_perm(h1, h0, sel[0]); ahi[1] = __builtin_amdgcn_perm(h1, h0, sel[1]);
    ahi[2] = __builtin_amdgcn_perm(h1, h0, sel[2]); ahi[3] = __builtin_amdgcn_perm(h1, h0, sel[3]);
    alo[0] = __builtin_amdgcn_perm(l1, l0, sel[0]); alo[1] = __builtin_amdgcn_perm(l1, l0, sel[1]);
    alo[2] = __builtin_amdgcn_perm(l1, l0, sel[2]); alo[3] = __builtin_amdgcn_perm(l1, l0, sel[3]);
}

template <int K>
__device__ __forceinline__ void spline_frag(float x, const float* __restrict__ knots,
                                            const unsigned* __restrict__ tbl, const SplineGeom& g,
                                            const FastGeom& fg, u32x4& ahi, u32x4& alo, unsigned woff = 0) {
    if constexpr (K == 3) make_spline_frag3(x, tbl, fg, ahi, alo, woff);
    else make_spline_frag<K>(x, knots, tbl, g, ahi, alo, woff);
}

// 8 fp32 values -> three truncated-bf16 fragments (v = v1 + v2 + v3 up to 2^-24)
__device__ __forceinline__ void split_bf16x3(const float (&v)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float a = v[2 * q], b = v[2 * q + 1];
        p1[q] = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
        a -= __uint_as_float(__float_as_uint(a) & 0xffff0000u);
        b -= __uint_as_float(__float_as_uint(b) & 0xffff0000u);
        p2[q] = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
        a -= __uint_as_float(__float_as_uint(a) & 0xffff0000u);
        b -= __uint_as_float(__float_as_uint(b) & 0xffff0000u);
        p3[q] = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
    }
}

__device__ __forceinline__ f32x16 mfma_f16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16_f16(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// ---- Gaussian RBF basis (FastKAN, fastkan.py:46-47) for the K == 0 instantiations of the split kernels:
// phi_g(z) = exp(-((z - c_g)/den)^2) = exp2(-t_g^2) with t_g = a*z - a*c_g, a = sqrt(log2 e)/den.
// z is the layer-normed input (fastkan.py:77-78) when ln_w != nullptr, else x itself.
struct RbfArgs {
    const float* centers;     // rbf.grid, num_grids device floats
    int ng;
    float a;                  // sqrt(log2 e) / denominator
    float k2;                 // d phi_g/dz = phi_g * t_g * k2,  k2 = -2 a ln 2
    const float* ln_w; const float* ln_b; const float* stats;   // stats[n] = (mean, rstd)
    const float* bias;        // forward: added to y (base_linear.bias), may be nullptr
    float* gz;                // input gradient: [N, in] gradient w.r.t. z when layernorm is on
    float* stats_out;         // forward, one-chunk layers: compute the LayerNorm row statistics HERE (from the rows the kernel
    float ln_eps;             //   loads anyway) and store them for the backward, instead of reading `stats`; nullptr: read
    float* colpart;           // weight gradient: [slabs][outP] column sums of gy per row slab (the base bias gradient rides in
                              // the kernel that reads gy anyway); nullptr: not wanted
    // (B-spline layers too, round 4) the layer input exists only as  x_affine[f] * x[.][f] + x_affine[in + f]  -- a BatchNorm1d
    // output that was never written: the XAFF instantiations of the input- and weight-gradient kernels apply it to the rows
    // they load
    const float* x_affine = nullptr;
    // (input-gradient kernel, XST instantiation, round 4) the column statistics of the gradient rows it stores, for the backward of
    // the BatchNorm1d whose folded output this layer read: st_partial[workgroup][2][in] receives sum gx and sum gx * xhat over
    // the workgroup's rows, xhat = (x - st_mean) * st_rstd on the RAW rows the kernel loads (x = that norm's input)
    const float* st_mean = nullptr; const float* st_rstd = nullptr; float* st_partial = nullptr;
};

// ca[g] = a * c_{8*window+g} (wave-uniform; slots >= num_grids repeat the last centre -- their packed weights
// are zero)
__device__ __forceinline__ void rbf_centers(const RbfArgs& rb, float (&ca)[8], int window = 0) {
#pragma unroll
    for (int g = 0; g < 8; ++g) ca[g] = rb.centers[min(8 * window + g, rb.ng - 1)] * rb.a;
}

// 8 RBF values of one scalar, scaled by 2^10, as fp16 hi / lo fragments
__device__ __forceinline__ void make_rbf_frag(float z, float a, const float (&ca)[8], u32x4& hi, u32x4& lo) {
    const float t0 = z * a;
    float v[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const float t = t0 - ca[g];
        v[g] = __builtin_amdgcn_exp2f(fmaf(-t, t, 10.0f));
    }
    split_f16x2(v, hi, lo);
}

// Cubic case of the same contraction with half the selects.  The window c0..c0+3 (c0 = m - 3) holds exactly one slot
// of every residue class mod 4, so the candidate for residue rho is d[rho], d[rho+4] or nothing (the coefficient does
// not exist): ONE v_perm_b32 per residue with a selector from a 16-entry LDS table indexed by m (dword selectors
// "low source" / "high source" / zero).  The four survivors then only need a ROTATION by c0 & 3 to line up with
// dN[0..3]: two more v_perm stages.  12 v_perm_b32, no compares, instead of 27 selects + 5 compares.
// Entry mm (0..15) = 8 dwords: [0..3] the residue selectors, [4] / [5] the two rotation-stage selectors of b = (mm + 1) & 3
// (round 3: computed per scalar they were 2 and + 2 compare + 2 select = 6 of the input gradient's ~45 VALU per scalar)
constexpr int kBarrelDw = 8;
__device__ __forceinline__ void build_barrel_table(unsigned* tbl /* LDS, 16*8 */, int tid) {
    if (tid < 128) {
        const int mm = tid >> 3, k = tid & 7, c0 = mm - 3;
        unsigned v = 0x03020100u;
        if (k < 4) {
            const int rho = k, t = c0 + ((rho - c0) & 3);   // the slot of residue rho inside [c0, c0+3]
            v = (mm >= 15) ? 0x0c0c0c0cu : (t == rho ? 0x03020100u : (t == rho + 4 ? 0x07060504u : 0x0c0c0c0cu));
        } else if (k < 6) {
            const unsigned bb = (unsigned)(mm + 1) & 3u;
            v = 0x03020100u + ((k == 4 ? (bb & 1u) : (bb >> 1)) * 0x04040404u);
        }
        tbl[tid] = v;
    }
}
// sel = dwords 0..3 of the entry, rot = dwords 4, 5
__device__ __forceinline__ float barrel_dot3(const float (&d)[8], const u32x4& sel, unsigned selA, unsigned selB,
                                             const float (&dN)[4]) {
    unsigned s4[4], r1[4];
#pragma unroll
    for (int rho = 0; rho < 4; ++rho)
        s4[rho] = __builtin_amdgcn_perm(__float_as_uint(d[rho + 4]), __float_as_uint(d[rho]), sel[rho]);
    // rotation e[r] = s4[(b + r) & 3], b = c0 & 3, as two v_perm stages: 0x03020100 keeps the low source, 0x07060504 takes the high one
#pragma unroll
    for (int i = 0; i < 4; ++i) r1[i] = __builtin_amdgcn_perm(s4[(i + 1) & 3], s4[i], selA);
    float e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = __uint_as_float(__builtin_amdgcn_perm(r1[(i + 2) & 3], r1[i], selB));
    return fmaf(e[3], dN[3], fmaf(e[2], dN[2], fmaf(e[1], dN[1], e[0] * dN[0])));
}

// wave-wide maximum of non-negative, NaN-free floats without touching LDS: DPP swaps inside each row of 16
// lanes (quad_perm, row_half_mirror, row_mirror), then four v_readlane + scalar max (bit patterns of
// non-negative floats order like unsigned integers).  __shfl_xor would cost six dependent ds_bpermute round trips.
__device__ __forceinline__ float wave_max_nonneg(float v) {
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0xB1, 0xf, 0xf, true)));   // quad_perm [1,0,3,2]
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x4E, 0xf, 0xf, true)));   // quad_perm [2,3,0,1]
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x141, 0xf, 0xf, true)));  // row_half_mirror
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x140, 0xf, 0xf, true)));  // row_mirror
    const unsigned u = __float_as_uint(v);
    const unsigned a = __builtin_amdgcn_readlane(u, 0), b = __builtin_amdgcn_readlane(u, 16);
    const unsigned c = __builtin_amdgcn_readlane(u, 32), d = __builtin_amdgcn_readlane(u, 48);
    return __uint_as_float(max(max(a, b), max(c, d)));
}

// power-of-two scale that brings |v| <= m below 2^10 (exact); 1 for m == 0 / non-finite
__device__ __forceinline__ int exp_for_max(float m) {
    if (!(m > 0.0f) || !(m <= 3.0e38f)) return 10;
    int ex;
    frexpf(m, &ex);
    return ex;                      // m < 2^ex ; scale = 2^(10-ex)
}

// One wave copies 1 KiB global -> LDS without touching registers (global_load_lds_dwordx4: lane l's 16 bytes land at
// lds_addr + 16 l; layout measured by tools/probes/lds_dma_probe.hip).  Inline asm on purpose: behind the builtin the
// compiler cannot tell which LDS bytes an in-flight copy targets and puts s_waitcnt vmcnt(0) in front of EVERY later
// ds_read, i.e. no overlap.  Completion is awaited explicitly (lds_dma_wait) before the barrier that publishes the buffer;
// the compiler's own vmcnt bookkeeping for the x loads stays safe (memory returns in order: it can only over-wait).
// M0 is written here and read by nothing else in these kernels.
__device__ __forceinline__ void lds_dma_1k(const unsigned char* g_lane, unsigned lds_addr /* wave-uniform */) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g_lane), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }


}  // namespace kagnn
