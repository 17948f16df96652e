// FastKAN layer on gfx950: LayerNorm row statistics + Gaussian RBF expansion evaluated in
// registers and fed straight into v_mfma_f32_32x32x2_f32 -- the [N, in, num_grids] basis tensor
// of the reference never exists.
//
// Reference behaviour replaced: node_classification_clean/fastkan.py:76-85 (FastKANLayer.forward:
// LayerNorm :77-78, RadialBasisFunction.forward :46-47, SplineLinear :81, base_linear(silu(x))
// :82-84) and the autograd backward of those.  Fragment mapping is the one of kan_fp32.hip:
// the two k-lanes of an MFMA are two input features, one MFMA per grid point g (plus one for
// the SiLU base branch).
#include "split_common.h"

namespace kagnn {

// split-precision kernels shared with the B-spline layer (K == 0 selects the RBF basis)
bool kan_split_fwd_ok(int in, int out, int G, int K);
size_t kan_split_pack_fwd_bytes(int in, int out, int C);
size_t kan_split_pack_dx_bytes(int in, int out, int C, int K);
size_t kan_split_dw_ws_bytes(long N, int in, int out, int C, int K);
void kan_split_dw_slabs(long N, int in, int out, int C, int K, long* slabs, long* outP);
int kan_split_pack_fwd_noscale(const float*, const float*, const float*, int, int, int, void*, hipStream_t);
int kan_split_pack_dx_noscale(const float*, const float*, const float*, int, int, int, int, void*, hipStream_t);
int kan_split_fwd_any(const float*, long, long, const float*, int, int, int, int, const void*, float*, long, const RbfArgs&, void*, size_t, hipStream_t);
size_t kan_split_fwd_ws_bytes(long N, int in, int out, int C);
int kan_split_dx_any(const float*, long, const float*, long, long, const float*, int, int, int, int, const void*, float*, long, const RbfArgs&, hipStream_t, int gx16);
int kan_split_dw_any(const float*, long, const float*, long, long, const float*, int, int, int, int, const float*, const float*, float*, float*, float*, float*, size_t, const RbfArgs&, hipStream_t);

int kan_f32_pack(const float*, const float*, const float*, int, int, int, float*, float*, hipStream_t);
size_t kan_f32_pack_fwd_bytes(int in, int out, int C);
size_t kan_f32_pack_dx_bytes(int in, int out, int C);
int kan_dw_reduce(const float* slab, long NS, long per_slab, float* gcat, hipStream_t st);
void dw_plan(long N, int in, int out, int* NBx, long* rpw);

struct LnArgs {
    const float* w; const float* b; float eps;   // w == nullptr: no layernorm
    const float* stats_in = nullptr;             // forward (exact-fp32 kernel): row statistics GIVEN by the caller (feature-sharded
                                                 // layer: they cover all ranks' columns) instead of taken from the rows loaded here
    // LayerNorm backward, feature-sharded layer: the two row sums  sum_f gz*gamma,  sum_f gz*gamma*zhat  over ALL ranks' columns
    // (already summed over the ranks) and 1 / (total column count); nullptr: the kernel's own sums over its `in` columns
    const float* row_sums = nullptr; float inv_total = 0.0f;
};

__device__ __forceinline__ float rbf_val(float z, float c, float inv_den) {
    const float d = (z - c) * inv_den;
    return __expf(-d * d);
}

// LayerNorm row statistics (mean, 1/sqrt(biased var + eps)), two-pass like torch; 16 lanes per row
__global__ __launch_bounds__(256) void fastkan_stats_kernel(const float* __restrict__ x, long ldx, long N,
                                                            int in, float eps, float* __restrict__ stats) {
    const long row = (blockIdx.x * 256L + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    const float* xr = x + min(row, N - 1) * ldx;
    float s = 0.0f;
    for (int f = l; f < in; f += 16) s += xr[f];
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)in;
    float v = 0.0f;
    for (int f = l; f < in; f += 16) { const float d = xr[f] - mean; v = fmaf(d, d, v); }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if (l == 0 && row < N) { stats[2 * row] = mean; stats[2 * row + 1] = rsqrtf(v / (float)in + eps); }
}

// the same for rows of <= 64 NV features (NV <= 16): a lane keeps its NV groups of 4 consecutive features in registers -- ONE pass
// over x, 16-byte loads when VEC (the scalar two-pass form above read every row twice: 114 us vs the 256 MB / ~4.5 TB/s
// = 57 us of a single pass at 1M x 64).  VEC or not, the same values meet in the same order: a column slice of a wider
// activation (unaligned) gives the bits of its contiguous copy.
template <int NV, bool VEC>
__global__ __launch_bounds__(256) void fastkan_stats_v4_kernel(const float* __restrict__ x, long ldx, long N,
                                                               int in, float eps, float* __restrict__ stats) {
    const long row = (blockIdx.x * 256L + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    const float* xr = x + min(row, N - 1) * ldx;
    float v[NV][4];
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = 4 * l + 64 * j;
        if (VEC) {
            const float4 t = c < in ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[j][0] = t.x; v[j][1] = t.y; v[j][2] = t.z; v[j][3] = t.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[j][i] = c + i < in ? xr[c + i] : 0.0f;
        }
        s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)in;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d = (4 * l + 64 * j + i < in) ? v[j][i] - mean : 0.0f;
            q = fmaf(d, d, q);
        }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) q += __shfl_xor(q, o);
    if (l == 0 && row < N) { stats[2 * row] = mean; stats[2 * row + 1] = rsqrtf(q / (float)in + eps); }
}

static int launch_stats(const float* x, long ldx, long N, int in, float eps, float* stats, hipStream_t st) {
    const bool vec = (in & 3) == 0 && (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const int grid = cdiv(N, 16);
#define L(NV) do { if (vec) fastkan_stats_v4_kernel<NV, true><<<grid, 256, 0, st>>>(x, ldx, N, in, eps, stats); \
                   else fastkan_stats_v4_kernel<NV, false><<<grid, 256, 0, st>>>(x, ldx, N, in, eps, stats); } while (0)
    if (in <= 64) L(1); else if (in <= 128) L(2); else if (in <= 256) L(4); else if (in <= 512) L(8);
    else if (in <= 1024) L(16);                       // (the skip read-out of the node models: 128 + 3 x 256 = 896 columns)
    else fastkan_stats_kernel<<<grid, 256, 0, st>>>(x, ldx, N, in, eps, stats);
#undef L
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// ------------------------------------------------------------------ feature-sharded layer: the LayerNorm exchange
// a rank holds `in` of the row's `P * in` columns; "LayerNorm statistics -> all-reduce of 2 floats / row".
// Each rank leaves (mean, M2 = sum of squared deviations from ITS mean) of its columns -- two-pass, like torch's LayerNorm and
// the kernels above --, the P pairs of a row are gathered and merged in RANK ORDER by Chan's update.  (Summing (sum x, sum x^2)
// over the ranks instead would lose the variance's digits whenever |mean| >> std, and its bits would depend on the collective's
// reduction order; the gathered form moves the same 2 floats per row and rank.)  16 lanes per row.
__global__ __launch_bounds__(256) void fastkan_row_moments_kernel(const float* __restrict__ x, long ldx, long N, int in,
                                                                  float* __restrict__ moments) {
    const long row = (blockIdx.x * 256L + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    const float* xr = x + min(row, N - 1) * ldx;
    float s = 0.0f;
    for (int f = l; f < in; f += 16) s += xr[f];
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)in;
    float v = 0.0f;
    for (int f = l; f < in; f += 16) { const float d = xr[f] - mean; v = fmaf(d, d, v); }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if (l == 0 && row < N) { moments[2 * row] = mean; moments[2 * row + 1] = v; }
}

// gathered[p][n] = (mean_p, M2_p) over `in` columns each -> stats[n] = (mean, 1 / sqrt(M2 / (P * in) + eps)); one thread per row
__global__ __launch_bounds__(256) void fastkan_merge_moments_kernel(const float* __restrict__ gathered, int P, long N, int in,
                                                                    float eps, float* __restrict__ stats) {
    const long row = blockIdx.x * 256L + threadIdx.x;
    if (row >= N) return;
    float mean = gathered[2 * row], m2 = gathered[2 * row + 1];
    const float nb = (float)in;
    for (int p = 1; p < P; ++p) {
        const float mb = gathered[((long)p * N + row) * 2], qb = gathered[((long)p * N + row) * 2 + 1];
        const float na = nb * (float)p, n = na + nb, d = mb - mean;
        mean = fmaf(d, nb / n, mean);
        m2 = m2 + qb + d * d * (na * nb / n);
    }
    stats[2 * row] = mean;
    stats[2 * row + 1] = rsqrtf(m2 / (nb * (float)P) + eps);
}

// the two row sums of the LayerNorm backward over this rank's columns:  sums[n] = (sum_f gz*gamma, sum_f gz*gamma*zhat),
// zhat = (x - mean) * rstd with the MERGED statistics; summed over the ranks by the caller (all-reduce of 2 floats / row),
// then fastkan_ln_bwd_kernel finishes with them (LnArgs::row_sums).  16 lanes per row, fixed order.
__global__ __launch_bounds__(256) void fastkan_ln_row_sums_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ gz,
                                                                  long N, int in, const float* __restrict__ gamma,
                                                                  const float* __restrict__ stats, float* __restrict__ sums) {
    const long row = (blockIdx.x * 256L + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    const long rc = min(row, N - 1);
    const float mean = stats[2 * rc], rstd = stats[2 * rc + 1];
    float s1 = 0.0f, s2 = 0.0f;
    for (int f = l; f < in; f += 16) {
        const float h = gz[rc * (long)in + f] * gamma[f];
        s1 += h;
        s2 = fmaf(h, (x[rc * ldx + f] - mean) * rstd, s2);
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (l == 0 && row < N) { sums[2 * row] = s1; sums[2 * row + 1] = s2; }
}

int fastkan_row_moments(const float* x, long ldx, long N, int in, float* moments, hipStream_t st) {
    if (N == 0) return KAGNN_OK;
    fastkan_row_moments_kernel<<<cdiv(N, 16), 256, 0, st>>>(x, ldx, N, in, moments);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

int fastkan_merge_moments(const float* gathered, int P, long N, int in, float eps, float* stats, hipStream_t st) {
    if (N == 0) return KAGNN_OK;
    fastkan_merge_moments_kernel<<<cdiv(N, 256), 256, 0, st>>>(gathered, P, N, in, eps, stats);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

static bool fk_stats_in_fwd() {                       // KAGNN_FASTKAN_STATS_IN_FWD=0: the separate statistics pass (A/B)
    static const bool on = [] { const char* e = getenv("KAGNN_FASTKAN_STATS_IN_FWD"); return e == nullptr || atoi(e) != 0; }();
    return on;
}

static bool fk_split(int in, int out, int ng, int mode) { return mode == 1 && kan_split_fwd_ok(in, out, ng, 0); }

static RbfArgs fk_rbf(const float* centers, int ng, float den, const float* lnw, const float* lnb,
                      const float* stats, const float* bias, float* gz) {
    RbfArgs rb{};
    rb.centers = centers; rb.ng = ng;
    rb.a = 1.2011224087864498f / den;                 // sqrt(log2 e) / denominator
    rb.k2 = -2.0f * rb.a * 0.6931471805599453f;
    rb.ln_w = lnw; rb.ln_b = lnb; rb.stats = stats; rb.bias = bias; rb.gz = gz;
    return rb;
}

// ------------------------------------------------------------------ forward
template <int OT>
__global__ __launch_bounds__(256) void fastkan_fwd_kernel(
    const float* __restrict__ x, long ldx, long N, int in, int ng, const float* __restrict__ centers,
    float inv_den, LnArgs ln, const float* __restrict__ pack, int ot0, int OT_total,
    const float* __restrict__ base_bias, float* __restrict__ y, long ldy, int out,
    float* __restrict__ stats) {
    __shared__ float s_c[kMaxKnots];
    if (threadIdx.x < ng) s_c[threadIdx.x] = centers[threadIdx.x];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + wave) * 32;
    if (row0 >= N) return;
    const int r = lane & 31, kh = lane >> 5;
    const long row = row0 + r;
    const bool rv = row < N;
    const int P = (in + 1) / 2, CT = ng + 1;
    const float* xr = x + (rv ? row : 0) * ldx;

    float mean = 0.0f, rstd = 1.0f;
    if (ln.w && ln.stats_in) {                    // (wave-uniform) the caller's statistics: the row is a column shard of a wider one
        const long rc = rv ? row : 0;
        mean = ln.stats_in[2 * rc]; rstd = ln.stats_in[2 * rc + 1];
    } else if (ln.w) {                            // two-pass mean / biased variance, eps inside the sqrt
        float s = 0.0f;
        for (int p = 0; p < P; ++p) {               // unconditional clamped loads, masked by a 0/1 factor
            const int f = p + kh * P;
            s = fmaf(xr[min(f, in - 1)], (rv && f < in) ? 1.0f : 0.0f, s);
        }
        s += __shfl_xor(s, 32);
        mean = s / (float)in;
        float v = 0.0f;
        for (int p = 0; p < P; ++p) {
            const int f = p + kh * P;
            const float d = (xr[min(f, in - 1)] - mean) * ((rv && f < in) ? 1.0f : 0.0f);
            v = fmaf(d, d, v);
        }
        v += __shfl_xor(v, 32);
        rstd = rsqrtf(v / (float)in + ln.eps);
        if (stats && kh == 0 && rv && ot0 == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
    }

    f32x16 acc[OT];
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

    for (int p = 0; p < P; ++p) {
        const int f = p + kh * P;
        const bool fv = rv && f < in;
        const float xv = xr[min(f, in - 1)];          // unconditional clamped load (masked below)
        const int fc = min(f, in - 1);
        const float z = ln.w ? fmaf((xv - mean) * rstd, ln.w[fc], ln.b[fc]) : xv;   // ln.w is wave-uniform
        const float sl = fv ? siluf(xv) : 0.0f;
        const float* wp = pack + ((long)p * CT * OT_total + ot0) * 64 + lane;
        for (int g = 0; g < ng; ++g) {
            const float a = fv ? rbf_val(z, s_c[g], inv_den) : 0.0f;
#pragma unroll
            for (int t = 0; t < OT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wp[((long)g * OT_total + t) * 64], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < OT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(sl, wp[((long)ng * OT_total + t) * 64], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < OT; ++t) {
        const int col = 32 * (ot0 + t) + r;
        const float bb = (base_bias && col < out) ? base_bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const long rr = row0 + mfma32_row(i, kh);
            if (rr < N && col < out) y[rr * ldy + col] = acc[t][i] + bb;
        }
    }
}

// ------------------------------------------------------------------ input gradient, stage 1
// D_g[n][f] = sum_o gy[n][o] * W[o][f][g];  gz[n][f] = sum_g D_g * d phi_g / dz  (gradient w.r.t.
// the layer-normed value), gb[n][f] = D_base * silu'(x).  Without layernorm gx = gz + gb directly.
constexpr int kFkGroup = 9;

__global__ __launch_bounds__(256) void fastkan_dx_kernel(
    const float* __restrict__ x, long ldx, const float* __restrict__ gy, long ldgy, long N, int in,
    int out, int ng, const float* __restrict__ centers, float inv_den, LnArgs ln,
    const float* __restrict__ stats, const float* __restrict__ pack, int OT_total,
    float* __restrict__ gx, long ldgx, float* __restrict__ gz_tmp /* [N,in] when layernorm */) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_c = smem;
    const int Q = 16 * OT_total, outP = 2 * Q, ldt = outP + 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* s_gy = smem + kMaxKnots + (long)wave * 32 * ldt;
    if (threadIdx.x < ng) s_c[threadIdx.x] = centers[threadIdx.x];
    const long row0 = ((long)blockIdx.x * (blockDim.x >> 6) + wave) * 32;
    for (int i = lane; i < 32 * outP; i += 64) {
        const int rr = i / outP, o = i - rr * outP;
        const long row = row0 + rr;
        const float gv = gy[min(row, N - 1) * ldgy + min(o, out - 1)];      // unconditional clamped load
        s_gy[rr * ldt + o] = (row < N && o < out) ? gv : 0.0f;
    }
    __syncthreads();
    if (row0 >= N) return;
    const int r = lane & 31, kh = lane >> 5;
    const int CT = ng + 1, FT = cdiv(in, 32);
    const float* arow = s_gy + r * ldt + kh * Q;

    for (int ft = 0; ft < FT; ++ft) {
        const int f = 32 * ft + r;
        float gzacc[16], gbacc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { gzacc[i] = 0.0f; gbacc[i] = 0.0f; }
        for (int c0 = 0; c0 < CT; c0 += kFkGroup) {
            f32x16 D[kFkGroup];
#pragma unroll
            for (int j = 0; j < kFkGroup; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) D[j][i] = 0.0f;
            const float* wp = pack + ((long)ft * CT + c0) * Q * 64 + lane;
            for (int q = 0; q < Q; ++q) {
                const float a = arow[q];
#pragma unroll
                for (int j = 0; j < kFkGroup; ++j)
                    if (c0 + j < CT)
                        D[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wp[((long)j * Q + q) * 64], D[j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const long rr = row0 + mfma32_row(i, kh);
                const bool ok = rr < N && f < in;
                const float xv = x[min(rr, N - 1) * ldx + min(f, in - 1)];       // clamped; !ok lanes never store
                float z = xv;
                if (ln.w) {                            // wave-uniform; clamped indices, no per-lane branch around loads
                    const long rc = min(rr, N - 1); const int fc = min(f, in - 1);
                    z = fmaf((xv - stats[2 * rc]) * stats[2 * rc + 1], ln.w[fc], ln.b[fc]);
                }
                const float sg = silu_gradf(xv);
#pragma unroll
                for (int j = 0; j < kFkGroup; ++j) {
                    const int c = c0 + j;
                    if (c < ng) {
                        const float d = (z - s_c[c]) * inv_den;
                        gzacc[i] = fmaf(D[j][i], __expf(-d * d) * (-2.0f * d * inv_den), gzacc[i]);
                    } else if (c == ng) {
                        gbacc[i] = D[j][i] * sg;
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const long rr = row0 + mfma32_row(i, kh);
            if (rr < N && f < in) {
                if (ln.w) { gz_tmp[rr * (long)in + f] = gzacc[i]; gx[rr * ldgx + f] = gbacc[i]; }
                else gx[rr * ldgx + f] = gzacc[i] + gbacc[i];
            }
        }
    }
}

// ------------------------------------------------------------------ input gradient, stage 2
// LayerNorm backward, one wave per row (persistent, so the per-feature sums for g_ln_weight /
// g_ln_bias stay in registers):  gh = gz*gamma;  gx += rstd * (gh - mean(gh) - zhat*mean(gh*zhat)).
constexpr int kLnMaxT = 64;     // features per lane: in <= 64*64

// T = features per lane, R = rows in flight per wave (more loads in flight for narrow rows).  T <= 16 keeps
// the rows (zhat, gh) in registers between the two passes; wider rows re-read them (L2-hot) so that only the
// per-feature sums occupy registers.  The four waves of a workgroup combine their sums through LDS in a
// fixed order: one partial row per workgroup.
template <int T, int R>
__global__ __launch_bounds__(256) void fastkan_ln_bwd_kernel(
    const float* __restrict__ x, long ldx, const float* __restrict__ gz, long N, int in, LnArgs ln,
    const float* __restrict__ stats, float* __restrict__ gx, long ldgx,
    float* __restrict__ partial /* [blocks][2][in] */) {
    constexpr bool CACHE = T <= 16;
    extern __shared__ float s_part[];              // [2][in]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wid = blockIdx.x * 4L + wave, nw = gridDim.x * 4L;
    float cw[T], cb[T], gam[CACHE ? T : 1];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        cw[t] = 0.0f; cb[t] = 0.0f;
        if constexpr (CACHE) gam[t] = ln.w[min(lane + 64 * t, in - 1)];
    }
    const float inv_n = 1.0f / (float)in;
    for (long row0 = wid * R; row0 < N; row0 += nw * R) {
        float zh[R][CACHE ? T : 1], gh[R][CACHE ? T : 1];
        float s1[R], s2[R], mean[R], rstd[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long row = min(row0 + r, N - 1);
            const float live = (row0 + r < N) ? 1.0f : 0.0f;         // rows past the end contribute zeros
            mean[r] = stats[2 * row]; rstd[r] = stats[2 * row + 1];
            s1[r] = 0.0f; s2[r] = 0.0f;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int f = lane + 64 * t;
                if (f < in) {
                    const float g = gz[row * (long)in + f] * live;
                    const float z = (x[row * ldx + f] - mean[r]) * rstd[r];
                    const float h = g * (CACHE ? gam[t] : ln.w[f]);
                    if constexpr (CACHE) { zh[r][t] = z; gh[r][t] = h; }
                    cw[t] = fmaf(g, z, cw[t]);
                    cb[t] += g;
                    s1[r] += h;
                    s2[r] = fmaf(h, z, s2[r]);
                }
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
            for (int r = 0; r < R; ++r) { s1[r] += __shfl_xor(s1[r], o); s2[r] += __shfl_xor(s2[r], o); }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (row0 + r >= N) break;
            const long row = row0 + r;
            const float m1 = ln.row_sums ? ln.row_sums[2 * row] * ln.inv_total : s1[r] * inv_n;
            const float m2 = ln.row_sums ? ln.row_sums[2 * row + 1] * ln.inv_total : s2[r] * inv_n;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int f = lane + 64 * t;
                if (f < in) {
                    float z, h;
                    if constexpr (CACHE) { z = zh[r][t]; h = gh[r][t]; }
                    else { z = (x[row * ldx + f] - mean[r]) * rstd[r]; h = gz[row * (long)in + f] * ln.w[f]; }
                    gx[row * ldgx + f] += rstd[r] * (h - m1 - z * m2);
                }
            }
        }
    }
    for (int w = 1; w < 4; ++w) {                    // waves 1..3 hand their sums to wave 0, in order
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int f = lane + 64 * t;
                if (f < in) { s_part[f] = cw[t]; s_part[in + f] = cb[t]; }
            }
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int f = lane + 64 * t;
                if (f < in) { cw[t] += s_part[f]; cb[t] += s_part[in + f]; }
            }
        }
    }
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int f = lane + 64 * t;
            if (f < in) { partial[(blockIdx.x * 2L + 0) * in + f] = cw[t]; partial[(blockIdx.x * 2L + 1) * in + f] = cb[t]; }
        }
    }
}

// The same with 16-byte accesses: a lane owns T4 groups of 4 consecutive columns (column group lane + 64 t).  Rows wider than a
// few hundred columns (the 896-wide skip read-out of the FastKAN node models) moved 2.4 GB per launch through 4-byte loads and
// read-modify-write stores at 2.5 TB/s (955 us at 169 343 x 896); same sums in the same order per column, so same bits.
template <int T4>
__global__ __launch_bounds__(256) void fastkan_ln_bwd_v4_kernel(
    const float* __restrict__ x, long ldx, const float* __restrict__ gz, long N, int in, LnArgs ln,
    const float* __restrict__ stats, float* __restrict__ gx, long ldgx,
    float* __restrict__ partial /* [blocks][2][in] */) {
    extern __shared__ float s_part[];              // [2][in]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wid = blockIdx.x * 4L + wave, nw = gridDim.x * 4L;
    const int q4 = in >> 2;                        // float4 groups per row
    float4 cw[T4], cb[T4], gam[T4];
#pragma unroll
    for (int t = 0; t < T4; ++t) {
        cw[t] = make_float4(0.f, 0.f, 0.f, 0.f); cb[t] = cw[t];
        const int c4 = min(lane + 64 * t, q4 - 1);
        gam[t] = *reinterpret_cast<const float4*>(ln.w + 4 * c4);
    }
    const float inv_n = 1.0f / (float)in;
    for (long row = wid; row < N; row += nw) {
        const float mean = stats[2 * row], rstd = stats[2 * row + 1];
        float4 zh[T4], gh[T4];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int t = 0; t < T4; ++t) {
            const int c4 = lane + 64 * t;
            if (c4 < q4) {
                const float4 g = *reinterpret_cast<const float4*>(gz + row * (long)in + 4 * c4);
                const float4 xv = *reinterpret_cast<const float4*>(x + row * ldx + 4 * c4);
                const float4 z = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
                const float4 h = make_float4(g.x * gam[t].x, g.y * gam[t].y, g.z * gam[t].z, g.w * gam[t].w);
                zh[t] = z; gh[t] = h;
                cw[t].x = fmaf(g.x, z.x, cw[t].x); cw[t].y = fmaf(g.y, z.y, cw[t].y); cw[t].z = fmaf(g.z, z.z, cw[t].z); cw[t].w = fmaf(g.w, z.w, cw[t].w);
                cb[t].x += g.x; cb[t].y += g.y; cb[t].z += g.z; cb[t].w += g.w;
                s1 += h.x; s2 = fmaf(h.x, z.x, s2);
                s1 += h.y; s2 = fmaf(h.y, z.y, s2);
                s1 += h.z; s2 = fmaf(h.z, z.z, s2);
                s1 += h.w; s2 = fmaf(h.w, z.w, s2);
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        const float m1 = ln.row_sums ? ln.row_sums[2 * row] * ln.inv_total : s1 * inv_n;
        const float m2 = ln.row_sums ? ln.row_sums[2 * row + 1] * ln.inv_total : s2 * inv_n;
#pragma unroll
        for (int t = 0; t < T4; ++t) {
            const int c4 = lane + 64 * t;
            if (c4 < q4) {
                float4* gp = reinterpret_cast<float4*>(gx + row * ldgx + 4 * c4);
                float4 v = *gp;
                v.x += rstd * (gh[t].x - m1 - zh[t].x * m2); v.y += rstd * (gh[t].y - m1 - zh[t].y * m2);
                v.z += rstd * (gh[t].z - m1 - zh[t].z * m2); v.w += rstd * (gh[t].w - m1 - zh[t].w * m2);
                *gp = v;
            }
        }
    }
    for (int w = 1; w < 4; ++w) {                    // waves 1..3 hand their sums to wave 0, in order
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < T4; ++t) {
                const int c4 = lane + 64 * t;
                if (c4 < q4) { *reinterpret_cast<float4*>(s_part + 4 * c4) = cw[t]; *reinterpret_cast<float4*>(s_part + in + 4 * c4) = cb[t]; }
            }
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < T4; ++t) {
                const int c4 = lane + 64 * t;
                if (c4 < q4) {
                    const float4 a = *reinterpret_cast<const float4*>(s_part + 4 * c4), b = *reinterpret_cast<const float4*>(s_part + in + 4 * c4);
                    cw[t].x += a.x; cw[t].y += a.y; cw[t].z += a.z; cw[t].w += a.w;
                    cb[t].x += b.x; cb[t].y += b.y; cb[t].z += b.z; cb[t].w += b.w;
                }
            }
        }
    }
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < T4; ++t) {
            const int c4 = lane + 64 * t;
            if (c4 < q4) {
                *reinterpret_cast<float4*>(partial + (blockIdx.x * 2L + 0) * in + 4 * c4) = cw[t];
                *reinterpret_cast<float4*>(partial + (blockIdx.x * 2L + 1) * in + 4 * c4) = cb[t];
            }
        }
    }
}

static int launch_ln_bwd(int blocks, const float* x, long ldx, const float* gz, long N, int in, LnArgs ln,
                         const float* stats, float* gx, long ldgx, float* partial, hipStream_t st) {
    const int T = cdiv(in, 64);
    const size_t lds = 2 * (size_t)in * sizeof(float);
    // wide rows, 16-byte accesses (the row sums s1 / s2 are taken in another order than the 4-byte form: results agree to
    // rounding, each form is deterministic)
    const bool v4 = in > 256 && in <= 2048 && (in & 3) == 0 && (ldx & 3) == 0 && (ldgx & 3) == 0 &&
                    ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gz) | reinterpret_cast<uintptr_t>(gx) |
                      reinterpret_cast<uintptr_t>(ln.w) | reinterpret_cast<uintptr_t>(partial)) & 15) == 0;
    if (v4) {
        const int T4 = cdiv(in, 256);
#define LV(TT) fastkan_ln_bwd_v4_kernel<TT><<<blocks, 256, lds, st>>>(x, ldx, gz, N, in, ln, stats, gx, ldgx, partial)
        if (T4 <= 2) LV(2); else if (T4 <= 4) LV(4); else LV(8);
#undef LV
        KAGNN_LAUNCH_CHECK();
        return KAGNN_OK;
    }
#define L(TT, RR) fastkan_ln_bwd_kernel<TT, RR><<<blocks, 256, lds, st>>>(x, ldx, gz, N, in, ln, stats, gx, ldgx, partial)
    // measured on MI355X: many single-row waves beat fewer multi-row ones (only very narrow rows take two)
    if (T <= 1) L(1, 2); else if (T <= 2) L(2, 2); else if (T <= 4) L(4, 1); else if (T <= 8) L(8, 1);
    else if (T <= 16) L(16, 1); else if (T <= 32) L(32, 1); else L(64, 1);
#undef L
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// out[j] = sum_w partial[w*stride + j], j < n  (fixed order).  One workgroup = 32 columns x 32 row groups (1024 threads:
// with 8 groups the ~2000 partial rows were a 256-deep chain of dependent loads per thread, 51 us per call);
// the groups are combined through LDS in a fixed order, so the result is deterministic.
constexpr int kSumGroups = 32;
__global__ __launch_bounds__(1024) void sum_partials_kernel(const float* __restrict__ partial, long W, long stride, long n,
                                                            float* __restrict__ out0, float* __restrict__ out1, long split) {
    __shared__ float s_p[kSumGroups][33];
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const long j = blockIdx.x * 32L + c;
    float a = 0.0f;
    if (j < n)
        for (long w = rg; w < W; w += kSumGroups) a += partial[w * stride + j];
    s_p[rg][c] = a;
    __syncthreads();
    if (rg == 0 && j < n) {
        float t = s_p[0][c];
#pragma unroll
        for (int g = 1; g < kSumGroups; ++g) t += s_p[g][c];
        if (j < split) out0[j] = t; else out1[j - split] = t;
    }
}

// column sums of gy (g_base_bias): workgroup b sums rows [b*rpb, ...) for all columns.  Thread = (row slot, 4 consecutive
// columns): the 256 / (F/4) row slots walk the rows in parallel with 16-byte loads and meet in LDS in slot order
// (deterministic).  (One thread per column walking ~1000 rows serially -- 64 active lanes per workgroup at F = 64 -- took
// 405 us per call at 1M x 64: 17 % of the FastKAN-GIN layer step.)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ a, long lda,
                                                             long N, int F, long rpb,
                                                             float* __restrict__ partial) {
    __shared__ float4 s_red[256];
    const long r0 = blockIdx.x * rpb, r1 = min(N, r0 + rpb);
    const int cl = min(256, cdiv(F, 4)), rs = 256 / cl;          // column groups per pass, row slots
    const int cg = threadIdx.x % cl, slot = threadIdx.x / cl;
    const bool vec = ((F & 3) == 0) && ((lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(a) & 15) == 0);
    for (int c0 = 0; c0 < F; c0 += 4 * cl) {                     // F > 1024: more than one pass (uniform trip count)
        const int c = c0 + 4 * cg;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (slot < rs && c < F) {
            if (vec) {
                for (long r = r0 + slot; r < r1; r += rs) {
                    const float4 v = *reinterpret_cast<const float4*>(a + r * lda + c);
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            } else {
                for (long r = r0 + slot; r < r1; r += rs) {
                    const float* row = a + r * lda + c;
                    acc.x += row[0];
                    if (c + 1 < F) acc.y += row[1];
                    if (c + 2 < F) acc.z += row[2];
                    if (c + 3 < F) acc.w += row[3];
                }
            }
        }
        __syncthreads();
        s_red[threadIdx.x] = acc;
        __syncthreads();
        if (slot == 0 && c < F) {
            for (int k = 1; k < rs; ++k) {                       // fixed order over the row slots
                const float4 v = s_red[k * cl + cg];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            float* o = partial + blockIdx.x * (long)F + c;
            o[0] = acc.x;
            if (c + 1 < F) o[1] = acc.y;
            if (c + 2 < F) o[2] = acc.z;
            if (c + 3 < F) o[3] = acc.w;
        }
    }
}

// ------------------------------------------------------------------ weight gradient
__global__ __launch_bounds__(256) void fastkan_dw_kernel(
    const float* __restrict__ x, long ldx, const float* __restrict__ gy, long ldgy, long N, int in,
    int out, int ng, const float* __restrict__ centers, float inv_den, LnArgs ln,
    const float* __restrict__ stats, int OT, long rows_per_wave, float* __restrict__ slab) {
    __shared__ float s_c[kMaxKnots];
    if (threadIdx.x < ng) s_c[threadIdx.x] = centers[threadIdx.x];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, kh = lane >> 5;
    const int ft = blockIdx.y / OT, ot = blockIdx.y % OT;
    const int FT = gridDim.y / OT;
    const long s = (long)blockIdx.x * 4 + wave;
    const long rbeg = s * rows_per_wave;
    const long rend = min(N, rbeg + rows_per_wave);
    const int CT = ng + 1;
    const int f = 32 * ft + r, o = 32 * ot + r;
    const bool fv = f < in, ov = o < out;
    const long inP = 32L * FT, outP = 32L * OT;
    const float gam = ln.w ? ln.w[min(f, in - 1)] : 1.0f, bet = ln.w ? ln.b[min(f, in - 1)] : 0.0f;

    for (int c0 = 0; c0 < CT; c0 += kFkGroup) {
        f32x16 D[kFkGroup];
#pragma unroll
        for (int j = 0; j < kFkGroup; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) D[j][i] = 0.0f;
        for (long n = rbeg + kh; n < rend + kh; n += 2) {
            const bool nv = n < rend;
            const bool live = nv && fv;
            const long nc = min(n, N - 1);                 // clamped unconditional loads; `live` masks the products
            const float xv = x[nc * ldx + min(f, in - 1)];
            const float b = gy[nc * ldgy + min(o, out - 1)];
            float z = xv;
            if (ln.w) z = fmaf((xv - stats[2 * nc]) * stats[2 * nc + 1], gam, bet);   // wave-uniform condition
            const float sl = siluf(xv);
#pragma unroll
            for (int j = 0; j < kFkGroup; ++j) {
                const int c = c0 + j;
                if (c < CT) {
                    float a = (c == ng) ? sl : rbf_val(z, s_c[min(c, ng - 1)], inv_den);
                    a = live ? a : 0.0f;
                    D[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, D[j], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kFkGroup; ++j) {
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

// g_spline_weight[o][f*ng+g] = gcat[g][f][o] ; g_base_weight[o][f] = gcat[ng][f][o]
__global__ void fastkan_dw_unpack_kernel(const float* __restrict__ gcat, int in, int out, int ng,
                                         long inP, long outP, float* __restrict__ g_sw,
                                         float* __restrict__ g_bw) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)in * out) return;
    const int o = i % out, f = i / out;
    for (int g = 0; g < ng; ++g)
        g_sw[((long)o * in + f) * ng + g] = gcat[((long)g * inP + f) * outP + o];
    if (g_bw) g_bw[(long)o * in + f] = gcat[((long)ng * inP + f) * outP + o];
}

// ------------------------------------------------------------------ host launchers
static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t fastkan_fwd_ws_bytes(long N, int in, int out, int ng, int mode) {
    if (fk_split(in, out, ng, mode)) return al256(kan_split_pack_fwd_bytes(in, out, ng)) + al256(kan_split_fwd_ws_bytes(N, in, out, ng));
    return al256(kan_f32_pack_fwd_bytes(in, out, ng)) + al256(kan_f32_pack_dx_bytes(in, out, ng));
}

int fastkan_fwd(const float* x, long ldx, long N, int in, int out, int ng, const float* centers,
                float den, const float* lnw, const float* lnb, float eps, const float* sw,
                const float* bw, const float* bb, float* y, long ldy, float* stats, void* ws,
                size_t ws_bytes, int mode, hipStream_t st, bool stats_given) {
    // stats_given: `stats` is an INPUT -- the row statistics over all ranks' columns of the feature-sharded layer
    if (ws_bytes < fastkan_fwd_ws_bytes(N, in, out, ng, mode)) return fail(KAGNN_ERR_ARG, "%s: workspace too small", "fastkan_fwd");
    if (lnw && !stats) return fail(KAGNN_ERR_ARG, "%s: row_stats is required with layernorm", "fastkan_fwd");
    if (fk_split(in, out, ng, mode)) {
        // one-chunk layers (<= 8 centres, in <= 64 with <= 64 outputs / in <= 32 with <= 128, one output block): the forward
        // kernel takes the row statistics from the rows it loads and stores them; anything else runs the statistics pass first
        const int cf = cdiv(min(out, 128), 32) <= 2 ? 64 : 32;
        const bool own_stats = lnw && !stats_given && ng <= 8 && in <= cf && out <= 128 && N > 0 && fk_stats_in_fwd();
        if (lnw && !own_stats && !stats_given) { int rc = launch_stats(x, ldx, N, in, eps, stats, st); if (rc) return rc; }
        { int rc = kan_split_pack_fwd_noscale(bw, sw, nullptr, in, out, ng, ws, st); if (rc) return rc; }
        char* part = static_cast<char*>(ws) + al256(kan_split_pack_fwd_bytes(in, out, ng));
        RbfArgs rb = fk_rbf(centers, ng, den, lnw, lnb, stats, bb, nullptr);
        if (own_stats) { rb.stats_out = stats; rb.ln_eps = eps; }
        return kan_split_fwd_any(x, ldx, N, nullptr, in, out, ng, 0, ws, y, ldy, rb, part,
                                 kan_split_fwd_ws_bytes(N, in, out, ng), st);
    }
    float* pf = (float*)ws;
    float* pd = (float*)((char*)ws + al256(kan_f32_pack_fwd_bytes(in, out, ng)));
    { int rc = kan_f32_pack(bw, sw, nullptr, in, out, ng, pf, pd, st); if (rc) return rc; }
    LnArgs ln{lnw, lnb, eps};
    if (stats_given) ln.stats_in = stats;
    const int OTt = cdiv(out, 32);
    dim3 grid(cdiv(N, 128));
    for (int ot0 = 0; ot0 < OTt; ot0 += 4) {
        const int n = min(4, OTt - ot0);
#define L(OTN) fastkan_fwd_kernel<OTN><<<grid, 256, 0, st>>>(x, ldx, N, in, ng, centers, 1.0f / den, ln, pf, ot0, OTt, bb, y, ldy, out, stats)
        if (n == 1) L(1); else if (n == 2) L(2); else if (n == 3) L(3); else L(4);
#undef L
        KAGNN_LAUNCH_CHECK();
    }
    return KAGNN_OK;
}

struct FkBwdPlan {
    size_t pack_f, pack_d, gz, gcat, slab, lnpart, colpart, total;
    int nb; long rpw; long NS; long per; int ln_blocks; int col_blocks; long col_rpb;
    bool split;
};

static FkBwdPlan fk_plan(long N, int in, int out, int ng, int mode) {
    FkBwdPlan p;
    p.split = fk_split(in, out, ng, mode);
    p.gz = al256((size_t)N * in * 4);
    if (p.split) {
        p.pack_f = 0;
        p.pack_d = al256(kan_split_pack_dx_bytes(in, out, ng, 0));
        p.gcat = 0;                                   // gcat + slabs live inside the split kernel's own workspace
        p.slab = al256(kan_split_dw_ws_bytes(N, in, out, ng, 0));
        p.nb = 0; p.rpw = 0; p.NS = 0; p.per = 0;
    } else {
        p.pack_f = al256(kan_f32_pack_fwd_bytes(in, out, ng));
        p.pack_d = al256(kan_f32_pack_dx_bytes(in, out, ng));
        dw_plan(N, in, out, &p.nb, &p.rpw);
        p.NS = (long)p.nb * 4;
        p.per = (long)(ng + 1) * 32 * cdiv(in, 32) * 32 * cdiv(out, 32);
        p.gcat = al256((size_t)p.per * 4);
        p.slab = al256((size_t)p.NS * p.per * 4);
    }
    p.ln_blocks = (int)max(1L, min(2048L, (N + 15) / 16));       // >= 4 rows per wave, up to 8192 waves
    p.lnpart = al256((size_t)p.ln_blocks * 2 * in * 4);
    p.col_blocks = (int)max(1L, min(512L, N / 256 + 1));
    p.col_rpb = (N + p.col_blocks - 1) / p.col_blocks;
    p.colpart = al256((size_t)p.col_blocks * out * 4);
    if (p.split) {                                    // the weight-gradient kernel leaves one row of column sums per row slab
        long slabs = 0, outP = 0;
        kan_split_dw_slabs(N, in, out, ng, 0, &slabs, &outP);
        p.colpart = al256((size_t)max(slabs * outP, (long)p.col_blocks * out) * 4);
    }
    p.total = p.pack_f + p.pack_d + p.gz + p.gcat + p.slab + p.lnpart + p.colpart;
    return p;
}

size_t fastkan_bwd_ws_bytes(long N, int in, int out, int ng, int mode) { return fk_plan(N, in, out, ng, mode).total; }

int fastkan_bwd(const float* x, long ldx, const float* gy, long ldgy, long N, int in, int out, int ng,
                const float* centers, float den, const float* lnw, const float* lnb, float eps,
                const float* sw, const float* bw, const float* stats, float* gx, long ldgx,
                float* g_lnw, float* g_lnb, float* g_sw, float* g_bw, float* g_bb, void* ws,
                size_t ws_bytes, int mode, hipStream_t st, int phase, float* row_sums, int in_total) {
    // phase 0: the whole backward.  Feature-sharded layer (the row is P column shards, `stats` covers all of them): phase 1 = all
    // of it EXCEPT the LayerNorm backward -- d loss / dz stays in the workspace, gx holds the base-branch part, row_sums[n] receives
    // the two row sums over this rank's columns; the caller sums them over the ranks; phase 2 (same shape arguments, same
    // workspace, untouched in between) = the LayerNorm backward with those sums: gx completed, g_lnw / g_lnb written.
    // phase 3 / 4: phase 1 in two halves -- 3 = the input-gradient half (gx, d loss / dz, row_sums), 4 = the weight-gradient half --
    // so that the caller's all-reduce of row_sums can run beside the weight gradient.
    const bool do_dx = phase != 4, do_dw = phase != 3, shard = phase == 1 || phase == 3 || phase == 4;
    const FkBwdPlan p = fk_plan(N, in, out, ng, mode);
    if (ws_bytes < p.total) return fail(KAGNN_ERR_ARG, "%s: workspace too small", "fastkan_bwd");
    if (lnw && in > 64 * kLnMaxT) return fail(KAGNN_ERR_UNSUPPORTED, "%s: layernorm backward supports input_dim <= 4096", "fastkan_bwd");
    char* q = (char*)ws;
    float* pf = (float*)q; q += p.pack_f;
    float* pd = (float*)q; q += p.pack_d;
    float* gz = (float*)q; q += p.gz;
    float* gcat = (float*)q; q += p.gcat;
    float* slab = (float*)q; q += p.slab;
    float* lnpart = (float*)q; q += p.lnpart;
    float* colpart = (float*)q;
    if (phase == 2) {
        if (!lnw) return KAGNN_OK;
        LnArgs ln{lnw, lnb, eps};
        ln.row_sums = row_sums; ln.inv_total = 1.0f / (float)in_total;
        if (N > 0) { int rc = launch_ln_bwd(p.ln_blocks, x, ldx, gz, N, in, ln, stats, gx, ldgx, lnpart, st); if (rc) return rc; }
        else KAGNN_HIP(hipMemsetAsync(lnpart, 0, (size_t)p.ln_blocks * 2 * in * sizeof(float), st));
        sum_partials_kernel<<<cdiv(2L * in, 32), 32 * kSumGroups, 0, st>>>(lnpart, p.ln_blocks, 2L * in, 2L * in, g_lnw, g_lnb, in);
        KAGNN_LAUNCH_CHECK();
        return KAGNN_OK;
    }
    if (p.split) {
        // same split-precision kernels as the B-spline layer, K == 0 selecting the RBF basis
        const RbfArgs rb = fk_rbf(centers, ng, den, lnw, lnb, stats, nullptr, gz);
        if (N > 0 && do_dx) {
            int rc = kan_split_pack_dx_noscale(bw, sw, nullptr, in, out, ng, 0, pd, st);
            if (rc) return rc;
            rc = kan_split_dx_any(x, ldx, gy, ldgy, N, nullptr, in, out, ng, 0, pd, gx, ldgx, rb, st, 0);
            if (rc) return rc;
        }
        if (lnw && shard) {
            if (N > 0 && do_dx) { fastkan_ln_row_sums_kernel<<<cdiv(N, 16), 256, 0, st>>>(x, ldx, gz, N, in, lnw, stats, row_sums); KAGNN_LAUNCH_CHECK(); }
        } else if (lnw) {
            LnArgs ln{lnw, lnb, eps};
            { int rc = launch_ln_bwd(p.ln_blocks, x, ldx, gz, N, in, ln, stats, gx, ldgx, lnpart, st); if (rc) return rc; }
            sum_partials_kernel<<<cdiv(2L * in, 32), 32 * kSumGroups, 0, st>>>(lnpart, p.ln_blocks, 2L * in, 2L * in, g_lnw, g_lnb, in);
            KAGNN_LAUNCH_CHECK();
        }
        if (!do_dw) return KAGNN_OK;
        // the base bias gradient (column sums of gy) rides in the weight-gradient kernel, which reads gy anyway: one row of
        // sums per row slab, folded in slab order (deterministic).  (A pass of its own over gy cost 0.08 ms per layer at 1M x 64.)
        RbfArgs rbw = rb;
        long slabs = 0, outP = 0;
        kan_split_dw_slabs(N, in, out, ng, 0, &slabs, &outP);
        rbw.colpart = (g_bb && N > 0) ? colpart : nullptr;
        { int rc = kan_split_dw_any(x, ldx, gy, ldgy, N, nullptr, in, out, ng, 0, sw, nullptr, g_bw, g_sw, nullptr,
                                    slab, p.slab, rbw, st); if (rc) return rc; }
        if (g_bb) {
            if (N > 0) {
                sum_partials_kernel<<<cdiv(out, 32), 32 * kSumGroups, 0, st>>>(colpart, slabs, outP, out, g_bb, g_bb, out);
                KAGNN_LAUNCH_CHECK();
            } else {
                KAGNN_HIP(hipMemsetAsync(g_bb, 0, (size_t)out * sizeof(float), st));
            }
        }
        return KAGNN_OK;
    }
    const float inv_den = 1.0f / den;
    LnArgs ln{lnw, lnb, eps};
    const int OTt = cdiv(out, 32), FT = cdiv(in, 32);
    if (N > 0 && do_dx) {
        int rc = kan_f32_pack(bw, sw, nullptr, in, out, ng, pf, pd, st);
        if (rc) return rc;
        int W = 4;                                    // waves per workgroup: as many as the gy tiles leave LDS for
        while (W > 1 && (kMaxKnots + (size_t)W * 32 * (32 * OTt + 1)) * sizeof(float) > 160 * 1024) W >>= 1;
        const size_t lds = (kMaxKnots + (size_t)W * 32 * (32 * OTt + 1)) * sizeof(float);
        if (lds > 160 * 1024) return fail(KAGNN_ERR_UNSUPPORTED, "%s: output_dim too large", "fastkan_bwd");
        if (lds > 64 * 1024)
            KAGNN_HIP(hipFuncSetAttribute((const void*)fastkan_dx_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fastkan_dx_kernel<<<cdiv(N, 32 * W), 64 * W, lds, st>>>(x, ldx, gy, ldgy, N, in, out, ng, centers, inv_den, ln, stats, pd, OTt, gx, ldgx, gz);
        KAGNN_LAUNCH_CHECK();
    }
    if (lnw && shard) {
        if (N > 0 && do_dx) { fastkan_ln_row_sums_kernel<<<cdiv(N, 16), 256, 0, st>>>(x, ldx, gz, N, in, lnw, stats, row_sums); KAGNN_LAUNCH_CHECK(); }
    } else if (lnw) {
        { int rc = launch_ln_bwd(p.ln_blocks, x, ldx, gz, N, in, ln, stats, gx, ldgx, lnpart, st); if (rc) return rc; }
        sum_partials_kernel<<<cdiv(2L * in, 32), 32 * kSumGroups, 0, st>>>(lnpart, p.ln_blocks, 2L * in, 2L * in, g_lnw, g_lnb, in);
        KAGNN_LAUNCH_CHECK();
    }
    if (!do_dw) return KAGNN_OK;
    dim3 grid(p.nb, FT * OTt);
    fastkan_dw_kernel<<<grid, 256, 0, st>>>(x, ldx, gy, ldgy, N, in, out, ng, centers, inv_den, ln, stats, OTt, p.rpw, slab);
    KAGNN_LAUNCH_CHECK();
    { int rc = kan_dw_reduce(slab, p.NS, p.per, gcat, st); if (rc) return rc; }
    fastkan_dw_unpack_kernel<<<cdiv((long)in * out, 256), 256, 0, st>>>(gcat, in, out, ng, 32L * FT, 32L * OTt, g_sw, g_bw);
    KAGNN_LAUNCH_CHECK();
    if (g_bb) {
        colsum_partial_kernel<<<p.col_blocks, 256, 0, st>>>(gy, ldgy, N, out, p.col_rpb, colpart);
        KAGNN_LAUNCH_CHECK();
        sum_partials_kernel<<<cdiv(out, 32), 32 * kSumGroups, 0, st>>>(colpart, p.col_blocks, out, out, g_bb, g_bb, out);
        KAGNN_LAUNCH_CHECK();
    }
    return KAGNN_OK;
}

}  // namespace kagnn
