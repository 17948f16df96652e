// The loss tail of the timing harness as one kernel each way: `out = softmax(logits); loss =
// CrossEntropyLoss()(out[mask], y[mask])` (node_classification_clean/time_model.py:43-45 -- the reference applies
// softmax BEFORE CrossEntropyLoss, which applies log_softmax again; `pre_softmax` keeps that, 0 is the plain
// softmax cross-entropy of utils.py's train loop).  In torch this is softmax, a boolean-mask gather (device->host
// sync for the row count), log_softmax and two single-block nll_loss reductions: ~4.6 ms at 1M nodes x 40 classes
// against 0.16 GB of logits.  Here: one pass over the logits forward (row statistics kept: 12 B/row), one backward;
// mean over the masked rows with the count taken on the device (no sync, capturable in a HIP graph); deterministic
// (per-workgroup partial sums, combined in a fixed order).
#include "common.h"

namespace kagnn {

template <int W>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = W / 2; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
template <int W>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = W / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// W lanes per row (power of two <= 64), KC classes per lane held in registers (C <= W * KC; KC == 0: any C, the row
// is re-read from L1/L2 for every pass), kRows rows per group and trip so that enough loads are in flight.
// stats[row] = (max, sum exp, sum exp of the second softmax); partial[2b], partial[2b+1] = this workgroup's loss sum
// and row count.
constexpr int kXentRows = 4;

template <int W, int KC>
__global__ __launch_bounds__(256) void xent_fwd_kernel(const float* __restrict__ z, long ld, long N, int C,
                                                        const long* __restrict__ y, const unsigned char* __restrict__ mask,
                                                        int pre, float* __restrict__ stats, float* __restrict__ partial) {
    constexpr int G = 256 / W, KR = KC > 0 ? KC : 1;
    __shared__ float s_sum[256], s_cnt[256];
    const int l = threadIdx.x & (W - 1), g = threadIdx.x / W;
    float bsum = 0.0f, bcnt = 0.0f;
    for (long base = (long)blockIdx.x * G * kXentRows; base < N; base += (long)gridDim.x * G * kXentRows) {
        float v[kXentRows][KR];
        long yv[kXentRows];
#pragma unroll
        for (int q = 0; q < kXentRows; ++q) {                 // all loads of the trip first
            const long r = min(base + q * G + g, N - 1);
            yv[q] = y[r];
            if (KC > 0) {
#pragma unroll
                for (int k = 0; k < KR; ++k) { const int c = l + W * k; v[q][k] = c < C ? z[r * ld + c] : -INFINITY; }
            }
        }
#pragma unroll
        for (int q = 0; q < kXentRows; ++q) {
            const long row = base + q * G + g;
            const bool valid = row < N;
            const float* zr = z + min(row, N - 1) * ld;
            float m = -INFINITY;
            if (KC > 0) {
#pragma unroll
                for (int k = 0; k < KR; ++k) m = fmaxf(m, v[q][k]);
            } else {
                for (int c = l; c < C; c += W) m = fmaxf(m, zr[c]);
            }
            m = group_max<W>(m);
            float s = 0.0f, e[KR];
            if (KC > 0) {
#pragma unroll
                for (int k = 0; k < KR; ++k) { e[k] = __expf(v[q][k] - m); s += e[k]; }       // exp(-inf) = 0 for c >= C
            } else {
                for (int c = l; c < C; c += W) s += __expf(zr[c] - m);
            }
            s = group_sum<W>(s);
            const bool label_ok = yv[q] >= 0 && yv[q] < C;
            const float zy = zr[label_ok ? yv[q] : 0];
            float s2 = 0.0f, loss;
            if (pre) {
                const float inv = 1.0f / s;                  // = the largest probability: the second softmax's shift
                if (KC > 0) {
#pragma unroll
                    for (int k = 0; k < KR; ++k) s2 += (l + W * k < C) ? __expf(e[k] * inv - inv) : 0.0f;
                } else {
                    for (int c = l; c < C; c += W) s2 += __expf(__expf(zr[c] - m) * inv - inv);
                }
                s2 = group_sum<W>(s2);
                loss = -(__expf(zy - m) * inv - inv - __logf(s2));
            } else {
                loss = -(zy - m - __logf(s));
            }
            if (!label_ok) loss = __builtin_nanf("");
            if (l == 0 && valid) {
                stats[row * 3 + 0] = m; stats[row * 3 + 1] = s; stats[row * 3 + 2] = s2;
                if (!mask || mask[row]) { bsum += loss; bcnt += 1.0f; }
            }
        }
    }
    s_sum[threadIdx.x] = bsum; s_cnt[threadIdx.x] = bcnt;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {                      // fixed tree => deterministic
        if (threadIdx.x < o) { s_sum[threadIdx.x] += s_sum[threadIdx.x + o]; s_cnt[threadIdx.x] += s_cnt[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s_sum[0]; partial[2 * blockIdx.x + 1] = s_cnt[0]; }
}

__global__ void xent_finish_kernel(const float* __restrict__ partial, int nb, float* __restrict__ loss,
                                   float* __restrict__ count) {
    __shared__ double s_sum[256], s_cnt[256];
    double a = 0.0, c = 0.0;
    for (int b = threadIdx.x; b < nb; b += 256) { a += partial[2 * b]; c += partial[2 * b + 1]; }
    s_sum[threadIdx.x] = a; s_cnt[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (threadIdx.x < o) { s_sum[threadIdx.x] += s_sum[threadIdx.x + o]; s_cnt[threadIdx.x] += s_cnt[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { *loss = (float)(s_sum[0] / s_cnt[0]); *count = (float)s_cnt[0]; }   // no rows: NaN, as torch
}

template <int W, int KC>
__global__ __launch_bounds__(256) void xent_bwd_kernel(const float* __restrict__ z, long ld, long N, int C,
                                                        const long* __restrict__ y, const unsigned char* __restrict__ mask,
                                                        int pre, const float* __restrict__ stats,
                                                        const float* __restrict__ count, const float* __restrict__ gloss,
                                                        float* __restrict__ gz, long ldg) {
    constexpr int G = 256 / W, KR = KC > 0 ? KC : 1;
    const int l = threadIdx.x & (W - 1), g = threadIdx.x / W;
    const float scale = gloss[0] / count[0];
    for (long base = (long)blockIdx.x * G * kXentRows; base < N; base += (long)gridDim.x * G * kXentRows) {
        float v[kXentRows][KR], st[kXentRows][3];
        long yv[kXentRows];
        bool on[kXentRows];
#pragma unroll
        for (int q = 0; q < kXentRows; ++q) {
            const long row = base + q * G + g, r = min(row, N - 1);
            yv[q] = y[r];
            on[q] = row < N && (!mask || mask[r]);
            st[q][0] = stats[r * 3]; st[q][1] = stats[r * 3 + 1]; st[q][2] = stats[r * 3 + 2];
            if (KC > 0) {
#pragma unroll
                for (int k = 0; k < KR; ++k) { const int c = l + W * k; v[q][k] = c < C ? z[r * ld + c] : -INFINITY; }
            }
        }
#pragma unroll
        for (int q = 0; q < kXentRows; ++q) {
            const long row = base + q * G + g;
            const bool valid = row < N;
            const float* zr = z + min(row, N - 1) * ld;
            const float m = st[q][0], inv = 1.0f / st[q][1], inv2 = 1.0f / st[q][2];
            float p[KR], d[KR], dot = 0.0f;
            if (KC > 0) {
#pragma unroll
                for (int k = 0; k < KR; ++k) {
                    p[k] = __expf(v[q][k] - m) * inv;                                  // 0 for c >= C
                    d[k] = pre ? __expf(p[k] - inv) * inv2 - (l + W * k == yv[q] ? 1.0f : 0.0f) : 0.0f;
                    dot = fmaf(d[k], p[k], dot);
                }
            } else if (pre) {
                for (int c = l; c < C; c += W) {
                    const float pc = __expf(zr[c] - m) * inv;
                    dot = fmaf(__expf(pc - inv) * inv2 - (c == yv[q] ? 1.0f : 0.0f), pc, dot);
                }
            }
            if (pre) dot = group_sum<W>(dot);
            if (!valid) continue;
            float* gr = gz + row * ldg;
            if (KC > 0) {
#pragma unroll
                for (int k = 0; k < KR; ++k) {
                    const int c = l + W * k;
                    const float val = pre ? p[k] * (d[k] - dot) : p[k] - (c == yv[q] ? 1.0f : 0.0f);
                    if (c < C) gr[c] = on[q] ? val * scale : 0.0f;
                }
            } else {
                for (int c = l; c < C; c += W) {
                    const float pc = __expf(zr[c] - m) * inv;
                    const float val = pre ? pc * (__expf(pc - inv) * inv2 - (c == yv[q] ? 1.0f : 0.0f) - dot)
                                          : pc - (c == yv[q] ? 1.0f : 0.0f);
                    gr[c] = on[q] ? val * scale : 0.0f;
                }
            }
        }
    }
}

// ---- few classes (C <= 64, every KAGNN dataset): one THREAD per row.  A wave per row spends ~100 wave instructions
// on 40 useful values (three cross-lane reductions); here a workgroup stages 256 rows through LDS with coalesced
// loads (row stride odd => conflict-free), each thread walks its own row, and the backward writes its rows back
// through the same tile.  ~10 wave instructions per row.
__device__ __forceinline__ void xent_stage_in(const float* __restrict__ z, long ld, long N, int C, long row0,
                                              float* tile, int stride) {
    const int c = threadIdx.x & 63, rq = threadIdx.x >> 6;
#pragma unroll 8
    for (int p = 0; p < 64; ++p) {
        const int r = 4 * p + rq;
        const long row = min(row0 + r, N - 1);
        if (c < C) tile[r * stride + c] = z[row * ld + c];
    }
}

__global__ __launch_bounds__(256) void xent_fwd_rows_kernel(const float* __restrict__ z, long ld, long N, int C,
                                                             const long* __restrict__ y, const unsigned char* __restrict__ mask,
                                                             int pre, float* __restrict__ stats, float* __restrict__ partial) {
    extern __shared__ float tile[];                       // [256][stride]
    __shared__ float s_sum[256], s_cnt[256];
    const int stride = C | 1;
    float bsum = 0.0f, bcnt = 0.0f;
    for (long row0 = (long)blockIdx.x * 256; row0 < N; row0 += (long)gridDim.x * 256) {
        __syncthreads();
        xent_stage_in(z, ld, N, C, row0, tile, stride);
        __syncthreads();
        const long row = row0 + threadIdx.x;
        const float* v = tile + threadIdx.x * stride;
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, v[c]);
        float s = 0.0f;
        for (int c = 0; c < C; ++c) s += __expf(v[c] - m);
        const long yv = y[min(row, N - 1)];
        const bool label_ok = yv >= 0 && yv < C;
        const float zy = v[label_ok ? yv : 0];
        float s2 = 0.0f, loss;
        if (pre) {
            const float inv = 1.0f / s;
            for (int c = 0; c < C; ++c) s2 += __expf(__expf(v[c] - m) * inv - inv);
            loss = -(__expf(zy - m) * inv - inv - __logf(s2));
        } else {
            loss = -(zy - m - __logf(s));
        }
        if (!label_ok) loss = __builtin_nanf("");
        if (row < N) {
            stats[row * 3 + 0] = m; stats[row * 3 + 1] = s; stats[row * 3 + 2] = s2;
            if (!mask || mask[row]) { bsum += loss; bcnt += 1.0f; }
        }
    }
    s_sum[threadIdx.x] = bsum; s_cnt[threadIdx.x] = bcnt;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {                      // fixed tree => deterministic
        if (threadIdx.x < o) { s_sum[threadIdx.x] += s_sum[threadIdx.x + o]; s_cnt[threadIdx.x] += s_cnt[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s_sum[0]; partial[2 * blockIdx.x + 1] = s_cnt[0]; }
}

__global__ __launch_bounds__(256) void xent_bwd_rows_kernel(const float* __restrict__ z, long ld, long N, int C,
                                                             const long* __restrict__ y, const unsigned char* __restrict__ mask,
                                                             int pre, const float* __restrict__ stats,
                                                             const float* __restrict__ count, const float* __restrict__ gloss,
                                                             float* __restrict__ gz, long ldg) {
    extern __shared__ float tile[];
    const int stride = C | 1;
    const float scale = gloss[0] / count[0];
    for (long row0 = (long)blockIdx.x * 256; row0 < N; row0 += (long)gridDim.x * 256) {
        __syncthreads();
        xent_stage_in(z, ld, N, C, row0, tile, stride);
        __syncthreads();
        const long row = row0 + threadIdx.x, r = min(row, N - 1);
        float* v = tile + threadIdx.x * stride;
        const bool on = row < N && (!mask || mask[r]);
        const float m = stats[r * 3], inv = 1.0f / stats[r * 3 + 1], inv2 = 1.0f / stats[r * 3 + 2];
        const long yv = y[r];
        const float sc = on ? scale : 0.0f;
        if (pre) {
            float dot = 0.0f;
            for (int c = 0; c < C; ++c) {
                const float p = __expf(v[c] - m) * inv;
                dot = fmaf(__expf(p - inv) * inv2 - (c == yv ? 1.0f : 0.0f), p, dot);
            }
            for (int c = 0; c < C; ++c) {
                const float p = __expf(v[c] - m) * inv;
                v[c] = p * (__expf(p - inv) * inv2 - (c == yv ? 1.0f : 0.0f) - dot) * sc;
            }
        } else {
            for (int c = 0; c < C; ++c) v[c] = (__expf(v[c] - m) * inv - (c == yv ? 1.0f : 0.0f)) * sc;
        }
        __syncthreads();
        const int c = threadIdx.x & 63, rq = threadIdx.x >> 6;
#pragma unroll 8
        for (int p = 0; p < 64; ++p) {
            const int rr = 4 * p + rq;
            if (c < C && row0 + rr < N) gz[(row0 + rr) * ldg + c] = tile[rr * stride + c];
        }
    }
}

static int xent_blocks(long N, int W) { return (int)max(1L, min((long)cdiv(N, (256 / W) * kXentRows), 4096L)); }
static int xent_width(int C) { int w = 4; while (w < C && w < 64) w <<= 1; return w; }

size_t xent_ws_bytes(long N) { return (size_t)2 * 4096 * sizeof(float); }

int xent_fwd(const float* z, long ld, long N, int C, const long* y, const unsigned char* mask, int pre, float* loss,
             float* stats, float* count, void* ws, size_t ws_bytes, hipStream_t st) {
    if (ws_bytes < xent_ws_bytes(N)) return fail(KAGNN_ERR_ARG, "%s: workspace too small", "xent_fwd");
    float* partial = (float*)ws;
    const int W = xent_width(C);
    int nb = N > 0 ? xent_blocks(N, W) : 0;
    // 9..64 classes: 8 lanes per row, up to 8 classes per lane in registers, no LDS, no barriers (0.115 vs 0.175 ms forward +
    // backward at 1M x 40 against the row-per-thread kernels over an LDS tile, which KAGNN_XENT_ROWS=1 still selects)
    static const bool w8 = [] { const char* e = getenv("KAGNN_XENT_ROWS"); return e == nullptr || atoi(e) == 0; }();
    if (w8 && C > 8 && C <= 64 && N > 0) {
        nb = xent_blocks(N, 8);
        if (C <= 40) xent_fwd_kernel<8, 5><<<nb, 256, 0, st>>>(z, ld, N, C, y, mask, pre, stats, partial);
        else xent_fwd_kernel<8, 8><<<nb, 256, 0, st>>>(z, ld, N, C, y, mask, pre, stats, partial);
    } else
    if (C <= 64 && N > 0) {
        nb = (int)min((long)cdiv(N, 256), 4096L);
        static unsigned long long big_lds = 0;                     // 256 rows x 65 floats is just over the 64 KB default
        if (auto first_use_ = first_use_on_this_device(big_lds)) {
            KAGNN_HIP(hipFuncSetAttribute((const void*)xent_fwd_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 68 * 1024));
            KAGNN_HIP(hipFuncSetAttribute((const void*)xent_bwd_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 68 * 1024));
        }
        xent_fwd_rows_kernel<<<nb, 256, (size_t)256 * (C | 1) * sizeof(float), st>>>(z, ld, N, C, y, mask, pre, stats, partial);
    } else
#define L(WW, KK) xent_fwd_kernel<WW, KK><<<nb, 256, 0, st>>>(z, ld, N, C, y, mask, pre, stats, partial)
    if (nb) switch (W) {
        case 4: L(4, 1); break; case 8: L(8, 1); break; case 16: L(16, 1); break; case 32: L(32, 1); break;
        default: if (C <= 64) L(64, 1); else if (C <= 128) L(64, 2); else if (C <= 256) L(64, 4); else L(64, 0);
    }
#undef L
    KAGNN_LAUNCH_CHECK();
    xent_finish_kernel<<<1, 256, 0, st>>>(partial, nb, loss, count);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

int xent_bwd(const float* z, long ld, long N, int C, const long* y, const unsigned char* mask, int pre,
             const float* stats, const float* count, const float* gloss, float* gz, long ldg, hipStream_t st) {
    if (N == 0) return KAGNN_OK;
    const int W = xent_width(C), nb = xent_blocks(N, W);
    static const bool w8 = [] { const char* e = getenv("KAGNN_XENT_ROWS"); return e == nullptr || atoi(e) == 0; }();
    if (w8 && C > 8 && C <= 64) {
        const int nb8 = xent_blocks(N, 8);
        if (C <= 40) xent_bwd_kernel<8, 5><<<nb8, 256, 0, st>>>(z, ld, N, C, y, mask, pre, stats, count, gloss, gz, ldg);
        else xent_bwd_kernel<8, 8><<<nb8, 256, 0, st>>>(z, ld, N, C, y, mask, pre, stats, count, gloss, gz, ldg);
        KAGNN_LAUNCH_CHECK();
        return KAGNN_OK;
    }
    if (C <= 64) {
        const int nbr = (int)min((long)cdiv(N, 256), 4096L);
        xent_bwd_rows_kernel<<<nbr, 256, (size_t)256 * (C | 1) * sizeof(float), st>>>(z, ld, N, C, y, mask, pre, stats, count, gloss, gz, ldg);
        KAGNN_LAUNCH_CHECK();
        return KAGNN_OK;
    }
#define L(WW, KK) xent_bwd_kernel<WW, KK><<<nb, 256, 0, st>>>(z, ld, N, C, y, mask, pre, stats, count, gloss, gz, ldg)
    switch (W) {
        case 4: L(4, 1); break; case 8: L(8, 1); break; case 16: L(16, 1); break; case 32: L(32, 1); break;
        default: if (C <= 64) L(64, 1); else if (C <= 128) L(64, 2); else if (C <= 256) L(64, 4); else L(64, 0);
    }
#undef L
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// ------------------------------------------------------------------ mean absolute error (the graph-regression scripts' loss:
// graph_regression/optuna_zinc.py:58 `torch.nn.L1Loss()(model(data).squeeze(), data.y)`), one launch each way instead of aten's
// sub / abs / mean and sign / expand / mul -- on a 256-molecule mini-batch the loss is 256 numbers and six launches.
// forward: loss = (1/n) sum |p_i - t_i|, one workgroup: thread j adds elements j, j + 1024, ... in order, the 1024 partial sums
// fold pairwise through LDS (fixed order: deterministic).  Meant for the [graphs] / [graphs, targets] predictions of a mini-batch;
// a million elements still work, at one workgroup's bandwidth.
__global__ __launch_bounds__(1024) void l1_loss_fwd_kernel(const float* __restrict__ p, const float* __restrict__ t, long n,
                                                           float* __restrict__ loss) {
    __shared__ float s_a[1024];
    float a = 0.0f;
    for (long i = threadIdx.x; i < n; i += 1024) a += fabsf(p[i] - t[i]);
    s_a[threadIdx.x] = a;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) s_a[threadIdx.x] += s_a[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = s_a[0] / (float)n;
}

// backward: g_p[i] = (g_loss * (1 / n)) * sign(p_i - t_i)   (sign(0) = sign(NaN) = 0, as aten's sign: `(0 < d) - (d < 0)`)
__global__ __launch_bounds__(256) void l1_loss_bwd_kernel(const float* __restrict__ p, const float* __restrict__ t, long n,
                                                          const float* __restrict__ g_loss, float* __restrict__ g_p) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= n) return;
    const float d = p[i] - t[i], g = g_loss[0] * (1.0f / (float)n);      // (aten divides by a host scalar as a multiplication by its reciprocal)
    g_p[i] = d > 0.0f ? g : d < 0.0f ? -g : g * 0.0f;
}

int l1_loss_fwd(const float* p, const float* t, long n, float* loss, hipStream_t st) {
    l1_loss_fwd_kernel<<<1, 1024, 0, st>>>(p, t, n, loss);        // n == 0: 0 / 0 = NaN, as aten's mean of an empty tensor
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

int l1_loss_bwd(const float* p, const float* t, long n, const float* g_loss, float* g_p, hipStream_t st) {
    if (n == 0) return KAGNN_OK;
    l1_loss_bwd_kernel<<<cdiv(n, 256), 256, 0, st>>>(p, t, n, g_loss, g_p);
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

// ------------------------------------------------------------------ Adam for the mini-batch training loop (graph_regression/
// optuna_zinc.py:49,62: `torch.optim.Adam(model.parameters(), lr=...)`, `optimizer.step()` per batch).  On a 256-molecule batch the
// whole step is ~0.8 ms of device work and torch's optimiser -- fused or not -- costs the HOST 0.2-0.3 ms per step in Python
// (state bookkeeping, tensor grouping, step counters as tensors); this is the same rule as one launch over all parameter tensors,
// driven by pointer tables in the kernel arguments.  fp32, no amsgrad, L2 weight decay as torch's (grad += wd * param):
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
constexpr int kAdamBatch = 64;      // (2.6 KB of kernel arguments; the ZINC model's 40 parameter tensors are ONE launch -- round 6: was 32 = two)
struct AdamBatch {
    float* p[kAdamBatch]; const float* g[kAdamBatch]; float* m[kAdamBatch]; float* v[kAdamBatch]; long n[kAdamBatch];
};
__global__ __launch_bounds__(256) void adam_step_kernel(const AdamBatch b, float step_size, float inv_bc2_sqrt, float beta1, float beta2,
                                                        float eps, float weight_decay) {
    const int k = blockIdx.y;
    const long n = b.n[k];
    float* __restrict__ p = b.p[k];
    const float* __restrict__ g = b.g[k];
    float* __restrict__ m = b.m[k];
    float* __restrict__ v = b.v[k];
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float gi = g[i];
        const float pi = p[i];
        if (weight_decay != 0.0f) gi = fmaf(weight_decay, pi, gi);
        const float mi = fmaf(beta1, m[i], (1.0f - beta1) * gi);
        const float vi = fmaf(beta2, v[i], (1.0f - beta2) * gi * gi);
        m[i] = mi; v[i] = vi;
        p[i] = pi - step_size * mi / (sqrtf(vi) * inv_bc2_sqrt + eps);
    }
}

int adam_step(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
              const long* numel, float lr, float beta1, float beta2, float eps, float weight_decay, long step, hipStream_t st) {
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    for (int k0 = 0; k0 < count; k0 += kAdamBatch) {
        AdamBatch b{};
        const int nb = min(kAdamBatch, count - k0);
        long nmax = 1;
        for (int k = 0; k < nb; ++k) {
            b.p[k] = params[k0 + k]; b.g[k] = grads[k0 + k]; b.m[k] = exp_avg[k0 + k]; b.v[k] = exp_avg_sq[k0 + k]; b.n[k] = numel[k0 + k];
            nmax = max(nmax, b.n[k]);
        }
        const unsigned gx = (unsigned)min((long)cdiv(nmax, 256 * 4), 256L);
        adam_step_kernel<<<dim3(max(gx, 1u), (unsigned)nb), 256, 0, st>>>(b, step_size, inv_bc2_sqrt, beta1, beta2, eps, weight_decay);
        KAGNN_LAUNCH_CHECK();
    }
    return KAGNN_OK;
}

}  // namespace kagnn
