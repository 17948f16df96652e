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

// W lanes per row (power of two <= 64), 256 / W rows per workgroup pass.  stats[row] = (max, sum exp, sum exp of the
// second softmax); partial[2b], partial[2b+1] = this workgroup's loss sum and row count.
template <int W>
__global__ __launch_bounds__(256) void xent_fwd_kernel(const float* __restrict__ z, long ld, long N, int C,
                                                        const long* __restrict__ y, const unsigned char* __restrict__ mask,
                                                        int pre, float* __restrict__ stats, float* __restrict__ partial) {
    constexpr int G = 256 / W;
    __shared__ float s_sum[256], s_cnt[256];
    const int l = threadIdx.x & (W - 1), g = threadIdx.x / W;
    float bsum = 0.0f, bcnt = 0.0f;
    for (long base = (long)blockIdx.x * G; base < N; base += (long)gridDim.x * G) {
        const long row = base + g;
        const bool valid = row < N;
        const float* zr = z + min(row, N - 1) * ld;
        float m = -INFINITY;
        for (int c = l; c < C; c += W) m = fmaxf(m, zr[c]);
        m = group_max<W>(m);
        float s = 0.0f;
        for (int c = l; c < C; c += W) s += __expf(zr[c] - m);
        s = group_sum<W>(s);
        const long yv = y[min(row, N - 1)];
        const bool label_ok = yv >= 0 && yv < C;
        const float zy = zr[label_ok ? yv : 0];
        float s2 = 0.0f, loss;
        if (pre) {
            const float inv = 1.0f / s;                      // = the largest probability: the second softmax's shift
            for (int c = l; c < C; c += W) s2 += __expf(__expf(zr[c] - m) * inv - inv);
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

template <int W>
__global__ __launch_bounds__(256) void xent_bwd_kernel(const float* __restrict__ z, long ld, long N, int C,
                                                        const long* __restrict__ y, const unsigned char* __restrict__ mask,
                                                        int pre, const float* __restrict__ stats,
                                                        const float* __restrict__ count, const float* __restrict__ gloss,
                                                        float* __restrict__ gz, long ldg) {
    constexpr int G = 256 / W;
    const int l = threadIdx.x & (W - 1), g = threadIdx.x / W;
    const float scale = gloss[0] / count[0];
    for (long base = (long)blockIdx.x * G; base < N; base += (long)gridDim.x * G) {
        const long row = base + g;
        const bool valid = row < N;
        const long r = min(row, N - 1);
        const float* zr = z + r * ld;
        const bool on = valid && (!mask || mask[r]);
        const float m = stats[r * 3], inv = 1.0f / stats[r * 3 + 1], inv2 = 1.0f / stats[r * 3 + 2];
        const long yv = y[r];
        float dot = 0.0f;
        if (pre) {
            for (int c = l; c < C; c += W) {
                const float p = __expf(zr[c] - m) * inv;
                const float d = __expf(p - inv) * inv2 - (c == yv ? 1.0f : 0.0f);
                dot = fmaf(d, p, dot);
            }
            dot = group_sum<W>(dot);
        }
        if (!valid) continue;
        float* gr = gz + row * ldg;
        for (int c = l; c < C; c += W) {
            const float p = __expf(zr[c] - m) * inv;
            float v;
            if (pre) v = p * (__expf(p - inv) * inv2 - (c == yv ? 1.0f : 0.0f) - dot);
            else v = p - (c == yv ? 1.0f : 0.0f);
            gr[c] = on ? v * scale : 0.0f;
        }
    }
}

static int xent_blocks(long N, int W) { return (int)max(1L, min((long)cdiv(N, 256 / W), 2048L)); }
static int xent_width(int C) { int w = 4; while (w < C && w < 64) w <<= 1; return w; }

size_t xent_ws_bytes(long N) { return (size_t)2 * 2048 * sizeof(float); }

int xent_fwd(const float* z, long ld, long N, int C, const long* y, const unsigned char* mask, int pre, float* loss,
             float* stats, float* count, void* ws, size_t ws_bytes, hipStream_t st) {
    if (ws_bytes < xent_ws_bytes(N)) return fail(KAGNN_ERR_ARG, "%s: workspace too small", "xent_fwd");
    float* partial = (float*)ws;
    const int W = xent_width(C), nb = N > 0 ? xent_blocks(N, W) : 0;
#define L(WW) xent_fwd_kernel<WW><<<nb, 256, 0, st>>>(z, ld, N, C, y, mask, pre, stats, partial)
    if (nb) switch (W) { case 4: L(4); break; case 8: L(8); break; case 16: L(16); break; case 32: L(32); break; default: L(64); }
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
#define L(WW) xent_bwd_kernel<WW><<<nb, 256, 0, st>>>(z, ld, N, C, y, mask, pre, stats, count, gloss, gz, ldg)
    switch (W) { case 4: L(4); break; case 8: L(8); break; case 16: L(16); break; case 32: L(32); break; default: L(64); }
#undef L
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

}  // namespace kagnn
