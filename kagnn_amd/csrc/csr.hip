// CSR construction from the reference's int64 edge_index: a stable LSD radix sort by key
// (rocPRIM device primitive), so perm == argsort(key, stable=True) bit-for-bit -- the integer
// contract SURVEY.md 8(c) G7 pins.  Built once per edge_index and cached by the host side; this
// replaces the per-forward index bookkeeping of torch_geometric's propagate().
#include "common.h"
#include <rocprim/rocprim.hpp>

namespace kagnn {

__global__ void csr_prepare_kernel(const int64_t* __restrict__ key, long E, long N,
                                   int* __restrict__ k32, int* __restrict__ ids,
                                   int* __restrict__ flags) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t k = key[e];
    if (k < 0 || k >= N) { atomicOr(flags, 1); k32[e] = 0; }
    else k32[e] = (int)k;
    ids[e] = (int)e;
}

__global__ void csr_finish_kernel(const int* __restrict__ ksorted, const int* __restrict__ perm,
                                  const int64_t* __restrict__ val, long E, long N,
                                  int* __restrict__ rowptr, int* __restrict__ col,
                                  int* __restrict__ flags) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t v = val[perm[e]];
    if (v < 0 || v >= N) atomicOr(flags, 2);
    col[e] = (int)v;
    const int k = ksorted[e];
    const int kprev = (e == 0) ? -1 : ksorted[e - 1];
    for (int r = kprev + 1; r <= k; ++r) rowptr[r] = (int)e;       // rows (kprev, k] start at e
    if (e == E - 1)
        for (long r = (long)k + 1; r <= N; ++r) rowptr[r] = (int)E;  // trailing empty rows + end
}

__global__ void csr_hub_kernel(const int* __restrict__ rowptr, long N, int T, int* __restrict__ seg,
                               long cap, int* __restrict__ counter) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int s = rowptr[i], t = rowptr[i + 1];
    if (t - s <= T) return;
    // segment length: short segments = more workgroups per hub (a hub kernel is pure latency otherwise); 64, not 32, at the default
    // threshold of 96: half the workgroups with four gathers per lane group in flight, -0.014 ms per step (profiles/r06_experiments.md 8)
    const int L = max(T / 4, 64);
    const int nseg = (t - s + L - 1) / L;
    const int base = atomicAdd(counter, nseg);
    for (int k = 0; k < nseg; ++k) {
        if (base + k < cap) {
            seg[3 * (base + k) + 0] = (int)i;
            seg[3 * (base + k) + 1] = s + k * L;
            seg[3 * (base + k) + 2] = min(t, s + (k + 1) * L);
        }
    }
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---------------------------------------------------------------------------------------------------------------
// Small graphs (E, N <= 65 536: the mini-batches of the graph-level models, BASELINE config 4 -- 256 molecules are ~6k nodes /
// ~13k edges, and the CSR is rebuilt for EVERY batch: graph_regression/optuna_zinc.py:56-66 hands each batch's edge_index to the
// convolutions): BOTH structures -- by destination and its transpose by source -- in ONE launch of two workgroups and without a host
// round trip, instead of 2 x (memset + prepare + ~8 rocPRIM launches + finish + hub) and 2 stream synchronisations.  Each workgroup
// runs a stable LSD radix sort of (key, edge id) with 4-bit digits: a thread owns a CONTIGUOUS block of the elements, counts its
// digits privately, the [digit][thread] table is scanned in LDS, and the thread scatters its block in order -- every pass is stable,
// so perm == argsort(key, stable=True) bit for bit, exactly what the rocPRIM path produces (tests: G7).  Node ids outside
// [0, N) are CLAMPED (no out-of-bounds access can follow) and flagged in flags[side]; the host binding reads the flags without
// blocking the stream (ops.GraphIndex: deferred validation).  No hub segments: every row goes to the row kernels.
constexpr int kSmallMax = 1 << 16;
constexpr int kSmallThreads = 1024;
bool csr_small_ok(long E, long N) { return E >= 1 && E <= kSmallMax && N >= 1 && N <= kSmallMax; }
size_t csr_small_workspace_bytes(long E) { return 2 * 4 * align256((size_t)E * 4); }

// INLDS: the (key, edge id) pairs live in LDS as ONE 32-bit word each (key << 16 | id: both < 65 536), two buffers of E words behind
// the counter table -- E <= kSmallLdsMax (a 256-molecule batch has ~13k edges).  With the pairs in global scratch the kernel is a
// chain of dependent L2 round trips per element and pass: 117 us for 12.7k edges, as long as the rocPRIM launches it replaces.
constexpr int kSmallLdsMax = 14336;           // 2 x 4 B x 14336 = 112 KiB + 32 KiB of counters
template <bool INLDS>
__global__ __launch_bounds__(kSmallThreads) void csr_small_pair_kernel(
    const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int E, int N, int bits,
    int* __restrict__ rowptr_d, int* __restrict__ col_d, int* __restrict__ perm_d,
    int* __restrict__ rowptr_s, int* __restrict__ col_s, int* __restrict__ perm_s,
    int* __restrict__ flags, int* __restrict__ ws, long ws_side /* ints per side */, long ws_arr /* ints per array */) {
    constexpr int T = kSmallThreads;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_csr[];
    unsigned short* s_cnt = reinterpret_cast<unsigned short*>(smem_csr);          // [16][T]: counts <= 64, exclusive offsets <= E - 1
    int* s_part = reinterpret_cast<int*>(smem_csr + 16 * T * 2);                   // [T / 64]
    unsigned* s_buf = reinterpret_cast<unsigned*>(smem_csr + 16 * T * 2 + 64);     // INLDS: two buffers of E packed words
    const int side = blockIdx.x, tid = threadIdx.x;
    const int64_t* key = side == 0 ? dst : src;
    const int64_t* val = side == 0 ? src : dst;
    int* rowptr = side == 0 ? rowptr_d : rowptr_s;
    int* col = side == 0 ? col_d : col_s;
    int* perm = side == 0 ? perm_d : perm_s;
    unsigned* bufA = INLDS ? s_buf : reinterpret_cast<unsigned*>(ws + side * ws_side);
    unsigned* bufB = INLDS ? s_buf + E : reinterpret_cast<unsigned*>(ws + side * ws_side + ws_arr);
    const int c = (E + T - 1) / T;                         // elements per thread, contiguous
    const int e0 = min(tid * c, E), e1 = min(e0 + c, E);
    int bad = 0;
    for (int e = e0; e < e1; ++e) {
        const int64_t k = key[e];
        const bool ok = k >= 0 && k < N;
        bad |= ok ? 0 : 1;
        bufA[e] = ((ok ? (unsigned)k : 0u) << 16) | (unsigned)e;          // (E <= 65 536: ids fit 16 bits; id 65 535 is the last one)
    }
    __syncthreads();
    const int passes = (bits + 3) / 4;
    unsigned* bs = bufA; unsigned* bd = bufB;
    for (int p = 0; p < passes; ++p) {
        const int shift = 16 + 4 * p;
        int cnt[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) cnt[d] = 0;
        for (int e = e0; e < e1; ++e) {
            const int d = (bs[e] >> shift) & 15;
#pragma unroll
            for (int q = 0; q < 16; ++q) cnt[q] += (q == d) ? 1 : 0;       // (register array: no dynamic indexing)
        }
#pragma unroll
        for (int d = 0; d < 16; ++d) s_cnt[d * T + tid] = (unsigned short)cnt[d];
        __syncthreads();
        // exclusive scan of the 16 T counters in (digit, thread) order: thread t owns entries [16 t, 16 t + 16)
        int run[16], sum = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) { run[j] = sum; sum += (int)s_cnt[16 * tid + j]; }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if ((tid & 63) >= o) incl += v; }
        if ((tid & 63) == 63) s_part[tid >> 6] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += s_part[w];
        base += incl - sum;
#pragma unroll
        for (int j = 0; j < 16; ++j) s_cnt[16 * tid + j] = (unsigned short)(base + run[j]);
        __syncthreads();
        int off[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) off[d] = (int)s_cnt[d * T + tid];
        for (int e = e0; e < e1; ++e) {
            const unsigned w = bs[e];
            const int d = (w >> shift) & 15;
            int pos = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) { pos = (q == d) ? off[q] : pos; off[q] += (q == d) ? 1 : 0; }
            bd[pos] = w;
        }
        __syncthreads();
        unsigned* t0 = bs; bs = bd; bd = t0;
    }
    // sorted (key, edge id) words -> perm, col, rowptr (as csr_finish_kernel)
    for (int e = tid; e < E; e += T) {
        const unsigned w = bs[e];
        const int pe = (int)(w & 0xffffu);
        perm[e] = pe;
        const int64_t v = val[pe];
        const bool ok = v >= 0 && v < N;
        bad |= ok ? 0 : 2;
        col[e] = ok ? (int)v : 0;
        const int k = (int)(w >> 16);
        const int kprev = (e == 0) ? -1 : (int)(bs[e - 1] >> 16);
        for (int r = kprev + 1; r <= k; ++r) rowptr[r] = e;
        if (e == E - 1)
            for (int r = k + 1; r <= N; ++r) rowptr[r] = E;
    }
    if (bad) atomicOr(flags + side, bad);
}

int csr_build_small(const int64_t* src, const int64_t* dst, long E, long N, int* rowptr, int* col, int* perm,
                    int* rowptr_t, int* col_t, int* perm_t, int* flags, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!csr_small_ok(E, N)) return fail(KAGNN_ERR_UNSUPPORTED, "%s: 1 <= E, N <= 65536", "csr_build_small");
    if (ws_bytes < csr_small_workspace_bytes(E)) return fail(KAGNN_ERR_ARG, "%s: workspace too small", "csr_build_small");
    int bits = 1;
    while ((1L << bits) < N && bits < 31) ++bits;
    const long arr = (long)(align256((size_t)E * 4) / 4);
    KAGNN_HIP(hipMemsetAsync(flags, 0, 2 * sizeof(int), st));
    const size_t lds_base = (size_t)16 * kSmallThreads * 2 + 64;
    if (E <= kSmallLdsMax) {
        const size_t lds = lds_base + 2 * (size_t)E * 4;
        static unsigned long long configured = 0;          // (per device: common.h)
        if (auto first_use_ = first_use_on_this_device(configured))
            KAGNN_HIP(hipFuncSetAttribute((const void*)csr_small_pair_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        csr_small_pair_kernel<true><<<2, kSmallThreads, lds, st>>>(src, dst, (int)E, (int)N, bits, rowptr, col, perm, rowptr_t, col_t, perm_t,
                                                                  flags, static_cast<int*>(ws), 4 * arr, arr);
    } else {
        csr_small_pair_kernel<false><<<2, kSmallThreads, lds_base, st>>>(src, dst, (int)E, (int)N, bits, rowptr, col, perm, rowptr_t, col_t,
                                                                        perm_t, flags, static_cast<int*>(ws), 4 * arr, arr);
    }
    KAGNN_LAUNCH_CHECK();
    return KAGNN_OK;
}

static int sort_temp_bytes(long E, int bits, size_t* out) {
    size_t tmp = 0;
    KAGNN_HIP(rocprim::radix_sort_pairs<rocprim::default_config, const int*, int*, const int*, int*>(
        nullptr, tmp, nullptr, nullptr, nullptr, nullptr, (size_t)E, 0, bits, nullptr, false));
    *out = tmp;
    return KAGNN_OK;
}

static int key_bits(long N) {
    int b = 1;
    while ((1L << b) < N && b < 31) ++b;
    return b;
}

int csr_workspace_bytes(long E, long N, size_t* bytes) {
    size_t tmp = 0;
    if (E > 0) { int rc = sort_temp_bytes(E, key_bits(N), &tmp); if (rc) return rc; }
    *bytes = 3 * align256((size_t)E * 4) + align256(tmp) + 256;
    return KAGNN_OK;
}

int csr_build(const int64_t* key, const int64_t* val, long E, long N, int* rowptr, int* col,
              int* perm, int T, int* hub_seg, long cap, int64_t* nseg_host, void* ws,
              size_t ws_bytes, hipStream_t st) {
    size_t need = 0;
    { int rc = csr_workspace_bytes(E, N, &need); if (rc) return rc; }
    if (ws_bytes < need) return fail(KAGNN_ERR_ARG, "%s: workspace too small", "csr_build");
    if (nseg_host) *nseg_host = 0;
    if (E == 0) {
        KAGNN_HIP(hipMemsetAsync(rowptr, 0, (size_t)(N + 1) * 4, st));
        KAGNN_HIP(hipStreamSynchronize(st));
        return KAGNN_OK;
    }
    char* p = static_cast<char*>(ws);
    int* flags = reinterpret_cast<int*>(p);             // [0] range errors, [1] hub counter
    p += 256;
    int* k32 = reinterpret_cast<int*>(p); p += align256((size_t)E * 4);
    int* ids = reinterpret_cast<int*>(p); p += align256((size_t)E * 4);
    int* ksorted = reinterpret_cast<int*>(p); p += align256((size_t)E * 4);
    void* tmp = p;
    size_t tmp_bytes = 0;
    const int bits = key_bits(N);
    { int rc = sort_temp_bytes(E, bits, &tmp_bytes); if (rc) return rc; }

    KAGNN_HIP(hipMemsetAsync(flags, 0, 256, st));
    csr_prepare_kernel<<<cdiv(E, 256), 256, 0, st>>>(key, E, N, k32, ids, flags);
    KAGNN_LAUNCH_CHECK();
    KAGNN_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, (const int*)k32, ksorted, (const int*)ids, perm,
                                        (size_t)E, 0, bits, st, false));
    csr_finish_kernel<<<cdiv(E, 256), 256, 0, st>>>(ksorted, perm, val, E, N, rowptr, col, flags);
    KAGNN_LAUNCH_CHECK();
    if (hub_seg && T > 0 && cap > 0) {
        csr_hub_kernel<<<cdiv(N, 256), 256, 0, st>>>(rowptr, N, T, hub_seg, cap, flags + 1);
        KAGNN_LAUNCH_CHECK();
    }
    int h[2] = {0, 0};
    KAGNN_HIP(hipMemcpyAsync(h, flags, sizeof(h), hipMemcpyDeviceToHost, st));
    KAGNN_HIP(hipStreamSynchronize(st));
    if (h[0]) return fail(KAGNN_ERR_ARG, "%s: edge_index holds node ids outside [0, num_nodes)", "csr_build");
    if (h[1] > cap) return fail(KAGNN_ERR_ARG, "%s: hub_seg capacity too small", "csr_build");
    if (nseg_host) *nseg_host = h[1];
    return KAGNN_OK;
}

}  // namespace kagnn
