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
    const int L = max(T / 4, 32);                      // segment length: short segments = more workgroups per hub,
    const int nseg = (t - s + L - 1) / L;             // a hub kernel is pure latency otherwise (<= T edges each, as documented)
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
