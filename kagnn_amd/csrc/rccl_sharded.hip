// libkagnn_rccl.so (include/kagnn_rccl.h): the feature-sharded KANLinear with its exchange step on a caller-owned
// ncclComm_t -- SURVEY.md 8(b)'s "sharded variants taking an ncclComm_t".  Host-side sequencing only, plus the two small
// layout kernels RCCL's contiguous-block collectives need; the KAN kernels are libkagnn_hip.so's (called through its C ABI,
// so both libraries can be rebuilt independently).  The reference has no multi-GPU code; contract:
// BASELINE.json north_star, SURVEY.md 8(e).  What each rank computes locally is KANLinear.forward on its input-feature
// slice (node_classification_clean/ekan.py:154-162) and its autograd backward.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <mutex>
#include <vector>
#include "../../include/kagnn_hip.h"
#include "../../include/kagnn_rccl.h"

#define KAGNN_RCCL_API extern "C" __attribute__((visibility("default")))

namespace {

thread_local char g_err[640] = "";

int fail(int code, const char* fmt, const char* a = "", const char* b = "") {
    snprintf(g_err, sizeof(g_err), fmt, a, b);
    return code;
}
#define R_ARG(cond, msg)                                                                               \
    do {                                                                                               \
        if (!(cond)) return fail(KAGNN_ERR_ARG, "%s: argument check failed: " msg, __func__);          \
    } while (0)
#define R_HIP(...)                                                                                     \
    do {                                                                                               \
        hipError_t e_ = (__VA_ARGS__);                                                                 \
        if (e_ != hipSuccess) return fail(KAGNN_ERR_HIP, "%s: HIP error -> %s", __func__, hipGetErrorString(e_)); \
    } while (0)
#define R_NCCL(...)                                                                                    \
    do {                                                                                               \
        ncclResult_t r_ = (__VA_ARGS__);                                                               \
        if (r_ != ncclSuccess) return fail(KAGNN_ERR_HIP, "%s: RCCL error -> %s", __func__, ncclGetErrorString(r_)); \
    } while (0)
// a failing libkagnn_hip call: carry its message
#define R_LIB(...)                                                                                     \
    do {                                                                                               \
        int c_ = (__VA_ARGS__);                                                                        \
        if (c_ != KAGNN_OK) { const char* m_ = kagnn_last_error(); return fail(c_, "%s: %s", __func__, m_ ? m_ : "?"); } \
    } while (0)

// ---- events: a small pool per device (creating one costs microseconds; an exchange needs one per row chunk)
struct EventPool {
    std::mutex mu;
    std::map<int, std::vector<hipEvent_t>> ev;
    size_t next = 0;
    hipError_t get(hipEvent_t* out) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> lk(mu);
        auto& v = ev[dev];
        if (v.size() < 64) {
            hipEvent_t h;
            e = hipEventCreateWithFlags(&h, hipEventDisableTiming);
            if (e != hipSuccess) return e;
            v.push_back(h);
            *out = h;
            return hipSuccess;
        }
        *out = v[next++ % v.size()];                 // (re-recording an event does not disturb waits already enqueued on it)
        return hipSuccess;
    }
};
EventPool g_events;

// ---- row chunks: whole 256-row kernel tiles from 4096 rows up (kagnn_amd/sharded.py::_chunk_bounds)
struct Bounds { long r0, r1; };
std::vector<Bounds> chunk_bounds(long n, int chunks) {
    std::vector<Bounds> b;
    if (n <= 0) { b.push_back({0, 0}); return b; }
    long c = chunks < 1 ? 1 : chunks;
    if (c > n) c = n;
    long per = (n + c - 1) / c;
    if (n >= 4096) per = (per + 255) / 256 * 256;
    for (long r = 0; r < n; r += per) b.push_back({r, r + per < n ? r + per : n});
    return b;
}
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// ---- layout kernels.  RCCL's reduce-scatter / all-gather move CONTIGUOUS per-rank blocks; the KAN kernels produce / consume
// row-major [rows, out].  blocks[p][r][c] <-> rows[r][p*w + c]; one thread per (row, float4 of a block) when w % 4 == 0.
template <bool TO_BLOCKS, int V>
__global__ __launch_bounds__(256) void rank_major_kernel(const float* __restrict__ src, float* __restrict__ dst, long n, int P, int w) {
    const int q = w / V;
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= n * q * P) return;
    // consecutive threads walk a ROW of the row-major side (coalesced there; the block side is coalesced per block)
    const long r = i / ((long)q * P);
    const int rem = (int)(i - r * ((long)q * P));
    const int p = rem / q, c = (rem - p * q) * V;
    const long row_major = r * ((long)P * w) + (long)p * w + c, blocked = ((long)p * n + r) * w + c;
    if constexpr (V == 4) {
        if (TO_BLOCKS) *reinterpret_cast<float4*>(dst + blocked) = *reinterpret_cast<const float4*>(src + row_major);
        else *reinterpret_cast<float4*>(dst + row_major) = *reinterpret_cast<const float4*>(src + blocked);
    } else {
        if (TO_BLOCKS) dst[blocked] = src[row_major];
        else dst[row_major] = src[blocked];
    }
}
template <bool TO_BLOCKS>
hipError_t rank_major(const float* src, float* dst, long n, int P, int w, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const bool v4 = (w % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const long items = n * (long)(v4 ? w / 4 : w) * P;
    const unsigned grid = (unsigned)((items + 255) / 256);
    if (v4) rank_major_kernel<TO_BLOCKS, 4><<<grid, 256, 0, st>>>(src, dst, n, P, w);
    else rank_major_kernel<TO_BLOCKS, 1><<<grid, 256, 0, st>>>(src, dst, n, P, w);
    return hipGetLastError();
}

struct FwdLayout { size_t part, blocks, ws, total, ws_bytes; };
struct BwdLayout { size_t gath, gfull, ws, total, ws_bytes; };

int layouts(long n, int in_local, int out, int G, int K, int mode, int row_chunks, FwdLayout* f, BwdLayout* b) {
    const size_t mat = align256((size_t)(n > 0 ? n : 0) * out * sizeof(float));
    size_t fws = 0, dws = 0;
    // the forward kernel's own scratch (non-zero only for few rows x many features): every chunk size that will occur
    for (const Bounds& c : chunk_bounds(n, row_chunks)) {
        size_t one = 0;
        R_LIB(kagnn_kan_fwd_workspace_bytes(c.r1 - c.r0 > 0 ? c.r1 - c.r0 : 1, in_local, out, G, K, mode, &one));
        fws = one > fws ? one : fws;
    }
    R_LIB(kagnn_kan_bwd_weight_workspace_bytes(n > 0 ? n : 1, in_local, out, G, K, mode, &dws));
    f->part = 0; f->blocks = mat; f->ws = 2 * mat; f->ws_bytes = fws; f->total = 2 * mat + align256(fws);
    b->gath = 0; b->gfull = mat; b->ws = 2 * mat; b->ws_bytes = dws; b->total = 2 * mat + align256(dws);
    return KAGNN_OK;
}

int check_common(const void* comm, int world, int rank, int out, long n, int row_chunks) {
    R_ARG(comm != nullptr, "comm is NULL");
    R_ARG(world >= 1 && rank >= 0 && rank < world, "world / rank");
    R_ARG(out > 0 && out % world == 0, "out_features must be divisible by the world size");
    R_ARG(n >= 0 && n < (1L << 31), "num_rows");
    R_ARG(row_chunks >= 1, "row_chunks >= 1");
    return KAGNN_OK;
}

}  // namespace

KAGNN_RCCL_API int kagnn_rccl_version(void) { return 100; }
KAGNN_RCCL_API const char* kagnn_rccl_last_error(void) { return g_err; }

KAGNN_RCCL_API int kagnn_rccl_unique_id(void* id128_host) {
    R_ARG(id128_host != nullptr, "id128_host is NULL");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    R_NCCL(ncclGetUniqueId(&id));
    memcpy(id128_host, &id, sizeof(id));
    return KAGNN_OK;
}

KAGNN_RCCL_API int kagnn_rccl_comm_init(const void* id128_host, int32_t world, int32_t rank, void** comm_out_host) {
    R_ARG(id128_host != nullptr && comm_out_host != nullptr, "NULL argument");
    R_ARG(world >= 1 && rank >= 0 && rank < world, "world / rank");
    ncclUniqueId id;
    memcpy(&id, id128_host, sizeof(id));
    ncclComm_t c = nullptr;
    R_NCCL(ncclCommInitRank(&c, world, id, rank));
    *comm_out_host = c;
    return KAGNN_OK;
}

KAGNN_RCCL_API int kagnn_rccl_comm_destroy(void* comm) {
    if (comm) R_NCCL(ncclCommDestroy(static_cast<ncclComm_t>(comm)));
    return KAGNN_OK;
}

KAGNN_RCCL_API int kagnn_sharded_kan_linear_workspace_bytes(int64_t num_rows, int32_t in_local, int32_t out_features,
                                                            int32_t grid_size, int32_t spline_order, int32_t mode, int32_t world,
                                                            int32_t row_chunks, size_t* fwd_bytes_host, size_t* bwd_bytes_host) {
    R_ARG(world >= 1 && out_features > 0 && out_features % world == 0, "out_features must be divisible by the world size");
    R_ARG(num_rows >= 0 && row_chunks >= 1, "num_rows / row_chunks");
    FwdLayout f; BwdLayout b;
    const int rc = layouts(num_rows, in_local, out_features, grid_size, spline_order, mode, row_chunks, &f, &b);
    if (rc != KAGNN_OK) return rc;
    if (fwd_bytes_host) *fwd_bytes_host = f.total;
    if (bwd_bytes_host) *bwd_bytes_host = b.total;
    return KAGNN_OK;
}

KAGNN_RCCL_API int kagnn_sharded_kan_linear_fwd(const float* x_slice, int64_t ldx, int64_t num_rows, const float* knots,
                                                int32_t in_local, int32_t out_features, int32_t grid_size, int32_t spline_order,
                                                int32_t mode, const void* pack_fwd, float* y_shard,
                                                void* comm, int32_t world, int32_t rank, int32_t row_chunks,
                                                void* workspace, size_t workspace_bytes, void* compute_stream, void* side_stream) {
    {
        const int rc = check_common(comm, world, rank, out_features, num_rows, row_chunks);
        if (rc != KAGNN_OK) return rc;
    }
    R_ARG(num_rows == 0 || (x_slice && y_shard && pack_fwd && knots), "NULL argument");
    R_ARG(side_stream != compute_stream, "side_stream must differ from compute_stream (the exchange of chunk c overlaps the kernel of chunk c+1)");
    FwdLayout L; BwdLayout unused;
    {
        const int rc = layouts(num_rows, in_local, out_features, grid_size, spline_order, mode, row_chunks, &L, &unused);
        if (rc != KAGNN_OK) return rc;
    }
    R_ARG(workspace_bytes >= L.total && (workspace || L.total == 0), "workspace too small (kagnn_sharded_kan_linear_workspace_bytes)");
    hipStream_t cs = static_cast<hipStream_t>(compute_stream), ss = static_cast<hipStream_t>(side_stream);
    ncclComm_t nc = static_cast<ncclComm_t>(comm);
    unsigned char* wsb = static_cast<unsigned char*>(workspace);
    float* part = reinterpret_cast<float*>(wsb + L.part);
    float* blocks = reinterpret_cast<float*>(wsb + L.blocks);
    const int w = out_features / world;
    for (const Bounds& c : chunk_bounds(num_rows, row_chunks)) {
        const long n = c.r1 - c.r0;
        float* pc = part + c.r0 * (long)out_features;          // this chunk's partial sums [n, out]
        float* bc = blocks + c.r0 * (long)out_features;        // ... and its rank-major blocks [P][n][w]
        if (n > 0) {
            R_LIB(kagnn_kan_linear_fwd(x_slice + c.r0 * ldx, ldx, n, knots, in_local, out_features, grid_size, spline_order, mode,
                                       pack_fwd, pc, out_features, wsb + L.ws, L.ws_bytes, compute_stream));
            R_HIP(rank_major<true>(pc, bc, n, world, w, cs));
        }
        hipEvent_t ev;
        R_HIP(g_events.get(&ev));
        R_HIP(hipEventRecord(ev, cs));
        R_HIP(hipStreamWaitEvent(ss, ev, 0));                  // the side stream waits for THIS chunk only
        R_NCCL(ncclReduceScatter(bc, y_shard + c.r0 * (long)w, (size_t)n * w, ncclFloat, ncclSum, nc, ss));
    }
    hipEvent_t done;
    R_HIP(g_events.get(&done));
    R_HIP(hipEventRecord(done, ss));
    R_HIP(hipStreamWaitEvent(cs, done, 0));
    return KAGNN_OK;
}

KAGNN_RCCL_API int kagnn_sharded_kan_linear_bwd(const float* x_slice, int64_t ldx, const float* gy_shard, int64_t num_rows,
                                                const float* knots, int32_t in_local, int32_t out_features, int32_t grid_size,
                                                int32_t spline_order, int32_t mode, const void* pack_dx,
                                                const float* spline_weight, const float* spline_scaler,
                                                float* gx_slice, int64_t ldgx,
                                                float* g_base_weight, float* g_spline_weight, float* g_spline_scaler,
                                                void* comm, int32_t world, int32_t rank, int32_t row_chunks,
                                                void* workspace, size_t workspace_bytes, void* compute_stream, void* side_stream) {
    {
        const int rc = check_common(comm, world, rank, out_features, num_rows, row_chunks);
        if (rc != KAGNN_OK) return rc;
    }
    R_ARG(g_spline_weight != nullptr && (num_rows == 0 || (x_slice && gy_shard && knots && spline_weight)), "NULL argument");
    R_ARG(gx_slice == nullptr || pack_dx != nullptr, "pack_dx is needed for gx_slice");
    R_ARG(side_stream != compute_stream, "side_stream must differ from compute_stream");
    FwdLayout unused; BwdLayout L;
    {
        const int rc = layouts(num_rows, in_local, out_features, grid_size, spline_order, mode, row_chunks, &unused, &L);
        if (rc != KAGNN_OK) return rc;
    }
    R_ARG(workspace_bytes >= L.total && (workspace || L.total == 0), "workspace too small (kagnn_sharded_kan_linear_workspace_bytes)");
    hipStream_t cs = static_cast<hipStream_t>(compute_stream), ss = static_cast<hipStream_t>(side_stream);
    ncclComm_t nc = static_cast<ncclComm_t>(comm);
    unsigned char* wsb = static_cast<unsigned char*>(workspace);
    float* gath = reinterpret_cast<float*>(wsb + L.gath);
    float* gfull = reinterpret_cast<float*>(wsb + L.gfull);
    const int w = out_features / world;
    const std::vector<Bounds> bounds = chunk_bounds(num_rows, row_chunks);
    // all gathers requested at once on the side stream (it waits for whatever produced gy_shard on the compute stream)
    hipEvent_t ready;
    R_HIP(g_events.get(&ready));
    R_HIP(hipEventRecord(ready, cs));
    R_HIP(hipStreamWaitEvent(ss, ready, 0));
    std::vector<hipEvent_t> landed(bounds.size());
    for (size_t i = 0; i < bounds.size(); ++i) {
        const Bounds& c = bounds[i];
        const long n = c.r1 - c.r0;
        R_NCCL(ncclAllGather(gy_shard + c.r0 * (long)w, gath + c.r0 * (long)out_features, (size_t)n * w, ncclFloat, nc, ss));
        R_HIP(g_events.get(&landed[i]));
        R_HIP(hipEventRecord(landed[i], ss));
    }
    for (size_t i = 0; i < bounds.size(); ++i) {
        const Bounds& c = bounds[i];
        const long n = c.r1 - c.r0;
        R_HIP(hipStreamWaitEvent(cs, landed[i], 0));           // THIS chunk's gather only
        if (n <= 0) continue;
        float* gc = gfull + c.r0 * (long)out_features;
        R_HIP(rank_major<false>(gath + c.r0 * (long)out_features, gc, n, world, w, cs));
        if (gx_slice)
            R_LIB(kagnn_kan_linear_bwd_input(x_slice + c.r0 * ldx, ldx, gc, out_features, n, knots, in_local, out_features,
                                             grid_size, spline_order, mode, pack_dx, gx_slice + c.r0 * ldgx, ldgx,
                                             KAGNN_DTYPE_F32, compute_stream));
    }
    // parameter gradients of this rank's slice: one pass over all rows (local: the parameters are sharded)
    if (num_rows == 0) {                                       // an empty shard: zero gradients
        const size_t oi = (size_t)out_features * in_local;
        if (g_base_weight) R_HIP(hipMemsetAsync(g_base_weight, 0, oi * sizeof(float), cs));
        if (g_spline_scaler) R_HIP(hipMemsetAsync(g_spline_scaler, 0, oi * sizeof(float), cs));
        R_HIP(hipMemsetAsync(g_spline_weight, 0, oi * (grid_size + spline_order) * sizeof(float), cs));
    } else
        R_LIB(kagnn_kan_linear_bwd_weight(x_slice, ldx, gfull, out_features, num_rows, knots, in_local, out_features, grid_size,
                                          spline_order, mode, spline_weight, spline_scaler, g_base_weight, g_spline_weight,
                                          g_spline_scaler, wsb + L.ws, L.ws_bytes, compute_stream));
    return KAGNN_OK;
}
