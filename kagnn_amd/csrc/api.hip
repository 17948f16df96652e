// extern "C" surface of libkagnn_hip.so (declared in include/kagnn_hip.h): argument validation
// and dispatch only -- the kernels live in the sibling .hip files.
#include "common.h"

namespace kagnn {
thread_local char g_err[512] = "";
thread_local bool g_half_products = false;      // KAGNN_PREC_HALF for the duration of an entry-point call (split_common.h)
thread_local DwDefer* g_dw_defer = nullptr;       // deferred weight-gradient slab reductions of a stack call (common.h)
thread_local MomDefer* g_mom_defer = nullptr;     // column moments whose finish is folded into the norm's apply kernel (common.h)
thread_local bool g_stack_prepacked = false;      // kagnn_kagin_model_fwd has packed the stack's layers together with the read-out's (one launch)

size_t aggregate_ws_bytes(long num_hub_seg, int F);
size_t aggregate_bf16_ws_bytes(long num_hub_seg, int F);
bool aggregate_bf16_ok(const void* x, long ldx, const void* out, long ldo, int out_bf16, int F, const float* bias);
int aggregate_sum_bf16(const void* x, long ldx, void* out, long ldo, int out_bf16, const int* rowptr, const int* col,
                       const float* ew, long N, int F, float self_scale, const float* in_scale, const float* out_scale,
                       const float* bias, int skip_self, const int* hub_seg, long num_hub_seg, int hub_threshold,
                       float* ws, size_t ws_bytes, hipStream_t st);
int rows_to_bf16(const float* x, long ldx, void* y, long ldy, long N, int F, hipStream_t st);
bool kan_sparse_fwd_agg_ok(const float* x, long ldx, long N, int in, int out, int G, int K);
size_t kan_sparse_fwd_agg_ws_bytes(long num_hub_seg, int in, int out);
int kan_sparse_fwd_agg(const float* x, long ldx, long N, const int* rowptr, const int* col, const int* hub_seg, long num_hub_seg,
                       int hub_threshold, float self_scale, const float* knots, int in, int out, int G, int K, const void* pack,
                       float* h0, long ldh, float* y, long ldy, void* ws, size_t ws_bytes, hipStream_t st);
int aggregate_sum(const AggArgs& a, const int* hub_seg, long num_hub_seg, float* ws, size_t ws_bytes, hipStream_t st);
bool aggregate_stats_ok(const AggArgs& a);
long aggregate_stats_rows(long N, int F, long num_hub_seg);
size_t bn_stats_fold_bytes(long B, int F);
int bn_sums_from_partials(float* ws, long B, int F, float* sums, hipStream_t st);
int bn_finish_partials(const float* partial, long B, int F, float* sums, hipStream_t st);
int bn_bwd_stats_given(const float*, long, int, const float*, const float*, const float*, float*, float*, float*, int, hipStream_t);
int gcn_deg_inv_sqrt(const int* rowptr, const int* col, long N, float* dis, hipStream_t st);
int gine_fwd(const float*, long, const float*, long, float*, long, const int*, const int*, const int*, long, int, float, hipStream_t);
int gine_bwd(const float*, long, const float*, long, const float*, long, float*, long, float*, long, const int*, const int*, const int*, long, int, float, hipStream_t, int gea_accumulate = 0);
int segment_pool(const float*, long, float*, long, const int*, long, int, int, hipStream_t);
int segment_bcast(const float*, long, float*, long, const int*, long, int, int, hipStream_t);
int embedding_fwd(const int64_t*, long, long, const float*, int, int, float*, long, int, hipStream_t);
int embedding_bwd(const int64_t*, long, long, const float*, long, int, int, float*, float*, size_t, hipStream_t);
size_t embedding_bwd_ws_bytes(long N, int V, int F);
int csr_workspace_bytes(long E, long N, size_t* bytes);
int csr_build(const int64_t*, const int64_t*, long, long, int*, int*, int*, int, int*, long, int64_t*, void*, size_t, hipStream_t);
bool csr_small_ok(long E, long N);
size_t csr_small_workspace_bytes(long E);
int csr_build_small(const int64_t*, const int64_t*, long, long, int*, int*, int*, int*, int*, int*, int*, void*, size_t, hipStream_t);

size_t kan_f32_pack_fwd_bytes(int in, int out, int C);
size_t kan_f32_pack_dx_bytes(int in, int out, int C);
int kan_f32_pack(const float*, const float*, const float*, int, int, int, float*, float*, hipStream_t);
int kan_f32_fwd(const float*, long, long, const float*, int, int, int, int, const float*, float*, long, bool, hipStream_t);
int kan_f32_dx(const float*, long, const float*, long, long, const float*, int, int, int, int, const float*, float*, long, bool, hipStream_t);
size_t kan_f32_dw_ws_bytes(long N, int in, int out, int C);
int kan_f32_dw(const float*, long, const float*, long, long, const float*, int, int, int, int, const float*, const float*, float*, float*, float*, float*, size_t, bool, hipStream_t);

size_t kan_split_pack_fwd_bytes(int in, int out, int C);
size_t kan_split_pack_dx_bytes(int in, int out, int C, int K);
int kan_split_pack_fwd_noscale(const float*, const float*, const float*, int, int, int, void*, hipStream_t);
int kan_split_pack_dx_noscale(const float*, const float*, const float*, int, int, int, int, void*, hipStream_t);
int kan_split_fwd(const float*, long, long, const float*, int, int, int, int, const void*, float*, long, void*, size_t, hipStream_t);
size_t kan_split_fwd_ws_bytes(long N, int in, int out, int C);
int kan_split_dx(const float*, long, const float*, long, long, const float*, int, int, int, int, const void*, float*, long, hipStream_t, int gx16, const float* x_affine);
size_t kan_split_dw_ws_bytes(long N, int in, int out, int C, int K);
int kan_split_dw(const float*, long, const float*, long, long, const float*, int, int, int, int, const float*, const float*, float*, float*, float*, float*, size_t, hipStream_t, const float* x_affine);
bool kan_split_fwd_ok(int in, int out, int G, int K);
bool kan_sparse_fwd_ok(int in, int out, int G, int K);
bool kan_fused_pack_ok(int in, int out, int C);
int kan_fused_pack(const float*, const float*, const float*, int, int, int, void*, void*, hipStream_t);
int kan_fused_pack_batch(int, const float* const*, const float* const*, const float* const*, const int*, const int*, int, void* const*, void* const*, hipStream_t);
size_t kan_sparse_pack_fwd_bytes(int in, int out, int C);
int kan_sparse_pack_fwd(const float*, const float*, const float*, int, int, int, void*, hipStream_t);
size_t kan_sparse_fwd_ws_bytes(long N, int in, int out, int C);
int kan_sparse_fwd(const float*, long, long, const float*, int, int, int, int, const void*, float*, long, void*, size_t, float*, float*, hipStream_t);
bool kan_sparse_fwd_parts_ok(const int*, int, int, int, int, int);
int kan_sparse_fwd_parts(const float* const*, const int*, const long*, int, long, const float*, int, int, int, int, const void*, float*, long, void*, size_t, hipStream_t, const float* const*);
bool kan_sparse_fwd_moments_ok(long N, int in, int out, int G, int K);
size_t kan_sparse_fwd_moments_ws_bytes(long N, int out);
int col_moments(const float*, long, long, int, float*, float*, void*, size_t, hipStream_t);
bool kan_split_dx_ok(int in, int out, int G, int K);
int kan_split_dx_stats_blocks(long N);
bool kan_split_dx_stats_ok(long N, int in, int out, int G, int K);
int kan_split_dx_stats(const float*, long, const float*, long, long, const float*, int, int, int, int, const void*, float*, long, hipStream_t,
                       const float*, const float*, const float*, float*);
bool kan_split_dw_ok(int in, int out, int G, int K);

int fastkan_fwd(const float*, long, long, int, int, int, const float*, float, const float*, const float*, float, const float*, const float*, const float*, float*, long, float*, void*, size_t, int, hipStream_t, bool stats_given = false);
int fastkan_row_moments(const float*, long, long, int, float*, hipStream_t);
int fastkan_merge_moments(const float*, int, long, int, float, float*, hipStream_t);
size_t fastkan_fwd_ws_bytes(long N, int in, int out, int ng, int mode);
size_t fastkan_bwd_ws_bytes(long N, int in, int out, int ng, int mode);
int fastkan_bwd(const float*, long, const float*, long, long, int, int, int, const float*, float, const float*, const float*, float, const float*, const float*, const float*, float*, long, float*, float*, float*, float*, float*, void*, size_t, int, hipStream_t, int phase = 0, float* row_sums = nullptr, int in_total = 0);
int gat_logits(const float*, long, long, int, int, const float*, const float*, float*, float*, hipStream_t);
int gat_fwd(const float*, long, const float*, const float*, const int*, const int*, long, int, int, const float*, float*, long, float*, float*, const int*, long, int, hipStream_t);
int gat_bwd(const float*, long, const float*, long, const float*, long, const float*, const float*, const float*, const float*, const float*, const int*, const int*, const int*, const int*, const int*, const int*, const float*, const float*, long, int, int, float*, float*, float*, float*, float*, long, const int*, long, int, hipStream_t);
int kan_bsplines(const float*, long, long, const float*, int, int, int, float*, hipStream_t);
size_t kan_grid_refit_ws_bytes(long N, int in);
int kan_grid_refit(const float*, long, long, const float*, const float*, int, int, int, int, const float*, const float*, float*, void*, size_t, hipStream_t);
size_t xent_ws_bytes(long N);
int xent_fwd(const float*, long, long, int, const long*, const unsigned char*, int, float*, float*, float*, void*, size_t, hipStream_t);
int xent_bwd(const float*, long, long, int, const long*, const unsigned char*, int, const float*, const float*, const float*, float*, long, hipStream_t);
int l1_loss_fwd(const float* p, const float* t, long n, float* loss, hipStream_t st);
int l1_loss_bwd(const float* p, const float* t, long n, const float* g_loss, float* g_p, hipStream_t st);
int adam_step(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
              const long* numel, float lr, float beta1, float beta2, float eps, float weight_decay, long step, hipStream_t st);
size_t gat_att_grad_ws_bytes(long N, int H, int C);
int gat_att_grad(const float*, long, const float*, const float*, long, int, int, float*, float*, void*, size_t, hipStream_t);
size_t bn_ws_bytes(long N, int F);
int bn_fwd(const float*, long, long, int, const float*, const float*, float*, float*, float, float, int, const float*, const float*, float, unsigned long long, float*, long, float*, float*, void*, size_t, hipStream_t);
int bn_bwd(const float*, long, const float*, long, long, int, const float*, const float*, const float*, int, float, unsigned long long, float*, long, float*, float*, void*, size_t, hipStream_t);
int bn_bwd_stats(const float*, long, const float*, long, long, int, const float*, const float*, const float*, float*, float*, float*, int, void*, size_t, hipStream_t);
int bn_fwd_partial_moments(const float*, long, long, int, const float*, const float*, float*, float*, float, float, const float*, int, float*, long, float*, float*, hipStream_t);
int bn_stats_affine(const float*, const float*, long, int, const float*, const float*, float*, float*, float, float, float*, float*, float*, hipStream_t);
bool kan_split_dx_bn_ok(long, int, int, int, int, const BnBack&, const void*);
int kan_split_dx_bn(const float*, long, const float*, long, long, const float*, int, int, int, int, const void*, float*, long, const BnBack&, hipStream_t);
int p2p_reduce_scatter(const float* const* parts, int P, int rank, long N, int out, long ld, float* y, long ldy, hipStream_t st);
int p2p_all_gather(const float* const* shards, int P, long N, int w, long lds, float* g, long ldg, hipStream_t st);
}  // namespace kagnn

using namespace kagnn;

// ---------------------------------------------------------------- stage timer (measurement aid; off by default)
// While enabled, every per-operation entry point -- ALSO when it runs inside kagnn_gin_kan_layer_fwd / _bwd* -- is bracketed by
// HIP events recorded on the stream it launches on, so that bench.py can time the dominant kernel live inside the timed region
// of the product's default path (one library call per convolution each way) instead of composing the layer from per-op calls.
#include <mutex>
#include <string>
#include <vector>
namespace {
struct StageRecord { const char* name; hipEvent_t a, b; int dev; };
constexpr int kStageMaxDevices = 64;
struct StageTimer {
    std::mutex mu;
    bool on = false;
    std::string only;                       // empty: every stage
    std::vector<StageRecord> rec;
    // events of earlier sessions, reused -- PER DEVICE: an event belongs to the device that was current when it was created, and
    // recording it on another device's stream fails (the ignored error used to show up as 0 ms timings: ADVICE r04)
    std::vector<hipEvent_t> pool[kStageMaxDevices];
} g_stage;
thread_local int g_stage_depth = 0;         // nested entry points (fwd_moments -> fwd): only the outermost is a stage
constexpr size_t kStageMaxRecords = 1u << 16;

struct StageScope {
    hipStream_t st;
    const char* name;
    hipEvent_t b = nullptr;
    bool outer;
    StageScope(const char* nm, void* stream) : st(as_stream(stream)), name(nm), outer(g_stage_depth++ == 0) {
        if (!outer || !g_stage.on) return;              // (unlocked read of a flag that only bench.py toggles, between steps)
        std::lock_guard<std::mutex> lk(g_stage.mu);
        if (!g_stage.on || (!g_stage.only.empty() && g_stage.only != nm) || g_stage.rec.size() >= kStageMaxRecords) return;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kStageMaxDevices) return;
        std::vector<hipEvent_t>& pool = g_stage.pool[dev];
        hipEvent_t ev[2] = {nullptr, nullptr};
        for (int i = 0; i < 2; ++i) {
            if (!pool.empty()) { ev[i] = pool.back(); pool.pop_back(); }
            else if (hipEventCreate(&ev[i]) != hipSuccess) {
                (void)hipGetLastError();
                if (i == 1) pool.push_back(ev[0]);      // a partial failure must not leak the first event
                return;
            }
        }
        (void)hipEventRecord(ev[0], st);
        b = ev[1];
        g_stage.rec.push_back(StageRecord{nm, ev[0], ev[1], dev});
    }
    ~StageScope() {
        --g_stage_depth;
        if (b) (void)hipEventRecord(b, st);
    }
};
}  // namespace
// KAGNN_PREC_HALF is KAGNN_PREC_SPLIT with ONE product per fp32 product: every routing decision below is the split mode's, the
// launchers of the three KAN kernels pick their HALF instantiation while the flag is up (shapes without one run the
// three-product kernels: more accurate, never less).  Entry points call each other with the rewritten mode, so a nested scope
// sees KAGNN_PREC_SPLIT and leaves the flag alone.
struct ModeScope {
    bool prev;
    explicit ModeScope(int32_t& mode) : prev(kagnn::g_half_products) {
        if (mode == KAGNN_PREC_HALF) { kagnn::g_half_products = true; mode = KAGNN_PREC_SPLIT; }
    }
    ~ModeScope() { kagnn::g_half_products = prev; }
};
#define KAGNN_STAGE(stream) StageScope stage_scope_(__func__, stream)
#define KAGNN_STAGE_AS(name, stream) StageScope stage_scope_(name, stream)

static int check_kan_dims(const char* fn, int in, int out, int G, int K, int mode) {
    if (in < 1 || out < 1) return fail(KAGNN_ERR_ARG, "%s: in_features/out_features must be >= 1", fn);
    if (K < 1 || K > kMaxOrder) return fail(KAGNN_ERR_UNSUPPORTED, "%s: spline_order must be 1..4", fn);
    if (G < 1 || G + 2 * K + 1 > kMaxKnots) return fail(KAGNN_ERR_UNSUPPORTED, "%s: grid_size out of range", fn);
    if (mode != KAGNN_PREC_FP32 && mode != KAGNN_PREC_SPLIT && mode != KAGNN_PREC_FP32_GRID) return fail(KAGNN_ERR_ARG, "%s: unknown precision mode", fn);
    return KAGNN_OK;
}
// the split path covers the hot shapes; everything else runs the exact-fp32 kernels (still HIP)
// the split kernels address activations through buffer descriptors with 32-bit byte offsets, re-opened at every
// workgroup tile (<= 256 rows forward / input gradient, <= 2^17 rows weight gradient): any N, rows up to 7680 floats
static bool fits32(long N, long ld) { (void)N; return ld <= 7680; }
static bool use_split_fwd(int in, int out, int G, int K, int mode) { return mode == KAGNN_PREC_SPLIT && kan_split_fwd_ok(in, out, G, K); }
static bool use_sparse_fwd(int in, int out, int G, int K, int mode) { return use_split_fwd(in, out, G, K, mode) && kan_sparse_fwd_ok(in, out, G, K); }
static bool use_split_dx(int in, int out, int G, int K, int mode) { return mode == KAGNN_PREC_SPLIT && kan_split_dx_ok(in, out, G, K); }
static bool use_split_dw(int in, int out, int G, int K, int mode) { return mode == KAGNN_PREC_SPLIT && kan_split_dw_ok(in, out, G, K); }

#pragma GCC visibility push(default)
extern "C" {

int kagnn_version(void) { return 260; }
const char* kagnn_last_error(void) { return g_err; }

int kagnn_stage_timer_enable(const char* only) {
    std::lock_guard<std::mutex> lk(g_stage.mu);
    g_stage.only = only ? only : "";
    g_stage.on = true;
    return KAGNN_OK;
}

int kagnn_stage_timer_disable(void) {
    std::lock_guard<std::mutex> lk(g_stage.mu);
    g_stage.on = false;
    return KAGNN_OK;
}

// Aggregates the records taken so far by stage name (waits for their events), hands the events back to the pool and clears
// the records.  names: caller's array of `capacity` char[64] slots; returns the number of distinct stages through *n_stages.
int kagnn_stage_timer_collect(char* names, int64_t* launches, double* total_ms, int32_t capacity, int32_t* n_stages) {
    KAGNN_CHECK_ARG(names && launches && total_ms && n_stages && capacity >= 1, "null output");
    std::lock_guard<std::mutex> lk(g_stage.mu);
    int n = 0;
    for (const StageRecord& r : g_stage.rec) {
        float ms = 0.0f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) {
            (void)hipGetLastError();
            ms = 0.0f;
        }
        g_stage.pool[r.dev].push_back(r.a);             // (before any `continue`: a record beyond `capacity` used to leak its events)
        g_stage.pool[r.dev].push_back(r.b);
        int k = 0;
        while (k < n && strncmp(names + 64 * k, r.name, 63) != 0) ++k;
        if (k == n) {
            if (n == capacity) continue;
            strncpy(names + 64 * n, r.name, 63);
            names[64 * n + 63] = 0;
            launches[n] = 0; total_ms[n] = 0.0;
            ++n;
        }
        launches[k] += 1;
        total_ms[k] += ms;
    }
    g_stage.rec.clear();
    *n_stages = n;
    return KAGNN_OK;
}

int kagnn_csr_workspace_bytes(int64_t E, int64_t N, size_t* bytes) {
    KAGNN_CHECK_ARG(bytes != nullptr && E >= 0 && N >= 0, "null output or negative size");
    KAGNN_CHECK_ARG(E < 2147483647LL && N < 2147483647LL, "N and E must fit int32");
    return csr_workspace_bytes(E, N, bytes);
}

int kagnn_csr_build(const int64_t* key, const int64_t* val, int64_t E, int64_t N, int32_t* rowptr,
                    int32_t* col, int32_t* perm, int32_t hub_threshold, int32_t* hub_seg,
                    int64_t hub_seg_capacity, int64_t* num_hub_seg_host, void* ws, size_t ws_bytes,
                    void* stream) {
    KAGNN_CHECK_ARG(E >= 0 && N >= 0 && E < 2147483647LL && N < 2147483647LL, "N and E must fit int32");
    KAGNN_CHECK_ARG(rowptr != nullptr, "rowptr is null");
    KAGNN_CHECK_ARG(E == 0 || (key && val && col && perm && ws), "null array");
    return csr_build(key, val, E, N, rowptr, col, perm, hub_threshold, hub_seg, hub_seg_capacity,
                     num_hub_seg_host, ws, ws_bytes, as_stream(stream));
}

int kagnn_csr_small_ok(int64_t E, int64_t N) { return csr_small_ok(E, N) ? 1 : 0; }

int kagnn_csr_small_workspace_bytes(int64_t E, size_t* bytes) {
    KAGNN_CHECK_ARG(bytes != nullptr && E >= 0, "null output or negative size");
    *bytes = csr_small_workspace_bytes(E);
    return KAGNN_OK;
}

int kagnn_csr_build_small(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int32_t* rowptr, int32_t* col, int32_t* perm,
                          int32_t* rowptr_t, int32_t* col_t, int32_t* perm_t, int32_t* flags, void* ws, size_t ws_bytes, void* stream) {
    KAGNN_CHECK_ARG(src && dst && rowptr && col && perm && rowptr_t && col_t && perm_t && flags && ws, "null array");
    return csr_build_small(src, dst, E, N, rowptr, col, perm, rowptr_t, col_t, perm_t, flags, ws, ws_bytes, as_stream(stream));
}

int kagnn_gcn_deg_inv_sqrt(const int32_t* rowptr, const int32_t* col, int64_t N, float* dis, void* stream) {
    KAGNN_CHECK_ARG(N >= 0 && rowptr && dis, "null array");
    return gcn_deg_inv_sqrt(rowptr, col, N, dis, as_stream(stream));
}

int kagnn_aggregate_sum_add(const float* x, int64_t ldx, float* out, int64_t ldo, const int32_t* rowptr,
                            const int32_t* col, const float* edge_weight, int64_t N, int32_t F,
                            float self_scale, const float* in_scale, const float* out_scale,
                            const float* bias, int32_t skip_self_loops, const int32_t* hub_seg,
                            int64_t num_hub_seg, int32_t hub_threshold, const float* addend, int64_t ld_addend,
                            void* workspace, size_t workspace_bytes, void* stream) {
    KAGNN_STAGE_AS("kagnn_aggregate_sum", stream);
    KAGNN_CHECK_ARG(N >= 0 && F >= 1, "bad shape");
    KAGNN_CHECK_ARG(N == 0 || (x && out && rowptr), "null array");
    KAGNN_CHECK_ARG(ldx >= F && ldo >= F && (!addend || ld_addend >= F), "leading dimension smaller than num_feat");
    KAGNN_CHECK_ARG(x != out, "in-place aggregation is not supported");
    AggArgs a{x, ldx, out, ldo, rowptr, col, edge_weight, N, F, self_scale, in_scale, out_scale, bias,
              skip_self_loops, hub_threshold > 0 ? hub_threshold : 0x7fffffff, addend, ld_addend};
    return aggregate_sum(a, hub_seg, num_hub_seg, static_cast<float*>(workspace), workspace_bytes, as_stream(stream));
}

int kagnn_aggregate_sum(const float* x, int64_t ldx, float* out, int64_t ldo, const int32_t* rowptr,
                        const int32_t* col, const float* edge_weight, int64_t N, int32_t F,
                        float self_scale, const float* in_scale, const float* out_scale,
                        const float* bias, int32_t skip_self_loops, const int32_t* hub_seg,
                        int64_t num_hub_seg, int32_t hub_threshold, void* workspace, size_t workspace_bytes,
                        void* stream) {
    return kagnn_aggregate_sum_add(x, ldx, out, ldo, rowptr, col, edge_weight, N, F, self_scale, in_scale, out_scale, bias,
                                   skip_self_loops, hub_seg, num_hub_seg, hub_threshold, nullptr, 0, workspace, workspace_bytes, stream);
}

// GIN-form aggregation of a matrix that exists only as  col_scale[c] * x[.][c] + col_shift[c]  -- the output of a training-mode
// BatchNorm1d whose normalising pass is folded into the aggregation that gathers it (reference node_classification_clean/
// models.py:198-200, `x = self.bns[i](self.convs[i](x, edge_index))` feeding the next GINConv):
//   out_i = col_scale * (self_scale * x_i + sum_{j->i} x_j) + (self_scale + deg_i) * col_shift   [+ addend_i]
// Unit edge weights (no edge_weight / in_scale / out_scale / bias / skip_self_loops).  col_scale / col_shift: F floats each.
int kagnn_aggregate_sum_affine(const float* x, int64_t ldx, float* out, int64_t ldo, const int32_t* rowptr, const int32_t* col,
                               int64_t N, int32_t F, float self_scale, const float* col_scale, const float* col_shift,
                               const int32_t* hub_seg, int64_t num_hub_seg, int32_t hub_threshold, const float* addend,
                               int64_t ld_addend, void* workspace, size_t workspace_bytes, void* stream) {
    KAGNN_STAGE_AS("kagnn_aggregate_sum", stream);
    KAGNN_CHECK_ARG(N >= 0 && F >= 1, "bad shape");
    KAGNN_CHECK_ARG(N == 0 || (x && out && rowptr), "null array");
    KAGNN_CHECK_ARG(ldx >= F && ldo >= F && (!addend || ld_addend >= F), "leading dimension smaller than num_feat");
    KAGNN_CHECK_ARG(x != out, "in-place aggregation is not supported");
    KAGNN_CHECK_ARG((col_scale == nullptr) == (col_shift == nullptr), "col_scale and col_shift must both be given or both be null");
    AggArgs a{x, ldx, out, ldo, rowptr, col, nullptr, N, F, self_scale, nullptr, nullptr, nullptr,
              0, hub_threshold > 0 ? hub_threshold : 0x7fffffff, addend, ld_addend};
    a.col_scale = col_scale; a.col_shift = col_shift;
    return aggregate_sum(a, hub_seg, num_hub_seg, static_cast<float*>(workspace), workspace_bytes, as_stream(stream));
}

int kagnn_aggregate_sum_bf16(const void* x, int64_t ldx, void* out, int64_t ldo, int32_t out_dtype, const int32_t* rowptr,
                             const int32_t* col, const float* edge_weight, int64_t N, int32_t F, float self_scale,
                             const float* in_scale, const float* out_scale, const float* bias, int32_t skip_self_loops,
                             const int32_t* hub_seg, int64_t num_hub_seg, int32_t hub_threshold, void* workspace,
                             size_t workspace_bytes, void* stream) {
    KAGNN_STAGE(stream);
    KAGNN_CHECK_ARG(N >= 0 && F >= 1, "bad shape");
    KAGNN_CHECK_ARG(N == 0 || (x && out && rowptr), "null array");
    KAGNN_CHECK_ARG(ldx >= F && ldo >= F, "leading dimension smaller than num_feat");
    KAGNN_CHECK_ARG(out_dtype == KAGNN_DTYPE_F32 || out_dtype == KAGNN_DTYPE_BF16, "out_dtype must be KAGNN_DTYPE_F32 or KAGNN_DTYPE_BF16");
    KAGNN_CHECK_ARG(x != out, "in-place aggregation is not supported");
    if (!aggregate_bf16_ok(x, ldx, out, ldo, out_dtype == KAGNN_DTYPE_BF16, F, bias))
        return fail(KAGNN_ERR_UNSUPPORTED, "%s: bf16 rows need num_feat % 8 == 0 (<= 512) and 16-byte aligned rows", __func__);
    return aggregate_sum_bf16(x, ldx, out, ldo, out_dtype == KAGNN_DTYPE_BF16, rowptr, col, edge_weight, N, F, self_scale, in_scale,
                              out_scale, bias, skip_self_loops, hub_seg, num_hub_seg, hub_threshold,
                              static_cast<float*>(workspace), workspace_bytes, as_stream(stream));
}

int kagnn_rows_to_bf16(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t N, int32_t F, void* stream) {
    KAGNN_CHECK_ARG(N >= 0 && F >= 1 && ldx >= F && ldy >= F, "bad shape");
    KAGNN_CHECK_ARG(N == 0 || (x && y), "null array");
    return rows_to_bf16(x, ldx, y, ldy, N, F, as_stream(stream));
}

int kagnn_aggregate_workspace_bytes(int64_t num_hub_seg, int32_t F, size_t* bytes_host) {
    KAGNN_CHECK_ARG(num_hub_seg >= 0 && F >= 1 && bytes_host, "bad argument");
    *bytes_host = aggregate_bf16_ws_bytes(num_hub_seg, F);       // (F rounded up to 8: covers the fp32 form's round-up to 4)
    return KAGNN_OK;
}

int kagnn_aggregate_gine(const float* x, int64_t ldx, const float* ea, int64_t lde, float* out,
                         int64_t ldo, const int32_t* rowptr, const int32_t* col, const int32_t* perm,
                         int64_t N, int32_t F, float self_scale, void* stream) {
    KAGNN_CHECK_ARG(N >= 0 && F >= 1, "bad shape");
    KAGNN_CHECK_ARG(N == 0 || (x && out && rowptr), "null array");
    return gine_fwd(x, ldx, ea, lde, out, ldo, rowptr, col, perm, N, F, self_scale, as_stream(stream));
}

int kagnn_aggregate_gine_bwd(const float* x, int64_t ldx, const float* ea, int64_t lde,
                             const float* gout, int64_t ldg, float* gx, int64_t ldgx, float* gea,
                             int64_t ldge, const int32_t* rowptr_t, const int32_t* col_t,
                             const int32_t* perm_t, int64_t N, int32_t F, float self_scale, void* stream) {
    KAGNN_CHECK_ARG(N >= 0 && F >= 1, "bad shape");
    KAGNN_CHECK_ARG(N == 0 || (x && gout && gx && rowptr_t), "null array");
    return gine_bwd(x, ldx, ea, lde, gout, ldg, gx, ldgx, gea, ldge, rowptr_t, col_t, perm_t, N, F,
                    self_scale, as_stream(stream));
}

int kagnn_segment_pool(const float* x, int64_t ldx, float* out, int64_t ldo, const int32_t* seg,
                       int64_t B, int32_t F, int32_t mean, void* stream) {
    KAGNN_CHECK_ARG(B >= 0 && F >= 1 && (B == 0 || (x && out && seg)), "bad argument");
    return segment_pool(x, ldx, out, ldo, seg, B, F, mean, as_stream(stream));
}

int kagnn_segment_broadcast(const float* g, int64_t ldg, float* gx, int64_t ldgx, const int32_t* seg,
                            int64_t B, int32_t F, int32_t mean, void* stream) {
    KAGNN_CHECK_ARG(B >= 0 && F >= 1 && (B == 0 || (g && gx && seg)), "bad argument");
    return segment_bcast(g, ldg, gx, ldgx, seg, B, F, mean, as_stream(stream));
}

// ---------------------------------------------------------------- efficient-KAN
int kagnn_embedding_fwd(const int64_t* index, int64_t index_stride, int64_t N, const float* table, int32_t V, int32_t F, float* out,
                        int64_t ldo, int32_t accumulate, void* stream) {
    KAGNN_CHECK_ARG(N >= 0 && V >= 1 && F >= 1 && index_stride >= 1 && ldo >= F, "bad shape");
    KAGNN_CHECK_ARG(N == 0 || (index && table && out), "null array");
    return embedding_fwd(index, index_stride, N, table, V, F, out, ldo, accumulate, as_stream(stream));
}

int kagnn_embedding_bwd_workspace_bytes(int64_t N, int32_t V, int32_t F, size_t* bytes) {
    KAGNN_CHECK_ARG(N >= 0 && V >= 1 && F >= 1 && bytes, "bad argument");
    *bytes = embedding_bwd_ws_bytes(N, V, F);
    return KAGNN_OK;
}

int kagnn_embedding_bwd(const int64_t* index, int64_t index_stride, int64_t N, const float* g, int64_t ldg, int32_t V, int32_t F,
                        float* g_table, void* workspace, size_t workspace_bytes, void* stream) {
    KAGNN_CHECK_ARG(N >= 0 && V >= 1 && F >= 1 && index_stride >= 1 && ldg >= F && g_table && workspace, "bad shape");
    KAGNN_CHECK_ARG(N == 0 || (index && g), "null array");
    return embedding_bwd(index, index_stride, N, g, ldg, V, F, g_table, static_cast<float*>(workspace), workspace_bytes, as_stream(stream));
}

int kagnn_kan_pack_bytes(int32_t in, int32_t out, int32_t G, int32_t K, int32_t mode,
                         size_t* fwd_bytes, size_t* dx_bytes) {
    ModeScope mode_scope_(mode);
    int rc = check_kan_dims(__func__, in, out, G, K, mode);
    if (rc) return rc;
    KAGNN_CHECK_ARG(fwd_bytes && dx_bytes, "null output");
    *fwd_bytes = use_sparse_fwd(in, out, G, K, mode) ? kan_sparse_pack_fwd_bytes(in, out, G + K)
               : use_split_fwd(in, out, G, K, mode) ? kan_split_pack_fwd_bytes(in, out, G + K) : kan_f32_pack_fwd_bytes(in, out, G + K);
    *dx_bytes = use_split_dx(in, out, G, K, mode) ? kan_split_pack_dx_bytes(in, out, G + K, K) : kan_f32_pack_dx_bytes(in, out, G + K);
    return KAGNN_OK;
}

int kagnn_kan_pack(const float* bw, const float* sw, const float* sc, int32_t in, int32_t out,
                   int32_t G, int32_t K, int32_t mode, void* pack_fwd, void* pack_dx, void* stream) {
    ModeScope mode_scope_(mode);
    KAGNN_STAGE(stream);
    int rc = check_kan_dims(__func__, in, out, G, K, mode);
    if (rc) return rc;
    KAGNN_CHECK_ARG(sw && pack_fwd && pack_dx, "null array");          // base_weight NULL = no SiLU branch
    const bool sf = use_split_fwd(in, out, G, K, mode), sd = use_split_dx(in, out, G, K, mode);
    // each workgroup derives the power-of-two weight scale itself; the hot case takes ONE launch for both layouts
    if (sd && use_sparse_fwd(in, out, G, K, mode) && kan_fused_pack_ok(in, out, G + K))
        return kan_fused_pack(bw, sw, sc, in, out, G + K, pack_fwd, pack_dx, as_stream(stream));
    if (sf) {
        rc = use_sparse_fwd(in, out, G, K, mode) ? kan_sparse_pack_fwd(bw, sw, sc, in, out, G + K, pack_fwd, as_stream(stream))
                                                 : kan_split_pack_fwd_noscale(bw, sw, sc, in, out, G + K, pack_fwd, as_stream(stream));
        if (rc) return rc;
    }
    if (sd) { rc = kan_split_pack_dx_noscale(bw, sw, sc, in, out, G + K, K, pack_dx, as_stream(stream)); if (rc) return rc; }
    if (!sf || !sd)
        return kan_f32_pack(bw, sw, sc, in, out, G + K, sf ? nullptr : (float*)pack_fwd, sd ? nullptr : (float*)pack_dx, as_stream(stream));
    return KAGNN_OK;
}

int kagnn_kan_pack_batch(int32_t n_layers, const float* const* bw, const float* const* sw, const float* const* sc,
                         const int32_t* in, const int32_t* out, int32_t G, int32_t K, int32_t mode,
                         void* const* pack_fwd, void* const* pack_dx, void* stream) {
    ModeScope mode_scope_(mode);
    KAGNN_STAGE(stream);
    KAGNN_CHECK_ARG(n_layers >= 1 && bw && sw && in && out && pack_fwd && pack_dx, "null array");
    for (int l = 0; l < n_layers; ++l) {
        int rc = check_kan_dims(__func__, in[l], out[l], G, K, mode);
        if (rc) return rc;
        KAGNN_CHECK_ARG(sw[l] && pack_fwd[l] && pack_dx[l], "null array");
        if (!(use_split_dx(in[l], out[l], G, K, mode) && use_sparse_fwd(in[l], out[l], G, K, mode)))
            return fail(KAGNN_ERR_UNSUPPORTED, "%s: only layers on the sparse-forward / split path batch their packs", __func__);
    }
    return kan_fused_pack_batch(n_layers, bw, sw, sc, in, out, G + K, pack_fwd, pack_dx, as_stream(stream));
}

int kagnn_kan_fwd_workspace_bytes(int64_t N, int32_t in, int32_t out, int32_t G, int32_t K, int32_t mode,
                                  size_t* bytes) {
    ModeScope mode_scope_(mode);
    int rc = check_kan_dims(__func__, in, out, G, K, mode);
    if (rc) return rc;
    KAGNN_CHECK_ARG(bytes && N >= 0, "bad argument");
    *bytes = use_sparse_fwd(in, out, G, K, mode) ? kan_sparse_fwd_ws_bytes(N, in, out, G + K)
           : use_split_fwd(in, out, G, K, mode) ? kan_split_fwd_ws_bytes(N, in, out, G + K) : 0;
    return KAGNN_OK;
}

int kagnn_kan_linear_fwd(const float* x, int64_t ldx, int64_t N, const float* knots, int32_t in,
                         int32_t out, int32_t G, int32_t K, int32_t mode, const void* pack_fwd,
                         float* y, int64_t ldy, void* ws, size_t ws_bytes, void* stream) {
    ModeScope mode_scope_(mode);
    KAGNN_STAGE(stream);
    int rc = check_kan_dims(__func__, in, out, G, K, mode);
    if (rc) return rc;
    KAGNN_CHECK_ARG(N >= 0 && ldx >= in && ldy >= out, "bad shape");
    if (N == 0) return KAGNN_OK;
    KAGNN_CHECK_ARG(x && knots && pack_fwd && y, "null array");
    if (use_split_fwd(in, out, G, K, mode)) {
        if (!(fits32(N, ldx) && fits32(N, ldy))) return fail(KAGNN_ERR_UNSUPPORTED, "%s: leading dimension > 7680 floats; call with KAGNN_PREC_FP32", __func__);
        if (use_sparse_fwd(in, out, G, K, mode))
            return kan_sparse_fwd(x, ldx, N, knots, in, out, G, K, pack_fwd, y, ldy, ws, ws_bytes, nullptr, nullptr, as_stream(stream));
        return kan_split_fwd(x, ldx, N, knots, in, out, G, K, pack_fwd, y, ldy, ws, ws_bytes, as_stream(stream));
    }
    return kan_f32_fwd(x, ldx, N, knots, in, out, G, K, (const float*)pack_fwd, y, ldy, mode == KAGNN_PREC_FP32_GRID, as_stream(stream));
}

// forward on an input given as column blocks [x_0 | x_1 | ...] that live in different buffers: the
// skip-concat read-out of the node models (reference node_classification_clean/models.py:202 `torch.cat(l, dim=1)` feeding
// `lay_out`) without building the concatenation, one launch, one write of y.  pack_fwd is the pack of the WHOLE layer.
int kagnn_kan_fwd_parts_ok(const int32_t* part_widths, int32_t num_parts, int32_t in, int32_t out, int32_t G, int32_t K, int32_t mode) {
    ModeScope mode_scope_(mode);
    if (!part_widths || check_kan_dims(__func__, in, out, G, K, mode)) return 0;
    return use_sparse_fwd(in, out, G, K, mode) && kan_sparse_fwd_parts_ok(part_widths, num_parts, in, out, G, K) ? 1 : 0;
}

int kagnn_kan_linear_fwd_parts(const float* const* x_parts, const int32_t* part_widths, const int64_t* part_ld, int32_t num_parts,
                               int64_t N, const float* knots, int32_t in, int32_t out, int32_t G, int32_t K, int32_t mode,
                               const void* pack_fwd, float* y, int64_t ldy, void* ws, size_t ws_bytes, void* stream) {
    ModeScope mode_scope_(mode);
    return kagnn_kan_linear_fwd_parts_affine(x_parts, part_widths, part_ld, nullptr, num_parts, N, knots, in, out, G, K, mode, pack_fwd, y, ldy,
                                             ws, ws_bytes, stream);
}

// part_affine (NULL, or per block NULL / 2 * width floats: the block's column scales, then its column shifts): the block is
// read as  scale * x + shift  -- a BatchNorm1d output that was never written (the skip read-out of the node models over
// normalised layer outputs, reference node_classification_clean/models.py:198-203).
int kagnn_kan_linear_fwd_parts_affine(const float* const* x_parts, const int32_t* part_widths, const int64_t* part_ld,
                                      const float* const* part_affine, int32_t num_parts,
                                      int64_t N, const float* knots, int32_t in, int32_t out, int32_t G, int32_t K, int32_t mode,
                                      const void* pack_fwd, float* y, int64_t ldy, void* ws, size_t ws_bytes, void* stream) {
    ModeScope mode_scope_(mode);
    KAGNN_STAGE_AS("kagnn_kan_linear_fwd_parts", stream);
    int rc = check_kan_dims(__func__, in, out, G, K, mode);
    if (rc) return rc;
    KAGNN_CHECK_ARG(x_parts && part_widths && part_ld && num_parts >= 1, "null block table");
    KAGNN_CHECK_ARG(N >= 0 && ldy >= out, "bad shape");
    if (!kagnn_kan_fwd_parts_ok(part_widths, num_parts, in, out, G, K, mode))
        return fail(KAGNN_ERR_UNSUPPORTED, "%s: blocks not covered (kagnn_kan_fwd_parts_ok): concatenate and call kagnn_kan_linear_fwd", __func__);
    if (N == 0) return KAGNN_OK;
    KAGNN_CHECK_ARG(knots && pack_fwd && y, "null array");
    if (!fits32(N, ldy)) return fail(KAGNN_ERR_UNSUPPORTED, "%s: leading dimension > 7680 floats", __func__);
    static_assert(sizeof(long) == sizeof(int64_t), "LP64");
    return kan_sparse_fwd_parts(x_parts, part_widths, reinterpret_cast<const long*>(part_ld), num_parts, N, knots, in, out, G, K, pack_fwd, y, ldy, ws, ws_bytes, as_stream(stream), part_affine);
}

// forward + column moments of its output (the statistics of the BatchNorm1d that follows a convolution)
static bool fused_moments(int64_t N, int32_t in, int32_t out, int32_t G, int32_t K, int32_t mode) {
    return use_split_fwd(in, out, G, K, mode) && use_sparse_fwd(in, out, G, K, mode) && kan_sparse_fwd_moments_ok(N, in, out, G, K);
}

int kagnn_kan_fwd_moments_workspace_bytes(int64_t N, int32_t in, int32_t out, int32_t G, int32_t K, int32_t mode,
                                          size_t* bytes) {
    ModeScope mode_scope_(mode);
    size_t b = 0;
    int rc = kagnn_kan_fwd_workspace_bytes(N, in, out, G, K, mode, &b);
    if (rc) return rc;
    const size_t m = fused_moments(N, in, out, G, K, mode) ? kan_sparse_fwd_moments_ws_bytes(N, out) : bn_ws_bytes(N, out);
    *bytes = b > m ? b : m;
    return KAGNN_OK;
}

int kagnn_kan_linear_fwd_moments(const float* x, int64_t ldx, int64_t N, const float* knots, int32_t in,
                                 int32_t out, int32_t G, int32_t K, int32_t mode, const void* pack_fwd,
                                 float* y, int64_t ldy, float* col_mean, float* col_m2, void* ws, size_t ws_bytes,
                                 void* stream) {
    ModeScope mode_scope_(mode);
    KAGNN_STAGE(stream);
    int rc = check_kan_dims(__func__, in, out, G, K, mode);
    if (rc) return rc;
    KAGNN_CHECK_ARG(N >= 1 && ldx >= in && ldy >= out, "bad shape (column moments need at least one row)");
    KAGNN_CHECK_ARG(x && knots && pack_fwd && y && col_mean && col_m2, "null array");
    size_t need = 0;
    rc = kagnn_kan_fwd_moments_workspace_bytes(N, in, out, G, K, mode, &need);
    if (rc) return rc;
    KAGNN_CHECK_ARG(need == 0 || (ws && ws_bytes >= need), "workspace too small (kagnn_kan_fwd_moments_workspace_bytes)");
    if (fused_moments(N, in, out, G, K, mode)) {
        if (!(fits32(N, ldx) && fits32(N, ldy))) return fail(KAGNN_ERR_UNSUPPORTED, "%s: leading dimension > 7680 floats; call with KAGNN_PREC_FP32", __func__);
        return kan_sparse_fwd(x, ldx, N, knots, in, out, G, K, pack_fwd, y, ldy, ws, ws_bytes, col_mean, col_m2, as_stream(stream));
    }
    rc = kagnn_kan_linear_fwd(x, ldx, N, knots, in, out, G, K, mode, pack_fwd, y, ldy, ws, ws_bytes, stream);
    if (rc) return rc;
    return col_moments(y, ldy, N, out, col_mean, col_m2, ws, ws_bytes, as_stream(stream));
}

int kagnn_kan_linear_bwd_input(const float* x, int64_t ldx, const float* gy, int64_t ldgy, int64_t N,
                               const float* knots, int32_t in, int32_t out, int32_t G, int32_t K,
                               int32_t mode, const void* pack_dx, void* gx, int64_t ldgx, int32_t gx_dtype, void* stream) {
    ModeScope mode_scope_(mode);
    return kagnn_kan_linear_bwd_input_affine(x, ldx, nullptr, gy, ldgy, N, knots, in, out, G, K, mode, pack_dx, gx, ldgx, gx_dtype, stream);
}

// x_affine (NULL, or 2 * in floats: column scales, then column shifts): the layer input is  scale * x + shift  -- a BatchNorm1d
// output that was never written; gx is the gradient with respect to THAT input (the norm's own backward takes it from there).
// Covered: split precision, cubic layers of <= 8 coefficients and <= 64 outputs (the read-out of the node models).
int kagnn_kan_linear_bwd_input_affine(const float* x, int64_t ldx, const float* x_affine, const float* gy, int64_t ldgy, int64_t N,
                                      const float* knots, int32_t in, int32_t out, int32_t G, int32_t K,
                                      int32_t mode, const void* pack_dx, void* gx, int64_t ldgx, int32_t gx_dtype, void* stream) {
    ModeScope mode_scope_(mode);
    KAGNN_STAGE_AS("kagnn_kan_linear_bwd_input", stream);
    int rc = check_kan_dims(__func__, in, out, G, K, mode);
    if (rc) return rc;
    KAGNN_CHECK_ARG(N >= 0 && ldx >= in && ldgy >= out && ldgx >= in, "bad shape");
    KAGNN_CHECK_ARG(gx_dtype == KAGNN_DTYPE_F32 || gx_dtype == KAGNN_DTYPE_BF16, "gx_dtype must be KAGNN_DTYPE_F32 or KAGNN_DTYPE_BF16");
    if (N == 0) return KAGNN_OK;
    KAGNN_CHECK_ARG(x && gy && knots && pack_dx && gx, "null array");
    if (use_split_dx(in, out, G, K, mode)) {
        if (!(fits32(N, ldx) && fits32(N, ldgy) && fits32(N, ldgx))) return fail(KAGNN_ERR_UNSUPPORTED, "%s: leading dimension > 7680 floats; call with KAGNN_PREC_FP32", __func__);
        return kan_split_dx(x, ldx, gy, ldgy, N, knots, in, out, G, K, pack_dx, static_cast<float*>(gx), ldgx, as_stream(stream),
                            gx_dtype == KAGNN_DTYPE_BF16, x_affine);
    }
    if (x_affine) return fail(KAGNN_ERR_UNSUPPORTED, "%s: an input affine is applied by the split-precision kernels only", __func__);
    if (gx_dtype != KAGNN_DTYPE_F32) return fail(KAGNN_ERR_UNSUPPORTED, "%s: bf16 gradient rows are produced by the split-precision kernels only", __func__);
    float* gxf = static_cast<float*>(gx);
    return kan_f32_dx(x, ldx, gy, ldgy, N, knots, in, out, G, K, (const float*)pack_dx, gxf, ldgx, mode == KAGNN_PREC_FP32_GRID, as_stream(stream));
}

// kagnn_kan_linear_bwd_input_affine that ALSO leaves the two column sums the backward of the folded BatchNorm1d starts from:
// sums[0][in] = sum_n gx, sums[1][in] = sum_n gx * xhat, xhat = (x - bn_mean) * bn_rstd on the raw rows (x = the norm's input).
// The read-out's gradient of the LAST convolution's output in the node models (reference node_classification_clean/models.py:198-203):
// that norm's incoming gradient is exactly this gx, so its statistics pass over (gx, x) goes away.  Covered (kagnn_kan_bwd_input_sums_ok):
// split precision, cubic layers of <= 8 coefficients, <= 64 inputs (a multiple of 4) and outputs, >= 32768 rows.
int kagnn_kan_bwd_input_sums_ok(int64_t N, int32_t in, int32_t out, int32_t G, int32_t K, int32_t mode) {
    ModeScope mode_scope_(mode);
    return mode == KAGNN_PREC_SPLIT && use_split_dx(in, out, G, K, mode) && kan_split_dx_stats_ok(N, in, out, G, K) ? 1 : 0;
}
int kagnn_kan_bwd_input_sums_workspace_bytes(int64_t N, int32_t in, size_t* bytes) {
    KAGNN_CHECK_ARG(N >= 0 && in >= 1 && bytes, "bad argument");
    *bytes = ((size_t)kan_split_dx_stats_blocks(N) + 1) * 2 * in * sizeof(float);
    return KAGNN_OK;
}
int kagnn_kan_linear_bwd_input_affine_sums(const float* x, int64_t ldx, const float* x_affine, const float* bn_mean, const float* bn_rstd,
                                           const float* gy, int64_t ldgy, int64_t N, const float* knots, int32_t in, int32_t out,
                                           int32_t G, int32_t K, int32_t mode, const void* pack_dx, float* gx, int64_t ldgx,
                                           float* sums, void* workspace, size_t workspace_bytes, void* stream) {
    ModeScope mode_scope_(mode);
    int rc = check_kan_dims(__func__, in, out, G, K, mode);
    if (rc) return rc;
    KAGNN_CHECK_ARG(N >= 1 && ldx >= in && ldgy >= out && ldgx >= in, "bad shape");
    KAGNN_CHECK_ARG(x && x_affine && bn_mean && bn_rstd && gy && knots && pack_dx && gx && sums && workspace, "null array");
    if (!kagnn_kan_bwd_input_sums_ok(N, in, out, G, K, mode)) return fail(KAGNN_ERR_UNSUPPORTED, "%s: shape not covered (kagnn_kan_bwd_input_sums_ok)", __func__);
    if (!(fits32(N, ldx) && fits32(N, ldgy) && fits32(N, ldgx))) return fail(KAGNN_ERR_UNSUPPORTED, "%s: leading dimension > 7680 floats", __func__);
    const int B = kan_split_dx_stats_blocks(N);
    KAGNN_CHECK_ARG(workspace_bytes >= ((size_t)B + 1) * 2 * in * sizeof(float), "workspace too small (kagnn_kan_bwd_input_sums_workspace_bytes)");
    float* partial = static_cast<float*>(workspace);
    {
        KAGNN_STAGE_AS("kagnn_kan_linear_bwd_input", stream);
        rc = kan_split_dx_stats(x, ldx, gy, ldgy, N, knots, in, out, G, K, pack_dx, gx, ldgx, as_stream(stream), x_affine, bn_mean, bn_rstd, partial);
        if (rc) return rc;
    }
    KAGNN_STAGE_AS("kagnn_batchnorm_bwd statistics fold", stream);
    return bn_finish_partials(partial, B, in, sums, as_stream(stream));
}

int kagnn_kan_bwd_weight_workspace_bytes(int64_t N, int32_t in, int32_t out, int32_t G, int32_t K,
                                         int32_t mode, size_t* bytes) {
    ModeScope mode_scope_(mode);
    int rc = check_kan_dims(__func__, in, out, G, K, mode);
    if (rc) return rc;
    KAGNN_CHECK_ARG(bytes && N >= 0, "bad argument");
    *bytes = use_split_dw(in, out, G, K, mode) ? kan_split_dw_ws_bytes(N, in, out, G + K, K)
                                                                   : kan_f32_dw_ws_bytes(N, in, out, G + K);
    return KAGNN_OK;
}

int kagnn_kan_linear_bwd_weight(const float* x, int64_t ldx, const float* gy, int64_t ldgy, int64_t N,
                                const float* knots, int32_t in, int32_t out, int32_t G, int32_t K,
                                int32_t mode, const float* sw, const float* sc, float* g_bw,
                                float* g_sw, float* g_sc, void* ws, size_t ws_bytes, void* stream) {
    ModeScope mode_scope_(mode);
    return kagnn_kan_linear_bwd_weight_affine(x, ldx, nullptr, gy, ldgy, N, knots, in, out, G, K, mode, sw, sc, g_bw, g_sw, g_sc, ws, ws_bytes, stream);
}

// x_affine: as kagnn_kan_linear_bwd_input_affine (the weight gradient of a layer whose input is a folded BatchNorm1d output)
int kagnn_kan_linear_bwd_weight_affine(const float* x, int64_t ldx, const float* x_affine, const float* gy, int64_t ldgy, int64_t N,
                                       const float* knots, int32_t in, int32_t out, int32_t G, int32_t K,
                                       int32_t mode, const float* sw, const float* sc, float* g_bw,
                                       float* g_sw, float* g_sc, void* ws, size_t ws_bytes, void* stream) {
    ModeScope mode_scope_(mode);
    KAGNN_STAGE_AS("kagnn_kan_linear_bwd_weight", stream);
    int rc = check_kan_dims(__func__, in, out, G, K, mode);
    if (rc) return rc;
    KAGNN_CHECK_ARG(N >= 0 && ldx >= in && ldgy >= out, "bad shape");
    KAGNN_CHECK_ARG(knots && sw && g_sw && ws, "null array");          // g_base_weight NULL: not wanted
    KAGNN_CHECK_ARG(N == 0 || (x && gy), "null array");
    KAGNN_CHECK_ARG((sc == nullptr) == (g_sc == nullptr), "spline_scaler and its gradient must both be given or both be null");
    if (use_split_dw(in, out, G, K, mode)) {
        if (!(fits32(N, ldx) && fits32(N, ldgy))) return fail(KAGNN_ERR_UNSUPPORTED, "%s: leading dimension > 7680 floats; call with KAGNN_PREC_FP32", __func__);
        return kan_split_dw(x, ldx, gy, ldgy, N, knots, in, out, G, K, sw, sc, g_bw, g_sw, g_sc, (float*)ws, ws_bytes, as_stream(stream), x_affine);
    }
    if (x_affine) return fail(KAGNN_ERR_UNSUPPORTED, "%s: an input affine is applied by the split-precision kernels only", __func__);
    return kan_f32_dw(x, ldx, gy, ldgy, N, knots, in, out, G, K, sw, sc, g_bw, g_sw, g_sc, (float*)ws, ws_bytes, mode == KAGNN_PREC_FP32_GRID, as_stream(stream));
}

// ---------------------------------------------------------------- FastKAN
static int check_fk(const char* fn, int in, int out, int ng, int mode) {
    if (in < 1 || out < 1) return fail(KAGNN_ERR_ARG, "%s: input_dim/output_dim must be >= 1", fn);
    if (ng < 1 || ng > kMaxKnots) return fail(KAGNN_ERR_UNSUPPORTED, "%s: num_grids out of range", fn);
    if (mode != KAGNN_PREC_FP32 && mode != KAGNN_PREC_SPLIT) return fail(KAGNN_ERR_ARG, "%s: unknown precision mode", fn);
    return KAGNN_OK;
}

// ---------------------------------------------------------------- adaptive grids (update_grid)
int kagnn_kan_bsplines(const float* x, int64_t ldx, int64_t N, const float* grid, int32_t in, int32_t G,
                       int32_t K, float* bases, void* stream) {
    int rc = check_kan_dims(__func__, in, 1, G, K, KAGNN_PREC_FP32_GRID);
    if (rc) return rc;
    KAGNN_CHECK_ARG(N >= 0 && ldx >= in, "bad shape");
    KAGNN_CHECK_ARG(N == 0 || (x && grid && bases), "null array");
    return kan_bsplines(x, ldx, N, grid, in, G, K, bases, as_stream(stream));
}

int kagnn_kan_grid_refit_workspace_bytes(int64_t N, int32_t in, int32_t G, int32_t K, size_t* bytes) {
    int rc = check_kan_dims(__func__, in, 1, G, K, KAGNN_PREC_FP32_GRID);
    if (rc) return rc;
    KAGNN_CHECK_ARG(bytes != nullptr && N >= 1, "null output or no rows");
    *bytes = kan_grid_refit_ws_bytes(N, in);
    return KAGNN_OK;
}

int kagnn_kan_grid_refit(const float* x, int64_t ldx, int64_t N, const float* grid_old, const float* grid_new,
                         int32_t in, int32_t out, int32_t G, int32_t K, const float* sw, const float* sc,
                         float* new_sw, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_kan_dims(__func__, in, out, G, K, KAGNN_PREC_FP32_GRID);
    if (rc) return rc;
    KAGNN_CHECK_ARG(N >= 1 && ldx >= in, "bad shape");
    KAGNN_CHECK_ARG(x && grid_old && grid_new && sw && new_sw && ws, "null array");
    KAGNN_CHECK_ARG(sw != new_sw, "the refit is not in place");
    return kan_grid_refit(x, ldx, N, grid_old, grid_new, in, out, G, K, sw, sc, new_sw, ws, ws_bytes, as_stream(stream));
}

int kagnn_fastkan_fwd_workspace_bytes(int64_t N, int32_t in, int32_t out, int32_t ng, int32_t mode, size_t* bytes) {
    ModeScope mode_scope_(mode);
    int rc = check_fk(__func__, in, out, ng, mode);
    if (rc) return rc;
    KAGNN_CHECK_ARG(bytes && N >= 0, "bad argument");
    *bytes = fastkan_fwd_ws_bytes(N, in, out, ng, mode);
    return KAGNN_OK;
}

int kagnn_fastkan_fwd(const float* x, int64_t ldx, int64_t N, int32_t in, int32_t out, int32_t ng,
                      const float* centers, float denominator, const float* ln_w, const float* ln_b,
                      float ln_eps, const float* spline_w, const float* base_w, const float* base_b,
                      float* y, int64_t ldy, float* row_stats, int32_t mode, void* ws, size_t ws_bytes, void* stream) {
    ModeScope mode_scope_(mode);
    KAGNN_STAGE(stream);
    int rc = check_fk(__func__, in, out, ng, mode);
    if (rc) return rc;
    KAGNN_CHECK_ARG(N >= 0 && ldx >= in && ldy >= out, "bad shape");
    if (N == 0) return KAGNN_OK;
    KAGNN_CHECK_ARG(x && centers && spline_w && y && ws, "null array");
    KAGNN_CHECK_ARG(denominator != 0.0f, "denominator is zero");
    KAGNN_CHECK_ARG((ln_w == nullptr) == (ln_b == nullptr), "layernorm weight and bias must both be given or both be null");
    if (mode == KAGNN_PREC_SPLIT && !(fits32(N, ldx) && fits32(N, ldy)))
        return fail(KAGNN_ERR_UNSUPPORTED, "%s: leading dimension > 7680 floats; call with KAGNN_PREC_FP32", __func__);
    return fastkan_fwd(x, ldx, N, in, out, ng, centers, denominator, ln_w, ln_b, ln_eps, spline_w, base_w,
                       base_b, y, ldy, row_stats, ws, ws_bytes, mode, as_stream(stream));
}

int kagnn_fastkan_bwd_workspace_bytes(int64_t N, int32_t in, int32_t out, int32_t ng, int32_t mode, size_t* bytes) {
    ModeScope mode_scope_(mode);
    int rc = check_fk(__func__, in, out, ng, mode);
    if (rc) return rc;
    KAGNN_CHECK_ARG(bytes && N >= 0, "bad argument");
    *bytes = fastkan_bwd_ws_bytes(N, in, out, ng, mode);
    return KAGNN_OK;
}

int kagnn_fastkan_bwd(const float* x, int64_t ldx, const float* gy, int64_t ldgy, int64_t N, int32_t in,
                      int32_t out, int32_t ng, const float* centers, float denominator,
                      const float* ln_w, const float* ln_b, float ln_eps, const float* spline_w,
                      const float* base_w, const float* row_stats, float* gx, int64_t ldgx,
                      float* g_ln_w, float* g_ln_b, float* g_spline_w, float* g_base_w,
                      float* g_base_b, int32_t mode, void* ws, size_t ws_bytes, void* stream) {
    ModeScope mode_scope_(mode);
    KAGNN_STAGE(stream);
    int rc = check_fk(__func__, in, out, ng, mode);
    if (rc) return rc;
    KAGNN_CHECK_ARG(N >= 0 && ldx >= in && ldgy >= out && ldgx >= in, "bad shape");
    KAGNN_CHECK_ARG(centers && spline_w && g_spline_w && ws, "null array");
    KAGNN_CHECK_ARG(N == 0 || (x && gy && gx), "null array");
    KAGNN_CHECK_ARG(ln_w == nullptr || (ln_b && row_stats && g_ln_w && g_ln_b), "layernorm needs bias, row_stats and both gradient outputs");
    KAGNN_CHECK_ARG(base_w == nullptr || (g_base_w && g_base_b), "base branch needs both gradient outputs");
    if (mode == KAGNN_PREC_SPLIT && !(fits32(N, ldx) && fits32(N, ldgy) && fits32(N, ldgx)))
        return fail(KAGNN_ERR_UNSUPPORTED, "%s: leading dimension > 7680 floats; call with KAGNN_PREC_FP32", __func__);
    return fastkan_bwd(x, ldx, gy, ldgy, N, in, out, ng, centers, denominator, ln_w, ln_b, ln_eps, spline_w,
                       base_w, row_stats, gx, ldgx, g_ln_w, g_ln_b, g_spline_w, g_base_w, g_base_b, ws,
                       ws_bytes, mode, as_stream(stream));
}

// ---------------------------------------------------------------- feature-sharded FastKAN layer
int kagnn_fastkan_row_moments(const float* x, int64_t ldx, int64_t N, int32_t in, float* moments, void* stream) {
    KAGNN_STAGE(stream);
    KAGNN_CHECK_ARG(N >= 0 && in >= 1 && ldx >= in, "bad shape");
    KAGNN_CHECK_ARG(N == 0 || (x && moments), "null array");
    return fastkan_row_moments(x, ldx, N, in, moments, as_stream(stream));
}

int kagnn_fastkan_merge_moments(const float* gathered, int32_t P, int64_t N, int32_t in, float ln_eps, float* row_stats,
                                void* stream) {
    KAGNN_STAGE(stream);
    KAGNN_CHECK_ARG(N >= 0 && in >= 1 && P >= 1, "bad shape");
    KAGNN_CHECK_ARG(N == 0 || (gathered && row_stats), "null array");
    return fastkan_merge_moments(gathered, P, N, in, ln_eps, row_stats, as_stream(stream));
}

int kagnn_fastkan_shard_fwd(const float* x, int64_t ldx, int64_t N, int32_t in, int32_t out, int32_t ng,
                            const float* centers, float denominator, const float* ln_w, const float* ln_b,
                            const float* row_stats, const float* spline_w, const float* base_w, const float* base_b,
                            float* y, int64_t ldy, int32_t mode, void* ws, size_t ws_bytes, void* stream) {
    ModeScope mode_scope_(mode);
    KAGNN_STAGE(stream);
    int rc = check_fk(__func__, in, out, ng, mode);
    if (rc) return rc;
    KAGNN_CHECK_ARG(N >= 0 && ldx >= in && ldy >= out, "bad shape");
    if (N == 0) return KAGNN_OK;
    KAGNN_CHECK_ARG(x && centers && spline_w && y && ws, "null array");
    KAGNN_CHECK_ARG(denominator != 0.0f, "denominator is zero");
    KAGNN_CHECK_ARG((ln_w == nullptr) == (ln_b == nullptr), "layernorm weight and bias must both be given or both be null");
    KAGNN_CHECK_ARG(ln_w == nullptr || row_stats, "layernorm on a column shard needs the merged row statistics");
    if (mode == KAGNN_PREC_SPLIT && !(fits32(N, ldx) && fits32(N, ldy)))
        return fail(KAGNN_ERR_UNSUPPORTED, "%s: leading dimension > 7680 floats; call with KAGNN_PREC_FP32", __func__);
    return fastkan_fwd(x, ldx, N, in, out, ng, centers, denominator, ln_w, ln_b, 0.0f, spline_w, base_w, base_b, y, ldy,
                       const_cast<float*>(row_stats), ws, ws_bytes, mode, as_stream(stream), ln_w != nullptr);
}

int kagnn_fastkan_shard_bwd(const float* x, int64_t ldx, const float* gy, int64_t ldgy, int64_t N, int32_t in,
                            int32_t out, int32_t ng, const float* centers, float denominator,
                            const float* ln_w, const float* ln_b, const float* spline_w,
                            const float* base_w, const float* row_stats, float* gx, int64_t ldgx,
                            float* row_sums, float* g_spline_w, float* g_base_w,
                            float* g_base_b, int32_t parts, int32_t mode, void* ws, size_t ws_bytes, void* stream) {
    ModeScope mode_scope_(mode);
    KAGNN_STAGE(stream);
    int rc = check_fk(__func__, in, out, ng, mode);
    if (rc) return rc;
    KAGNN_CHECK_ARG(N >= 0 && ldx >= in && ldgy >= out && ldgx >= in, "bad shape");
    KAGNN_CHECK_ARG(parts >= 1 && parts <= 3, "parts: 1 = input-gradient half, 2 = weight-gradient half, 3 = both");
    KAGNN_CHECK_ARG(centers && spline_w && ws, "null array");
    KAGNN_CHECK_ARG(N == 0 || (x && gy), "null array");
    KAGNN_CHECK_ARG(!(parts & 1) || N == 0 || gx, "null array");
    KAGNN_CHECK_ARG(!(parts & 1) || ln_w == nullptr || (ln_b && row_stats && (N == 0 || row_sums)), "layernorm needs bias, row_stats and row_sums");
    KAGNN_CHECK_ARG(!(parts & 2) || (g_spline_w && (base_w == nullptr || g_base_w)), "the weight-gradient half needs its outputs");
    KAGNN_CHECK_ARG(ln_w == nullptr || (ln_b && row_stats), "layernorm needs bias and row_stats");
    if (mode == KAGNN_PREC_SPLIT && !(fits32(N, ldx) && fits32(N, ldgy) && fits32(N, ldgx)))
        return fail(KAGNN_ERR_UNSUPPORTED, "%s: leading dimension > 7680 floats; call with KAGNN_PREC_FP32", __func__);
    return fastkan_bwd(x, ldx, gy, ldgy, N, in, out, ng, centers, denominator, ln_w, ln_b, 0.0f, spline_w,
                       base_w, row_stats, gx, ldgx, nullptr, nullptr, g_spline_w, g_base_w, g_base_b, ws,
                       ws_bytes, mode, as_stream(stream), parts == 3 ? 1 : parts == 1 ? 3 : 4, row_sums, 0);
}

int kagnn_fastkan_shard_bwd_finish(const float* x, int64_t ldx, int64_t N, int32_t in, int32_t in_total, int32_t out,
                                   int32_t ng, const float* ln_w, const float* ln_b, const float* row_stats,
                                   const float* row_sums, float* gx, int64_t ldgx, float* g_ln_w, float* g_ln_b,
                                   int32_t mode, void* ws, size_t ws_bytes, void* stream) {
    ModeScope mode_scope_(mode);
    KAGNN_STAGE(stream);
    int rc = check_fk(__func__, in, out, ng, mode);
    if (rc) return rc;
    KAGNN_CHECK_ARG(N >= 0 && ldx >= in && ldgx >= in && in_total >= in, "bad shape");
    KAGNN_CHECK_ARG(ln_w && ln_b && g_ln_w && g_ln_b && ws, "null array");
    KAGNN_CHECK_ARG(N == 0 || (x && gx && row_stats && row_sums), "null array");
    return fastkan_bwd(x, ldx, nullptr, out, N, in, out, ng, nullptr, 1.0f, ln_w, ln_b, 0.0f, nullptr,
                       nullptr, row_stats, gx, ldgx, g_ln_w, g_ln_b, nullptr, nullptr, nullptr, ws,
                       ws_bytes, mode, as_stream(stream), 2, const_cast<float*>(row_sums), in_total);
}

// ---------------------------------------------------------------- BatchNorm1d
int kagnn_batchnorm_workspace_bytes(int64_t N, int32_t F, size_t* bytes) {
    KAGNN_CHECK_ARG(bytes && N >= 0 && F >= 1, "bad argument");
    *bytes = bn_ws_bytes(N, F);
    return KAGNN_OK;
}

int kagnn_batchnorm_fwd(const float* x, int64_t ldx, int64_t N, int32_t F, const float* weight, const float* bias,
                        float* running_mean, float* running_var, float momentum, float eps, int32_t training,
                        const float* col_mean, const float* col_m2, float dropout_p, uint64_t dropout_seed,
                        float* y, int64_t ldy, float* save_mean, float* save_rstd, void* ws, size_t ws_bytes,
                        void* stream) {
    KAGNN_STAGE(stream);
    KAGNN_CHECK_ARG(N >= 0 && F >= 1 && ldx >= F && ldy >= F, "bad shape");
    if (N == 0) return KAGNN_OK;
    KAGNN_CHECK_ARG(x && y && save_mean && save_rstd && ws, "null array");
    KAGNN_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "running_mean and running_var must both be given or both be null");
    KAGNN_CHECK_ARG(training || running_mean, "eval mode needs the running statistics");
    KAGNN_CHECK_ARG((col_mean == nullptr) == (col_m2 == nullptr), "col_mean and col_m2 must both be given or both be null");
    KAGNN_CHECK_ARG(dropout_p >= 0.0f && dropout_p <= 1.0f, "dropout_p outside [0, 1]");
    return bn_fwd(x, ldx, N, F, weight, bias, running_mean, running_var, momentum, eps, training, col_mean, col_m2, dropout_p,
                  dropout_seed, y, ldy, save_mean, save_rstd, ws, ws_bytes, as_stream(stream));
}

// The statistics half of a training-mode BatchNorm1d forward whose normalising pass is folded into the kernels that read its
// output (round 4; reference node_classification_clean/models.py:198-200): from the column moments (mean, M2) the producing
// kernel left behind -> save_mean, save_rstd (for the backward), the running statistics update, and the per-column affine
//   affine[0][c] = gamma[c] * rstd[c],   affine[1][c] = beta[c] - mean[c] * affine[0][c]
// that kagnn_aggregate_sum_affine / kagnn_gin_kan_layer_fwd_affine / kagnn_kan_linear_*_affine apply to the rows they load.
int kagnn_batchnorm_stats_affine(const float* col_mean, const float* col_m2, int64_t N, int32_t F, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float momentum, float eps, float* save_mean,
                                 float* save_rstd, float* affine, void* stream) {
    KAGNN_CHECK_ARG(N >= 1 && F >= 1 && col_mean && col_m2 && save_mean && save_rstd && affine, "bad argument");
    KAGNN_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "running_mean and running_var must both be given or both be null");
    return bn_stats_affine(col_mean, col_m2, N, F, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_rstd, affine,
                           as_stream(stream));
}

int kagnn_batchnorm_bwd(const float* x, int64_t ldx, const float* gy, int64_t ldgy, int64_t N, int32_t F,
                        const float* weight, const float* save_mean, const float* save_rstd, int32_t training,
                        float dropout_p, uint64_t dropout_seed,
                        float* gx, int64_t ldgx, float* g_weight, float* g_bias, void* ws, size_t ws_bytes,
                        void* stream) {
    KAGNN_STAGE(stream);
    KAGNN_CHECK_ARG(N >= 0 && F >= 1 && ldx >= F && ldgy >= F && (gx == nullptr || ldgx >= F), "bad shape");
    if (N == 0) return KAGNN_OK;
    KAGNN_CHECK_ARG(x && gy && save_mean && save_rstd && ws, "null array");
    KAGNN_CHECK_ARG(dropout_p >= 0.0f && dropout_p <= 1.0f, "dropout_p outside [0, 1]");
    return bn_bwd(x, ldx, gy, ldgy, N, F, weight, save_mean, save_rstd, training, dropout_p, dropout_seed, gx, ldgx, g_weight,
                  g_bias, ws, ws_bytes, as_stream(stream));
}

// ---------------------------------------------------------------- GAT attention aggregation
int kagnn_gat_logits(const float* xh, int64_t ldx, int64_t N, int32_t H, int32_t C, const float* att_src,
                     const float* att_dst, float* a_src, float* a_dst, void* stream) {
    KAGNN_CHECK_ARG(N >= 0 && H >= 1 && C >= 1 && ldx >= (int64_t)H * C, "bad shape");
    if (N == 0) return KAGNN_OK;
    KAGNN_CHECK_ARG(xh && att_src && att_dst && a_src && a_dst, "null array");
    return gat_logits(xh, ldx, N, H, C, att_src, att_dst, a_src, a_dst, as_stream(stream));
}

int kagnn_gat_fwd(const float* xh, int64_t ldx, const float* a_src, const float* a_dst, const int32_t* rowptr,
                  const int32_t* col, int64_t N, int32_t H, int32_t C, const float* bias, float* out, int64_t ldo,
                  float* row_max, float* row_sum, const int32_t* hub_seg, int64_t num_hub_seg, int32_t hub_threshold,
                  void* stream) {
    KAGNN_CHECK_ARG(N >= 0 && H >= 1 && C >= 1 && ldx >= (int64_t)H * C && ldo >= (int64_t)H * C, "bad shape");
    if (N == 0) return KAGNN_OK;
    KAGNN_CHECK_ARG(xh && a_src && a_dst && rowptr && out && row_max && row_sum, "null array");
    return gat_fwd(xh, ldx, a_src, a_dst, rowptr, col, N, H, C, bias, out, ldo, row_max, row_sum, hub_seg, num_hub_seg,
                   hub_threshold, as_stream(stream));
}

int kagnn_gat_bwd(const float* xh, int64_t ldx, const float* gout, int64_t ldg, const float* out, int64_t ldo,
                  const float* bias, const float* a_src, const float* a_dst, const float* row_max,
                  const float* row_sum, const int32_t* rowptr, const int32_t* col, const int32_t* perm,
                  const int32_t* rowptr_t, const int32_t* col_t, const int32_t* perm_t, const float* att_src,
                  const float* att_dst, int64_t N, int32_t H, int32_t C, float* edge_scratch, float* self_scratch,
                  float* g_dst, float* g_src, float* gx, int64_t ldgx, const int32_t* hub_seg, int64_t num_hub_seg,
                  int32_t hub_threshold, void* stream) {
    KAGNN_CHECK_ARG(N >= 0 && H >= 1 && C >= 1 && ldx >= (int64_t)H * C && ldg >= (int64_t)H * C && ldo >= (int64_t)H * C &&
                    ldgx >= (int64_t)H * C, "bad shape");
    if (N == 0) return KAGNN_OK;
    KAGNN_CHECK_ARG(xh && gout && out && a_src && a_dst && row_max && row_sum && rowptr && rowptr_t && att_src && att_dst &&
                    self_scratch && g_dst && g_src && gx, "null array");
    return gat_bwd(xh, ldx, gout, ldg, out, ldo, bias, a_src, a_dst, row_max, row_sum, rowptr, col, perm, rowptr_t, col_t,
                   perm_t, att_src, att_dst, N, H, C, edge_scratch, self_scratch, g_dst, g_src, gx, ldgx, hub_seg, num_hub_seg,
                   hub_threshold, as_stream(stream));
}

int kagnn_gat_att_grad_workspace_bytes(int64_t N, int32_t H, int32_t C, size_t* bytes) {
    KAGNN_CHECK_ARG(bytes && N >= 0 && H >= 1 && C >= 1, "bad argument");
    *bytes = gat_att_grad_ws_bytes(N, H, C);
    return KAGNN_OK;
}

int kagnn_gat_att_grad(const float* xh, int64_t ld, const float* g_src, const float* g_dst, int64_t N, int32_t H,
                       int32_t C, float* g_att_src, float* g_att_dst, void* ws, size_t ws_bytes, void* stream) {
    KAGNN_CHECK_ARG(N >= 0 && H >= 1 && C >= 1 && ld >= (int64_t)H * C, "bad shape");
    KAGNN_CHECK_ARG(g_att_src && g_att_dst && ws && (N == 0 || (xh && g_src && g_dst)), "null array");
    return gat_att_grad(xh, ld, g_src, g_dst, N, H, C, g_att_src, g_att_dst, ws, ws_bytes, as_stream(stream));
}

// ---------------------------------------------------------------- harness loss
int kagnn_softmax_xent_workspace_bytes(int64_t N, size_t* bytes) {
    KAGNN_CHECK_ARG(bytes != nullptr && N >= 0, "null output or negative size");
    *bytes = xent_ws_bytes(N);
    return KAGNN_OK;
}

int kagnn_softmax_xent_fwd(const float* logits, int64_t ld, int64_t N, int32_t C, const int64_t* labels,
                           const uint8_t* mask, int32_t pre_softmax, float* loss, float* row_stats, float* count,
                           void* ws, size_t ws_bytes, void* stream) {
    KAGNN_CHECK_ARG(N >= 0 && C >= 1 && ld >= C, "bad shape");
    KAGNN_CHECK_ARG(loss && count && ws && (N == 0 || (logits && labels && row_stats)), "null array");
    return xent_fwd(logits, ld, N, C, (const long*)labels, mask, pre_softmax, loss, row_stats, count, ws, ws_bytes,
                    as_stream(stream));
}

int kagnn_softmax_xent_bwd(const float* logits, int64_t ld, int64_t N, int32_t C, const int64_t* labels,
                           const uint8_t* mask, int32_t pre_softmax, const float* row_stats, const float* count,
                           const float* g_loss, float* g_logits, int64_t ldg, void* stream) {
    KAGNN_CHECK_ARG(N >= 0 && C >= 1 && ld >= C && ldg >= C, "bad shape");
    KAGNN_CHECK_ARG(N == 0 || (logits && labels && row_stats && count && g_loss && g_logits), "null array");
    return xent_bwd(logits, ld, N, C, (const long*)labels, mask, pre_softmax, row_stats, count, g_loss, g_logits, ldg,
                    as_stream(stream));
}

// mean absolute error of two contiguous fp32 vectors (torch.nn.L1Loss, reduction = "mean"): graph_regression/optuna_zinc.py:58
int kagnn_l1_loss_fwd(const float* pred, const float* target, int64_t n, float* loss, void* stream) {
    KAGNN_STAGE(stream);
    KAGNN_CHECK_ARG(n >= 0 && loss && (n == 0 || (pred && target)), "null array or negative size");
    return l1_loss_fwd(pred, target, n, loss, as_stream(stream));
}

int kagnn_l1_loss_bwd(const float* pred, const float* target, int64_t n, const float* g_loss, float* g_pred, void* stream) {
    KAGNN_STAGE(stream);
    KAGNN_CHECK_ARG(n >= 0 && (n == 0 || (pred && target && g_loss && g_pred)), "null array or negative size");
    return l1_loss_bwd(pred, target, n, g_loss, g_pred, as_stream(stream));
}

// one Adam update of `count` fp32 parameter tensors (HOST arrays of device pointers and element counts; step = 1, 2, ...: the
// number of this update, for the bias corrections): torch.optim.Adam's rule without amsgrad, one launch per 32 tensors
int kagnn_adam_step(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                    const int64_t* numel, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, void* stream) {
    KAGNN_STAGE(stream);
    KAGNN_CHECK_ARG(count >= 0 && step >= 1 && (count == 0 || (params && grads && exp_avg && exp_avg_sq && numel)), "null array or bad count / step");
    KAGNN_CHECK_ARG(lr >= 0.0f && beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f && eps >= 0.0f, "bad hyper-parameter");
    for (int k = 0; k < count; ++k)
        KAGNN_CHECK_ARG(numel[k] >= 0 && (numel[k] == 0 || (params[k] && grads[k] && exp_avg[k] && exp_avg_sq[k])), "null tensor");
    static_assert(sizeof(long) == sizeof(int64_t), "LP64");
    return adam_step(count, params, grads, exp_avg, exp_avg_sq, reinterpret_cast<const long*>(numel), lr, beta1, beta2, eps, weight_decay,
                     (long)step, as_stream(stream));
}


// ---------------------------------------------------------------- direct peer-to-peer exchange (p2p.hip)

int kagnn_p2p_reduce_scatter(const float* const* parts, int32_t world, int32_t rank, int64_t N, int32_t out, int64_t ld, float* y,
                             int64_t ldy, void* stream) {
    KAGNN_STAGE(stream);
    KAGNN_CHECK_ARG(parts && N >= 0 && out >= 1 && ld >= out && (N == 0 || y), "bad argument");
    return p2p_reduce_scatter(parts, world, rank, N, out, ld, y, ldy, as_stream(stream));
}

int kagnn_p2p_all_gather(const float* const* shards, int32_t world, int64_t N, int32_t w, int64_t lds, float* g, int64_t ldg,
                         void* stream) {
    KAGNN_STAGE(stream);
    KAGNN_CHECK_ARG(shards && N >= 0 && w >= 1 && lds >= w && ldg >= (int64_t)w * world && (N == 0 || g), "bad argument");
    return p2p_all_gather(shards, world, N, w, lds, g, ldg, as_stream(stream));
}

// ---------------------------------------------------------------- one KAN-GIN convolution per call
static size_t al256z(size_t b) { return (b + 255) & ~(size_t)255; }

int kagnn_gin_kan_layer_workspace_bytes(int64_t N, int32_t L, const int32_t* widths, int32_t G, int32_t K, int32_t mode,
                                        int64_t num_hub_seg, int64_t num_hub_seg_t, size_t* fwd_bytes, size_t* bwd_bytes) {
    ModeScope mode_scope_(mode);
    KAGNN_CHECK_ARG(N >= 0 && L >= 1 && L <= 8 && widths && fwd_bytes && bwd_bytes, "bad argument");
    size_t fw = 0, dw = 0;
    int wmax = 0;
    for (int l = 0; l < L; ++l) {
        int rc = check_kan_dims(__func__, widths[l], widths[l + 1], G, K, mode);
        if (rc) return rc;
        size_t b = 0;
        rc = (l == L - 1 ? kagnn_kan_fwd_moments_workspace_bytes : kagnn_kan_fwd_workspace_bytes)(N, widths[l], widths[l + 1], G, K, mode, &b);
        if (rc) return rc;
        fw = b > fw ? b : fw;
        rc = kagnn_kan_bwd_weight_workspace_bytes(N, widths[l], widths[l + 1], G, K, mode, &b); if (rc) return rc;
        dw = b > dw ? b : dw;
        wmax = widths[l] > wmax ? widths[l] : wmax;
    }
    const size_t hub_f = aggregate_bf16_ws_bytes(num_hub_seg, widths[0]), hub_t = aggregate_bf16_ws_bytes(num_hub_seg_t, widths[0]);
    // (+ the hub-row fix-up of the aggregation fused into the first KANLinear, narrow first layers: kan_sparse_fwd_agg)
    const size_t fuse_b = widths[0] <= 32 ? kan_sparse_fwd_agg_ws_bytes(num_hub_seg, widths[0], widths[1]) : 0;
    *fwd_bytes = al256z(hub_f) + al256z(fw) + al256z(fuse_b) + 256;
    // backward: hub partials | dW slabs | two ping-pong gradient matrices [N, max width] (fp32)
    *bwd_bytes = al256z(hub_t) + al256z(dw) + 2 * al256z((size_t)N * wmax * sizeof(float)) + 256;
    return KAGNN_OK;
}

// GINE message passing around the same chain (reference graph_regression/models.py:98,107-119: GINEConv(KAN)): the aggregation of
// the forward is kagnn_aggregate_gine (relu(x_j + e_ij) messages, edge attributes in ORIGINAL edge order through `perm`), the last
// step of the backward kagnn_aggregate_gine_bwd on the transposed structure (also the edge-attribute gradient)
struct GineStage {
    const float* x; int64_t ldx; const float* ea; int64_t lde; const int32_t* perm;      // forward: perm of the CSR; backward: of its transpose
    float* g_ea; int64_t ldge;                                                           // backward only (g_ea may be null)
    int accumulate_g_ea = 0;                                                             // backward: g_ea += (the stack's later convolutions)
    int prepacked = 0;                                                                   // forward: the packs were made by the caller (one launch for a whole stack)
};

static int layer_fwd_impl(const void* x, int32_t x_dtype, int64_t ldx, int64_t N, const int32_t* rowptr, const int32_t* col,
                          const int32_t* hub_seg, int64_t num_hub_seg, int32_t hub_threshold, float self_scale,
                          const float* in_col_scale, const float* in_col_shift,
                          int32_t L, const int32_t* widths, const float* const* bw, const float* const* sw,
                          const float* const* sc, const float* knots, int32_t G, int32_t K, int32_t mode,
                          float* const* acts, void* const* pack_fwd, void* const* pack_dx, float* col_mean,
                          float* col_m2, void* workspace, size_t workspace_bytes, void* stream, const char* fn,
                          const GineStage* gine = nullptr) {
    (void)fn;
    KAGNN_CHECK_ARG(N >= 0 && L >= 1 && L <= 8 && widths && bw && sw && acts && pack_fwd && pack_dx, "bad argument");
    KAGNN_CHECK_ARG((in_col_scale == nullptr) == (in_col_shift == nullptr), "in_col_scale and in_col_shift must both be given or both be null");
    KAGNN_CHECK_ARG(!in_col_scale || x_dtype == KAGNN_DTYPE_F32, "the column affine of the gathered matrix needs fp32 rows");
    KAGNN_CHECK_ARG((col_mean == nullptr) == (col_m2 == nullptr), "col_mean and col_m2 must both be given or both be null");
    size_t need_f = 0, need_b = 0;
    int rc = kagnn_gin_kan_layer_workspace_bytes(N, L, widths, G, K, mode, num_hub_seg, 0, &need_f, &need_b);
    if (rc) return rc;
    KAGNN_CHECK_ARG(workspace && workspace_bytes >= need_f, "workspace too small (kagnn_gin_kan_layer_workspace_bytes)");
    if (N == 0) return KAGNN_OK;
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    const size_t hub_b = al256z(aggregate_bf16_ws_bytes(num_hub_seg, widths[0]));
    // The aggregation fused INTO the first KANLinear (one kernel, north_star's producer -> consumer form) for narrow first
    // layers (<= 32 features: the per-rank slices of the feature-sharded layer), split precision, fp32 rows: KAGNN_FUSE_AGG=1.
    // Off by default -- bit-identical to the two launches (tests/test_gpu_models.py) but not faster: a forward tile pays the
    // gather's two dependent round trips with 2-3 waves per SIMD to hide them, the stand-alone kernel has 8 (N = 1M, E = 10M,
    // layer forward: 0.60 vs 0.56 ms at 8 input features, 1.06 vs 0.62 at 32; profiles/r03_experiments.md).
    const char* fuse_e = getenv("KAGNN_FUSE_AGG");
    const bool fuse_env = fuse_e != nullptr && atoi(fuse_e) != 0;
    const bool fuse = fuse_env && !gine && !in_col_scale && x_dtype == KAGNN_DTYPE_F32 && mode == KAGNN_PREC_SPLIT && !(L == 1 && col_mean) &&
                      use_sparse_fwd(widths[0], widths[1], G, K, mode) &&
                      kan_sparse_fwd_agg_ok(static_cast<const float*>(x), ldx, N, widths[0], widths[1], G, K);
    // 1. h0 = self_scale * x_i + sum_{j -> i} x_j      (GINE: sum_{j -> i} relu(x_j + e_ij))
    if (gine)
        rc = kagnn_aggregate_gine(static_cast<const float*>(x), ldx, gine->ea, gine->lde, acts[0], widths[0], rowptr, col, gine->perm, N,
                                  widths[0], self_scale, stream);
    else if (fuse)
        rc = KAGNN_OK;                       // (produced by the first forward kernel, step 3)
    else if (x_dtype == KAGNN_DTYPE_BF16)
        rc = kagnn_aggregate_sum_bf16(x, ldx, acts[0], widths[0], KAGNN_DTYPE_F32, rowptr, col, nullptr, N, widths[0], self_scale,
                                      nullptr, nullptr, nullptr, 0, hub_seg, num_hub_seg, hub_threshold, ws, hub_b, stream);
    else if (in_col_scale)
        rc = kagnn_aggregate_sum_affine(static_cast<const float*>(x), ldx, acts[0], widths[0], rowptr, col, N, widths[0], self_scale,
                                        in_col_scale, in_col_shift, hub_seg, num_hub_seg, hub_threshold, nullptr, 0, ws, hub_b, stream);
    else
        rc = kagnn_aggregate_sum(static_cast<const float*>(x), ldx, acts[0], widths[0], rowptr, col, nullptr, N, widths[0],
                                 self_scale, nullptr, nullptr, nullptr, 0, hub_seg, num_hub_seg, hub_threshold, ws, hub_b, stream);
    if (rc) return rc;
    // 2. weight packs: one launch for the whole chain where the shapes allow it
    const bool prepacked = gine && gine->prepacked;
    bool batched = L >= 2 && !prepacked;
    int in_[8], out_[8];
    for (int l = 0; l < L; ++l) {
        in_[l] = widths[l]; out_[l] = widths[l + 1];
        batched = batched && use_split_dx(in_[l], out_[l], G, K, mode) && use_sparse_fwd(in_[l], out_[l], G, K, mode) &&
                  kan_fused_pack_ok(in_[l], out_[l], G + K);
    }
    if (batched) {
        rc = kagnn_kan_pack_batch(L, bw, sw, sc, in_, out_, G, K, mode, pack_fwd, pack_dx, stream);
        if (rc) return rc;
    } else if (!prepacked) {
        for (int l = 0; l < L; ++l) {
            rc = kagnn_kan_pack(bw[l], sw[l], sc ? sc[l] : nullptr, in_[l], out_[l], G, K, mode, pack_fwd[l], pack_dx[l], stream);
            if (rc) return rc;
        }
    }
    // 3. the chain
    for (int l = 0; l < L; ++l) {
        if (l == 0 && fuse) {
            const size_t fw_b = need_f - 256 - hub_b - al256z(kan_sparse_fwd_agg_ws_bytes(num_hub_seg, widths[0], widths[1]));
            KAGNN_STAGE_AS("kagnn_kan_linear_fwd+aggregate_sum (one kernel)", stream);
            rc = kan_sparse_fwd_agg(static_cast<const float*>(x), ldx, N, rowptr, col, hub_seg, num_hub_seg, hub_threshold, self_scale,
                                    knots, in_[0], out_[0], G, K, pack_fwd[0], acts[0], in_[0], acts[1], out_[0],
                                    ws + hub_b + fw_b, need_f - hub_b - fw_b, as_stream(stream));
            if (rc) return rc;
            continue;
        }
        if (l == L - 1 && col_mean)          // the convolution's output: its column moments for the norm that follows
            rc = kagnn_kan_linear_fwd_moments(acts[l], in_[l], N, knots, in_[l], out_[l], G, K, mode, pack_fwd[l], acts[l + 1],
                                              out_[l], col_mean, col_m2, ws + hub_b, need_f - hub_b, stream);
        else
            rc = kagnn_kan_linear_fwd(acts[l], in_[l], N, knots, in_[l], out_[l], G, K, mode, pack_fwd[l], acts[l + 1], out_[l],
                                      ws + hub_b, need_f - hub_b, stream);
        if (rc) return rc;
    }
    return KAGNN_OK;
}

int kagnn_gin_kan_layer_fwd(const void* x, int32_t x_dtype, int64_t ldx, int64_t N, const int32_t* rowptr, const int32_t* col,
                            const int32_t* hub_seg, int64_t num_hub_seg, int32_t hub_threshold, float self_scale,
                            int32_t L, const int32_t* widths, const float* const* bw, const float* const* sw,
                            const float* const* sc, const float* knots, int32_t G, int32_t K, int32_t mode,
                            float* const* acts, void* const* pack_fwd, void* const* pack_dx, float* col_mean,
                            float* col_m2, void* workspace, size_t workspace_bytes, void* stream) {
    ModeScope mode_scope_(mode);
    return layer_fwd_impl(x, x_dtype, ldx, N, rowptr, col, hub_seg, num_hub_seg, hub_threshold, self_scale, nullptr, nullptr, L, widths,
                          bw, sw, sc, knots, G, K, mode, acts, pack_fwd, pack_dx, col_mean, col_m2, workspace, workspace_bytes, stream, __func__);
}

// The same on an input that exists only as  in_col_scale * x + in_col_shift  (the previous layer's BatchNorm1d, folded into this
// layer's aggregation: kagnn_aggregate_sum_affine); acts[0] receives the aggregate of the NORMALISED rows, as before.
int kagnn_gin_kan_layer_fwd_affine(const float* x, int64_t ldx, int64_t N, const int32_t* rowptr, const int32_t* col,
                                   const int32_t* hub_seg, int64_t num_hub_seg, int32_t hub_threshold, float self_scale,
                                   const float* in_col_scale, const float* in_col_shift,
                                   int32_t L, const int32_t* widths, const float* const* bw, const float* const* sw,
                                   const float* const* sc, const float* knots, int32_t G, int32_t K, int32_t mode,
                                   float* const* acts, void* const* pack_fwd, void* const* pack_dx, float* col_mean,
                                   float* col_m2, void* workspace, size_t workspace_bytes, void* stream) {
    ModeScope mode_scope_(mode);
    return layer_fwd_impl(x, KAGNN_DTYPE_F32, ldx, N, rowptr, col, hub_seg, num_hub_seg, hub_threshold, self_scale, in_col_scale,
                          in_col_shift, L, widths, bw, sw, sc, knots, G, K, mode, acts, pack_fwd, pack_dx, col_mean, col_m2, workspace,
                          workspace_bytes, stream, __func__);
}

// the BatchNorm1d (training mode) that follows the layer, for kagnn_gin_kan_layer_bwd_bn
struct BnStage { const float* y; int64_t ldy; const float* weight; const float* mean; const float* rstd; float* g_weight; float* g_bias; };
static size_t bn_stage_bytes(int64_t N, int out) { return al256z(bn_ws_bytes(N, out)) + al256z(4 * (size_t)((out + 63) & ~63) * sizeof(float)); }
// the statistics of the PREVIOUS norm's backward, produced by this layer's transposed aggregation (kagnn_gin_kan_layer_bwd_bn_sums):
// prev_y = that norm's input (this convolution's forward input before the folded affine), its saved mean / rstd, sums = out [2][in]
struct StatsOut { const float* y; int64_t ldy; const float* mean; const float* rstd; float* sums; };
static size_t stats_out_bytes(int64_t N, int f0, int64_t num_hub_seg_t) { return al256z(bn_stats_fold_bytes(aggregate_stats_rows(N, f0, num_hub_seg_t), f0)); }

static int layer_bwd_impl(const float* gy, int64_t ldgy, int64_t N, const int32_t* rowptr_t, const int32_t* col_t,
                          const int32_t* hub_seg_t, int64_t num_hub_seg_t, int32_t hub_threshold, float self_scale,
                          int32_t L, const int32_t* widths, const float* const* sw, const float* const* sc,
                          const float* knots, int32_t G, int32_t K, int32_t mode, const float* const* acts,
                          const void* const* pack_dx, void* gx, int32_t gx_dtype, int64_t ldgx, int32_t bf16_gather,
                          const float* gx_addend, int64_t ld_addend, const BnStage* bn,
                          float* const* g_bw, float* const* g_sw, float* const* g_sc, void* workspace,
                          size_t workspace_bytes, void* stream, const char* fn,
                          const float* bn_sums_in = nullptr, const StatsOut* so = nullptr, const GineStage* gine = nullptr) {
    KAGNN_CHECK_ARG(N >= 0 && L >= 1 && L <= 8 && widths && sw && acts && pack_dx && g_sw, "bad argument");
    KAGNN_CHECK_ARG(!gx_addend || (gx && gx_dtype == KAGNN_DTYPE_F32 && !bf16_gather && ld_addend >= widths[0]),
                    "gx_addend needs an fp32 gx and fp32 gather operands");
    size_t need_f = 0, need_b = 0;
    int rc = kagnn_gin_kan_layer_workspace_bytes(N, L, widths, G, K, mode, 0, num_hub_seg_t, &need_f, &need_b);
    if (rc) return rc;
    const size_t bn_b = bn ? bn_stage_bytes(N, widths[L]) : 0;
    const size_t so_b = so ? stats_out_bytes(N, widths[0], num_hub_seg_t) : 0;
    KAGNN_CHECK_ARG(!so || (so->y && so->mean && so->rstd && so->sums && so->ldy >= widths[0] && gx && gx_dtype == KAGNN_DTYPE_F32 && !bf16_gather),
                    "the previous norm's statistics need its input, mean, rstd and an fp32 gx");
    KAGNN_CHECK_ARG(!bn_sums_in || bn, "bn_sums belongs to the BatchNorm stage");
    if (!(workspace && workspace_bytes >= need_b + bn_b + so_b))
        return fail(KAGNN_ERR_ARG, bn ? "%s: workspace too small (kagnn_gin_kan_layer_workspace_bytes + kagnn_gin_kan_layer_bwd_bn_workspace_bytes)"
                                      : "%s: workspace too small (kagnn_gin_kan_layer_workspace_bytes)", fn);
    if (N == 0) return KAGNN_OK;
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    int wmax = 0;
    size_t dwb = 0;
    for (int l = 0; l < L; ++l) {
        wmax = widths[l] > wmax ? widths[l] : wmax;
        size_t b = 0;
        kagnn_kan_bwd_weight_workspace_bytes(N, widths[l], widths[l + 1], G, K, mode, &b);
        dwb = b > dwb ? b : dwb;
    }
    const size_t hub_b = al256z(aggregate_bf16_ws_bytes(num_hub_seg_t, widths[0])), dw_b = al256z(dwb);
    const size_t g_b = al256z((size_t)N * wmax * sizeof(float));
    unsigned char* gbuf[2] = {ws + hub_b + dw_b, ws + hub_b + dw_b + g_b};
    // where layer l's weight gradient keeps its row slabs: the shared area -- or, inside a stack call that reduces the slabs of all
    // its layers in one launch at the end (DwDefer, common.h), a piece of that call's arena
    auto dw_area = [&](int l, unsigned char*& area, size_t& bytes) {
        area = ws + hub_b; bytes = dw_b;
        kagnn::DwDefer* d = kagnn::g_dw_defer;
        if (d == nullptr) return;
        size_t b = 0;
        if (kagnn_kan_bwd_weight_workspace_bytes(N, widths[l], widths[l + 1], G, K, mode, &b) != KAGNN_OK) return;
        b = al256z(b);
        if (d->used + b <= d->arena_bytes) { area = d->arena + d->used; bytes = b; d->used += b; }
    };
    const float* g = gy;
    long ldg = ldgy;
    int cur = 0;
    bool gh0_bf16 = false;
    // the normalisation's backward: statistics pass (column sums -> g_weight, g_bias, the per-column table), then EITHER the
    // last layer's input-gradient kernel applies it to the rows it loads and leaves them for the weight gradient (no pass
    // of its own), OR -- shapes that kernel does not cover -- the stand-alone pass writes them
    bool bn_in_dx = false;
    BnBack bnb{};
    if (bn) {
        const int out = widths[L], ldt = (out + 63) & ~63;
        KAGNN_CHECK_ARG(bn->y && bn->mean && bn->rstd && bn->ldy >= out && N >= 2, "bad BatchNorm stage");
        unsigned char* bws = ws + need_b;
        float* tab = reinterpret_cast<float*>(bws + al256z(bn_ws_bytes(N, out)));
        bnb = BnBack{bn->y, (long)bn->ldy, tab, ldt, reinterpret_cast<float*>(gbuf[1]), (long)out};
        const int in = widths[L - 1];
        bn_in_dx = mode == KAGNN_PREC_SPLIT && use_split_dx(in, out, G, K, mode) && out <= wmax && !(L == 1 && (gx == nullptr || bf16_gather)) &&
                   fits32(N, ldg) && kan_split_dx_bn_ok(ldg, in, out, G, K, bnb, g);
        if (bn_in_dx && bn_sums_in) {        // the column sums came with the gradient (the aggregation that produced g left them)
            KAGNN_STAGE_AS("kagnn_batchnorm_bwd statistics given (in ..._layer_bwd_bn)", stream);
            rc = bn_bwd_stats_given(bn_sums_in, N, out, bn->weight, bn->mean, bn->rstd, bn->g_weight, bn->g_bias, tab, ldt, as_stream(stream));
            if (rc) return rc;
        } else if (bn_in_dx) {
            KAGNN_STAGE_AS("kagnn_batchnorm_bwd statistics (in ..._layer_bwd_bn)", stream);
            rc = bn_bwd_stats(bn->y, bn->ldy, g, ldg, N, out, bn->weight, bn->mean, bn->rstd, bn->g_weight, bn->g_bias, tab, ldt, bws,
                              bn_ws_bytes(N, out), as_stream(stream));
            if (rc) return rc;
        } else {
            // (out may exceed the chain's widest INPUT, which sizes the ping-pong matrices: then the stage's own matrix is needed)
            if (out > wmax) return fail(KAGNN_ERR_UNSUPPORTED, "%s: a BatchNorm stage wider than every layer input is not covered", fn);
            KAGNN_STAGE_AS("kagnn_batchnorm_bwd", stream);
            rc = bn_bwd(bn->y, bn->ldy, g, ldg, N, out, bn->weight, bn->mean, bn->rstd, 1, 0.0f, 0ULL, reinterpret_cast<float*>(gbuf[1]), out,
                        bn->g_weight, bn->g_bias, bws, bn_ws_bytes(N, out), as_stream(stream));
            if (rc) return rc;
            g = reinterpret_cast<const float*>(gbuf[1]); ldg = out;
        }
    }
    for (int l = L - 1; l >= 0; --l) {
        const int in = widths[l], out = widths[l + 1];
        const bool fused_bn = bn_in_dx && l == L - 1;
        if (fused_bn) {            // input gradient FIRST: it produces the normalised-backward rows the weight gradient reads
            {
                KAGNN_STAGE_AS("kagnn_kan_linear_bwd_input", stream);       // (+ the norm's element-wise backward on the rows it loads)
                rc = kan_split_dx_bn(acts[l], in, g, ldg, N, knots, in, out, G, K, pack_dx[l], reinterpret_cast<float*>(gbuf[0]), in, bnb,
                                     as_stream(stream));
            }
            if (rc) return rc;
            unsigned char* dwa; size_t dwn;
            dw_area(l, dwa, dwn);
            rc = kagnn_kan_linear_bwd_weight(acts[l], in, bnb.gy_out, bnb.ldo, N, knots, in, out, G, K, mode, sw[l], sc ? sc[l] : nullptr,
                                             g_bw ? g_bw[l] : nullptr, g_sw[l], g_sc ? g_sc[l] : nullptr, dwa, dwn, stream);
            if (rc) return rc;
            if (l == 0 && gx == nullptr) break;
            g = reinterpret_cast<const float*>(gbuf[0]); ldg = in; cur = 1;
            continue;
        }
        unsigned char* dwa; size_t dwn;
        dw_area(l, dwa, dwn);
        rc = kagnn_kan_linear_bwd_weight(acts[l], in, g, ldg, N, knots, in, out, G, K, mode, sw[l], sc ? sc[l] : nullptr,
                                         g_bw ? g_bw[l] : nullptr, g_sw[l], g_sc ? g_sc[l] : nullptr, dwa, dwn, stream);
        if (rc) return rc;
        if (l == 0 && gx == nullptr) break;
        // the gathered matrix of the transposed aggregation leaves the dX kernel as bf16 when the mode asks for it
        const bool b16 = l == 0 && bf16_gather && mode == KAGNN_PREC_SPLIT && K == 3 && G + K <= 8 && out <= 128 && in % 8 == 0 &&
                         in <= 512 /* the bf16 aggregation's row limit (aggregate_bf16_ok): wider first layers keep fp32 rows */ &&
                         use_split_dx(in, out, G, K, mode);
        // (a stand-alone BatchNorm pass left its rows in gbuf[1]: the first input gradient then writes gbuf[0])
        rc = kagnn_kan_linear_bwd_input(acts[l], in, g, ldg, N, knots, in, out, G, K, mode, pack_dx[l], gbuf[cur], in,
                                        b16 ? KAGNN_DTYPE_BF16 : KAGNN_DTYPE_F32, stream);
        if (rc) return rc;
        g = reinterpret_cast<const float*>(gbuf[cur]); ldg = in; cur ^= 1;
        gh0_bf16 = b16;
    }
    if (gx == nullptr) return KAGNN_OK;
    const int f0 = widths[0];
    if (gine)        // GINE: gradient of the relu(x_j + e_ij) messages on the transposed structure -> gx and the edge-attribute gradient
        return gine_bwd(gine->x, gine->ldx, gine->ea, gine->lde, g, ldg, static_cast<float*>(gx), ldgx, gine->g_ea, gine->ldge,
                        rowptr_t, col_t, gine->perm, N, f0, self_scale, as_stream(stream), gine->accumulate_g_ea);
    if (gh0_bf16 || gx_dtype == KAGNN_DTYPE_BF16) {
        const void* src = g;
        if (!gh0_bf16) {                          // fp32 d loss / d h0 but a bf16 result wanted: convert, then the bf16 kernel
            rc = kagnn_rows_to_bf16(g, ldg, gbuf[cur], f0, N, f0, stream);
            if (rc) return rc;
            src = gbuf[cur];
        }
        return kagnn_aggregate_sum_bf16(src, f0, gx, ldgx, gx_dtype, rowptr_t, col_t, nullptr, N, f0, self_scale, nullptr, nullptr,
                                        nullptr, 0, hub_seg_t, num_hub_seg_t, hub_threshold, ws, hub_b, stream);
    }
    if (so) {       // the transposed aggregation also leaves the column statistics of gx for the previous norm's backward
        AggArgs a{g, ldg, static_cast<float*>(gx), ldgx, rowptr_t, col_t, nullptr, N, f0, self_scale, nullptr, nullptr, nullptr,
                  0, hub_threshold > 0 ? hub_threshold : 0x7fffffff, gx_addend, ld_addend};
        float* partial = reinterpret_cast<float*>(ws + need_b + bn_b);
        a.st_y = so->y; a.st_ldy = so->ldy; a.st_mean = so->mean; a.st_rstd = so->rstd; a.st_partial = partial;
        KAGNN_CHECK_ARG(ldg >= f0 && ldgx >= f0 && (!gx_addend || ld_addend >= f0), "leading dimension smaller than num_feat");
        if (!aggregate_stats_ok(a)) return fail(KAGNN_ERR_UNSUPPORTED, "%s: the previous norm's statistics need 17..256 input features in 16-byte aligned fp32 rows", fn);
        {
            KAGNN_STAGE_AS("kagnn_aggregate_sum", stream);
            rc = aggregate_sum(a, hub_seg_t, num_hub_seg_t, reinterpret_cast<float*>(ws), hub_b, as_stream(stream));
            if (rc) return rc;
        }
        KAGNN_STAGE_AS("kagnn_batchnorm_bwd statistics fold", stream);
        const bool hubs = num_hub_seg_t > 0 && hub_seg_t != nullptr && hub_threshold > 0;
        return bn_sums_from_partials(partial, aggregate_stats_rows(N, f0, hubs ? num_hub_seg_t : 0), f0, so->sums, as_stream(stream));
    }
    return kagnn_aggregate_sum_add(g, ldg, static_cast<float*>(gx), ldgx, rowptr_t, col_t, nullptr, N, f0, self_scale, nullptr,
                                   nullptr, nullptr, 0, hub_seg_t, num_hub_seg_t, hub_threshold, gx_addend, ld_addend, ws, hub_b, stream);
}

// gx_addend (optional, fp32 [N, widths[0]]): gx = <the layer's input gradient> + gx_addend, added inside the transposed
// aggregation's epilogue -- the skip-concat models hand the read-out's gradient of the same activation in here instead of
// letting the tape sum the two in a pass of its own (reference node_classification_clean/models.py:196-202)
int kagnn_gin_kan_layer_bwd_add(const float* gy, int64_t ldgy, int64_t N, const int32_t* rowptr_t, const int32_t* col_t,
                                const int32_t* hub_seg_t, int64_t num_hub_seg_t, int32_t hub_threshold, float self_scale,
                                int32_t L, const int32_t* widths, const float* const* sw, const float* const* sc,
                                const float* knots, int32_t G, int32_t K, int32_t mode, const float* const* acts,
                                const void* const* pack_dx, void* gx, int32_t gx_dtype, int64_t ldgx, int32_t bf16_gather,
                                const float* gx_addend, int64_t ld_addend,
                                float* const* g_bw, float* const* g_sw, float* const* g_sc, void* workspace,
                                size_t workspace_bytes, void* stream) {
    ModeScope mode_scope_(mode);
    return layer_bwd_impl(gy, ldgy, N, rowptr_t, col_t, hub_seg_t, num_hub_seg_t, hub_threshold, self_scale, L, widths, sw, sc, knots, G, K,
                          mode, acts, pack_dx, gx, gx_dtype, ldgx, bf16_gather, gx_addend, ld_addend, nullptr, g_bw, g_sw, g_sc, workspace,
                          workspace_bytes, stream, __func__);
}

// The backward of  BatchNorm1d(KAN(aggregate(x)))  in training mode -- the convolution plus the norm that follows it in
// every node model (reference node_classification_clean/models.py:198-200) -- given g = d loss / d (norm output):
// the norm's statistics pass (-> g_bn_weight, g_bn_bias), then the chain's backward with the norm's element-wise backward
// applied INSIDE the last layer's input-gradient kernel (no normalisation-backward pass over [N, out]), then the transposed
// aggregation (+ gx_addend).  y = the norm's input (the chain's output), bn_mean / bn_rstd = the statistics its forward saved.
// Workspace: kagnn_gin_kan_layer_workspace_bytes' backward size + kagnn_gin_kan_layer_bwd_bn_workspace_bytes.
int kagnn_gin_kan_layer_bwd_bn_workspace_bytes(int64_t N, int32_t out, size_t* bytes) {
    KAGNN_CHECK_ARG(N >= 0 && out >= 1 && bytes, "bad argument");
    *bytes = bn_stage_bytes(N, out);
    return KAGNN_OK;
}

int kagnn_gin_kan_layer_bwd_bn(const float* g, int64_t ldg, const float* y, int64_t ldy, const float* bn_weight,
                               const float* bn_mean, const float* bn_rstd, float* g_bn_weight, float* g_bn_bias,
                               int64_t N, const int32_t* rowptr_t, const int32_t* col_t,
                               const int32_t* hub_seg_t, int64_t num_hub_seg_t, int32_t hub_threshold, float self_scale,
                               int32_t L, const int32_t* widths, const float* const* sw, const float* const* sc,
                               const float* knots, int32_t G, int32_t K, int32_t mode, const float* const* acts,
                               const void* const* pack_dx, void* gx, int32_t gx_dtype, int64_t ldgx, int32_t bf16_gather,
                               const float* gx_addend, int64_t ld_addend,
                               float* const* g_bw, float* const* g_sw, float* const* g_sc, void* workspace,
                               size_t workspace_bytes, void* stream) {
    ModeScope mode_scope_(mode);
    const BnStage bn{y, ldy, bn_weight, bn_mean, bn_rstd, g_bn_weight, g_bn_bias};
    return layer_bwd_impl(g, ldg, N, rowptr_t, col_t, hub_seg_t, num_hub_seg_t, hub_threshold, self_scale, L, widths, sw, sc, knots, G, K,
                          mode, acts, pack_dx, gx, gx_dtype, ldgx, bf16_gather, gx_addend, ld_addend, &bn, g_bw, g_sw, g_sc, workspace,
                          workspace_bytes, stream, __func__);
}

// kagnn_gin_kan_layer_bwd_bn with the norms' backward STATISTICS travelling with the gradients (round 4): in the node models the
// gradient g arriving at layer l's norm is produced by layer l+1's transposed aggregation (+ the skip gradient it adds), so
//   * prev_y / prev_mean / prev_rstd / prev_sums (all or none): this call's transposed aggregation ALSO leaves
//     prev_sums[0][in] = sum_n gx, prev_sums[1][in] = sum_n gx * xhat_prev  (xhat_prev = (prev_y - prev_mean) * prev_rstd; prev_y is
//     the previous norm's input = this convolution's input before the folded affine) -- from partial sums in the row kernel's
//     epilogue, folded in a fixed order;
//   * bn_sums (or NULL): [2][out] sums for THIS norm made that way by the next layer's call -- the statistics pass over g and y is
//     skipped (only when the norm's element-wise backward runs inside the input-gradient kernel; otherwise ignored).
// Extra workspace behind kagnn_gin_kan_layer_bwd_bn's: kagnn_gin_kan_layer_bwd_bn_sums_workspace_bytes (0 without prev_sums).
int kagnn_gin_kan_layer_bwd_bn_sums_workspace_bytes(int64_t N, int32_t in_features, int64_t num_hub_seg_t, size_t* bytes) {
    KAGNN_CHECK_ARG(N >= 0 && in_features >= 1 && num_hub_seg_t >= 0 && bytes, "bad argument");
    *bytes = stats_out_bytes(N, in_features, num_hub_seg_t);
    return KAGNN_OK;
}

int kagnn_gin_kan_layer_bwd_bn_sums(const float* g, int64_t ldg, const float* y, int64_t ldy, const float* bn_weight,
                                    const float* bn_mean, const float* bn_rstd, float* g_bn_weight, float* g_bn_bias,
                                    const float* bn_sums,
                                    const float* prev_y, int64_t ld_prev_y, const float* prev_mean, const float* prev_rstd,
                                    float* prev_sums,
                                    int64_t N, const int32_t* rowptr_t, const int32_t* col_t,
                                    const int32_t* hub_seg_t, int64_t num_hub_seg_t, int32_t hub_threshold, float self_scale,
                                    int32_t L, const int32_t* widths, const float* const* sw, const float* const* sc,
                                    const float* knots, int32_t G, int32_t K, int32_t mode, const float* const* acts,
                                    const void* const* pack_dx, void* gx, int32_t gx_dtype, int64_t ldgx, int32_t bf16_gather,
                                    const float* gx_addend, int64_t ld_addend,
                                    float* const* g_bw, float* const* g_sw, float* const* g_sc, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    ModeScope mode_scope_(mode);
    const BnStage bn{y, ldy, bn_weight, bn_mean, bn_rstd, g_bn_weight, g_bn_bias};
    const StatsOut so{prev_y, ld_prev_y, prev_mean, prev_rstd, prev_sums};
    KAGNN_CHECK_ARG((prev_sums == nullptr) == (prev_y == nullptr), "prev_y and prev_sums come together");
    return layer_bwd_impl(g, ldg, N, rowptr_t, col_t, hub_seg_t, num_hub_seg_t, hub_threshold, self_scale, L, widths, sw, sc, knots, G, K,
                          mode, acts, pack_dx, gx, gx_dtype, ldgx, bf16_gather, gx_addend, ld_addend, &bn, g_bw, g_sw, g_sc, workspace,
                          workspace_bytes, stream, __func__, bn_sums, prev_sums ? &so : nullptr);
}

int kagnn_gin_kan_layer_bwd(const float* gy, int64_t ldgy, int64_t N, const int32_t* rowptr_t, const int32_t* col_t,
                            const int32_t* hub_seg_t, int64_t num_hub_seg_t, int32_t hub_threshold, float self_scale,
                            int32_t L, const int32_t* widths, const float* const* sw, const float* const* sc,
                            const float* knots, int32_t G, int32_t K, int32_t mode, const float* const* acts,
                            const void* const* pack_dx, void* gx, int32_t gx_dtype, int64_t ldgx, int32_t bf16_gather,
                            float* const* g_bw, float* const* g_sw, float* const* g_sc, void* workspace,
                            size_t workspace_bytes, void* stream) {
    ModeScope mode_scope_(mode);
    return kagnn_gin_kan_layer_bwd_add(gy, ldgy, N, rowptr_t, col_t, hub_seg_t, num_hub_seg_t, hub_threshold, self_scale, L, widths, sw,
                                       sc, knots, G, K, mode, acts, pack_dx, gx, gx_dtype, ldgx, bf16_gather, nullptr, 0, g_bw, g_sw,
                                       g_sc, workspace, workspace_bytes, stream);
}

// ---- the same ONE call per convolution each way around GINE message passing (BASELINE config 4: the ZINC-shaped mini-batch step is
// host- and launch-bound, graph_regression/models.py:107-119, optuna_zinc.py:56-66).  Forward = kagnn_aggregate_gine + one pack
// launch + the chain (column moments of the output for the BatchNorm1d that follows, when col_mean is given); backward = [the norm's
// statistics pass and its element-wise backward inside the last input-gradient kernel, when bn_y is given] + the chain's
// dW / dX + kagnn_aggregate_gine_bwd.  Same kernels, same order, same bits as the per-operation composition.  fp32 rows; the
// structure arrays are those of kagnn_csr_build (forward: by destination; backward: by source), small graphs: no hub segments.
// Workspace: kagnn_gin_kan_layer_workspace_bytes (num_hub_seg = 0) [+ kagnn_gin_kan_layer_bwd_bn_workspace_bytes].
int kagnn_gine_kan_layer_fwd(const float* x, int64_t ldx, const float* edge_attr, int64_t lde, int64_t N, const int32_t* rowptr,
                             const int32_t* col, const int32_t* perm, float self_scale,
                             int32_t L, const int32_t* widths, const float* const* bw, const float* const* sw,
                             const float* const* sc, const float* knots, int32_t G, int32_t K, int32_t mode,
                             float* const* acts, void* const* pack_fwd, void* const* pack_dx, float* col_mean,
                             float* col_m2, void* workspace, size_t workspace_bytes, void* stream) {
    ModeScope mode_scope_(mode);
    KAGNN_CHECK_ARG(N == 0 || (x && edge_attr && perm && widths && ldx >= widths[0] && lde >= widths[0]), "null array or short leading dimension");
    const GineStage gs{x, ldx, edge_attr, lde, perm, nullptr, 0};
    return layer_fwd_impl(x, KAGNN_DTYPE_F32, ldx, N, rowptr, col, nullptr, 0, 0, self_scale, nullptr, nullptr, L, widths, bw, sw, sc,
                          knots, G, K, mode, acts, pack_fwd, pack_dx, col_mean, col_m2, workspace, workspace_bytes, stream, __func__, &gs);
}

int kagnn_gine_kan_layer_bwd(const float* g, int64_t ldg, const float* bn_y, int64_t ld_bn_y, const float* bn_weight,
                             const float* bn_mean, const float* bn_rstd, float* g_bn_weight, float* g_bn_bias,
                             const float* x, int64_t ldx, const float* edge_attr, int64_t lde, int64_t N,
                             const int32_t* rowptr_t, const int32_t* col_t, const int32_t* perm_t, float self_scale,
                             int32_t L, const int32_t* widths, const float* const* sw, const float* const* sc,
                             const float* knots, int32_t G, int32_t K, int32_t mode, const float* const* acts,
                             const void* const* pack_dx, float* gx, int64_t ldgx, float* g_edge_attr, int64_t ldge,
                             float* const* g_bw, float* const* g_sw, float* const* g_sc, void* workspace,
                             size_t workspace_bytes, void* stream) {
    ModeScope mode_scope_(mode);
    KAGNN_CHECK_ARG(N == 0 || (x && edge_attr && perm_t && gx && widths && ldx >= widths[0] && lde >= widths[0] && ldgx >= widths[0]),
                    "null array or short leading dimension (gx is required: the edge-attribute gradient comes out of the same kernel)");
    KAGNN_CHECK_ARG(!g_edge_attr || ldge >= widths[0], "short leading dimension of g_edge_attr");
    const GineStage gs{x, ldx, edge_attr, lde, perm_t, g_edge_attr, ldge};
    const BnStage bn{bn_y, ld_bn_y, bn_weight, bn_mean, bn_rstd, g_bn_weight, g_bn_bias};
    return layer_bwd_impl(g, ldg, N, rowptr_t, col_t, nullptr, 0, 0, self_scale, L, widths, sw, sc, knots, G, K, mode, acts, pack_dx,
                          gx, KAGNN_DTYPE_F32, ldgx, 0, nullptr, 0, bn_y ? &bn : nullptr, g_bw, g_sw, g_sc, workspace, workspace_bytes,
                          stream, __func__, nullptr, nullptr, &gs);
}

// ---- the WHOLE message-passing stack of a graph-level model in one call each way (round 5): nconv x {GINE convolution around a KAN
// chain of L layers -> training-mode BatchNorm1d}, every chain hidden -> ... -> hidden with the same widths (reference
// graph_regression/models.py:107-119: `for i in range(n_layers): x = self.bn[i](self.conv[i](x, edge_index, edge_attr))`).  On a
// 256-molecule mini-batch a convolution is ~100 us of device work; as one tape node per convolution the HOST spent ~100 us per node
// each way on argument marshalling and allocations -- the step was host-bound at twice its device time.  Forward: ONE pack launch for
// all nconv * L layers, then per convolution kagnn_aggregate_gine, the chain (column moments from the last kernel) and the
// normalising pass -> h[i].  Backward: per convolution (last first) the norm's statistics pass, its element-wise backward inside
// the last input-gradient kernel, dW / dX, kagnn_aggregate_gine_bwd; the edge-attribute gradients of the nconv convolutions add
// up in g_edge_attr in place.  Same kernels and orders as nconv calls of kagnn_gine_kan_layer_fwd / _bwd: same bits.
// Array arguments: widths [L + 1] (widths[0] == widths[L]); per layer, convolution-major [nconv * L]: base_weight, spline_weight,
// spline_scaler, pack_fwd, pack_dx, g_*; acts [nconv * (L + 1)]; per convolution [nconv]: self_scale / momentum / eps (HOST floats),
// bn_weight, bn_bias, running_mean, running_var (device; the last two NULL arrays or NULL entries: no running statistics), h,
// save_mean, save_rstd, g_bn_weight, g_bn_bias.  Workspace: kagnn_gine_kan_stack_workspace_bytes.
// the backward's arena of weight-gradient row slabs: one area per layer of the stack, reduced in ONE launch at the end of the call
static size_t gine_stack_dw_arena_bytes(int64_t N, int nconv, int L, const int32_t* widths, int G, int K, int mode) {
    if (nconv * L > kagnn::kDwDeferMax) return 0;
    size_t a = 0;
    for (int l = 0; l < L; ++l) {
        size_t b = 0;
        if (kagnn_kan_bwd_weight_workspace_bytes(N, widths[l], widths[l + 1], G, K, mode, &b) != KAGNN_OK) return 0;
        a += al256z(b);
    }
    return a * (size_t)nconv;
}

int kagnn_gine_kan_stack_workspace_bytes(int64_t N, int32_t nconv, int32_t L, const int32_t* widths, int32_t G, int32_t K, int32_t mode,
                                         size_t* fwd_bytes, size_t* bwd_bytes) {
    ModeScope mode_scope_(mode);
    KAGNN_CHECK_ARG(N >= 0 && nconv >= 1 && L >= 1 && L <= 8 && widths && fwd_bytes && bwd_bytes, "bad argument");
    KAGNN_CHECK_ARG(widths[0] == widths[L], "every convolution of the stack maps hidden -> hidden");
    size_t f = 0, b = 0, bn = 0, bw = 0;
    int rc = kagnn_gin_kan_layer_workspace_bytes(N, L, widths, G, K, mode, 0, 0, &f, &b);
    if (rc) return rc;
    rc = kagnn_batchnorm_workspace_bytes(N, widths[L], &bn); if (rc) return rc;
    rc = kagnn_gin_kan_layer_bwd_bn_workspace_bytes(N, widths[L], &bw); if (rc) return rc;
    *fwd_bytes = al256z(f) + al256z(bn) + al256z(2 * (size_t)widths[L] * sizeof(float)) + 256;
    *bwd_bytes = al256z(b + bw) + 2 * al256z((size_t)N * widths[0] * sizeof(float)) + gine_stack_dw_arena_bytes(N, nconv, L, widths, G, K, mode) + 256;
    return KAGNN_OK;
}

static bool mom_defer_enabled() {             // OPT-IN (KAGNN_MOM_DEFER=1): measured slower on the device, see below; read per call for the test
    const char* e = getenv("KAGNN_MOM_DEFER");
    return e != nullptr && atoi(e) != 0;
}

int kagnn_gine_kan_stack_fwd(const float* x, int64_t ldx, const float* edge_attr, int64_t lde, int64_t N, const int32_t* rowptr,
                             const int32_t* col, const int32_t* perm, const float* self_scale, int32_t nconv,
                             int32_t L, const int32_t* widths, const float* const* bw, const float* const* sw,
                             const float* const* sc, const float* knots, int32_t G, int32_t K, int32_t mode,
                             float* const* acts, void* const* pack_fwd, void* const* pack_dx,
                             const float* const* bn_weight, const float* const* bn_bias, float* const* running_mean,
                             float* const* running_var, const float* momentum, const float* eps,
                             float* const* h, float* const* save_mean, float* const* save_rstd,
                             void* workspace, size_t workspace_bytes, void* stream) {
    ModeScope mode_scope_(mode);
    KAGNN_CHECK_ARG(nconv >= 1 && L >= 1 && L <= 8 && widths && self_scale && bw && sw && acts && pack_fwd && pack_dx && bn_weight && bn_bias &&
                    momentum && eps && h && save_mean && save_rstd, "null array");
    KAGNN_CHECK_ARG(widths[0] == widths[L] && N >= 2, "hidden -> hidden chains, at least two rows (batch statistics)");
    size_t need_f = 0, need_b = 0, lf = 0, lb = 0, bnb = 0;
    int rc = kagnn_gine_kan_stack_workspace_bytes(N, nconv, L, widths, G, K, mode, &need_f, &need_b);
    if (rc) return rc;
    KAGNN_CHECK_ARG(workspace && workspace_bytes >= need_f, "workspace too small (kagnn_gine_kan_stack_workspace_bytes)");
    rc = kagnn_gin_kan_layer_workspace_bytes(N, L, widths, G, K, mode, 0, 0, &lf, &lb); if (rc) return rc;
    rc = kagnn_batchnorm_workspace_bytes(N, widths[L], &bnb); if (rc) return rc;
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    unsigned char* ws_bn = ws + al256z(lf);
    float* mom = reinterpret_cast<float*>(ws_bn + al256z(bnb));
    const int H = widths[L];
    // one pack launch for the whole stack where the shapes allow it (<= 16 layers on the sparse-forward / split path)
    int in_[16], out_[16];
    bool batch = nconv * L <= 16;
    for (int k = 0; k < nconv * L && batch; ++k) {
        in_[k] = widths[k % L]; out_[k] = widths[k % L + 1];
        batch = use_split_dx(in_[k], out_[k], G, K, mode) && use_sparse_fwd(in_[k], out_[k], G, K, mode) && kan_fused_pack_ok(in_[k], out_[k], G + K);
    }
    if (batch && !kagnn::g_stack_prepacked) { rc = kagnn_kan_pack_batch(nconv * L, bw, sw, sc, in_, out_, G, K, mode, pack_fwd, pack_dx, stream); if (rc) return rc; }
    const float* in = x;
    int64_t ldin = ldx;
    for (int i = 0; i < nconv; ++i) {
        GineStage gs{in, ldin, edge_attr, lde, perm, nullptr, 0};
        gs.prepacked = batch ? 1 : 0;
        // (round 6, opt-in: KAGNN_MOM_DEFER=1) the last forward kernel leaves its <= 32 per-workgroup moment rows where they are and the
        // norm's apply kernel folds them: no moments_finish launch, same merge order, same bits.  One launch fewer per convolution, but
        // every one of the apply kernel's ~370 workgroups repeats the 24-row merge chain: 15.7 us against 5.0 + 5.1 for the two
        // launches (profiles/r06_experiments.md 3) -- and a launch costs the host ~1 us.  Off by default.
        kagnn::MomDefer md{nullptr, 0};
        {
            struct MomScope {
                kagnn::MomDefer* prev;
                explicit MomScope(kagnn::MomDefer* d) : prev(kagnn::g_mom_defer) { kagnn::g_mom_defer = d; }
                ~MomScope() { kagnn::g_mom_defer = prev; }
            } mom_scope_(mom_defer_enabled() ? &md : nullptr);
            rc = layer_fwd_impl(in, KAGNN_DTYPE_F32, ldin, N, rowptr, col, nullptr, 0, 0, self_scale[i], nullptr, nullptr, L, widths, bw + i * L,
                                sw + i * L, sc ? sc + i * L : nullptr, knots, G, K, mode, acts + i * (L + 1), pack_fwd + i * L, pack_dx + i * L,
                                mom, mom + H, ws, al256z(lf), stream, __func__, &gs);
        }
        if (rc) return rc;
        if (md.P > 0) {
            KAGNN_STAGE_AS("kagnn_batchnorm_fwd", stream);
            rc = bn_fwd_partial_moments(acts[i * (L + 1) + L], H, N, H, bn_weight[i], bn_bias[i], running_mean ? running_mean[i] : nullptr,
                                        running_var ? running_var[i] : nullptr, momentum[i], eps[i], md.partial, md.P, h[i], H,
                                        save_mean[i], save_rstd[i], as_stream(stream));
        } else {
            rc = kagnn_batchnorm_fwd(acts[i * (L + 1) + L], H, N, H, bn_weight[i], bn_bias[i], running_mean ? running_mean[i] : nullptr,
                                     running_var ? running_var[i] : nullptr, momentum[i], eps[i], 1, mom, mom + H, 0.0f, 0ULL, h[i], H,
                                     save_mean[i], save_rstd[i], ws_bn, bnb, stream);
        }
        if (rc) return rc;
        in = h[i]; ldin = H;
    }
    return KAGNN_OK;
}

int kagnn_gine_kan_stack_bwd(const float* g, int64_t ldg, const float* x, int64_t ldx, const float* edge_attr, int64_t lde, int64_t N,
                             const int32_t* rowptr_t, const int32_t* col_t, const int32_t* perm_t, const float* self_scale,
                             int32_t nconv, int32_t L, const int32_t* widths, const float* const* sw, const float* const* sc,
                             const float* knots, int32_t G, int32_t K, int32_t mode, const float* const* acts,
                             const void* const* pack_dx, const float* const* h, const float* const* bn_weight,
                             const float* const* save_mean, const float* const* save_rstd,
                             float* gx, int64_t ldgx, float* g_edge_attr, int64_t ldge, float* const* g_bn_weight, float* const* g_bn_bias,
                             float* const* g_bw, float* const* g_sw, float* const* g_sc, void* workspace, size_t workspace_bytes,
                             void* stream) {
    ModeScope mode_scope_(mode);
    KAGNN_CHECK_ARG(nconv >= 1 && L >= 1 && L <= 8 && widths && self_scale && sw && acts && pack_dx && h && bn_weight && save_mean && save_rstd &&
                    g_bn_weight && g_bn_bias && g_sw && gx && x && edge_attr, "null array");
    KAGNN_CHECK_ARG(widths[0] == widths[L] && N >= 2 && ldgx >= widths[0], "hidden -> hidden chains, at least two rows");
    size_t need_f = 0, need_b = 0, lf = 0, lb = 0, bwb = 0;
    int rc = kagnn_gine_kan_stack_workspace_bytes(N, nconv, L, widths, G, K, mode, &need_f, &need_b);
    if (rc) return rc;
    KAGNN_CHECK_ARG(workspace && workspace_bytes >= need_b, "workspace too small (kagnn_gine_kan_stack_workspace_bytes)");
    rc = kagnn_gin_kan_layer_workspace_bytes(N, L, widths, G, K, mode, 0, 0, &lf, &lb); if (rc) return rc;
    rc = kagnn_gin_kan_layer_bwd_bn_workspace_bytes(N, widths[L], &bwb); if (rc) return rc;
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    const int H = widths[0];
    const size_t gbytes = al256z((size_t)N * H * sizeof(float));
    float* pp[2] = {reinterpret_cast<float*>(ws + al256z(lb + bwb)), reinterpret_cast<float*>(ws + al256z(lb + bwb) + gbytes)};
    // every layer's row slabs in an area of their own, all nconv * L slab reductions in one launch after the last convolution
    kagnn::DwDefer defer{};
    defer.arena = ws + al256z(lb + bwb) + 2 * gbytes;
    defer.arena_bytes = gine_stack_dw_arena_bytes(N, nconv, L, widths, G, K, mode);
    struct DeferScope {
        kagnn::DwDefer* prev;
        explicit DeferScope(kagnn::DwDefer* d) : prev(kagnn::g_dw_defer) { kagnn::g_dw_defer = d; }
        ~DeferScope() { kagnn::g_dw_defer = prev; }
    } defer_scope_(defer.arena_bytes ? &defer : nullptr);
    const float* gcur = g;
    int64_t ldcur = ldg;
    for (int i = nconv - 1; i >= 0; --i) {
        const float* in = i == 0 ? x : h[i - 1];
        const int64_t ldin = i == 0 ? ldx : H;
        float* gout = i == 0 ? gx : pp[i & 1];
        const int64_t ldo = i == 0 ? ldgx : H;
        GineStage gs{in, ldin, edge_attr, lde, perm_t, g_edge_attr, ldge};
        gs.accumulate_g_ea = i < nconv - 1 ? 1 : 0;
        const BnStage bn{acts[i * (L + 1) + L], H, bn_weight[i], save_mean[i], save_rstd[i], g_bn_weight[i], g_bn_bias[i]};
        rc = layer_bwd_impl(gcur, ldcur, N, rowptr_t, col_t, nullptr, 0, 0, self_scale[i], L, widths, sw + i * L, sc ? sc + i * L : nullptr, knots,
                            G, K, mode, acts + i * (L + 1), pack_dx + i * L, gout, KAGNN_DTYPE_F32, ldo, 0, nullptr, 0, &bn,
                            g_bw ? g_bw + i * L : nullptr, g_sw + i * L, g_sc ? g_sc + i * L : nullptr, ws, al256z(lb + bwb), stream, __func__,
                            nullptr, nullptr, &gs);
        if (rc) return rc;
        gcur = gout; ldcur = ldo;
    }
    {
        KAGNN_STAGE_AS("kagnn_kan_linear_bwd_weight (slab reductions of the stack)", stream);
        return kagnn::dw_defer_flush(as_stream(stream));
    }
}

// ---------------------------------------------------------------- the whole graph-regression model per call (round 6)
// KAGIN.forward of the reference's graph_regression/models.py:107-119 and its backward as ONE library call each way: the sequence of
// this file's own entry points that kagnn_amd/graph_ops.py::_KaginModelFn runs from Python, with the same arguments in the same
// order -- the same kernels, the same bits -- minus ~17 ctypes round trips, ~45 tensor allocations and their pointer tables.
namespace {
struct KmLayout {
    // `saved`: byte offsets
    size_t x0, ea, acts, h, stats, packs, pooled, ro_act[KAGNN_MODEL_MAX_READOUT], ro_pf[KAGNN_MODEL_MAX_READOUT], ro_pd[KAGNN_MODEL_MAX_READOUT], saved_total;
    size_t fb, db;                       // one stack layer's forward / input-gradient pack, 256-aligned
    size_t csr[6];                       // rowptr, col, perm, rowptr_t, col_t, perm_t (int32) when the library builds the CSR itself
    size_t fwd_csr_ws, csr_ws_bytes;     // ... and that build's scratch inside the forward workspace
    // workspaces: byte offsets of the fixed parts, then the shared scratch of the sub-calls
    size_t fwd_scratch, fwd_total;
    size_t bwd_gy[2], bwd_gh, bwd_gx0, bwd_gea, bwd_scratch, bwd_total;
    size_t grads_floats;
    size_t g_atom[KAGNN_MODEL_MAX_TABLES], g_bond[KAGNN_MODEL_MAX_TABLES], g_bn_w[KAGNN_MODEL_MAX_CONVS], g_bn_b[KAGNN_MODEL_MAX_CONVS];
    size_t g_bw[KAGNN_MODEL_MAX_LAYERS], g_sw[KAGNN_MODEL_MAX_LAYERS], g_sc[KAGNN_MODEL_MAX_LAYERS];
    size_t g_ro_bw[KAGNN_MODEL_MAX_READOUT], g_ro_sw[KAGNN_MODEL_MAX_READOUT], g_ro_sc[KAGNN_MODEL_MAX_READOUT];   // float offsets into grads
    bool ro_batch;                       // the read-out's packs in one launch (kagnn_kan_pack_batch)
};

// a stream + two events of the library's own, per host thread and device (created on first use, never destroyed: process lifetime)
struct SideStream { hipStream_t st; hipEvent_t fork, join; };
SideStream* side_stream() {
    thread_local SideStream pool[64];
    thread_local bool made[64] = {};
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) return nullptr;
    if (!made[d]) {
        if (hipStreamCreateWithFlags(&pool[d].st, hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&pool[d].fork, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&pool[d].join, hipEventDisableTiming) != hipSuccess) return nullptr;
        made[d] = true;
    }
    return &pool[d];
}

int km_check(const kagnn_kagin_model_t* m, const char* fn) {
    if (!m) return fail(KAGNN_ERR_ARG, "%s: null model", fn);
    const bool ok = m->num_nodes >= 2 && m->num_edges >= 0 && m->num_graphs >= 1 && m->hidden >= 1 && m->hidden <= 64 &&
                    m->num_atom_tables >= 1 && m->num_atom_tables <= KAGNN_MODEL_MAX_TABLES && m->num_bond_tables >= 1 &&
                    m->num_bond_tables <= KAGNN_MODEL_MAX_TABLES && m->x_stride >= m->num_atom_tables && m->e_stride >= m->num_bond_tables &&
                    m->num_convs >= 1 && m->num_convs <= KAGNN_MODEL_MAX_CONVS && m->num_layers >= 1 && m->num_layers <= 8 &&
                    m->num_convs * m->num_layers <= KAGNN_MODEL_MAX_LAYERS && m->num_readout >= 1 && m->num_readout <= KAGNN_MODEL_MAX_READOUT &&
                    m->readout_widths[0] == m->hidden;
    if (!ok) return fail(KAGNN_ERR_ARG, "%s: sizes outside the limits of kagnn_kagin_model_t (include/kagnn_hip.h)", fn);
    return KAGNN_OK;
}

int km_layout(const kagnn_kagin_model_t* m, KmLayout& L, const char* fn) {
    int rc = km_check(m, fn);
    if (rc) return rc;
    const size_t N = (size_t)m->num_nodes, E = (size_t)m->num_edges, B = (size_t)m->num_graphs, H = (size_t)m->hidden;
    const int nconv = (int)m->num_convs, nl = (int)m->num_layers, G = (int)m->grid_size, K = (int)m->spline_order, mode = (int)m->mode;
    const int C = G + K;
    size_t fb = 0, db = 0;
    rc = kagnn_kan_pack_bytes((int)H, (int)H, G, K, mode, &fb, &db); if (rc) return rc;
    L.fb = al256z(fb); L.db = al256z(db);
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += al256z(bytes); return at; };
    L.x0 = take(N * H * 4);
    L.ea = take((E ? E : 1) * H * 4);
    L.acts = take((size_t)nconv * (nl + 1) * N * H * 4);
    L.h = take((size_t)nconv * N * H * 4);
    L.stats = take((size_t)nconv * 2 * H * 4);
    L.packs = take((size_t)nconv * nl * (L.fb + L.db));
    L.pooled = take(B * H * 4);
    const int nr = (int)m->num_readout;
    bool batch = nr >= 2 && (int)m->readout_spline_order == 3 && (int)m->readout_grid_size + 3 <= 8;
    for (int i = 0; i < nr; ++i) {
        const int fin = (int)m->readout_widths[i], fout = (int)m->readout_widths[i + 1], rm = (int)m->readout_modes[i];
        if (fin < 1 || fout < 1) return fail(KAGNN_ERR_ARG, "%s: read-out widths", fn);
        L.ro_act[i] = i == 0 ? L.pooled : take(B * (size_t)fin * 4);          // input of read-out layer i
        size_t pf = 0, pd = 0;
        rc = kagnn_kan_pack_bytes(fin, fout, (int)m->readout_grid_size, (int)m->readout_spline_order, rm, &pf, &pd); if (rc) return rc;
        L.ro_pf[i] = take(pf); L.ro_pd[i] = take(pd);
        batch = batch && rm == (int)m->readout_modes[0] && (rm == KAGNN_PREC_SPLIT || rm == KAGNN_PREC_HALF) && fout <= 64;
    }
    L.ro_batch = batch;
    for (int k = 0; k < 6; ++k) L.csr[k] = 0;
    L.csr_ws_bytes = 0;
    if (m->edge_src) {
        if (!m->edge_dst || !m->csr_flags || !kagnn_csr_small_ok((int64_t)E, (int64_t)N))
            return fail(KAGNN_ERR_ARG, "%s: edge_src needs edge_dst, csr_flags and a graph kagnn_csr_build_small covers", fn);
        for (int k = 0; k < 6; ++k) L.csr[k] = take(((k % 3 == 0) ? N + 1 : (E ? E : 1)) * sizeof(int32_t));
        rc = kagnn_csr_small_workspace_bytes((int64_t)E, &L.csr_ws_bytes); if (rc) return rc;
    }
    L.saved_total = o + 256;
    // forward workspace: the stack's, the read-out forwards' split-K scratch
    int32_t widths[9];
    for (int l = 0; l <= nl; ++l) widths[l] = (int32_t)H;
    size_t sf = 0, sb = 0;
    rc = kagnn_gine_kan_stack_workspace_bytes((int64_t)N, nconv, nl, widths, G, K, mode, &sf, &sb); if (rc) return rc;
    size_t scratch_f = sf, scratch_b = sb;
    for (int i = 0; i < nr; ++i) {
        const int fin = (int)m->readout_widths[i], fout = (int)m->readout_widths[i + 1], rm = (int)m->readout_modes[i];
        size_t a = 0, b = 0;
        rc = kagnn_kan_fwd_workspace_bytes((int64_t)B, fin, fout, (int)m->readout_grid_size, (int)m->readout_spline_order, rm, &a); if (rc) return rc;
        rc = kagnn_kan_bwd_weight_workspace_bytes((int64_t)B, fin, fout, (int)m->readout_grid_size, (int)m->readout_spline_order, rm, &b); if (rc) return rc;
        scratch_f = scratch_f > a ? scratch_f : a;
        scratch_b = scratch_b > b ? scratch_b : b;
    }
    for (int t = 0; t < (int)m->num_atom_tables; ++t) {
        size_t a = 0;
        rc = kagnn_embedding_bwd_workspace_bytes((int64_t)N, (int)m->atom_rows[t], (int)H, &a); if (rc) return rc;
        scratch_b = scratch_b > a ? scratch_b : a;
    }
    for (int t = 0; t < (int)m->num_bond_tables; ++t) {
        size_t a = 0;
        rc = kagnn_embedding_bwd_workspace_bytes((int64_t)E, (int)m->bond_rows[t], (int)H, &a); if (rc) return rc;
        scratch_b = scratch_b > a ? scratch_b : a;
    }
    L.fwd_scratch = 0; L.fwd_csr_ws = al256z(scratch_f); L.fwd_total = al256z(scratch_f) + al256z(L.csr_ws_bytes) + 256;
    size_t wmax = 1;
    for (int i = 0; i <= nr; ++i) wmax = wmax > (size_t)m->readout_widths[i] ? wmax : (size_t)m->readout_widths[i];
    o = 0;
    L.bwd_gy[0] = take(B * wmax * 4); L.bwd_gy[1] = take(B * wmax * 4);
    L.bwd_gh = take(N * H * 4); L.bwd_gx0 = take(N * H * 4); L.bwd_gea = take((E ? E : 1) * H * 4);
    L.bwd_scratch = o; L.bwd_total = o + al256z(scratch_b) + 256;
    // the flat gradient buffer (floats)
    size_t g = 0;
    for (int t = 0; t < (int)m->num_atom_tables; ++t) { L.g_atom[t] = g; g += (size_t)m->atom_rows[t] * H; }
    for (int t = 0; t < (int)m->num_bond_tables; ++t) { L.g_bond[t] = g; g += (size_t)m->bond_rows[t] * H; }
    for (int i = 0; i < nconv; ++i) {
        L.g_bn_w[i] = g; g += H; L.g_bn_b[i] = g; g += H;
        for (int l = 0; l < nl; ++l) {
            const int k = i * nl + l;
            L.g_bw[k] = g; g += H * H; L.g_sw[k] = g; g += H * H * C; L.g_sc[k] = g; g += H * H;
        }
    }
    for (int i = 0; i < nr; ++i) {
        const size_t fin = (size_t)m->readout_widths[i], fout = (size_t)m->readout_widths[i + 1];
        const size_t Cr = (size_t)(m->readout_grid_size + m->readout_spline_order);
        L.g_ro_bw[i] = g; g += fout * fin; L.g_ro_sw[i] = g; g += fout * fin * Cr;
        L.g_ro_sc[i] = g; if (m->readout_spline_scaler[i]) g += fout * fin;
    }
    L.grads_floats = g;
    return KAGNN_OK;
}
}  // namespace

int kagnn_kagin_model_struct_bytes(void) { return (int)sizeof(kagnn_kagin_model_t); }

int kagnn_kagin_model_sizes(const kagnn_kagin_model_t* m, size_t* saved_bytes, size_t* fwd_ws, size_t* bwd_ws, size_t* grads_floats) {
    KAGNN_CHECK_ARG(saved_bytes && fwd_ws && bwd_ws && grads_floats, "null output");
    KmLayout L;
    int rc = km_layout(m, L, __func__);
    if (rc) return rc;
    *saved_bytes = L.saved_total; *fwd_ws = L.fwd_total; *bwd_ws = L.bwd_total; *grads_floats = L.grads_floats;
    return KAGNN_OK;
}

int kagnn_kagin_model_fwd(const kagnn_kagin_model_t* m, void* stream) {
    KmLayout L;
    int rc = km_layout(m, L, __func__);
    if (rc) return rc;
    KAGNN_CHECK_ARG(m->saved && m->workspace && m->out && m->x_index && (m->rowptr || m->edge_src) && m->seg_ptr && m->knots, "null array");
    KAGNN_CHECK_ARG(m->num_edges == 0 || (m->e_index && (m->edge_src || (m->col && m->perm))), "null edge array");       // (a batch of single atoms has none)
    KAGNN_CHECK_ARG((size_t)m->saved_bytes >= L.saved_total && (size_t)m->workspace_bytes >= L.fwd_total,
                    "saved / workspace too small (kagnn_kagin_model_sizes)");
    const int64_t N = m->num_nodes, E = m->num_edges, B = m->num_graphs;
    const int H = (int)m->hidden, nconv = (int)m->num_convs, nl = (int)m->num_layers, G = (int)m->grid_size, K = (int)m->spline_order, mode = (int)m->mode;
    unsigned char* sv = static_cast<unsigned char*>(m->saved);
    unsigned char* ws = static_cast<unsigned char*>(m->workspace);
    float* x0 = reinterpret_cast<float*>(sv + L.x0);
    float* ea = reinterpret_cast<float*>(sv + L.ea);
    // the batch's CSR + transpose, built by the library on a stream of its own: nothing before the first GINE aggregation depends on
    // it, so the one-workgroup-per-direction sort (55 us for a 256-molecule batch) runs beside the encoders and the weight packs
    const int32_t* rowptr = m->rowptr; const int32_t* col = m->col; const int32_t* perm = m->perm;
    SideStream* side = nullptr;
    if (m->edge_src) {
        int32_t* a[6];
        for (int k = 0; k < 6; ++k) a[k] = reinterpret_cast<int32_t*>(sv + L.csr[k]);
        rowptr = a[0]; col = a[1]; perm = a[2];
        side = side_stream();
        if (side == nullptr) return fail(KAGNN_ERR_HIP, "%s: no side stream", __func__);
        KAGNN_HIP(hipEventRecord(side->fork, as_stream(stream)));
        KAGNN_HIP(hipStreamWaitEvent(side->st, side->fork, 0));
        rc = kagnn_csr_build_small(m->edge_src, m->edge_dst, E, N, a[0], a[1], a[2], a[3], a[4], a[5], m->csr_flags, ws + L.fwd_csr_ws,
                                   L.csr_ws_bytes, side->st);
        if (rc) return rc;
        KAGNN_HIP(hipEventRecord(side->join, side->st));
    }
    // encoders: sum over the feature columns of one table each (models.py:244-281)
    for (int t = 0; t < (int)m->num_atom_tables; ++t) {
        rc = kagnn_embedding_fwd(m->x_index + t, m->x_stride, N, m->atom_table[t], (int32_t)m->atom_rows[t], H, x0, H, t > 0, stream);
        if (rc) return rc;
    }
    if (E == 0) { KAGNN_HIP(hipMemsetAsync(ea, 0, (size_t)H * sizeof(float), as_stream(stream))); }   // (a batch of single atoms: a row nothing reads)
    for (int t = 0; t < (int)m->num_bond_tables; ++t) {
        rc = kagnn_embedding_fwd(m->e_index + t, m->e_stride, E, m->bond_table[t], (int32_t)m->bond_rows[t], H, ea, H, t > 0, stream);
        if (rc) return rc;
    }
    // the GINE stack
    int32_t widths[9];
    for (int l = 0; l <= nl; ++l) widths[l] = H;
    float* acts[KAGNN_MODEL_MAX_CONVS * 9];
    void* pf[KAGNN_MODEL_MAX_LAYERS]; void* pd[KAGNN_MODEL_MAX_LAYERS];
    float* h[KAGNN_MODEL_MAX_CONVS]; float* mean[KAGNN_MODEL_MAX_CONVS]; float* rstd[KAGNN_MODEL_MAX_CONVS];
    const size_t hs = (size_t)N * H * sizeof(float);
    for (int k = 0; k < nconv * (nl + 1); ++k) acts[k] = reinterpret_cast<float*>(sv + L.acts + (size_t)k * hs);
    for (int k = 0; k < nconv * nl; ++k) { pf[k] = sv + L.packs + (size_t)k * L.fb; pd[k] = sv + L.packs + (size_t)nconv * nl * L.fb + (size_t)k * L.db; }
    for (int i = 0; i < nconv; ++i) {
        h[i] = reinterpret_cast<float*>(sv + L.h + (size_t)i * hs);
        mean[i] = reinterpret_cast<float*>(sv + L.stats) + (size_t)(2 * i) * H;
        rstd[i] = reinterpret_cast<float*>(sv + L.stats) + (size_t)(2 * i + 1) * H;
    }
    // ONE pack launch for the stack's layers AND the read-out's where they share grid, order and mode (each layer's pack depends on
    // its own weights only: the same bits as the two launches of the per-operation path)
    const int nr = (int)m->num_readout, rG = (int)m->readout_grid_size, rK = (int)m->readout_spline_order;
    void* rpf[KAGNN_MODEL_MAX_READOUT]; void* rpd[KAGNN_MODEL_MAX_READOUT];
    int32_t rin[KAGNN_MODEL_MAX_READOUT], rout[KAGNN_MODEL_MAX_READOUT];
    for (int i = 0; i < nr; ++i) { rpf[i] = sv + L.ro_pf[i]; rpd[i] = sv + L.ro_pd[i]; rin[i] = (int32_t)m->readout_widths[i]; rout[i] = (int32_t)m->readout_widths[i + 1]; }
    bool packed_all = false;
    {
        int32_t md = mode;
        ModeScope mode_scope_(md);
        bool ok = nconv * nl + nr <= 16 && L.ro_batch && rG == G && rK == K && (int)m->readout_modes[0] == mode &&
                  use_split_dx(H, H, G, K, md) && use_sparse_fwd(H, H, G, K, md) && kan_fused_pack_ok(H, H, G + K);
        for (int i = 0; i < nr && ok; ++i)
            ok = use_split_dx(rin[i], rout[i], G, K, md) && use_sparse_fwd(rin[i], rout[i], G, K, md) && kan_fused_pack_ok(rin[i], rout[i], G + K);
        if (ok) {
            const float* abw[16]; const float* asw[16]; const float* asc[16]; int32_t ain[16], aout[16]; void* apf[16]; void* apd[16];
            int n = 0;
            for (int k = 0; k < nconv * nl; ++k, ++n) { abw[n] = m->base_weight[k]; asw[n] = m->spline_weight[k]; asc[n] = m->spline_scaler[k]; ain[n] = H; aout[n] = H; apf[n] = pf[k]; apd[n] = pd[k]; }
            for (int i = 0; i < nr; ++i, ++n) { abw[n] = m->readout_base_weight[i]; asw[n] = m->readout_spline_weight[i]; asc[n] = m->readout_spline_scaler[i]; ain[n] = rin[i]; aout[n] = rout[i]; apf[n] = rpf[i]; apd[n] = rpd[i]; }
            rc = kagnn_kan_pack_batch(n, abw, asw, asc, ain, aout, G, K, mode, apf, apd, stream);
            if (rc) return rc;
            packed_all = true;
        }
    }
    struct PrepackedScope {
        bool prev;
        explicit PrepackedScope(bool on) : prev(kagnn::g_stack_prepacked) { kagnn::g_stack_prepacked = on; }
        ~PrepackedScope() { kagnn::g_stack_prepacked = prev; }
    };
    if (side) KAGNN_HIP(hipStreamWaitEvent(as_stream(stream), side->join, 0));
    {
    PrepackedScope prepacked_scope_(packed_all);
    rc = kagnn_gine_kan_stack_fwd(x0, H, ea, H, N, rowptr, col, perm, m->self_scale, nconv, nl, widths, m->base_weight, m->spline_weight,
                                  m->spline_scaler, m->knots, G, K, mode, acts, pf, pd, m->bn_weight, m->bn_bias,
                                  const_cast<float* const*>(m->running_mean), const_cast<float* const*>(m->running_var), m->momentum, m->eps, h, mean, rstd,
                                  ws + L.fwd_scratch, (size_t)m->workspace_bytes - L.fwd_scratch, stream);
    }
    if (rc) return rc;
    // global_add_pool, then the read-out chain
    float* pooled = reinterpret_cast<float*>(sv + L.pooled);
    rc = kagnn_segment_pool(h[nconv - 1], H, pooled, H, m->seg_ptr, B, H, 0, stream);
    if (rc) return rc;
    if (L.ro_batch && !packed_all) {
        rc = kagnn_kan_pack_batch(nr, m->readout_base_weight, m->readout_spline_weight, m->readout_spline_scaler, rin, rout, rG, rK,
                                  (int32_t)m->readout_modes[0], rpf, rpd, stream);
        if (rc) return rc;
    }
    for (int i = 0; i < nr; ++i) {
        const int rm = (int)m->readout_modes[i];
        if (!L.ro_batch && !packed_all) {
            rc = kagnn_kan_pack(m->readout_base_weight[i], m->readout_spline_weight[i], m->readout_spline_scaler[i], rin[i], rout[i], rG, rK, rm, rpf[i], rpd[i], stream);
            if (rc) return rc;
        }
        const float* xin = reinterpret_cast<const float*>(sv + L.ro_act[i]);
        float* y = i + 1 < nr ? reinterpret_cast<float*>(sv + L.ro_act[i + 1]) : m->out;
        size_t wb = 0;
        rc = kagnn_kan_fwd_workspace_bytes(B, rin[i], rout[i], rG, rK, rm, &wb); if (rc) return rc;
        rc = kagnn_kan_linear_fwd(xin, rin[i], B, m->readout_knots[i], rin[i], rout[i], rG, rK, rm, rpf[i], y, rout[i],
                                  wb ? ws + L.fwd_scratch : nullptr, wb, stream);
        if (rc) return rc;
    }
    return KAGNN_OK;
}

int kagnn_kagin_model_bwd(const kagnn_kagin_model_t* m, void* stream) {
    KmLayout L;
    int rc = km_layout(m, L, __func__);
    if (rc) return rc;
    KAGNN_CHECK_ARG(m->saved && m->workspace && m->g_out && m->grads && m->x_index && (m->rowptr_t || m->edge_src) && m->seg_ptr && m->knots, "null array");
    KAGNN_CHECK_ARG(m->num_edges == 0 || (m->e_index && (m->edge_src || (m->col_t && m->perm_t))), "null edge array");
    KAGNN_CHECK_ARG((size_t)m->saved_bytes >= L.saved_total && (size_t)m->workspace_bytes >= L.bwd_total,
                    "saved / workspace too small (kagnn_kagin_model_sizes)");
    const int64_t N = m->num_nodes, E = m->num_edges, B = m->num_graphs;
    const int H = (int)m->hidden, nconv = (int)m->num_convs, nl = (int)m->num_layers, G = (int)m->grid_size, K = (int)m->spline_order, mode = (int)m->mode;
    unsigned char* sv = static_cast<unsigned char*>(m->saved);
    unsigned char* ws = static_cast<unsigned char*>(m->workspace);
    float* gr = m->grads;
    const int nr = (int)m->num_readout, rG = (int)m->readout_grid_size, rK = (int)m->readout_spline_order;
    KAGNN_CHECK_ARG(m->ld_g_out >= m->readout_widths[nr], "ld_g_out smaller than the model's output width");
    // read-out, last layer first: input gradient, then weight gradient (the order of graph_ops._KaginModelFn.backward)
    const float* gy = m->g_out;
    int64_t ldgy = m->ld_g_out;
    for (int i = nr - 1; i >= 0; --i) {
        const int fin = (int)m->readout_widths[i], fout = (int)m->readout_widths[i + 1], rm = (int)m->readout_modes[i];
        const float* xin = reinterpret_cast<const float*>(sv + L.ro_act[i]);
        float* gx = reinterpret_cast<float*>(ws + L.bwd_gy[i & 1]);
        rc = kagnn_kan_linear_bwd_input(xin, fin, gy, ldgy, B, m->readout_knots[i], fin, fout, rG, rK, rm, sv + L.ro_pd[i], gx, fin, KAGNN_DTYPE_F32, stream);
        if (rc) return rc;
        size_t wb = 0;
        rc = kagnn_kan_bwd_weight_workspace_bytes(B, fin, fout, rG, rK, rm, &wb); if (rc) return rc;
        rc = kagnn_kan_linear_bwd_weight(xin, fin, gy, ldgy, B, m->readout_knots[i], fin, fout, rG, rK, rm, m->readout_spline_weight[i],
                                         m->readout_spline_scaler[i], gr + L.g_ro_bw[i], gr + L.g_ro_sw[i],
                                         m->readout_spline_scaler[i] ? gr + L.g_ro_sc[i] : nullptr, ws + L.bwd_scratch, wb, stream);
        if (rc) return rc;
        gy = gx; ldgy = fin;
    }
    // pool backward, the stack, the encoders
    float* gh = reinterpret_cast<float*>(ws + L.bwd_gh);
    rc = kagnn_segment_broadcast(gy, ldgy, gh, H, m->seg_ptr, B, H, 0, stream);
    if (rc) return rc;
    int32_t widths[9];
    for (int l = 0; l <= nl; ++l) widths[l] = H;
    const float* acts[KAGNN_MODEL_MAX_CONVS * 9];
    const void* pd[KAGNN_MODEL_MAX_LAYERS];
    const float* h[KAGNN_MODEL_MAX_CONVS]; const float* mean[KAGNN_MODEL_MAX_CONVS]; const float* rstd[KAGNN_MODEL_MAX_CONVS];
    float* g_bn_w[KAGNN_MODEL_MAX_CONVS]; float* g_bn_b[KAGNN_MODEL_MAX_CONVS];
    float* g_bw[KAGNN_MODEL_MAX_LAYERS]; float* g_sw[KAGNN_MODEL_MAX_LAYERS]; float* g_sc[KAGNN_MODEL_MAX_LAYERS];
    const size_t hs = (size_t)N * H * sizeof(float);
    for (int k = 0; k < nconv * (nl + 1); ++k) acts[k] = reinterpret_cast<const float*>(sv + L.acts + (size_t)k * hs);
    for (int k = 0; k < nconv * nl; ++k) {
        pd[k] = sv + L.packs + (size_t)nconv * nl * L.fb + (size_t)k * L.db;
        g_bw[k] = gr + L.g_bw[k]; g_sw[k] = gr + L.g_sw[k]; g_sc[k] = gr + L.g_sc[k];
    }
    for (int i = 0; i < nconv; ++i) {
        h[i] = reinterpret_cast<const float*>(sv + L.h + (size_t)i * hs);
        mean[i] = reinterpret_cast<const float*>(sv + L.stats) + (size_t)(2 * i) * H;
        rstd[i] = reinterpret_cast<const float*>(sv + L.stats) + (size_t)(2 * i + 1) * H;
        g_bn_w[i] = gr + L.g_bn_w[i]; g_bn_b[i] = gr + L.g_bn_b[i];
    }
    const float* x0 = reinterpret_cast<const float*>(sv + L.x0);
    const float* ea = reinterpret_cast<const float*>(sv + L.ea);
    float* gx0 = reinterpret_cast<float*>(ws + L.bwd_gx0);
    float* gea = reinterpret_cast<float*>(ws + L.bwd_gea);
    const int32_t* rowptr_t = m->rowptr_t; const int32_t* col_t = m->col_t; const int32_t* perm_t = m->perm_t;
    if (m->edge_src) {                      // (built by the forward into `saved`)
        rowptr_t = reinterpret_cast<const int32_t*>(static_cast<unsigned char*>(m->saved) + L.csr[3]);
        col_t = reinterpret_cast<const int32_t*>(static_cast<unsigned char*>(m->saved) + L.csr[4]);
        perm_t = reinterpret_cast<const int32_t*>(static_cast<unsigned char*>(m->saved) + L.csr[5]);
    }
    rc = kagnn_gine_kan_stack_bwd(gh, H, x0, H, ea, H, N, rowptr_t, col_t, perm_t, m->self_scale, nconv, nl, widths, m->spline_weight,
                                  m->spline_scaler, m->knots, G, K, mode, acts, pd, h, m->bn_weight, mean, rstd, gx0, H, gea, H, g_bn_w, g_bn_b, g_bw, g_sw,
                                  g_sc, ws + L.bwd_scratch, (size_t)m->workspace_bytes - L.bwd_scratch, stream);
    if (rc) return rc;
    for (int t = 0; t < (int)m->num_atom_tables; ++t) {
        size_t wb = 0;
        rc = kagnn_embedding_bwd_workspace_bytes(N, (int32_t)m->atom_rows[t], H, &wb); if (rc) return rc;
        rc = kagnn_embedding_bwd(m->x_index + t, m->x_stride, N, gx0, H, (int32_t)m->atom_rows[t], H, gr + L.g_atom[t], ws + L.bwd_scratch, wb, stream);
        if (rc) return rc;
    }
    for (int t = 0; t < (int)m->num_bond_tables; ++t) {
        size_t wb = 0;
        rc = kagnn_embedding_bwd_workspace_bytes(E, (int32_t)m->bond_rows[t], H, &wb); if (rc) return rc;
        rc = kagnn_embedding_bwd(m->e_index + t, m->e_stride, E, gea, H, (int32_t)m->bond_rows[t], H, gr + L.g_bond[t], ws + L.bwd_scratch, wb, stream);
        if (rc) return rc;
    }
    return KAGNN_OK;
}

}  // extern "C"
#pragma GCC visibility pop
