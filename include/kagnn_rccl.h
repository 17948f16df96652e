/*
 * kagnn_rccl.h -- C ABI of libkagnn_rccl.so: the feature-sharded KANLinear of the multi-GPU layer with its exchange step
 * done by RCCL on a communicator the CALLER owns ("... and the sharded variants taking an ncclComm_t /
 * process-group").  A separate shared object on purpose: libkagnn_hip.so (include/kagnn_hip.h) depends on libamdhip64
 * only, so single-GPU users do not carry librccl; this one links librccl.so and libkagnn_hip.so.
 *
 * The reference has no multi-GPU code; the contract is BASELINE.json's north_star: "spline coefficients
 * sharded across 8 GPUs with all-reduce via RCCL".  What is sharded is KANLinear.forward (node_classification_clean/
 * ekan.py:154-162): every rank holds the input-feature slice [lo, hi) of base_weight / spline_weight / spline_scaler and
 * the same columns of the activation; its partial sums over ALL outputs are closed by one reduce-scatter along `out`
 * (forward) and one all-gather (backward) -- the all-reduce of north_star with each rank keeping only the columns the
 * next sharded layer needs.  Same scheme as kagnn_amd/sharded.py::ShardedGIKANLayer(comm="rccl"), which goes through
 * torch.distributed; here the whole exchange is library code:
 *
 *   forward, per row chunk c (rows [r0, r1)):
 *     compute stream : kagnn_kan_linear_fwd on the chunk -> partial sums [rows, out]; one staging kernel -> rank-major
 *                      blocks [P][rows][out/P] (block p is what rank p keeps)
 *     side stream    : waits for that chunk only; ncclReduceScatter(blocks -> y_shard rows [r0, r1))
 *                      => the exchange of chunk c runs beside the KAN kernel of chunk c+1
 *   backward:
 *     side stream    : ncclAllGather of every chunk of gy_shard, all requested at once
 *     compute stream : per chunk, waits for ITS gather only; one kernel -> gathered gradient [rows, out] in its final
 *                      layout; kagnn_kan_linear_bwd_input on the chunk; after the last chunk ONE kagnn_kan_linear_bwd_weight
 *                      over all rows (parameter gradients are local: the parameters are sharded)
 *
 * Conventions as in kagnn_hip.h (device pointers, row-major fp32 + leading dimensions, streams as void*, 0 = success,
 * kagnn_rccl_last_error()).  `comm` is an ncclComm_t passed as void*; `world` / `rank` must be the communicator's.
 * Every rank of the communicator must make the same call with the same shapes and row_chunks.  y_shard / gy_shard are
 * CONTIGUOUS [num_rows, out_features / world] (RCCL writes / reads them in place).  Nothing falls back to a CPU path.
 */
#ifndef KAGNN_RCCL_H
#define KAGNN_RCCL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int kagnn_rccl_version(void);          /* 100 */
const char* kagnn_rccl_last_error(void);

/* Convenience for hosts that do not bind RCCL themselves (the ctypes host of kagnn_amd/rccl.py): a 128-byte ncclUniqueId
 * (rank 0 makes it, the host broadcasts the bytes by whatever means it has), ncclCommInitRank on the CURRENT device,
 * ncclCommDestroy.  A C / C++ host that already owns an ncclComm_t passes it to the entry points below and never calls these. */
int kagnn_rccl_unique_id(void* id128_host);
int kagnn_rccl_comm_init(const void* id128_host, int32_t world, int32_t rank, void** comm_out_host);
int kagnn_rccl_comm_destroy(void* comm);

/* scratch sizes for one KANLinear at `num_rows` rows split into `row_chunks` chunks: forward = partial sums + rank-major blocks (2 * N * out * 4 bytes) + the
 * forward kernel's own workspace; backward = gathered blocks + gathered gradient + the weight-gradient partial sums */
int kagnn_sharded_kan_linear_workspace_bytes(int64_t num_rows, int32_t in_local, int32_t out_features, int32_t grid_size,
                                             int32_t spline_order, int32_t mode, int32_t world, int32_t row_chunks,
                                             size_t* fwd_bytes_host, size_t* bwd_bytes_host);

/* y_shard[N, out/world] = (sum over ranks of  silu(x_p) @ base_weight_p^T + bases(x_p) @ (spline_weight_p * scaler_p)^T)[:, my columns]
 *   x_slice  [N, in_local] (ldx): this rank's input-feature columns;  pack_fwd: kagnn_kan_pack of this rank's parameter slice
 *   row_chunks >= 1 (clamped to the row count; chunks are whole 256-row tiles from 4096 rows up)
 * On return the result is ordered on compute_stream (it has waited for the side stream).                              */
int kagnn_sharded_kan_linear_fwd(const float* x_slice, int64_t ldx, int64_t num_rows, const float* knots,
                                 int32_t in_local, int32_t out_features, int32_t grid_size, int32_t spline_order,
                                 int32_t mode, const void* pack_fwd, float* y_shard,
                                 void* comm, int32_t world, int32_t rank, int32_t row_chunks,
                                 void* workspace, size_t workspace_bytes, void* compute_stream, void* side_stream);

/* backward of the above: gy_shard[N, out/world] is d loss / d y_shard; gx_slice[N, in_local] (ldgx; NULL = not wanted) and the
 * gradients of this rank's parameter slice (as kagnn_kan_linear_bwd_weight: g_base_weight / g_spline_scaler may be NULL). */
int kagnn_sharded_kan_linear_bwd(const float* x_slice, int64_t ldx, const float* gy_shard, int64_t num_rows,
                                 const float* knots, int32_t in_local, int32_t out_features, int32_t grid_size,
                                 int32_t spline_order, int32_t mode, const void* pack_dx,
                                 const float* spline_weight, const float* spline_scaler,
                                 float* gx_slice, int64_t ldgx,
                                 float* g_base_weight, float* g_spline_weight, float* g_spline_scaler,
                                 void* comm, int32_t world, int32_t rank, int32_t row_chunks,
                                 void* workspace, size_t workspace_bytes, void* compute_stream, void* side_stream);

#ifdef __cplusplus
}
#endif
#endif /* KAGNN_RCCL_H */
