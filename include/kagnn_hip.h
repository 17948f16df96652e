/*
 * kagnn_hip.h -- C ABI of libkagnn_hip.so, the MI355X (gfx950) implementation of the
 * KAN-GNN layer hot path of RomanBresson/KAGNN.
 *
 * The reference has NO native boundary (it is 100 % Python on torch / torch_geometric); the
 * entry points below are what a ctypes binding inside the reference's layers would call in
 * place of the aten-op sequences cited on each function (paths relative to the reference
 * repository root).  INTEGRATION.md shows that binding.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in _host;
 *  - matrices are row-major fp32 with an explicit leading dimension (elements) so column
 *    slices of a wider activation can be passed without a copy;
 *  - index arrays produced by this library are int32 (validated: N, E < 2^31); the edge list
 *    handed in is the reference's int64 `edge_index`;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); no call synchronises
 *    the device except kagnn_csr_build (documented there);
 *  - return value 0 = success, negative = error; kagnn_last_error() returns a message for the
 *    calling thread's most recent failure.  Nothing falls back to a CPU path.
 */
#ifndef KAGNN_HIP_H
#define KAGNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KAGNN_OK 0
#define KAGNN_ERR_ARG (-1)      /* bad argument (shape / range / null)            */
#define KAGNN_ERR_HIP (-2)      /* a HIP runtime call or kernel launch failed      */
#define KAGNN_ERR_UNSUPPORTED (-3)

/* precision of the dense contraction (the `mode` argument of the KAN entry points) */
#define KAGNN_PREC_FP32 0       /* v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate    */
#define KAGNN_PREC_SPLIT 1      /* fp16 hi/lo split operands, 3 MFMAs per product, fp32 accumulate */
#define KAGNN_PREC_FP32_GRID 2  /* exact fp32, and `knots` is the whole [in, G+2k+1] grid buffer: per-feature,
                                 * non-uniform knot rows (what KANLinear.update_grid leaves behind)       */
#define KAGNN_PREC_HALF 3       /* BUILD-DEFINED reduced precision (BASELINE config 2 "bf16"; the reference is fp32 only, ekan.py:154-162):
                                 * KAGNN_PREC_SPLIT's kernels with ONE fp16 product per fp32 product -- bases, SiLU, gy and the packed
                                 * weights are evaluated in fp32 and rounded once (RNE, 11 significant bits) after the same exact
                                 * power-of-two scaling, fp32 accumulate: a third of the matrix-core work, ~5e-4 from the fp32 result.
                                 * Cubic layers of <= 8 coefficients; any other shape runs the three-product kernels (more accurate).
                                 * Takes the same packs as KAGNN_PREC_SPLIT.  Never the headline.                              */

/* element type of an activation / gradient matrix where an entry point accepts more than fp32 */
#define KAGNN_DTYPE_F32 0
#define KAGNN_DTYPE_BF16 1

int kagnn_version(void);          /* 260 = this header (round 6: + the feature-sharded FastKAN entry points kagnn_fastkan_row_moments .. _shard_bwd_finish, + kagnn_kagin_model_*; 250 = 240 + KAGNN_PREC_HALF; 240 = 230 + kagnn_gin_kan_layer_bwd_bn_sums; 230 = 220 + the *_affine entry points of a folded BatchNorm1d; 220 = 210 + the stage timer) */
const char* kagnn_last_error(void);

/* Stage timer -- a measurement aid, off by default (no reference counterpart: the reference times whole epochs with
 * time.time(), node_classification_clean/time_model.py:38-47).  While enabled, every per-operation entry point below
 * (aggregation, weight packs, KANLinear forward / input gradient / weight gradient, FastKAN forward / backward, BatchNorm) is
 * bracketed by HIP events recorded on the stream it launches on -- ALSO when it runs inside kagnn_gin_kan_layer_fwd / _bwd*, so
 * a caller can time one kernel live inside the product's one-call-per-convolution path.  `only`: NULL = every stage, else the
 * one stage name to record (e.g. "kagnn_aggregate_sum").  kagnn_stage_timer_collect waits for the recorded events, sums them by
 * stage name into the caller's arrays (`names`: `capacity` slots of 64 chars) and clears the records. */
int kagnn_stage_timer_enable(const char* only);
int kagnn_stage_timer_disable(void);
int kagnn_stage_timer_collect(char* names, int64_t* launches, double* total_ms, int32_t capacity, int32_t* n_stages);

/* ------------------------------------------------------------------------------------------
 * Graph structure.  Replaces the per-call `index_select` / `scatter_add_` bookkeeping of
 * torch_geometric's MessagePassing.propagate (called from node_classification_clean/
 * models.py:48-56 via GINConv and :31-37 via GCNConv) by a CSR built once per edge_index.
 *
 * kagnn_csr_build: stable sort of the E edges by `key` (values 0..N-1).
 *   rowptr[N+1]  = exclusive prefix sum of the key histogram
 *   perm[E]      = argsort(key, stable)                (bit-exact contract)
 *   col[E]       = val[perm]
 * (key=dst,val=src) gives the forward structure, (key=src,val=dst) its transpose for backward.
 * Rows whose degree exceeds `hub_threshold` are split into segments of L = max(hub_threshold/4, 64)
 * edges, listed in hub_seg[3*i+{0,1,2}] = {row, e_begin, e_end} (a row's segments are contiguous
 * in e and its first one starts at rowptr[row]); *num_hub_seg_host receives their count (never
 * more than E/L + E/hub_threshold + 1, so size hub_seg for 3x that many int32).  This call
 * synchronises `stream` once (to return the count).
 * ------------------------------------------------------------------------------------------ */
int kagnn_csr_workspace_bytes(int64_t num_edges, int64_t num_nodes, size_t* bytes_host);
int kagnn_csr_build(const int64_t* key, const int64_t* val, int64_t num_edges, int64_t num_nodes,
                    int32_t* rowptr, int32_t* col, int32_t* perm,
                    int32_t hub_threshold, int32_t* hub_seg, int64_t hub_seg_capacity,
                    int64_t* num_hub_seg_host,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Small graphs (1 <= E, N <= 65 536; kagnn_csr_small_ok == 1) -- the mini-batches of the graph-level models, whose CSR is rebuilt
 * per batch (reference graph_regression/optuna_zinc.py:56-66 -> torch_geometric's propagate bookkeeping): BOTH structures in ONE
 * launch and with NO host synchronisation.  src / dst = edge_index[0] / edge_index[1] (int64); (rowptr, col, perm) by destination as
 * kagnn_csr_build(key = dst, val = src) makes them, (rowptr_t, col_t, perm_t) by source; same stable order, bit for bit.  Node ids
 * outside [0, num_nodes) are clamped to 0 (no out-of-bounds access follows) and reported in the DEVICE array flags[2] (bit 0: a key,
 * bit 1: a value; [0] destination side, [1] source side), which the caller reads when convenient.  No hub segments.
 * Workspace: kagnn_csr_small_workspace_bytes(E).                                                                                */
int kagnn_csr_small_ok(int64_t num_edges, int64_t num_nodes);
int kagnn_csr_small_workspace_bytes(int64_t num_edges, size_t* bytes_host);
int kagnn_csr_build_small(const int64_t* src, const int64_t* dst, int64_t num_edges, int64_t num_nodes, int32_t* rowptr,
                          int32_t* col, int32_t* perm, int32_t* rowptr_t, int32_t* col_t, int32_t* perm_t, int32_t* flags,
                          void* workspace, size_t workspace_bytes, void* stream);

/* in-degree normalisation of GCN (torch_geometric gcn_norm as used by KAGCNConv,
 * node_classification_clean/models.py:31-37): dis[i] = (1 + #non-loop in-edges of i)^-1/2.
 * `rowptr/col` must be the by-destination CSR of the ORIGINAL edge list. */
int kagnn_gcn_deg_inv_sqrt(const int32_t* rowptr, const int32_t* col, int64_t num_nodes,
                           float* dis, void* stream);

/* ------------------------------------------------------------------------------------------
 * Neighbour aggregation (sum).  Replaces x.index_select(0,row) -> scatter_add_(0,col) of
 * MessagePassing.propagate:
 *
 *   out[i,:] = out_scale[i] * ( self_scale * s(i) * x[i,:] + sum_{e in row i} w(e) * s(col[e]) * x[col[e],:] ) + bias[:]
 *
 * with s(j) = in_scale[j] (or 1 when NULL), w(e) = edge_weight[e] in CSR order (or 1 when NULL),
 * out_scale NULL = 1, bias NULL = 0.  `skip_self_loops` != 0 drops edges with col[e]==i (GCN's
 * add_remaining_self_loops).  GIN: self_scale = 1+eps, everything else NULL.  GCN: in_scale =
 * out_scale = dis, self_scale = 1, skip_self_loops = 1, bias = conv bias.  The backward of an
 * aggregation is the same call on the transposed CSR.
 * Deterministic (bit-reproducible run to run): rows listed in hub_seg are summed per segment into
 * `workspace` (kagnn_aggregate_workspace_bytes(num_hub_seg, num_feat) bytes; may be NULL when num_hub_seg
 * is 0) and folded into the row in segment order -- no atomics anywhere.
 * ------------------------------------------------------------------------------------------ */
int kagnn_aggregate_workspace_bytes(int64_t num_hub_seg, int32_t num_feat, size_t* bytes_host);
int kagnn_aggregate_sum(const float* x, int64_t ldx, float* out, int64_t ldo,
                        const int32_t* rowptr, const int32_t* col, const float* edge_weight,
                        int64_t num_nodes, int32_t num_feat, float self_scale,
                        const float* in_scale, const float* out_scale, const float* bias,
                        int32_t skip_self_loops,
                        const int32_t* hub_seg, int64_t num_hub_seg, int32_t hub_threshold,
                        void* workspace, size_t workspace_bytes, void* stream);
/* out = <the aggregation above> + addend (fp32 [num_nodes, num_feat], leading dimension ld_addend; NULL = none), added
 * last, in the kernel's epilogue: where the tape would otherwise sum two gradients of one activation in a pass of its own
 * (the skip-concat models: the read-out's gradient of h_l and the next convolution's -- reference
 * node_classification_clean/models.py:196-202).  Bit-identical to kagnn_aggregate_sum followed by the addition.        */
int kagnn_aggregate_sum_add(const float* x, int64_t ldx, float* out, int64_t ldo,
                            const int32_t* rowptr, const int32_t* col, const float* edge_weight,
                            int64_t num_nodes, int32_t num_feat, float self_scale,
                            const float* in_scale, const float* out_scale, const float* bias,
                            int32_t skip_self_loops,
                            const int32_t* hub_seg, int64_t num_hub_seg, int32_t hub_threshold,
                            const float* addend, int64_t ld_addend,
                            void* workspace, size_t workspace_bytes, void* stream);

/* The same aggregation with bf16 GATHER OPERANDS (`KAGNN_ACT=bf16`; BASELINE.json config 2 -- the reference has no
 * reduced-precision path, SURVEY.md 8(d) makes this a build-defined mode): `x` rows are bf16 (leading dimension in
 * ELEMENTS, a multiple of 8; 16-byte aligned), sums are accumulated in fp32 and written as fp32 (`out_dtype` =
 * KAGNN_DTYPE_F32: the forward, feeding the KAN layer) or bf16 with ONE round-to-nearest-even per element
 * (KAGNN_DTYPE_BF16: the input gradient of a bf16 activation).  Halves the E * num_feat * 4 bytes of the gather.
 * num_feat % 8 == 0, <= 512 (else KAGNN_ERR_UNSUPPORTED: convert to fp32 and call kagnn_aggregate_sum).
 * Deterministic; workspace as for kagnn_aggregate_sum.  kagnn_rows_to_bf16 converts fp32 rows (RNE). */
int kagnn_aggregate_sum_bf16(const void* x, int64_t ldx, void* out, int64_t ldo, int32_t out_dtype,
                             const int32_t* rowptr, const int32_t* col, const float* edge_weight,
                             int64_t num_nodes, int32_t num_feat, float self_scale,
                             const float* in_scale, const float* out_scale, const float* bias,
                             int32_t skip_self_loops,
                             const int32_t* hub_seg, int64_t num_hub_seg, int32_t hub_threshold,
                             void* workspace, size_t workspace_bytes, void* stream);
int kagnn_rows_to_bf16(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t num_rows, int32_t num_feat,
                       void* stream);

/* GINE message: out[i,:] = self_scale*x[i,:] + sum_e relu(x[col[e],:] + edge_attr[perm[e],:])
 * (torch_geometric GINEConv as used by graph_regression/models.py:98,113).              */
int kagnn_aggregate_gine(const float* x, int64_t ldx, const float* edge_attr, int64_t lde,
                         float* out, int64_t ldo, const int32_t* rowptr, const int32_t* col,
                         const int32_t* perm, int64_t num_nodes, int32_t num_feat,
                         float self_scale, void* stream);
/* backward of the GINE message, run on the TRANSPOSED structure (rows = source nodes j, built
 * with key=src,val=dst): for e in row j with target i = col_t[e] and original edge id
 * pe = perm_t[e]:  g = gout[i,:] * (x[j,:] + edge_attr[pe,:] > 0);  g_edge_attr[pe,:] = g;
 * gx[j,:] = self_scale*gout[j,:] + sum_e g.  Deterministic (no atomics); g_edge_attr may be NULL. */
int kagnn_aggregate_gine_bwd(const float* x, int64_t ldx, const float* edge_attr, int64_t lde,
                             const float* gout, int64_t ldg, float* gx, int64_t ldgx,
                             float* g_edge_attr, int64_t ldge,
                             const int32_t* rowptr_t, const int32_t* col_t, const int32_t* perm_t,
                             int64_t num_nodes, int32_t num_feat, float self_scale, void* stream);

/* segmented sum over a SORTED batch vector given as segment offsets (global_add_pool,
 * graph_regression/models.py:117): out[b,:] = sum_{i in [seg[b],seg[b+1])} x[i,:];
 * `mean` != 0 divides by max(count,1) (global_mean_pool).  Backward = kagnn_segment_broadcast. */
int kagnn_segment_pool(const float* x, int64_t ldx, float* out, int64_t ldo,
                       const int32_t* seg_ptr, int64_t num_segments, int32_t num_feat,
                       int32_t mean, void* stream);
int kagnn_segment_broadcast(const float* gout, int64_t ldg, float* gx, int64_t ldgx,
                            const int32_t* seg_ptr, int64_t num_segments, int32_t num_feat,
                            int32_t mean, void* stream);

/* ------------------------------------------------------------------------------------------
 * efficient-KAN layer.  Replaces KANLinear.forward (node_classification_clean/ekan.py:154-162:
 * b_splines :79-112 + scaled_spline_weight :146-152 + two F.linear) and its autograd backward.
 * The [N, in, G+k] basis tensor is never materialised.
 *
 *  knots        : ONE row of the module's `grid` buffer, G+2k+1 fp32 values (uniform grid; the host
 *                 side verifies all rows equal and uniform) -- or, with mode KAGNN_PREC_FP32_GRID,
 *                 the whole buffer [in, G+2k+1] with increasing, possibly non-uniform rows
 *  base_weight  [out,in] or NULL (no SiLU branch), spline_weight [out,in,G+k], spline_scaler [out,in] or NULL
 *
 * kagnn_kan_pack rearranges (base_weight | spline_weight*scaler) into the MFMA fragment order
 * used by fwd (`pack_fwd`) and by the input-gradient kernel (`pack_dx`); call it whenever the
 * parameters change.  Sizes from kagnn_kan_pack_bytes.
 * ------------------------------------------------------------------------------------------ */
int kagnn_kan_pack_bytes(int32_t in_features, int32_t out_features, int32_t grid_size,
                         int32_t spline_order, int32_t mode, size_t* fwd_bytes_host,
                         size_t* dx_bytes_host);
int kagnn_kan_pack(const float* base_weight, const float* spline_weight,
                   const float* spline_scaler, int32_t in_features, int32_t out_features,
                   int32_t grid_size, int32_t spline_order, int32_t mode,
                   void* pack_fwd, void* pack_dx, void* stream);

/* kagnn_kan_pack for the layers of a chain (<= 8, same grid_size / spline_order / mode) in ONE launch -- a pack
 * launch is ~20 us of latency, not work.  Arrays of n_layers HOST-side entries (device pointers / sizes); `sc` may
 * be NULL (no scalers) or hold NULL entries.  KAGNN_ERR_UNSUPPORTED unless every layer takes the sparse-forward /
 * split-precision path with out_features <= 64 (the caller then packs layer by layer).                          */
int kagnn_kan_pack_batch(int32_t n_layers, const float* const* base_weight, const float* const* spline_weight,
                         const float* const* spline_scaler, const int32_t* in_features,
                         const int32_t* out_features, int32_t grid_size, int32_t spline_order, int32_t mode,
                         void* const* pack_fwd, void* const* pack_dx, void* stream);

/* y[N,out] = silu(x) @ base_weight^T + bases(x) @ (spline_weight*scaler)^T.
 * Inputs with few rows and many features (Cora: 2708 x 1433) split the feature loop over more
 * workgroups and sum per-split partial outputs in a fixed order; that needs a scratch buffer of
 * kagnn_kan_fwd_workspace_bytes() bytes (0 for tall inputs: workspace may then be NULL).        */
int kagnn_kan_fwd_workspace_bytes(int64_t num_rows, int32_t in_features, int32_t out_features,
                                  int32_t grid_size, int32_t spline_order, int32_t mode,
                                  size_t* bytes_host);
int kagnn_kan_linear_fwd(const float* x, int64_t ldx, int64_t num_rows, const float* knots,
                         int32_t in_features, int32_t out_features, int32_t grid_size,
                         int32_t spline_order, int32_t mode, const void* pack_fwd,
                         float* y, int64_t ldy, void* workspace, size_t workspace_bytes,
                         void* stream);

/* The same forward on an input given as COLUMN BLOCKS [x_0 | x_1 | ...] held in different buffers: replaces `self.lay_out(torch.cat(l, dim=1))` of the node models (reference
 * node_classification_clean/models.py:202-203, 256-257) without building the concatenation -- one launch, y written once.
 * x_parts / part_widths / part_ld are HOST arrays of num_parts device pointers / widths / leading dimensions; pack_fwd is
 * the pack of the whole layer (in_features = sum of the widths).  Covered: cubic layers of <= 8 coefficients in the split
 * mode, every block a multiple of 64 columns, <= 512 columns in all, 16-byte aligned rows (leading dimensions: multiples
 * of 4, <= 7680); kagnn_kan_fwd_parts_ok() says (1 / 0) whether a shape is.
 * Anything else returns KAGNN_ERR_UNSUPPORTED (concatenate and call kagnn_kan_linear_fwd).                          */
int kagnn_kan_fwd_parts_ok(const int32_t* part_widths, int32_t num_parts, int32_t in_features, int32_t out_features,
                           int32_t grid_size, int32_t spline_order, int32_t mode);
int kagnn_kan_linear_fwd_parts(const float* const* x_parts, const int32_t* part_widths, const int64_t* part_ld,
                               int32_t num_parts, int64_t num_rows, const float* knots, int32_t in_features, int32_t out_features,
                               int32_t grid_size, int32_t spline_order, int32_t mode, const void* pack_fwd,
                               float* y, int64_t ldy, void* workspace, size_t workspace_bytes, void* stream);

/* The same forward, which also leaves the COLUMN MOMENTS of y in col_mean[out] (mean over the rows) and
 * col_m2[out] (sum of squared deviations from it): the batch statistics of the BatchNorm1d that follows every
 * convolution (node_classification_clean/models.py:198-200), handed to kagnn_batchnorm_fwd so that it needs no
 * statistics pass over y.  Cubic-spline split-precision layers accumulate them in the forward kernel's epilogue
 * (per wave over its row tiles, merged in a fixed order by the pairwise update of Chan et al.: deterministic, no
 * cancellation); every other layer shape runs the plain forward and one column pass.  num_rows >= 1; the workspace
 * (kagnn_kan_fwd_moments_workspace_bytes) is required.                                                          */
int kagnn_kan_fwd_moments_workspace_bytes(int64_t num_rows, int32_t in_features, int32_t out_features,
                                          int32_t grid_size, int32_t spline_order, int32_t mode,
                                          size_t* bytes_host);
int kagnn_kan_linear_fwd_moments(const float* x, int64_t ldx, int64_t num_rows, const float* knots,
                                 int32_t in_features, int32_t out_features, int32_t grid_size,
                                 int32_t spline_order, int32_t mode, const void* pack_fwd,
                                 float* y, int64_t ldy, float* col_mean, float* col_m2,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* gx[N,in] = d loss / d x given gy[N,out] (x is the saved layer input; bases' derivatives are
 * recomputed, nothing but x was saved).  gx_dtype = KAGNN_DTYPE_F32, or KAGNN_DTYPE_BF16 (split-precision
 * B-spline layers with <= 128 outputs: the rows the transposed aggregation gathers next, rounded once) with
 * ldgx in bf16 elements.                                                                    */
int kagnn_kan_linear_bwd_input(const float* x, int64_t ldx, const float* gy, int64_t ldgy,
                               int64_t num_rows, const float* knots, int32_t in_features,
                               int32_t out_features, int32_t grid_size, int32_t spline_order,
                               int32_t mode, const void* pack_dx, void* gx, int64_t ldgx,
                               int32_t gx_dtype, void* stream);

/* parameter gradients: g_base_weight[out,in] (NULL when not wanted), g_spline_weight[out,in,G+k],
 * g_spline_scaler[out,in] (NULL when the layer has no scaler).  `workspace` holds the per-wave
 * partial sums (size from kagnn_kan_bwd_weight_workspace_bytes); deterministic reduction.   */
int kagnn_kan_bwd_weight_workspace_bytes(int64_t num_rows, int32_t in_features,
                                         int32_t out_features, int32_t grid_size,
                                         int32_t spline_order, int32_t mode, size_t* bytes_host);
int kagnn_kan_linear_bwd_weight(const float* x, int64_t ldx, const float* gy, int64_t ldgy,
                                int64_t num_rows, const float* knots, int32_t in_features,
                                int32_t out_features, int32_t grid_size, int32_t spline_order,
                                int32_t mode, const float* spline_weight,
                                const float* spline_scaler, float* g_base_weight,
                                float* g_spline_weight, float* g_spline_scaler,
                                void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * One KAN-GIN convolution per call -- GIKANLayer.forward (node_classification_clean/models.py:48-56: GINConv around
 * make_kan's KAN chain, ekan.py:270-275) and its backward; the `gin_kan_fused_fwd / _bwd` of SURVEY.md 8(b).  The call
 * sequences the library's own kernels on `stream` (aggregation, ONE weight-pack launch for the chain, the KANLinear
 * forwards; on the way back per layer the weight and input gradients, then the transposed aggregation); the caller owns
 * every buffer:
 *   widths[L+1]        layer widths: in_features of layer 0 ... out_features of layer L-1 (L <= 8, same grid / order / mode)
 *   acts[L+1]          fp32 [N, widths[l]]: acts[0] receives the aggregate h0, acts[l+1] the output of layer l (acts[L] = y);
 *                      the backward reads acts[0..L-1] (the only saved tensors) -- contiguous, ld = width
 *   pack_fwd/pack_dx   per layer, sized by kagnn_kan_pack_bytes; written by _fwd, pack_dx read by _bwd
 *   x                  [N, widths[0]] fp32 or bf16 (x_dtype; bf16 = the gather-operand mode of kagnn_aggregate_sum_bf16)
 *   _bwd: gx           d loss / d x in gx_dtype, or NULL; `bf16_gather` != 0 lets the first layer's input-gradient kernel
 *                      write d loss / d h0 as bf16 rows for the transposed aggregation (split precision, cubic, <= 8
 *                      coefficients; otherwise it stays fp32); g_base_weight / g_spline_weight / g_spline_scaler per layer
 *   (rowptr, col, hub_seg): by destination for _fwd, the transposed (by source) structure for _bwd
 *   _fwd: col_mean / col_m2  [widths[L]] each or both NULL: the column moments of y for the BatchNorm1d that follows
 *                      (kagnn_kan_linear_fwd_moments; hand them to kagnn_batchnorm_fwd)
 * Workspace sizes from kagnn_gin_kan_layer_workspace_bytes (hub-segment counts of the two directions).  Deterministic.
 * ------------------------------------------------------------------------------------------ */
int kagnn_gin_kan_layer_workspace_bytes(int64_t num_nodes, int32_t num_layers, const int32_t* widths,
                                        int32_t grid_size, int32_t spline_order, int32_t mode,
                                        int64_t num_hub_seg, int64_t num_hub_seg_t,
                                        size_t* fwd_bytes_host, size_t* bwd_bytes_host);
int kagnn_gin_kan_layer_fwd(const void* x, int32_t x_dtype, int64_t ldx, int64_t num_nodes,
                            const int32_t* rowptr, const int32_t* col, const int32_t* hub_seg,
                            int64_t num_hub_seg, int32_t hub_threshold, float self_scale,
                            int32_t num_layers, const int32_t* widths, const float* const* base_weight,
                            const float* const* spline_weight, const float* const* spline_scaler,
                            const float* knots, int32_t grid_size, int32_t spline_order, int32_t mode,
                            float* const* acts, void* const* pack_fwd, void* const* pack_dx,
                            float* col_mean, float* col_m2,
                            void* workspace, size_t workspace_bytes, void* stream);
int kagnn_gin_kan_layer_bwd(const float* gy, int64_t ldgy, int64_t num_nodes, const int32_t* rowptr_t,
                            const int32_t* col_t, const int32_t* hub_seg_t, int64_t num_hub_seg_t,
                            int32_t hub_threshold, float self_scale, int32_t num_layers, const int32_t* widths,
                            const float* const* spline_weight, const float* const* spline_scaler,
                            const float* knots, int32_t grid_size, int32_t spline_order, int32_t mode,
                            const float* const* acts, const void* const* pack_dx, void* gx, int32_t gx_dtype,
                            int64_t ldgx, int32_t bf16_gather, float* const* g_base_weight,
                            float* const* g_spline_weight, float* const* g_spline_scaler,
                            void* workspace, size_t workspace_bytes, void* stream);
/* the same backward with gx = <input gradient> + gx_addend (fp32 [num_nodes, widths[0]]; needs an fp32 gx and
 * bf16_gather == 0): the addition rides in the transposed aggregation's epilogue (kagnn_aggregate_sum_add)            */
int kagnn_gin_kan_layer_bwd_add(const float* gy, int64_t ldgy, int64_t num_nodes, const int32_t* rowptr_t,
                                const int32_t* col_t, const int32_t* hub_seg_t, int64_t num_hub_seg_t,
                                int32_t hub_threshold, float self_scale, int32_t num_layers, const int32_t* widths,
                                const float* const* spline_weight, const float* const* spline_scaler,
                                const float* knots, int32_t grid_size, int32_t spline_order, int32_t mode,
                                const float* const* acts, const void* const* pack_dx, void* gx, int32_t gx_dtype,
                                int64_t ldgx, int32_t bf16_gather, const float* gx_addend, int64_t ld_addend,
                                float* const* g_base_weight, float* const* g_spline_weight,
                                float* const* g_spline_scaler, void* workspace, size_t workspace_bytes, void* stream);

/* The backward of  BatchNorm1d(KAN(aggregate(x)))  in TRAINING mode -- a convolution plus the norm that follows it in every
 * node model (reference node_classification_clean/models.py:198-200) -- given g = d loss / d (norm output) [N, widths[L]]:
 * the norm's statistics pass (column sums -> g_bn_weight, g_bn_bias; either may be NULL), then the chain's backward with the
 * norm's element-wise backward applied INSIDE the last layer's input-gradient kernel (cubic layers of <= 8 coefficients with
 * 32 / 64 outputs; other shapes run the stand-alone pass), then the transposed aggregation (+ gx_addend).  y = the norm's
 * input (= the chain's output), bn_mean / bn_rstd = the batch statistics kagnn_batchnorm_fwd saved, bn_weight NULL = no
 * affine.  Workspace: the backward size of kagnn_gin_kan_layer_workspace_bytes PLUS
 * kagnn_gin_kan_layer_bwd_bn_workspace_bytes(num_nodes, widths[L]).  Needs num_nodes >= 2.                             */
int kagnn_gin_kan_layer_bwd_bn_workspace_bytes(int64_t num_nodes, int32_t out_features, size_t* bytes_host);
int kagnn_gin_kan_layer_bwd_bn(const float* g, int64_t ldg, const float* y, int64_t ldy, const float* bn_weight,
                               const float* bn_mean, const float* bn_rstd, float* g_bn_weight, float* g_bn_bias,
                               int64_t num_nodes, const int32_t* rowptr_t, const int32_t* col_t,
                               const int32_t* hub_seg_t, int64_t num_hub_seg_t, int32_t hub_threshold, float self_scale,
                               int32_t num_layers, const int32_t* widths, const float* const* spline_weight,
                               const float* const* spline_scaler, const float* knots, int32_t grid_size,
                               int32_t spline_order, int32_t mode, const float* const* acts,
                               const void* const* pack_dx, void* gx, int32_t gx_dtype, int64_t ldgx, int32_t bf16_gather,
                               const float* gx_addend, int64_t ld_addend, float* const* g_base_weight,
                               float* const* g_spline_weight, float* const* g_spline_scaler, void* workspace,
                               size_t workspace_bytes, void* stream);

/* kagnn_gin_kan_layer_bwd_bn with the norms' backward STATISTICS travelling with the gradients (round 4; reference
 * node_classification_clean/models.py:198-200 -- in the node models the gradient g arriving at layer l's norm is what layer
 * l+1's transposed aggregation produces, plus the skip gradient added there):
 *   prev_y [N, widths[0]] (ld_prev_y), prev_mean, prev_rstd, prev_sums -- all four or none: this call's transposed aggregation
 *     ALSO leaves prev_sums[0][widths[0]] = sum_n gx, prev_sums[1][widths[0]] = sum_n gx * xhat, xhat = (prev_y - prev_mean) *
 *     prev_rstd, where prev_y is the PREVIOUS norm's input (this convolution's input before the folded affine): the two column
 *     sums that norm's backward starts from.  Partial sums in the row kernel's epilogue (one row pair per workgroup, hub rows
 *     from the merge kernel), folded in a fixed order: deterministic.  Needs an fp32 gx, 17..256 input features (a multiple of
 *     4), 16-byte aligned rows; otherwise KAGNN_ERR_UNSUPPORTED.
 *   bn_sums [2][widths[L]] or NULL: the same two sums for THIS norm, made by the next layer's call -- the statistics pass over
 *     g and y is skipped when the norm's element-wise backward runs inside the input-gradient kernel (ignored otherwise).
 * Workspace: kagnn_gin_kan_layer_bwd_bn's, plus kagnn_gin_kan_layer_bwd_bn_sums_workspace_bytes when prev_sums is given.  */
int kagnn_gin_kan_layer_bwd_bn_sums_workspace_bytes(int64_t num_nodes, int32_t in_features, int64_t num_hub_seg_t,
                                                    size_t* bytes_host);
int kagnn_gin_kan_layer_bwd_bn_sums(const float* g, int64_t ldg, const float* y, int64_t ldy, const float* bn_weight,
                                    const float* bn_mean, const float* bn_rstd, float* g_bn_weight, float* g_bn_bias,
                                    const float* bn_sums,
                                    const float* prev_y, int64_t ld_prev_y, const float* prev_mean, const float* prev_rstd,
                                    float* prev_sums,
                                    int64_t num_nodes, const int32_t* rowptr_t, const int32_t* col_t,
                                    const int32_t* hub_seg_t, int64_t num_hub_seg_t, int32_t hub_threshold, float self_scale,
                                    int32_t num_layers, const int32_t* widths, const float* const* spline_weight,
                                    const float* const* spline_scaler, const float* knots, int32_t grid_size,
                                    int32_t spline_order, int32_t mode, const float* const* acts,
                                    const void* const* pack_dx, void* gx, int32_t gx_dtype, int64_t ldgx, int32_t bf16_gather,
                                    const float* gx_addend, int64_t ld_addend, float* const* g_base_weight,
                                    float* const* g_spline_weight, float* const* g_spline_scaler, void* workspace,
                                    size_t workspace_bytes, void* stream);

/* Embedding-table encoders of the graph-level models (reference graph_regression/models.py:244-281, AtomEncoder / BondEncoder:
 * out = sum over the integer feature columns of one table lookup each), one launch per column each way:
 *   fwd: out[i, :] (+)= table[index[i * index_stride], :]      (accumulate != 0: add to out -- the 2nd, 3rd, ... column)
 *   bwd: g_table[v, :] = sum_{i : index[i * index_stride] == v} g[i, :]   rows in order inside blocks of 128, blocks in order:
 *        deterministic (no atomics)
 * index: int64 (a column of the [N, columns] feature matrix: index_stride = columns); table / g_table: [V, F] contiguous.  An index
 * outside [0, V) -- a device-side assert in torch.nn.functional.embedding -- gives a NaN row in fwd and is skipped in bwd.          */
int kagnn_embedding_fwd(const int64_t* index, int64_t index_stride, int64_t num_rows, const float* table, int32_t num_embeddings,
                        int32_t num_feat, float* out, int64_t ldo, int32_t accumulate, void* stream);
int kagnn_embedding_bwd_workspace_bytes(int64_t num_rows, int32_t num_embeddings, int32_t num_feat, size_t* bytes_host);
int kagnn_embedding_bwd(const int64_t* index, int64_t index_stride, int64_t num_rows, const float* g, int64_t ldg,
                        int32_t num_embeddings, int32_t num_feat, float* g_table, void* workspace, size_t workspace_bytes,
                        void* stream);      /* num_embeddings <= 512; two launches: per-row-block partial tables, then their sum in block order */

/* ---- one library call per GINE convolution each way (round 5; BASELINE config 4 -- the ZINC-shaped mini-batch step is launch- and
 * host-bound).  Replaces: torch_geometric GINEConv around ekan.KAN + the BatchNorm1d that follows it, reference
 * graph_regression/models.py:98,107-119 (called per mini-batch from optuna_zinc.py:56-66).
 * Forward:  acts[0] = self_scale * x_i + sum_{j->i} relu(x_j + edge_attr[perm[e]]) (kagnn_aggregate_gine), ONE pack launch for the
 *   chain, the chain's KANLinear forwards into acts[1..L]; col_mean / col_m2 (both or NULL): the column moments of acts[L] from the
 *   last kernel's epilogue, for kagnn_batchnorm_fwd(col_mean, col_m2).
 * Backward: g = gradient of the chain's output -- or, with bn_y != NULL, of the training-mode BatchNorm1d that follows (bn_y = the
 *   norm's input = acts[L], bn_mean / bn_rstd as saved by its forward; g_bn_weight / g_bn_bias receive its parameter gradients):
 *   statistics pass, then the norm's element-wise backward inside the last layer's input-gradient kernel where the shape allows it
 *   (kagnn_gin_kan_layer_bwd_bn), dW / dX per layer, then kagnn_aggregate_gine_bwd on the TRANSPOSED structure: gx [N, widths[0]]
 *   (required) and g_edge_attr [E, widths[0]] in original edge order (may be NULL).
 * Same kernels and summation orders as the per-operation entry points: bit-identical results.  fp32 rows, no hub segments (mini-batches
 * of small graphs).  Workspace: kagnn_gin_kan_layer_workspace_bytes(..., num_hub_seg = 0, num_hub_seg_t = 0) forward / backward sizes,
 * the backward plus kagnn_gin_kan_layer_bwd_bn_workspace_bytes(num_nodes, widths[L]) when bn_y is given.                              */
int kagnn_gine_kan_layer_fwd(const float* x, int64_t ldx, const float* edge_attr, int64_t lde, int64_t num_nodes,
                             const int32_t* rowptr, const int32_t* col, const int32_t* perm, float self_scale,
                             int32_t num_layers, const int32_t* widths, const float* const* base_weight,
                             const float* const* spline_weight, const float* const* spline_scaler, const float* knots,
                             int32_t grid_size, int32_t spline_order, int32_t mode, float* const* acts,
                             void* const* pack_fwd, void* const* pack_dx, float* col_mean, float* col_m2,
                             void* workspace, size_t workspace_bytes, void* stream);
int kagnn_gine_kan_layer_bwd(const float* g, int64_t ldg, const float* bn_y, int64_t ld_bn_y, const float* bn_weight,
                             const float* bn_mean, const float* bn_rstd, float* g_bn_weight, float* g_bn_bias,
                             const float* x, int64_t ldx, const float* edge_attr, int64_t lde, int64_t num_nodes,
                             const int32_t* rowptr_t, const int32_t* col_t, const int32_t* perm_t, float self_scale,
                             int32_t num_layers, const int32_t* widths, const float* const* spline_weight,
                             const float* const* spline_scaler, const float* knots, int32_t grid_size,
                             int32_t spline_order, int32_t mode, const float* const* acts, const void* const* pack_dx,
                             float* gx, int64_t ldgx, float* g_edge_attr, int64_t ldge, float* const* g_base_weight,
                             float* const* g_spline_weight, float* const* g_spline_scaler, void* workspace,
                             size_t workspace_bytes, void* stream);

/* ---- the WHOLE message-passing stack of a graph-level model in one call each way (round 5):  num_convs x { GINE convolution around a KAN
 * chain of num_layers layers -> training-mode BatchNorm1d }, every chain hidden -> ... -> hidden with the same `widths` (widths[0] ==
 * widths[num_layers]).  Replaces the loop of reference graph_regression/models.py:107-119
 * (`for i in range(n_layers): x = self.bn[i](self.conv[i](x, edge_index, edge_attr))`, dropout 0) as called per mini-batch from
 * optuna_zinc.py:56-66: on a 256-molecule batch a convolution is ~100 us of device work, and as one tape node per convolution the host
 * spent about as long per node each way on argument marshalling -- the step was host-bound at twice its device time.
 * Forward: ONE pack launch for all layers of all convolutions, then per convolution kagnn_aggregate_gine, the chain (column moments
 * from the last kernel) and the normalising pass -> h[i] (the next convolution's input).  Backward: g = gradient of h[num_convs - 1];
 * per convolution, last first: the norm's statistics pass, its element-wise backward inside the last input-gradient kernel, dW / dX,
 * kagnn_aggregate_gine_bwd -> the previous convolution's g; gx = gradient of x; the edge-attribute gradients of all convolutions add
 * up in g_edge_attr [E, widths[0]] (may be NULL).  Same kernels and summation orders as num_convs calls of
 * kagnn_gine_kan_layer_fwd / _bwd: bit-identical.
 * Arrays: per layer, convolution-major [num_convs * num_layers]: base_weight, spline_weight, spline_scaler, pack_fwd, pack_dx, g_*;
 * acts [num_convs * (num_layers + 1)] (acts[i * (L + 1)] = the aggregated input, ... + L = the chain's output = the norm's input);
 * per convolution [num_convs]: self_scale, momentum, eps (HOST floats); bn_weight, bn_bias, running_mean, running_var (the last two
 * arrays or entries may be NULL), h, save_mean, save_rstd, g_bn_weight, g_bn_bias (device).  num_nodes >= 2.                          */
int kagnn_gine_kan_stack_workspace_bytes(int64_t num_nodes, int32_t num_convs, int32_t num_layers, const int32_t* widths,
                                         int32_t grid_size, int32_t spline_order, int32_t mode, size_t* fwd_bytes_host,
                                         size_t* bwd_bytes_host);
int kagnn_gine_kan_stack_fwd(const float* x, int64_t ldx, const float* edge_attr, int64_t lde, int64_t num_nodes,
                             const int32_t* rowptr, const int32_t* col, const int32_t* perm, const float* self_scale,
                             int32_t num_convs, int32_t num_layers, const int32_t* widths, const float* const* base_weight,
                             const float* const* spline_weight, const float* const* spline_scaler, const float* knots,
                             int32_t grid_size, int32_t spline_order, int32_t mode, float* const* acts, void* const* pack_fwd,
                             void* const* pack_dx, const float* const* bn_weight, const float* const* bn_bias,
                             float* const* running_mean, float* const* running_var, const float* momentum, const float* eps,
                             float* const* h, float* const* save_mean, float* const* save_rstd, void* workspace,
                             size_t workspace_bytes, void* stream);
int kagnn_gine_kan_stack_bwd(const float* g, int64_t ldg, const float* x, int64_t ldx, const float* edge_attr, int64_t lde,
                             int64_t num_nodes, const int32_t* rowptr_t, const int32_t* col_t, const int32_t* perm_t,
                             const float* self_scale, int32_t num_convs, int32_t num_layers, const int32_t* widths,
                             const float* const* spline_weight, const float* const* spline_scaler, const float* knots,
                             int32_t grid_size, int32_t spline_order, int32_t mode, const float* const* acts,
                             const void* const* pack_dx, const float* const* h, const float* const* bn_weight,
                             const float* const* save_mean, const float* const* save_rstd, float* gx, int64_t ldgx,
                             float* g_edge_attr, int64_t ldge, float* const* g_bn_weight, float* const* g_bn_bias,
                             float* const* g_base_weight, float* const* g_spline_weight, float* const* g_spline_scaler,
                             void* workspace, size_t workspace_bytes, void* stream);

/* kagnn_kan_linear_bwd_input_affine that ALSO leaves the two column sums the backward of the folded BatchNorm1d starts from (round 4):
 * sums[0][in] = sum_n gx, sums[1][in] = sum_n gx * xhat, xhat = (x - bn_mean) * bn_rstd on the raw rows x (= that norm's input).
 * The read-out's gradient of the LAST convolution's output in the node models (node_classification_clean/models.py:198-203): that
 * norm's incoming gradient IS this gx, so its statistics pass goes away (hand `sums` to kagnn_gin_kan_layer_bwd_bn_sums as bn_sums).
 * Per workgroup one partial row pair (lanes add their 8 rows, the row groups meet by shuffles, the waves in LDS in order), folded by
 * one launch: deterministic.  Covered (kagnn_kan_bwd_input_sums_ok == 1): split precision, cubic layers of <= 8 coefficients,
 * <= 64 inputs (a multiple of 4) and outputs, >= 32768 rows; else KAGNN_ERR_UNSUPPORTED.                                        */
int kagnn_kan_bwd_input_sums_ok(int64_t num_rows, int32_t in_features, int32_t out_features, int32_t grid_size, int32_t spline_order,
                                int32_t mode);
int kagnn_kan_bwd_input_sums_workspace_bytes(int64_t num_rows, int32_t in_features, size_t* bytes_host);
int kagnn_kan_linear_bwd_input_affine_sums(const float* x, int64_t ldx, const float* x_affine, const float* bn_mean,
                                           const float* bn_rstd, const float* gy, int64_t ldgy, int64_t num_rows, const float* knots,
                                           int32_t in_features, int32_t out_features, int32_t grid_size, int32_t spline_order,
                                           int32_t mode, const void* pack_dx, float* gx, int64_t ldgx, float* sums,
                                           void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Adaptive grids.  Replaces the device work of KANLinear.update_grid (ekan.py:164-211) and the dense
 * b_splines (:79-112).  `grid*` are whole grid buffers [in, G+2k+1] with increasing rows.
 * ------------------------------------------------------------------------------------------ */
/* bases[N, in, G+k] = b_splines(x) on per-feature knot rows (ekan.py:79-112).                  */
int kagnn_kan_bsplines(const float* x, int64_t ldx, int64_t num_rows, const float* grid,
                       int32_t in_features, int32_t grid_size, int32_t spline_order, float* bases,
                       void* stream);

/* the coefficient refit of update_grid (ekan.py:169-177 + curve2coeff :114-144 + :211):
 *   new_spline_weight[o,f,:] = argmin_s || bases_new(x[:,f]) s - bases_old(x[:,f]) (spline_weight*scaler)[o,f,:] ||
 * through per-feature fp64 Gram matrices (one streaming pass over x; no [N,in,out] intermediate), solved by
 * Cholesky; a basis with no sample in its support gets coefficient 0.  G+k <= 16.  Not in place.           */
int kagnn_kan_grid_refit_workspace_bytes(int64_t num_rows, int32_t in_features, int32_t grid_size,
                                         int32_t spline_order, size_t* bytes_host);
int kagnn_kan_grid_refit(const float* x, int64_t ldx, int64_t num_rows, const float* grid_old,
                         const float* grid_new, int32_t in_features, int32_t out_features,
                         int32_t grid_size, int32_t spline_order, const float* spline_weight,
                         const float* spline_scaler, float* new_spline_weight, void* workspace,
                         size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * FastKAN layer.  Replaces FastKANLayer.forward (node_classification_clean/fastkan.py:76-85:
 * LayerNorm :77-78, RadialBasisFunction :46-47, SplineLinear :81, base_linear(silu(x)) :82-84)
 * and its autograd backward.  centers = the module's `rbf.grid` parameter (num_grids fp32
 * values, device pointer), denominator as in fastkan.py:44.  `precision` as for the KAN layer
 * (KAGNN_PREC_SPLIT covers num_grids <= 16; the host sums wider layers over groups of centres).  ln_weight/ln_bias NULL = no layernorm; base_weight NULL = no base branch.
 *   spline_weight [out, in*num_grids]  (in major, grid minor), base_weight [out,in], base_bias [out]
 * ------------------------------------------------------------------------------------------ */
int kagnn_fastkan_fwd_workspace_bytes(int64_t num_rows, int32_t in_features,
                                      int32_t out_features, int32_t num_grids,
                                      int32_t precision, size_t* bytes_host);
int kagnn_fastkan_fwd(const float* x, int64_t ldx, int64_t num_rows, int32_t in_features,
                      int32_t out_features, int32_t num_grids, const float* centers,
                      float denominator, const float* ln_weight, const float* ln_bias,
                      float ln_eps, const float* spline_weight, const float* base_weight,
                      const float* base_bias, float* y, int64_t ldy,
                      float* row_stats /* [N,2] = mean, rstd; required when ln_weight != NULL */,
                      int32_t precision, void* workspace, size_t workspace_bytes, void* stream);
int kagnn_fastkan_bwd_workspace_bytes(int64_t num_rows, int32_t in_features,
                                      int32_t out_features, int32_t num_grids,
                                      int32_t precision, size_t* bytes_host);
/* all gradients of one layer: gx[N,in], g_ln_weight[in], g_ln_bias[in],
 * g_spline_weight[out,in*num_grids], g_base_weight[out,in], g_base_bias[out]; `row_stats` is the
 * array kagnn_fastkan_fwd wrote.  No gradient is produced for `centers` (rbf.grid has
 * requires_grad=False, fastkan.py:43).                                                      */
int kagnn_fastkan_bwd(const float* x, int64_t ldx, const float* gy, int64_t ldgy,
                      int64_t num_rows, int32_t in_features, int32_t out_features,
                      int32_t num_grids, const float* centers, float denominator,
                      const float* ln_weight, const float* ln_bias, float ln_eps,
                      const float* spline_weight, const float* base_weight,
                      const float* row_stats, float* gx, int64_t ldgx, float* g_ln_weight,
                      float* g_ln_bias, float* g_spline_weight, float* g_base_weight,
                      float* g_base_bias, int32_t precision, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Feature-sharded FastKAN layer (no counterpart in the reference, which is single-device: this is
 * FastKANLayer.forward, fastkan.py:76-85, when a rank holds `in_features` of the row's P * in_features input columns and the
 * matching slices of layernorm.weight / .bias, spline_linear.weight[:, columns * num_grids], base_linear.weight[:, columns]).
 * LayerNorm (fastkan.py:77-78) is the one operation of the layer that reduces over the sharded axis; the exchange is
 * 2 floats per row each way:
 *   forward : kagnn_fastkan_row_moments  -> moments[n] = (mean, sum of squared deviations from it) over the local columns;
 *             the caller gathers the P arrays ([P][N][2], rank order);  kagnn_fastkan_merge_moments merges a row's P pairs
 *             in rank order (Chan's update: no cancellation, same bits on every rank) -> row_stats[n] = (mean, rstd) over all
 *             P * in_features columns;  kagnn_fastkan_shard_fwd = kagnn_fastkan_fwd with row_stats GIVEN: y = this rank's
 *             partial sums over its columns for ALL outputs (base_bias on one rank only); the caller reduce-scatters them.
 *   backward: kagnn_fastkan_shard_bwd (gy = the gathered [N, out] gradient) = everything but the LayerNorm backward:
 *             g_spline_weight / g_base_weight / g_base_bias (may be NULL) of the local slices, gx = the base-branch part,
 *             d loss / dz kept in `workspace`, row_sums[n] = (sum_f gz*gamma, sum_f gz*gamma*zhat) over the local columns;
 *             the caller sums row_sums over the ranks (all-reduce);  kagnn_fastkan_shard_bwd_finish completes gx and writes
 *             g_ln_weight / g_ln_bias -- SAME shape arguments and SAME workspace (kagnn_fastkan_bwd_workspace_bytes), not
 *             touched between the two calls.  Without layernorm (ln_weight NULL) shard_bwd alone is the whole backward.
 * ------------------------------------------------------------------------------------------ */
int kagnn_fastkan_row_moments(const float* x, int64_t ldx, int64_t num_rows, int32_t in_features,
                              float* moments /* [N,2] */, void* stream);
int kagnn_fastkan_merge_moments(const float* gathered /* [P][N][2] */, int32_t num_ranks, int64_t num_rows,
                                int32_t in_features /* per rank */, float ln_eps, float* row_stats /* [N,2] */,
                                void* stream);
int kagnn_fastkan_shard_fwd(const float* x, int64_t ldx, int64_t num_rows, int32_t in_features,
                            int32_t out_features, int32_t num_grids, const float* centers,
                            float denominator, const float* ln_weight, const float* ln_bias,
                            const float* row_stats, const float* spline_weight, const float* base_weight,
                            const float* base_bias, float* y, int64_t ldy, int32_t precision,
                            void* workspace, size_t workspace_bytes, void* stream);
int kagnn_fastkan_shard_bwd(const float* x, int64_t ldx, const float* gy, int64_t ldgy,
                            int64_t num_rows, int32_t in_features, int32_t out_features,
                            int32_t num_grids, const float* centers, float denominator,
                            const float* ln_weight, const float* ln_bias, const float* spline_weight,
                            const float* base_weight, const float* row_stats, float* gx, int64_t ldgx,
                            float* row_sums /* [N,2] */, float* g_spline_weight, float* g_base_weight,
                            float* g_base_bias,
                            int32_t parts /* 1 = input-gradient half (gx, dz, row_sums), 2 = weight-gradient half, 3 = both:
                                             two calls (1 then 2) let the all-reduce of row_sums run beside the weight gradient */,
                            int32_t precision, void* workspace, size_t workspace_bytes, void* stream);
int kagnn_fastkan_shard_bwd_finish(const float* x, int64_t ldx, int64_t num_rows, int32_t in_features,
                                   int32_t in_features_total, int32_t out_features, int32_t num_grids,
                                   const float* ln_weight, const float* ln_bias, const float* row_stats,
                                   const float* row_sums, float* gx, int64_t ldgx, float* g_ln_weight,
                                   float* g_ln_bias, int32_t precision, void* workspace,
                                   size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm1d over node rows (the epilogue after every convolution of the node / graph models:
 * node_classification_clean/models.py:195-202, torch.nn.BatchNorm1d semantics).  x, y, gy, gx are
 * [num_rows, num_features] row-major with leading dimensions in elements; statistics per column.
 *   training != 0: batch statistics (biased variance) normalise; running_mean / running_var (may be
 *                  NULL) are updated with `momentum` (unbiased variance), save_mean / save_rstd
 *                  receive the batch mean and 1/sqrt(var + eps) for the backward.
 *   training == 0: running statistics normalise; save_mean / save_rstd receive running_mean and
 *                  1/sqrt(running_var + eps).
 * weight / bias may be NULL (affine=False).  The backward writes gx (may be NULL), g_weight and
 * g_bias (either may be NULL) for the same `training` flag as the forward.  Deterministic.
 *
 * Fused epilogue (models.py:198-201: `x = dropout(bn(conv(x)))`):
 *   col_mean / col_m2 (both or neither, training only): the column moments of x from the kernel that produced
 *                  it (kagnn_kan_linear_fwd_moments, kagnn_gin_kan_layer_fwd) -- the statistics pass is skipped;
 *   dropout_p > 0 (training only): y = dropout(bn(x), p) in the same pass; element (row, col) is kept with
 *                  probability 1-p (resolution 2^-16) and scaled by 1/(1-p), decided by a counter-based hash of
 *                  (dropout_seed, row, col / 4) -- a pure function of the seed, NOT torch's Philox stream.  The
 *                  backward regenerates the decisions from the same (dropout_p, dropout_seed): `gy` is the
 *                  gradient of the dropped-out y.  No mask tensor exists.
 * ------------------------------------------------------------------------------------------ */
int kagnn_batchnorm_workspace_bytes(int64_t num_rows, int32_t num_features, size_t* bytes_host);
int kagnn_batchnorm_fwd(const float* x, int64_t ldx, int64_t num_rows, int32_t num_features,
                        const float* weight, const float* bias, float* running_mean,
                        float* running_var, float momentum, float eps, int32_t training,
                        const float* col_mean, const float* col_m2, float dropout_p,
                        uint64_t dropout_seed, float* y,
                        int64_t ldy, float* save_mean, float* save_rstd, void* workspace,
                        size_t workspace_bytes, void* stream);
int kagnn_batchnorm_bwd(const float* x, int64_t ldx, const float* gy, int64_t ldgy,
                        int64_t num_rows, int32_t num_features, const float* weight,
                        const float* save_mean, const float* save_rstd, int32_t training,
                        float dropout_p, uint64_t dropout_seed, float* gx,
                        int64_t ldgx, float* g_weight, float* g_bias, void* workspace,
                        size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Attention aggregation of KAGATConv / FASTKAGATConv (node_classification_clean/models.py:39-46,
 * 76-83: torch_geometric GATConv with `lin` = a KAN layer; heads concatenated, negative slope 0.2,
 * existing self loops removed and one added per node, no attention dropout).
 *   xh [N, heads*channels] = lin(x);  att_src / att_dst [heads, channels];  bias [heads*channels] or NULL
 *   kagnn_gat_logits: a_src[N,heads], a_dst[N,heads]
 *   kagnn_gat_fwd   : out [N, heads*channels], and the per-(node, head) softmax maximum / normaliser
 *                     (row_max, row_sum: the only extra state the backward needs)
 *   kagnn_gat_bwd   : gx = d loss / d xh (all three paths), g_src[N,heads] / g_dst[N,heads] = d loss / d a_src,
 *                     a_dst (the caller contracts them with xh for the att_src / att_dst gradients);
 *                     edge_scratch [E*heads] and self_scratch [N*heads] floats.  CSR arrays as built by
 *                     kagnn_csr_build: (rowptr, col, perm) by destination, (rowptr_t, col_t, perm_t) by source;
 *                     hub_seg / num_hub_seg / hub_threshold = the by-destination hub segments (rows with more
 *                     edges get a whole workgroup per head; NULL / 0 disables that).
 * ------------------------------------------------------------------------------------------ */
int kagnn_gat_logits(const float* xh, int64_t ldx, int64_t num_nodes, int32_t heads, int32_t channels,
                     const float* att_src, const float* att_dst, float* a_src, float* a_dst,
                     void* stream);
int kagnn_gat_fwd(const float* xh, int64_t ldx, const float* a_src, const float* a_dst,
                  const int32_t* rowptr, const int32_t* col, int64_t num_nodes, int32_t heads,
                  int32_t channels, const float* bias, float* out, int64_t ldo, float* row_max,
                  float* row_sum, const int32_t* hub_seg, int64_t num_hub_seg, int32_t hub_threshold,
                  void* stream);
int kagnn_gat_bwd(const float* xh, int64_t ldx, const float* gout, int64_t ldg, const float* out,
                  int64_t ldo, const float* bias, const float* a_src, const float* a_dst,
                  const float* row_max, const float* row_sum, const int32_t* rowptr,
                  const int32_t* col, const int32_t* perm, const int32_t* rowptr_t,
                  const int32_t* col_t, const int32_t* perm_t, const float* att_src,
                  const float* att_dst, int64_t num_nodes, int32_t heads, int32_t channels,
                  float* edge_scratch, float* self_scratch, float* g_dst, float* g_src, float* gx,
                  int64_t ldgx, const int32_t* hub_seg, int64_t num_hub_seg, int32_t hub_threshold,
                  void* stream);

/* gradients of GATConv's attention vectors from the per-node terms kagnn_gat_bwd returns:
 *   g_att_src[h,c] = sum_n g_src[n,h] * xh[n, h*C + c],  g_att_dst likewise with g_dst   (heads*channels <= 1024).
 * One pass over xh, deterministic.                                                             */
int kagnn_gat_att_grad_workspace_bytes(int64_t num_nodes, int32_t heads, int32_t channels, size_t* bytes_host);
int kagnn_gat_att_grad(const float* xh, int64_t ld, const float* g_src, const float* g_dst, int64_t num_nodes,
                       int32_t heads, int32_t channels, float* g_att_src, float* g_att_dst,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Loss tail of the timing harness (node_classification_clean/time_model.py:43-45):
 *   out = softmax(logits, dim=1); loss = CrossEntropyLoss()(out[mask], y[mask])
 * pre_softmax = 1 keeps the reference's softmax-before-CrossEntropyLoss (which applies log_softmax again);
 * 0 is plain softmax cross-entropy (utils.py's training loop).  mean over the masked rows; `mask` [N] bytes
 * (torch.bool storage) or NULL = every row; labels int64 in [0, num_classes) (anything else: loss = NaN).
 * loss / count: device scalars (count = number of masked rows, kept for the backward; no host sync);
 * row_stats [N,3] is the only tensor saved.  Deterministic.  g_logits rows outside the mask are written as 0. */
int kagnn_softmax_xent_workspace_bytes(int64_t num_rows, size_t* bytes_host);
int kagnn_softmax_xent_fwd(const float* logits, int64_t ld, int64_t num_rows, int32_t num_classes,
                           const int64_t* labels, const uint8_t* mask, int32_t pre_softmax, float* loss,
                           float* row_stats, float* count, void* workspace, size_t workspace_bytes,
                           void* stream);
int kagnn_softmax_xent_bwd(const float* logits, int64_t ld, int64_t num_rows, int32_t num_classes,
                           const int64_t* labels, const uint8_t* mask, int32_t pre_softmax,
                           const float* row_stats, const float* count, const float* g_loss,
                           float* g_logits, int64_t ldg, void* stream);

/* Loss of the graph-regression scripts (graph_regression/optuna_zinc.py:58: torch.nn.L1Loss()(model(data).squeeze(), data.y)):
 *   loss = mean |pred - target| over n contiguous fp32 elements; g_pred = (g_loss / n) * sign(pred - target).
 * loss / g_loss: device scalars; one launch each way, deterministic (fixed summation order), no host sync. */
int kagnn_l1_loss_fwd(const float* pred, const float* target, int64_t n, float* loss, void* stream);
int kagnn_l1_loss_bwd(const float* pred, const float* target, int64_t n, const float* g_loss, float* g_pred, void* stream);

/* Optimiser of the same scripts (optuna_zinc.py:49,62: torch.optim.Adam(model.parameters(), lr), optimizer.step() per batch): one
 * update of `count` fp32 tensors in one launch per 32 tensors.  HOST arrays of device pointers / element counts; `step` = 1, 2, ...
 * (bias corrections 1 - beta^step are computed on the host in double).  torch's rule without amsgrad:
 *   g += weight_decay * p;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps) */
int kagnn_adam_step(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                    const int64_t* numel, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, void* stream);

/* ------------------------------------------------------------------------------------------
 * Direct peer-to-peer exchange steps of the feature-sharded layer (no reference counterpart: the reference has no
 * multi-GPU code; contract = BASELINE.json north_star, SURVEY.md 8(b)/(e): "sharded variants", "hand-rolled
 * direct P2P over hipIpcMemHandle peer buffers").  `parts` / `shards`: HOST arrays of `world` DEVICE pointers -- entry p is
 * rank p's exchange buffer as mapped into THIS process (hipIpcOpenMemHandle; own entry = own buffer).  The caller orders the
 * ranks (every peer's buffer complete before the call, not overwritten before every reader is done).
 *   reduce_scatter: y[n][c] = sum_p parts[p][n*ld + rank*(out/world) + c], c < out/world, summed in rank order (deterministic)
 *   all_gather:     g[n][p*w + c] = shards[p][n*lds + c]
 * out/world (resp. w) and the leading dimensions must be multiples of 4 floats, bases 16-byte aligned; world <= 16. */
int kagnn_p2p_reduce_scatter(const float* const* parts, int32_t world, int32_t rank, int64_t num_rows, int32_t out_features,
                             int64_t ld, float* y, int64_t ldy, void* stream);
int kagnn_p2p_all_gather(const float* const* shards, int32_t world, int64_t num_rows, int32_t shard_width, int64_t lds,
                         float* g, int64_t ldg, void* stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm1d folded into its consumers (round 4).  Reference: node_classification_clean/models.py:198-203 --
 * `x = self.bns[i](self.convs[i](x, edge_index))` feeds the next GINConv and the skip read-out `lay_out(torch.cat(l, dim=1))`.
 * In training mode without dropout the normalised matrix need not exist: kagnn_batchnorm_stats_affine turns the column
 * moments the convolution's last forward kernel produced into (save_mean, save_rstd, running statistics, per-column affine
 * a = gamma * rstd, b = beta - mean * a), and every consumer reads the convolution's raw output y as  a * y + b:
 *   - the next convolution's aggregation: kagnn_aggregate_sum_affine / kagnn_gin_kan_layer_fwd_affine
 *       out_i = a * (self_scale * y_i + sum_{j->i} y_j) + (self_scale + deg_i) * b      (unit edge weights)
 *   - the read-out: kagnn_kan_linear_fwd_parts_affine (per column block), kagnn_kan_linear_bwd_input_affine,
 *     kagnn_kan_linear_bwd_weight_affine (cubic layers, <= 8 coefficients, <= 64 outputs, split precision).
 * Gradients are taken with respect to the NORMALISED input; the norm's own backward (kagnn_gin_kan_layer_bwd_bn,
 * kagnn_batchnorm_bwd) goes on from there, so nothing else changes.  `affine` arrays: 2 * width floats, scales then shifts,
 * 16-byte aligned.
 */
int kagnn_batchnorm_stats_affine(const float* col_mean, const float* col_m2, int64_t num_rows, int32_t num_feat,
                                 const float* gamma, const float* beta, float* running_mean, float* running_var,
                                 float momentum, float eps, float* save_mean, float* save_rstd, float* affine, void* stream);
int kagnn_aggregate_sum_affine(const float* x, int64_t ldx, float* out, int64_t ldo, const int32_t* rowptr, const int32_t* col,
                               int64_t num_nodes, int32_t num_feat, float self_scale, const float* col_scale,
                               const float* col_shift, const int32_t* hub_seg, int64_t num_hub_seg, int32_t hub_threshold,
                               const float* addend, int64_t ld_addend, void* workspace, size_t workspace_bytes, void* stream);
int kagnn_gin_kan_layer_fwd_affine(const float* x, int64_t ldx, int64_t num_nodes, const int32_t* rowptr, const int32_t* col,
                                   const int32_t* hub_seg, int64_t num_hub_seg, int32_t hub_threshold, float self_scale,
                                   const float* in_col_scale, const float* in_col_shift,
                                   int32_t num_layers, const int32_t* widths, const float* const* base_weight,
                                   const float* const* spline_weight, const float* const* spline_scaler, const float* knots,
                                   int32_t grid_size, int32_t spline_order, int32_t mode, float* const* acts,
                                   void* const* pack_fwd, void* const* pack_dx, float* col_mean, float* col_m2,
                                   void* workspace, size_t workspace_bytes, void* stream);
int kagnn_kan_linear_fwd_parts_affine(const float* const* x_parts, const int32_t* part_widths, const int64_t* part_ld,
                                      const float* const* part_affine, int32_t num_parts, int64_t num_rows, const float* knots,
                                      int32_t in_features, int32_t out_features, int32_t grid_size, int32_t spline_order,
                                      int32_t mode, const void* pack_fwd, float* y, int64_t ldy, void* workspace,
                                      size_t workspace_bytes, void* stream);
int kagnn_kan_linear_bwd_input_affine(const float* x, int64_t ldx, const float* x_affine, const float* gy, int64_t ldgy,
                                      int64_t num_rows, const float* knots, int32_t in_features, int32_t out_features,
                                      int32_t grid_size, int32_t spline_order, int32_t mode, const void* pack_dx, void* gx,
                                      int64_t ldgx, int32_t gx_dtype, void* stream);
int kagnn_kan_linear_bwd_weight_affine(const float* x, int64_t ldx, const float* x_affine, const float* gy, int64_t ldgy,
                                       int64_t num_rows, const float* knots, int32_t in_features, int32_t out_features,
                                       int32_t grid_size, int32_t spline_order, int32_t mode, const float* spline_weight,
                                       const float* spline_scaler, float* g_base_weight, float* g_spline_weight,
                                       float* g_spline_scaler, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * The WHOLE graph-regression model per call (round 6; BASELINE config 4).  Replaces, per mini-batch, `KAGIN.forward`
 * of the reference's graph_regression/models.py:107-119 -- `AtomEncoder` / `BondEncoder` embedding sums (:244-281),
 * `n_layers x {GINEConv(KAN) -> BatchNorm1d}` (:98,108-114), `global_add_pool` (:117), the KAN read-out (:118) -- and its
 * autograd backward, called from graph_regression/optuna_zinc.py:56-66.  No new kernels: the call sequences this header's own
 * entry points (kagnn_embedding_fwd, kagnn_gine_kan_stack_fwd, kagnn_segment_pool, kagnn_kan_pack_batch / kagnn_kan_pack,
 * kagnn_kan_linear_fwd; on the way back kagnn_kan_linear_bwd_input / _bwd_weight, kagnn_segment_broadcast,
 * kagnn_gine_kan_stack_bwd, kagnn_embedding_bwd) on `stream` in exactly the order and with exactly the arguments the
 * per-operation host code uses -- same bits.  What it removes is the HOST: a 256-molecule step is ~0.75 ms of device work and
 * the per-operation binding spent as long again in ~19 calls, ~45 allocations and their pointer tables.  Everything the backward
 * needs lives in ONE caller-owned `saved` buffer whose layout only the library knows (kagnn_kagin_model_sizes); all parameter
 * gradients land in ONE flat fp32 buffer `grads`, in the order: atom tables, bond tables, then per convolution bn_weight, bn_bias
 * and per layer base_weight, spline_weight, spline_scaler, then per read-out layer base_weight, spline_weight, spline_scaler
 * (absent scalers take no room).  All fields are 8 bytes wide except the three float arrays at the end.
 * Limits: num_convs * num_layers <= 16, hidden -> hidden chains of width <= 64 on the split path (kagnn_gine_kan_stack_*),
 * <= 8 read-out layers, <= 16 tables per encoder with <= 512 rows, training-mode affine BatchNorm1d, >= 2 nodes.          */
#define KAGNN_MODEL_MAX_LAYERS 16
#define KAGNN_MODEL_MAX_CONVS 16
#define KAGNN_MODEL_MAX_TABLES 16
#define KAGNN_MODEL_MAX_READOUT 8
typedef struct kagnn_kagin_model {
    int64_t num_nodes, num_edges, num_graphs, hidden;
    int64_t num_atom_tables, num_bond_tables, x_stride, e_stride;     /* index matrices: int64 [N, x_stride] / [E, e_stride]; table t reads column t */
    int64_t num_convs, num_layers, grid_size, spline_order, mode;
    int64_t num_readout, readout_grid_size, readout_spline_order;
    int64_t readout_widths[KAGNN_MODEL_MAX_READOUT + 1];               /* hidden, ..., outputs */
    int64_t readout_modes[KAGNN_MODEL_MAX_READOUT];
    int64_t atom_rows[KAGNN_MODEL_MAX_TABLES], bond_rows[KAGNN_MODEL_MAX_TABLES];
    const int64_t* x_index; const int64_t* e_index;
    const float* atom_table[KAGNN_MODEL_MAX_TABLES]; const float* bond_table[KAGNN_MODEL_MAX_TABLES];
    const int32_t* rowptr; const int32_t* col; const int32_t* perm;        /* CSR by destination (kagnn_csr_build_small) */
    const int32_t* rowptr_t; const int32_t* col_t; const int32_t* perm_t;  /* ... and its transpose */
    const int32_t* seg_ptr;                                                /* [num_graphs + 1] node offsets */
    /* edge_src != NULL: the library builds the CSR + transpose of THIS batch itself (kagnn_csr_build_small: num_edges, num_nodes
     * <= 65 536) into `saved`, on a stream of its own beside the encoders and the weight packs, and ignores the six arrays above;
     * csr_flags (2 ints, device) receives the out-of-range-id flags of that build for the caller's validation                    */
    const int64_t* edge_src; const int64_t* edge_dst; int32_t* csr_flags;
    const float* knots;
    const float* base_weight[KAGNN_MODEL_MAX_LAYERS]; const float* spline_weight[KAGNN_MODEL_MAX_LAYERS];
    const float* spline_scaler[KAGNN_MODEL_MAX_LAYERS];
    const float* bn_weight[KAGNN_MODEL_MAX_CONVS]; const float* bn_bias[KAGNN_MODEL_MAX_CONVS];
    float* running_mean[KAGNN_MODEL_MAX_CONVS]; float* running_var[KAGNN_MODEL_MAX_CONVS];      /* NULL: not tracked */
    const float* readout_knots[KAGNN_MODEL_MAX_READOUT];
    const float* readout_base_weight[KAGNN_MODEL_MAX_READOUT]; const float* readout_spline_weight[KAGNN_MODEL_MAX_READOUT];
    const float* readout_spline_scaler[KAGNN_MODEL_MAX_READOUT];      /* NULL entries: no scaler */
    void* saved; int64_t saved_bytes;                                 /* forward writes, backward reads */
    void* workspace; int64_t workspace_bytes;                         /* scratch of the call (forward / backward sizes differ) */
    float* out;                                                       /* forward: [num_graphs, readout_widths[num_readout]] */
    const float* g_out; int64_t ld_g_out;                             /* backward: gradient of `out` */
    float* grads;                                                     /* backward: all parameter gradients, flat */
    float self_scale[KAGNN_MODEL_MAX_CONVS];                          /* 1 + eps of every GINE convolution */
    float momentum[KAGNN_MODEL_MAX_CONVS]; float eps[KAGNN_MODEL_MAX_CONVS];
} kagnn_kagin_model_t;
int kagnn_kagin_model_struct_bytes(void);                             /* sizeof(kagnn_kagin_model_t): a binding checks its mirror */
int kagnn_kagin_model_sizes(const kagnn_kagin_model_t* model, size_t* saved_bytes_host, size_t* fwd_workspace_bytes_host,
                            size_t* bwd_workspace_bytes_host, size_t* grads_floats_host);
int kagnn_kagin_model_fwd(const kagnn_kagin_model_t* model, void* stream);
int kagnn_kagin_model_bwd(const kagnn_kagin_model_t* model, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KAGNN_HIP_H */
