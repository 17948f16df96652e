// efficient-KAN layer, split-precision mode (KAGNN_PREC_SPLIT) -- placeholder translation unit
// during bring-up: kan_split_supported() answers "no", so api.hip routes every shape to the
// exact-fp32 MFMA kernels in kan_fp32.hip.
#include "common.h"
namespace kagnn {
bool kan_split_supported(int, int, int, int) { return false; }
size_t kan_split_pack_fwd_bytes(int, int, int) { return 0; }
size_t kan_split_pack_dx_bytes(int, int, int) { return 0; }
size_t kan_split_dw_ws_bytes(long, int, int, int) { return 0; }
int kan_split_pack(const float*, const float*, const float*, int, int, int, void*, void*, hipStream_t) { return fail(KAGNN_ERR_UNSUPPORTED, "%s: split mode not built", "kan_split_pack"); }
int kan_split_fwd(const float*, long, long, const float*, int, int, int, int, const void*, float*, long, hipStream_t) { return fail(KAGNN_ERR_UNSUPPORTED, "%s: split mode not built", "kan_split_fwd"); }
int kan_split_dx(const float*, long, const float*, long, long, const float*, int, int, int, int, const void*, float*, long, hipStream_t) { return fail(KAGNN_ERR_UNSUPPORTED, "%s: split mode not built", "kan_split_dx"); }
int kan_split_dw(const float*, long, const float*, long, long, const float*, int, int, int, int, const float*, const float*, float*, float*, float*, float*, size_t, hipStream_t) { return fail(KAGNN_ERR_UNSUPPORTED, "%s: split mode not built", "kan_split_dw"); }
}  // namespace kagnn
