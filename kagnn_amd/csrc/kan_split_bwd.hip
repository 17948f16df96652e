// split-precision backward kernels of the efficient-KAN layer (bring-up: not enabled yet; api.hip
// asks kan_split_dx_ok / kan_split_dw_ok and routes to the exact-fp32 kernels when they say no).
#include "common.h"
namespace kagnn {
size_t kan_split_pack_dx_bytes(int, int, int) { return 0; }
size_t kan_split_dw_ws_bytes(long, int, int, int) { return 0; }
int kan_split_pack_dx(const float*, const float*, const float*, int, int, int, void*, hipStream_t) { return fail(KAGNN_ERR_UNSUPPORTED, "%s: not built", "kan_split_pack_dx"); }
int kan_split_dx(const float*, long, const float*, long, long, const float*, int, int, int, int, const void*, float*, long, hipStream_t) { return fail(KAGNN_ERR_UNSUPPORTED, "%s: not built", "kan_split_dx"); }
int kan_split_dw(const float*, long, const float*, long, long, const float*, int, int, int, int, const float*, const float*, float*, float*, float*, float*, size_t, hipStream_t) { return fail(KAGNN_ERR_UNSUPPORTED, "%s: not built", "kan_split_dw"); }
}  // namespace kagnn
