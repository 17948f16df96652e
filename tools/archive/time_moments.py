"""GPU box: KANLinear forward with / without the column-moments epilogue vs the separate statistics pass (1M x 64 -> 64)."""
import torch
import kagnn_amd
from kagnn_amd import ops

dev = "cuda:0"
torch.manual_seed(0)
layer = kagnn_amd.KANLinear(64, 64, grid_size=5, spline_order=3).to(dev)
x = (torch.randn(1_000_000, 64) * 0.5).to(dev)
args = (layer.base_weight.contiguous(), layer.spline_weight.contiguous(), layer.spline_scaler.contiguous(), layer._knots(), 5, 3,
        ops.PREC_SPLIT)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


y = ops._kan_fwd_raw(x, *args)[0]
rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
print("forward              ms", round(timed(lambda: ops._kan_fwd_raw(x, *args)), 4))
print("forward + moments    ms", round(timed(lambda: ops._kan_fwd_raw(x, *args, moments=True)), 4))
mom = ops._kan_fwd_raw(x, *args, moments=True)[2]
print("batch_norm (own statistics pass) ms", round(timed(lambda: ops._batchnorm_fwd_raw(y, None, None, rm, rv, True, 0.1, 1e-5)), 4))
print("batch_norm (given moments)       ms", round(timed(lambda: ops._batchnorm_fwd_raw(y, None, None, rm, rv, True, 0.1, 1e-5, mom)), 4))
print("batch_norm + dropout 0.5         ms", round(timed(lambda: ops._batchnorm_fwd_raw(y, None, None, rm, rv, True, 0.1, 1e-5, mom, 0.5, 7)), 4))
print("aten dropout 0.5 on [1M, 64]     ms", round(timed(lambda: torch.nn.functional.dropout(y, 0.5, True)), 4))
