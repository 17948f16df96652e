import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import kagnn_amd
from kagnn_amd import ops
from oracle import kan_oracle as orc
z = np.load('tests/golden/g9_harness.npz')
T = lambda a: torch.from_numpy(np.asarray(a))
n, e, fin, hid, classes, G, k = [int(v) for v in z["cfg"]]
kind = 'gin'
model = kagnn_amd.GKAN_Nodes(kind, 2, fin, hid, classes, skip=True, grid_size=G, spline_order=k, hidden_layers=2)
pre = f"{kind}.init."
sd = {n_[len(pre):]: T(z[n_]) for n_ in z.files if n_.startswith(pre)}
model.load_state_dict(sd)
model = model.cuda().train()
x, ei = T(z["x"]), T(z["edge_index"])
xd, eid = x.cuda(), ei.cuda()
gi = ops.GraphIndex(eid, n)
# stage by stage
h_gpu = ops.aggregate_sum(xd, gi, self_scale=1.0)
h_cpu = orc.sum_aggregate(x, ei) + x
print("agg", float((h_gpu.cpu() - h_cpu).abs().max()))
def layer_p(prefix):
    return {q: sd[prefix + q] for q in ("base_weight", "spline_weight", "spline_scaler", "grid")}
l0, l1 = layer_p("convs.0.nn.layers.0."), layer_p("convs.0.nn.layers.1.")
a_cpu = orc.kan_linear_forward(h_cpu, l0["base_weight"], l0["spline_weight"], l0["spline_scaler"], l0["grid"], k)
a_gpu = model.convs[0].nn.layers[0](h_gpu)
print("kan0", float((a_gpu.detach().cpu() - a_cpu).abs().max()), float(a_cpu.abs().max()), float(h_cpu.abs().max()))
for mode in (0, 1):
    model.convs[0].nn.layers[0].precision = mode
    a_g = model.convs[0].nn.layers[0](h_gpu)
    d = (a_g.detach().cpu() - a_cpu).abs()
    print(" mode", mode, float(d.max()), "bad rows", int((d.max(1).values > 1e-3).sum()))
    bad = (d.max(1).values > 1e-3).nonzero().view(-1)[:5]
    for r in bad:
        print("   row", int(r), h_cpu[r].abs().max().item(), d[r].max().item())
print("---- continuing")
import torch.nn.functional as F
model.convs[0].nn.layers[0].precision = None
b_cpu = orc.kan_linear_forward(a_cpu, l1["base_weight"], l1["spline_weight"], l1["spline_scaler"], l1["grid"], k)
for mode in (0, 1):
    model.convs[0].nn.layers[1].precision = mode
    b_gpu = model.convs[0].nn.layers[1](a_gpu.detach())
    print("kan1 mode", mode, float((b_gpu.detach().cpu() - b_cpu).abs().max()), float(b_cpu.abs().max()))
model.convs[0].nn.layers[1].precision = None
bn_cpu = F.batch_norm(b_cpu, None, None, sd["bns.0.weight"], sd["bns.0.bias"], True, 0.1, 1e-5)
bn_gpu = model.bns[0](b_gpu.detach())
print("bn0", float((bn_gpu.detach().cpu() - bn_cpu).abs().max()))
h2_cpu = orc.sum_aggregate(bn_cpu, ei) + bn_cpu
h2_gpu = ops.aggregate_sum(bn_gpu.detach(), gi, self_scale=1.0)
print("agg2", float((h2_gpu.cpu() - h2_cpu).abs().max()), float(h2_cpu.abs().max()))
print("---- conv1")
l0, l1 = layer_p("convs.1.nn.layers.0."), layer_p("convs.1.nn.layers.1.")
c_cpu = orc.kan_linear_forward(h2_cpu, l0["base_weight"], l0["spline_weight"], l0["spline_scaler"], l0["grid"], k)
c_gpu = model.convs[1].nn.layers[0](h2_gpu)
print("kan10", float((c_gpu.detach().cpu() - c_cpu).abs().max()), float(c_cpu.abs().max()))
d_cpu = orc.kan_linear_forward(c_cpu, l1["base_weight"], l1["spline_weight"], l1["spline_scaler"], l1["grid"], k)
d_gpu = model.convs[1].nn.layers[1](c_gpu.detach())
print("kan11", float((d_gpu.detach().cpu() - d_cpu).abs().max()), float(d_cpu.abs().max()))
bn1_cpu = F.batch_norm(d_cpu, None, None, sd["bns.1.weight"], sd["bns.1.bias"], True, 0.1, 1e-5)
bn1_gpu = model.bns[1](d_gpu.detach())
print("bn1", float((bn1_gpu.detach().cpu() - bn1_cpu).abs().max()))
cat_cpu = torch.cat([x, bn_cpu, bn1_cpu], 1)
cat_gpu = torch.cat([xd, bn_gpu.detach(), bn1_gpu.detach()], 1)
lo = layer_p("lay_out.")
o_cpu = orc.kan_linear_forward(cat_cpu, lo["base_weight"], lo["spline_weight"], lo["spline_scaler"], lo["grid"], k)
for mode in (0, 1):
    model.lay_out.precision = mode
    o_gpu = model.lay_out(cat_gpu)
    print("lay_out mode", mode, float((o_gpu.detach().cpu() - o_cpu).abs().max()), float(o_cpu.abs().max()))
print("vs fixture logits0", float((o_cpu - T(z["gin.logits0"])).abs().max()))
model.lay_out.precision = None
full = model(xd, eid)
print("full model vs staged", float((full.detach().cpu() - o_cpu).abs().max()))
