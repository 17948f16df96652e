"""Pin oracle/kan_oracle.py to the golden vectors generated from the reference's own layers
(tests/golden/make_golden.py) and, when /root/reference is present, to the live import."""
import numpy as np
import pytest
import torch

from oracle import kan_oracle as orc

torch.set_num_threads(1)
T = lambda a: torch.from_numpy(np.asarray(a))


def close(a, b, tol=1e-6):
    a, b = T(a).double(), T(b).double()
    scale = max(1e-300, float(b.abs().max()))
    assert a.shape == b.shape
    assert float((a - b).abs().max()) <= tol * scale, float((a - b).abs().max())


def test_g1_bspline_table_bit_exact(golden):
    z = golden("g1_bsplines")
    for (G, k) in [(5, 3), (4, 3), (8, 3), (1, 1), (2, 1), (8, 4), (32, 4), (3, 2)]:
        x, grid, want = T(z[f"x_G{G}_k{k}"]), T(z[f"grid_G{G}_k{k}"]), z[f"bases_G{G}_k{k}"]
        assert torch.equal(orc.make_knots(2, G, k), grid)          # knot buffer bit-identical
        got = orc.bspline_bases(x, grid, k).numpy()
        np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
        np.testing.assert_array_equal(np.nan_to_num(got), np.nan_to_num(want))   # same op order => bit-exact


def _kanlinear_case(z, tag):
    p = {k: T(z[f"{tag}.{k}"]) for k in ("base_weight", "spline_weight", "spline_scaler", "grid")}
    return p, T(z[f"{tag}.x"]), T(z[f"{tag}.gy"])


def test_g2_kanlinear_fwd_bwd(golden):
    z = golden("g2_kanlinear")
    i = 0
    while f"shape_{i}" in z:
        fi, fo, G, k = [int(v) for v in z[f"shape_{i}"]]
        tag = f"{fi}_{fo}_{G}_{k}"
        p, x, gy = _kanlinear_case(z, tag)
        x = x.requires_grad_(True)
        ps = {n: (v.requires_grad_(True) if n != "grid" else v) for n, v in p.items()}
        y = orc.kan_linear_forward(x, ps["base_weight"], ps["spline_weight"], ps["spline_scaler"], ps["grid"], k)
        y.backward(gy)
        close(y.detach(), z[f"{tag}.y"])
        close(x.grad, z[f"{tag}.gx"])
        close(ps["base_weight"].grad, z[f"{tag}.g_base_weight"])
        close(ps["spline_weight"].grad, z[f"{tag}.g_spline_weight"])
        close(ps["spline_scaler"].grad, z[f"{tag}.g_spline_scaler"])
        i += 1
    assert i == 10


def test_g3_kan_chain(golden):
    z = golden("g3_kan_chain")
    i = 0
    while f"cfg_{i}" in z:
        cfg = [int(v) for v in z[f"cfg_{i}"]]
        sizes, k = cfg[:-2], cfg[-1]
        tag = "kan_" + "_".join(map(str, sizes))
        layers = []
        for li in range(len(sizes) - 1):
            layers.append({n: T(z[f"{tag}.layers.{li}.{n}"]) for n in
                           ("base_weight", "spline_weight", "spline_scaler", "grid")})
        for L in layers:
            for n in ("base_weight", "spline_weight", "spline_scaler"):
                L[n].requires_grad_(True)
        x = T(z[f"{tag}.x"]).requires_grad_(True)
        y = orc.kan_forward(x, layers, k)
        y.backward(T(z[f"{tag}.gy"]))
        close(y.detach(), z[f"{tag}.y"])
        close(x.grad, z[f"{tag}.gx"])
        for li, L in enumerate(layers):
            for n in ("base_weight", "spline_weight", "spline_scaler"):
                close(L[n].grad, z[f"{tag}.grad.layers.{li}.{n}"], 2e-6)
        i += 1
    assert i == 5


FK_KEYS = ("layernorm.weight", "layernorm.bias", "rbf.grid", "spline_linear.weight",
           "base_linear.weight", "base_linear.bias")


def test_g4_fastkan(golden):
    z = golden("g4_fastkan")
    i = 0
    while f"shape_{i}" in z:
        fi, fo, ng = [int(v) for v in z[f"shape_{i}"]]
        tag = f"fk_{fi}_{fo}_{ng}"
        p = {n: T(z[f"{tag}.{n}"]) for n in FK_KEYS}
        for n in FK_KEYS:
            if n != "rbf.grid":
                p[n].requires_grad_(True)
        x = T(z[f"{tag}.x"]).requires_grad_(True)
        y = orc.fastkan_forward(x, [p])
        y.backward(T(z[f"{tag}.gy"]))
        close(y.detach(), z[f"{tag}.y"])
        close(x.grad, z[f"{tag}.gx"])
        for n in FK_KEYS:
            if n != "rbf.grid":
                close(p[n].grad, z[f"{tag}.grad.{n}"], 2e-6)
        i += 1
    assert i == 5
    tag = "fastkan_48_72_24"
    layers = [{n: T(z[f"{tag}.layers.{li}.{n}"]) for n in FK_KEYS} for li in range(2)]
    x = T(z[f"{tag}.x"]).requires_grad_(True)
    y = orc.fastkan_forward(x, layers)
    y.backward(T(z[f"{tag}.gy"]))
    close(y.detach(), z[f"{tag}.y"])
    close(x.grad, z[f"{tag}.gx"])


def test_g7_csr_bit_exact(golden):
    z = golden("g7_csr")
    for g in ("small", "plaw"):
        ei, n = T(z[f"{g}.edge_index"]), int(z[f"{g}.num_nodes"][0])
        rp, col, perm = orc.csr_by_key(ei[1], ei[0], n)
        np.testing.assert_array_equal(rp.numpy(), z[f"{g}.rowptr"])
        np.testing.assert_array_equal(col.numpy(), z[f"{g}.col"])
        np.testing.assert_array_equal(perm.numpy(), z[f"{g}.perm"])
        rp, col, perm = orc.csr_by_key(ei[0], ei[1], n)
        np.testing.assert_array_equal(rp.numpy(), z[f"{g}.rowptr_t"])
        np.testing.assert_array_equal(col.numpy(), z[f"{g}.col_t"])
        # structural properties that do not depend on who wrote the fixture
        assert rp[0] == 0 and rp[-1] == ei.size(1) and bool((rp[1:] >= rp[:-1]).all())
        assert torch.equal(torch.sort(perm).values, torch.arange(ei.size(1)))


def test_g5_gin_composition(golden):
    z = golden("g5_gin")
    g7 = golden("g7_csr")
    for g in ("small", "plaw"):
        ei = T(g7[f"{g}.edge_index"])
        pre = f"{g}.kan"
        layers = [{n: T(z[f"{pre}.layers.{li}.{n}"]) for n in
                   ("base_weight", "spline_weight", "spline_scaler", "grid")} for li in range(2)]
        y, gx, grads = orc.kan_gin_layer_fwd_bwd(T(z[f"{pre}.x"]), ei, layers, 3, T(z[f"{pre}.gy"]))
        close(y, z[f"{pre}.y"], 2e-6)
        close(gx, z[f"{pre}.gx"], 2e-6)
        for li in range(2):
            for n in ("base_weight", "spline_weight", "spline_scaler"):
                close(grads[li][n], z[f"{pre}.grad.layers.{li}.{n}"], 5e-6)
        x0 = T(z[f"{pre}.x"])
        close(orc.sum_aggregate(x0, ei) + x0, z[f"{pre}.agg"])
        # independent check of the aggregation: dense adjacency product in fp64
        n = x0.size(0)
        A = torch.zeros(n, n, dtype=torch.float64)
        A.index_put_((ei[1], ei[0]), torch.ones(ei.size(1), dtype=torch.float64), accumulate=True)
        close(A @ x0.double() + x0.double(), z[f"{pre}.agg"], 1e-5)


def test_g6_gcn_norm_properties(golden):
    z = golden("g6_gcn")
    g7 = golden("g7_csr")
    for g in ("small", "plaw"):
        ei, n = T(g7[f"{g}.edge_index"]), int(g7[f"{g}.num_nodes"][0])
        ei2, w = orc.gcn_norm(ei, n)
        np.testing.assert_array_equal(ei2.numpy(), z[f"{g}.gcn.norm_edge_index"])
        close(w, z[f"{g}.gcn.norm_weight"])
        # exactly one self loop per node, appended last, non-loop edges kept in order
        assert torch.equal(ei2[:, -n:], torch.arange(n).repeat(2, 1))
        keep = ei[0] != ei[1]
        assert torch.equal(ei2[:, :-n], ei[:, keep])
        # symmetric normalisation: w_e * sqrt(deg[dst] * deg[src]) == 1
        deg = torch.zeros(n).scatter_add_(0, ei2[1], torch.ones(ei2.size(1)))
        close(w * torch.sqrt(deg[ei2[0]] * deg[ei2[1]]), torch.ones_like(w), 1e-5)


def test_live_reference_matches_oracle(reference_modules):
    """Fresh (non-fixture) comparison against the imported reference, incl. timing-shape case."""
    ref_ekan, ref_fastkan = reference_modules
    torch.manual_seed(7)
    layer = ref_ekan.KANLinear(24, 18, grid_size=6, spline_order=2)
    x = torch.randn(500, 24)
    want = layer(x)
    sd = layer.state_dict()
    got = orc.kan_linear_forward(x, sd["base_weight"], sd["spline_weight"], sd["spline_scaler"], sd["grid"], 2)
    assert torch.equal(got, want.detach())      # same op sequence => bit-identical on CPU
    fk = ref_fastkan.FastKAN([24, 12, 6], num_grids=5)
    sd = fk.state_dict()
    layers = [{n: sd[f"layers.{li}.{n}"] for n in FK_KEYS} for li in range(2)]
    close(orc.fastkan_forward(x, layers), fk(x).detach(), 1e-6)


def test_g10_update_grid(golden):
    """oracle.update_grid against the reference's KANLinear.update_grid: knots bit-identical (same op order),
    refitted coefficients to the accuracy of an fp32 least-squares solve; then the layer on its adaptive grid."""
    z = golden("g10_update_grid")
    i = 0
    while f"shape_{i}" in z:
        fi, fo, G, k = [int(v) for v in z[f"shape_{i}"]]
        tag = f"{fi}_{fo}_{G}_{k}"
        p = {n: T(z[f"{tag}.before.{n}"]) for n in ("base_weight", "spline_weight", "spline_scaler", "grid")}
        for step in range(2):
            xb = T(z[f"{tag}.u{step}.x"])
            grid, fit = orc.update_grid(xb, p, G, k)
            assert torch.equal(grid, T(z[f"{tag}.u{step}.grid"]))
            close(fit, z[f"{tag}.u{step}.spline_weight"], tol=2e-4)
            _, fit64 = orc.update_grid(xb, p, G, k, solve_dtype=torch.float64)
            close(fit64, z[f"{tag}.u{step}.spline_weight"], tol=2e-4)
            p = dict(p, grid=T(z[f"{tag}.u{step}.grid"]), spline_weight=T(z[f"{tag}.u{step}.spline_weight"]))
        x, gy = T(z[f"{tag}.x"]).requires_grad_(True), T(z[f"{tag}.gy"])
        np.testing.assert_array_equal(orc.bspline_bases(x.detach(), p["grid"], k).numpy(), z[f"{tag}.bases"])
        ps = {n: (v.clone().requires_grad_(True) if n != "grid" else v) for n, v in p.items()}
        y = orc.kan_linear_forward(x, ps["base_weight"], ps["spline_weight"], ps["spline_scaler"], ps["grid"], k)
        y.backward(gy)
        close(y.detach(), z[f"{tag}.y"])
        close(x.grad, z[f"{tag}.gx"])
        close(ps["spline_weight"].grad, z[f"{tag}.g_spline_weight"])
        close(ps["spline_scaler"].grad, z[f"{tag}.g_spline_scaler"])
        i += 1
    assert i == 5


# ------------------------------------------------------------------ round-2 fixtures
def test_g4b_fastkan_wide(golden):
    """the FastKAN shapes SURVEY 8(c) lists: (128, 256, 4) and the skip-concat read-out (896, 40, 4)"""
    z = golden("g4b_fastkan_wide")
    i = 0
    while f"shape_{i}" in z:
        fi, fo, ng = [int(v) for v in z[f"shape_{i}"]]
        tag = f"fk_{fi}_{fo}_{ng}"
        p = {n: T(z[f"{tag}.{n}"]) for n in FK_KEYS}
        for n in FK_KEYS:
            if n != "rbf.grid":
                p[n].requires_grad_(True)
        x = T(z[f"{tag}.x"]).requires_grad_(True)
        y = orc.fastkan_forward(x, [p])
        y.backward(T(z[f"{tag}.gy"]))
        close(y.detach(), z[f"{tag}.y"], 2e-6)
        close(x.grad, z[f"{tag}.gx"], 2e-6)
        for n in FK_KEYS:
            if n != "rbf.grid":
                close(p[n].grad, z[f"{tag}.grad.{n}"], 5e-6)
        i += 1
    assert i == 2


def test_g5b_gin_on_10k_node_powerlaw_graph(golden):
    z = golden("g5b_gin_plaw10k")
    ei = T(z["edge_index"])
    assert torch.equal(ei, orc.powerlaw_graph(10000, 100000, seed=0))          # the recipe is part of the contract
    layers = [{n: T(z[f"kan.layers.{li}.{n}"]) for n in ("base_weight", "spline_weight", "spline_scaler", "grid")}
              for li in range(2)]
    y, gx, grads = orc.kan_gin_layer_fwd_bwd(T(z["kan.x"]), ei, layers, 3, T(z["kan.gy"]))
    close(y, z["kan.y"], 2e-6)
    close(gx, z["kan.gx"], 2e-6)
    for li in range(2):
        for n in ("base_weight", "spline_weight", "spline_scaler"):
            close(grads[li][n], z[f"kan.grad.layers.{li}.{n}"], 1e-5)
    fl = [{n: T(z[f"fastkan.layers.{li}.{n}"]) for n in FK_KEYS} for li in range(2)]
    x = T(z["fastkan.x"]).requires_grad_(True)
    y = orc.gin_conv(x, ei, lambda h: orc.fastkan_forward(h, fl))
    y.backward(T(z["fastkan.gy"]))
    close(y.detach(), z["fastkan.y"], 2e-6)
    close(x.grad, z["fastkan.gx"], 2e-6)


@pytest.mark.parametrize("name,arch", [("g9_harness", "kan"), ("g11_fastkan_harness", "fastkan")])
def test_node_model_restatement_matches_reference_made_logits(golden, name, arch):
    """oracle.node_model_forward (GKAN_Nodes / GFASTKAN_Nodes.forward, models.py:192-203,246-257) against logits the
    reference's own KAN / FastKAN modules produced inside the restated message passing (G9, G11); chunked rows must
    not change anything"""
    z = golden(name)
    x, ei = T(z["x"]), T(z["edge_index"])
    for kind in ("gin", "gcn"):
        pre = f"{kind}.init."
        st = {k[len(pre):]: T(z[k]) for k in z.files if k.startswith(pre)}
        for chunk in (None, 97):
            out = orc.node_model_forward(x, ei, st, arch, kind, 2, 3, chunk=chunk)
            close(out, z[f"{kind}.logits0"], 2e-5)


def test_g12_fastkan_gcn_conv(golden):
    z, g7 = golden("g12_fastkan_gcn"), golden("g7_csr")
    for g in ("small", "plaw"):
        pre = f"{g}.fgcn"
        ei = T(g7[f"{g}.edge_index"])
        p = {n: T(z[f"{pre}.lin.{n}"]) for n in FK_KEYS}
        x = T(z[f"{pre}.x"]).requires_grad_(True)
        bias = T(z[f"{pre}.bias"]).requires_grad_(True)
        y = orc.gcn_conv(x, ei, lambda h: orc.fastkan_forward(h, [p]), bias)
        y.backward(T(z[f"{pre}.gy"]))
        close(y.detach(), z[f"{pre}.y"], 2e-6)
        close(x.grad, z[f"{pre}.gx"], 2e-6)
        close(bias.grad, z[f"{pre}.grad.bias"], 2e-6)


def test_gcn_norm_sparse_adds_a_unit_loop_on_top_of_the_diagonal():
    """torch-sparse ``edge_index`` (time_model.py:70-80): add_self_loops semantics -- independent dense fp64 check"""
    g = torch.Generator().manual_seed(7)
    n = 40
    dense = (torch.rand(n, n, generator=g) < 0.1).double() * torch.rand(n, n, generator=g).double()
    dense[3, 3] = 0.7                                        # an existing loop keeps its weight AND gets +1
    dense[:, 5] = 0; dense[5, :] = 0                         # an isolated node: only the added loop
    ei, w = orc.gcn_norm_sparse(dense.to_sparse())
    a_hat = dense + torch.eye(n, dtype=torch.float64)
    dis = a_hat.sum(1).pow(-0.5)
    want = dis[:, None] * a_hat * dis[None, :]
    got = torch.zeros(n, n, dtype=torch.float64).index_put_((ei[1], ei[0]), w, accumulate=True)
    close(got, want, 1e-12)


@pytest.mark.parametrize("kind", ["kan", "fastkan"])
def test_g8b_zinc_batch_whole_model_restatement(golden, kind):
    """BASELINE config 4's model at its real batch shape: ``oracle.graph_regression_forward`` (the oracle's own layer
    restatements + gine_conv + global_add_pool + batch statistics) reproduces the predictions and the L1 loss of the
    fixture that was made with the reference's ekan.KAN / fastkan.FastKAN modules; its fp64 gradients agree with the
    fixture's fp32 ones to the error of an fp32 chain this deep (three BatchNorms, relu kinks in 400k messages)."""
    z = golden("g8b_zinc_batch")
    pre = f"{kind}.state."
    x, ei, ea, batch, y = T(z["x"]), T(z["edge_index"]), T(z["edge_attr"]), T(z["batch"]), T(z[f"{kind}.y"])
    st32 = {k[len(pre):]: T(z[k]) for k in z.files if k.startswith(pre)}
    pred = orc.graph_regression_forward(x, ei, ea, batch, 256, st32, kind, 3)
    close(pred, z[f"{kind}.pred"], 2e-5)
    assert abs(float((pred.squeeze() - y).abs().mean()) - float(z[f"{kind}.loss"])) < 1e-5
    st = {k: (v.double().requires_grad_(True) if v.is_floating_point() else v) for k, v in st32.items()}
    pred64 = orc.graph_regression_forward(x, ei, ea, batch, 256, st, kind, 3)
    close(pred64.detach(), z[f"{kind}.pred"], 5e-5)
    (pred64.squeeze() - y.double()).abs().mean().backward()
    checked = zero = 0
    names = [k for k in z.files if k.startswith(f"{kind}.grad.")]
    gmax = max(float(st[k[len(kind) + 6:]].grad.abs().max()) for k in names)
    for k in names:
        g64 = st[k[len(kind) + 6:]].grad
        if float(g64.abs().max()) <= 1e-12 * gmax:
            # a bias in front of a training-mode BatchNorm: the gradient is IDENTICALLY zero (the batch mean is subtracted), fp64
            # leaves 1e-17, and what the reference's fp32 run stored is the rounding noise of a cancelling sum over 5 932 rows --
            # bound the noise (1e-5 of the model's largest gradient) instead of comparing it with zero (round 5: `close` scales
            # by the reference's own maximum, so this case no longer hides behind max(1, .))
            assert float(np.abs(z[k]).max()) <= 1e-5 * gmax, k
            zero += 1
            continue
        close(g64, z[k], 5e-3)
        checked += 1
    assert checked >= 20 and zero <= 4, (checked, zero)


G13_MODELS = [("KAGIN", "gin", "kan"), ("FASTKAGIN", "gin", "fastkan"), ("KAGCN", "gcn", "kan"), ("FASTKAGCN", "gcn", "fastkan")]


@pytest.mark.parametrize("name,family,arch", G13_MODELS, ids=[m[0] for m in G13_MODELS])
@pytest.mark.parametrize("bi", [0, 1], ids=["16graphs", "empty+single-node"])
def test_g13_graph_classification_restatement(golden, name, family, arch, bi):
    """the graph-CLASSIFICATION callers (graph_classification/models.py:95-119,125-151,174-194,245-265): ``oracle.
    graph_classification_forward`` reproduces the log-probabilities, d/dx and every parameter gradient of fixture G13, which was made
    with the reference's own graph_classification/ekan.py / fastkan.py modules inside the restated GIN / GCN convolutions, torch's
    BatchNorm1d, global_add_pool / global_mean_pool and log_softmax -- on the G8-sized 16-graph batch and on a batch with an empty
    and a single-node graph (VERDICT r05 missing 4)."""
    z = golden("g13_graph_classification")
    b = f"b{bi}."
    x, ei, batch, ng, gout = T(z[b + "x"]), T(z[b + "edge_index"]), T(z[b + "batch"]), int(z[b + "num_graphs"]), T(z[b + "g_out"])
    pre = f"{b}{name}.init."
    st32 = {k[len(pre):]: T(z[k]) for k in z.files if k.startswith(pre)}
    out32 = orc.graph_classification_forward(x, ei, batch, ng, st32, arch, family, 2)
    close(out32, z[f"{b}{name}.out"], 2e-5)
    frozen = ("grid", "rbf.grid", "eps", "running_mean", "running_var", "num_batches_tracked")
    st = {k: (v.double().requires_grad_(True) if v.is_floating_point() and not k.endswith(frozen) else v.double() if v.is_floating_point() else v)
          for k, v in st32.items()}
    xr = x.double().requires_grad_(True)
    out = orc.graph_classification_forward(xr, ei, batch, ng, st, arch, family, 2)
    assert out.shape == (ng, 3)
    close(out.detach(), z[f"{b}{name}.out"], 2e-5)
    out.backward(gout.double())
    close(xr.grad, z[f"{b}{name}.gx"], 2e-4)
    names = [k for k in z.files if k.startswith(f"{b}{name}.grad.")]
    gmax = max(float(np.abs(z[k]).max()) for k in names)
    checked = 0
    for k in names:
        g64 = st[k[len(f"{b}{name}.grad."):]].grad
        if float(g64.abs().max()) <= 1e-12 * gmax:          # a bias right in front of a training-mode BatchNorm: identically zero
            assert float(np.abs(z[k]).max()) <= 1e-5 * gmax, k
            continue
        close(g64, z[k], 2e-4)
        checked += 1
    assert checked >= 8, checked
    if bi == 1:                                              # the empty graph: zero pooled row -> the read-out of a zero vector
        empty = out.detach()[1]
        zero_in = orc.graph_classification_forward(torch.zeros(1, 16, dtype=torch.float64), torch.zeros(2, 0, dtype=torch.int64),
                                                   torch.zeros(1, dtype=torch.int64), 1, {k: v.detach() for k, v in st.items()}, arch, family, 0)
        close(empty, zero_in[0], 1e-12)
