"""Pin the oracle's C restatement (oracle/kan_ref.c, fp64) to the golden vectors generated from the
reference's own layers, and to the torch-op oracle."""
import numpy as np
import torch

from oracle import build_c as cref
from oracle import kan_oracle as orc
from helpers import KAN_KEYS


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


def test_c_bases_match_golden_table(golden):
    z = golden("g1_bsplines")
    for (G, k) in [(5, 3), (4, 3), (8, 3), (1, 1), (2, 1), (8, 4), (32, 4), (3, 2)]:
        x, grid, want = z[f"x_G{G}_k{k}"], z[f"grid_G{G}_k{k}"], z[f"bases_G{G}_k{k}"]
        got = cref.bspline_bases(x, grid, G, k)
        np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
        assert rel(np.nan_to_num(got), np.nan_to_num(want)) < 2e-6


def test_c_kanlinear_matches_golden(golden):
    z = golden("g2_kanlinear")
    i = 0
    while f"shape_{i}" in z:
        fi, fo, G, k = [int(v) for v in z[f"shape_{i}"]]
        tag = f"{fi}_{fo}_{G}_{k}"
        p = {n: z[f"{tag}.{n}"] for n in KAN_KEYS}
        y = cref.kan_linear_fwd(z[f"{tag}.x"], p, G, k)
        gx, gbw, gsw, gsc = cref.kan_linear_bwd(z[f"{tag}.x"], z[f"{tag}.gy"], p, G, k)
        assert rel(y, z[f"{tag}.y"]) < 5e-6
        assert rel(gx, z[f"{tag}.gx"]) < 5e-6
        assert rel(gbw, z[f"{tag}.g_base_weight"]) < 5e-6
        assert rel(gsw, z[f"{tag}.g_spline_weight"]) < 5e-6
        assert rel(gsc, z[f"{tag}.g_spline_scaler"]) < 5e-6
        i += 1
    assert i == 10


def test_c_csr_and_aggregate(golden):
    z = golden("g7_csr")
    g5 = golden("g5_gin")
    for g in ("small", "plaw"):
        ei, n = z[f"{g}.edge_index"], int(z[f"{g}.num_nodes"][0])
        rp, col, perm = cref.csr_build(ei[1], ei[0], n)
        np.testing.assert_array_equal(rp, z[f"{g}.rowptr"])
        np.testing.assert_array_equal(col, z[f"{g}.col"])
        np.testing.assert_array_equal(perm, z[f"{g}.perm"])
        x = g5[f"{g}.kan.x"]
        assert rel(cref.aggregate(x, ei[0], ei[1], None, 1.0), g5[f"{g}.kan.agg"]) < 2e-6
    # agreement with the torch-op oracle on a fresh graph, weighted
    ei = orc.powerlaw_graph(500, 4000, seed=9)
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(500, 12, generator=gen)
    w = torch.rand(4000, generator=gen)
    want = orc.sum_aggregate(x.double(), ei, 500, w.double()) + 0.5 * x.double()
    assert rel(cref.aggregate(x.numpy(), ei[0].numpy(), ei[1].numpy(), w.numpy(), 0.5), want.numpy()) < 1e-6
