#!/usr/bin/env python3
"""Randomised graphs (hubs, isolated nodes, self loops, duplicate edges) and widths through the sum aggregation and the
GAT attention aggregation, forward and backward, against the fp64 oracle.
usage: python tools/fuzz_graph.py [cases] [seed]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from kagnn_amd import ops
from oracle import kan_oracle as orc
from helpers import assert_close

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DEV = "cuda:0"
bad = 0


def random_graph(n, e, gen, hub):
    src = torch.randint(0, n, (e,), generator=gen)
    dst = torch.randint(0, n, (e,), generator=gen)
    if hub and e > 10:
        k = min(e // 2, rng.choice([600, 1500, 5000]))
        dst[:k] = int(torch.randint(0, n, (1,), generator=gen))        # one destination with a long neighbour list
        src[k:k + k // 2] = int(torch.randint(0, n, (1,), generator=gen))   # and one source (hub of the transposed graph)
    return torch.stack([src, dst])


for it in range(cases):
    n = rng.choice([1, 2, 7, 100, 1000, 5000, 20000])
    e = rng.choice([0, 1, 5, n, 4 * n, 20 * n])
    gen = torch.Generator().manual_seed(it)
    ei = random_graph(n, e, gen, hub=rng.random() < 0.5)
    kind = rng.choice(["sum", "sum", "gat"])
    try:
        gi = ops.GraphIndex(ei.to(DEV), n)
        if kind == "sum":
            f = rng.choice([1, 3, 4, 8, 12, 16, 31, 32, 64, 100, 128, 260])
            tag = f"case {it}: sum n={n} e={e} f={f}"
            x = torch.randn(n, f, generator=gen); gy = torch.randn(n, f, generator=gen)
            xr = x.double().requires_grad_(True)
            want = orc.sum_aggregate(xr, ei, n) + 1.25 * xr
            want.backward(gy.double())
            xd = x.to(DEV).requires_grad_(True)
            got = ops.aggregate_sum(xd, gi, self_scale=1.25)
            got.backward(gy.to(DEV))
            assert_close(got, want.detach(), what="fwd"); assert_close(xd.grad, xr.grad, what="bwd")
        else:
            h, c = rng.choice([(1, 8), (2, 16), (4, 16), (8, 8), (4, 4), (3, 24), (2, 64), (1, 100)])
            tag = f"case {it}: gat n={n} e={e} heads={h} channels={c}"
            xh = torch.randn(n, h * c, generator=gen) * 0.7
            a_s, a_d = torch.randn(1, h, c, generator=gen) * 0.5, torch.randn(1, h, c, generator=gen) * 0.5
            b = torch.randn(h * c, generator=gen) * 0.1
            gy = torch.randn(n, h * c, generator=gen)
            leaves = [t.double().requires_grad_(True) for t in (xh, a_s, a_d, b)]
            want = orc.gat_conv(leaves[0], ei, lambda t: t, leaves[1], leaves[2], leaves[3], h)
            want.backward(gy.double())
            dl = [t.to(DEV).requires_grad_(True) for t in (xh, a_s, a_d, b)]
            got = ops.gat_aggregate(dl[0], dl[1], dl[2], dl[3], gi, h, c)
            got.backward(gy.to(DEV))
            assert_close(got, want.detach(), what="fwd")
            for nme, a, w in zip(("xh", "att_src", "att_dst", "bias"), dl, leaves):
                # the attention-vector gradients are sums over all nodes of (logit gradient) x (features): where a node has one
                # incoming edge its softmax is the constant 1 and the exact term is 0, so what fp32 adds up is rounding noise
                # that grows like sqrt(n) -- scale the bound with it instead of comparing noise with 0 (n = 20 000, e = 0: 5e-5)
                # (round 5: assert_close bounds by tol * max|reference|, and the reference here can be 0 or pure cancellation --
                # the noise term is the ABSOLUTE floor, at the operands' unit scale)
                noise = 2e-5 * max(1.0, 0.05 * n ** 0.5) if nme.startswith("att_") else 0.0
                assert_close(a.grad, w.grad, 2e-5, what="g_" + nme, noise=noise)
        print("ok  ", tag, flush=True)
    except Exception as ex:
        bad += 1
        print("FAIL", tag if "tag" in dir() else f"case {it}", "->", str(ex)[:200], flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
