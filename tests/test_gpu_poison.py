"""No kernel of the path reads memory it did not write.

Every scratch buffer of the library comes from the caller (``*_workspace_bytes`` + a pointer: include/kagnn_hip.h), i.e. from
torch's caching allocator, and so do the outputs.  Each case below runs forward + backward with the allocator's free blocks
filled with one of five patterns beforehand (zeros, 1e30, NaN, 0xFF bytes, -3e38) and with every cached weight pack rebuilt under
that pattern: all outputs and gradients must be the same bits whatever the pattern.  A read of an unwritten padding row / slot /
partial-sum slab -- which parity tests at "nice" shapes and a freshly zeroed heap never see -- follows the pattern instead.
(Found nothing when written, round 5; it stays as the guard.)

And no kernel WRITES outside the buffers it was handed: the same cases with every device `torch.empty` of the host layer
(outputs, gradients, workspaces, CSR arrays) wrapped in two 64 KB guard bands, which must be intact afterwards."""
import math

import pytest
import torch

import kagnn_amd
from kagnn_amd import ops
from oracle import kan_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PATTERNS = ("zeros", "huge", "nan", "ff", "neg")


# torch's caching allocator: requests <= 1 MB are carved from 2 MB segments (small pool), requests in (1, 10) MB from 20 MB
# segments, larger ones get their own segment rounded to 2 MB; a freed block keeps its contents and merges with free neighbours
# of the same segment.  To leave NO unpoisoned free byte: release the cache, fill the holes of the segments live tensors sit in
# with graded small blocks first, then add whole segments (1 MB pairs, ten 2 MB per 20 MB, exact 20 MB, and big ones).
_SIZES = ((512, 2000), (4096, 1000), (65536, 400), (1 << 20, 400),
          ((1 << 20) + 512, 100), (2 << 20, 200), (20 << 20, 40), (64 << 20, 8), (256 << 20, 4), (1 << 30, 1))


def _poison(kind):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    blocks = []
    for nbytes, count in _SIZES:
        for _ in range(count):
            t = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
            if kind == "zeros":
                t.zero_()
            elif kind == "ff":
                t.fill_(255)
            else:
                t[:nbytes // 4 * 4].view(torch.float32).fill_({"huge": 1e30, "nan": float("nan"), "neg": -3e38}[kind])
            blocks.append(t)
    torch.cuda.synchronize()
    del blocks                       # back to the allocator's free lists, contents intact: the next allocations land on them


def _same(a, b):
    return torch.equal(a, b) or (bool((torch.isnan(a) == torch.isnan(b)).all()) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)))


def _under_every_pattern(module, run, what):
    """``run()`` -> list of tensors; parameters get a new version before each run so that no cached pack survives."""
    ref = None
    buffers = None if module is None else [b.detach().clone() for b in module.buffers()]
    for kind in PATTERNS:
        _poison(kind)
        if module is not None:
            module.zero_grad(set_to_none=True)
            with torch.no_grad():
                for p in module.parameters():
                    p.add_(0.0)
                for b, b0 in zip(module.buffers(), buffers):          # running statistics start from the same values every time
                    b.copy_(b0)
        got = [t.detach().clone() for t in run()]
        torch.cuda.synchronize()
        if ref is None:
            ref = got
            assert module is None or all(bool(torch.isfinite(t).all()) for t in ref), what
            continue
        assert len(got) == len(ref)
        for k, (a, b) in enumerate(zip(got, ref)):
            assert _same(a, b), f"{what}: tensor {k} follows the heap pattern '{kind}' (max diff {float((a - b).abs().max())})"


def test_the_poison_reaches_fresh_allocations():
    """teeth of the method: after `_poison`, fresh `torch.empty` buffers of the sizes the path allocates (a few hundred bytes to
    ~100 MB) hold the pattern, and a computation that reads one is caught by `_under_every_pattern`"""
    for nfloat in (64, 3000, 10_000, 700_000, 5_000_000, 30_000_000):
        _poison("huge")
        t = torch.empty(nfloat, device=DEV)
        assert float((t == 1e30).float().mean()) > 0.99, nfloat
        _poison("nan")
        t = torch.empty(nfloat, device=DEV)
        assert float(torch.isnan(t).float().mean()) > 0.99, nfloat
    with pytest.raises(AssertionError, match="follows the heap pattern"):
        _under_every_pattern(None, lambda: [torch.nan_to_num(torch.empty(5000, device=DEV), nan=1.0, posinf=2.0, neginf=3.0).clamp(-5, 5)], "unwritten read")


def _fwd_bwd(module, x, *args, gy_seed=7):
    def run():
        xr = x.clone().requires_grad_(True)
        y = module(xr, *args)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(gy_seed)).to(DEV)
        y.backward(gy)
        return [y, xr.grad] + [p.grad for p in module.parameters() if p.grad is not None]
    return run



# ---------------------------------------------------------------------------------------------------------------- the cases
def _kanlinear(n, fin, fout, grid):
    def build():
        torch.manual_seed(n)
        layer = kagnn_amd.KANLinear(fin, fout, grid_size=grid, spline_order=3).to(DEV)
        x = (torch.randn(n, fin, generator=torch.Generator().manual_seed(1)) * 0.6).to(DEV)
        return layer, _fwd_bwd(layer, x)
    return build


def _fastkan(n, fin, hidden, fout, grid):
    def build():
        torch.manual_seed(n)
        net = kagnn_amd.FastKAN([fin, hidden, fout], num_grids=grid).to(DEV)
        x = (torch.randn(n, fin, generator=torch.Generator().manual_seed(2)) * 0.8).to(DEV)
        return net, _fwd_bwd(net, x)
    return build


def _gin(flavour, n, e, f, hidden):
    """the one-call GIN convolution (aggregation incl. hub rows + chain, forward and backward) on a power-law graph, CSR included"""
    def build():
        ei = orc.powerlaw_graph(n, e, seed=3).to(DEV)
        torch.manual_seed(n)
        if flavour == "kan":
            conv = kagnn_amd.GIKANLayer(f, hidden, grid_size=5 if hidden < 128 else 8, spline_order=3, hidden_dim=hidden, nb_layers=2).to(DEV)
        else:
            conv = kagnn_amd.GIFASTKANLayer(f, hidden, grid_size=4, hidden_dim=hidden, nb_layers=2).to(DEV)
        x = (torch.randn(n, f, generator=torch.Generator().manual_seed(4)) * 0.5).to(DEV)

        def run():
            g = ops.GraphIndex(ei, n)
            return _fwd_bwd(conv, x, g)() + [g.rowptr, g.col, g.rowptr_t, g.col_t]
        return conv, run
    return build


def _node_model(arch, kind, dropout):
    """whole node-classification step with every fold on (norm statistics out of the forward, norm backward inside dX, skip
    gradients, one-launch read-out, in-kernel dropout), CSR rebuilt each run"""
    def build():
        n, f, hidden, classes = 3001, 40, 64, 7
        ei = orc.powerlaw_graph(n, 10 * n, seed=5).to(DEV)
        torch.manual_seed(11)
        if arch == "kan":
            model = kagnn_amd.GKAN_Nodes(kind, 3, f, hidden, classes, skip=True, grid_size=5, spline_order=3, hidden_layers=2, dropout=dropout).to(DEV)
        else:
            model = kagnn_amd.GFASTKAN_Nodes(kind, 3, f, hidden, classes, skip=True, grid_size=4, hidden_layers=2, dropout=dropout).to(DEV)
        x = (torch.randn(n, f, generator=torch.Generator().manual_seed(6)) * 0.4).to(DEV)
        y = torch.randint(0, classes, (n,), generator=torch.Generator().manual_seed(8)).to(DEV)

        def run():
            g = ops.GraphIndex(ei, n)
            torch.manual_seed(99)                                   # dropout masks
            loss = ops.softmax_cross_entropy(model(x, g), y)
            loss.backward()
            return [loss] + [p.grad for p in model.parameters() if p.grad is not None] + \
                   [b for b in model.buffers() if b.dtype.is_floating_point] + [g.rowptr, g.col, g.rowptr_t, g.col_t]
        return model, run
    return build


def _graph_level(flavour):
    """the ZINC-shaped mini-batch step: embedding encoders, single-launch CSR, the GINE stack as one tape node, pooling, read-out"""
    def build():
        from types import SimpleNamespace
        B, H = 48, 64
        gen = torch.Generator().manual_seed(21)
        sizes = torch.randint(9, 38, (B,), generator=gen)
        n = int(sizes.sum()); off = torch.cumsum(sizes, 0) - sizes
        src, dst, batch = [], [], []
        for b in range(B):
            nb = int(sizes[b]); eb = 2 * nb + 1
            src.append(torch.randint(0, nb, (eb,), generator=gen) + off[b]); dst.append(torch.randint(0, nb, (eb,), generator=gen) + off[b])
            batch.append(torch.full((nb,), b))
        e = sum(len(s_) for s_ in src)
        d = SimpleNamespace(x=torch.randint(0, 21, (n, 1), generator=gen).to(DEV), edge_index=torch.stack([torch.cat(src), torch.cat(dst)]).to(DEV),
                            edge_attr=torch.randint(0, 4, (e,), generator=gen).to(DEV), batch=torch.cat(batch).to(DEV), num_graphs=B,
                            y=torch.randn(B, generator=gen).to(DEV))
        torch.manual_seed(5)
        if flavour == "kan":
            m = kagnn_amd.KAGINRegression(1, 1, 4, H, 2, 4, 3, 1, 0.0, True)
        else:
            m = kagnn_amd.FASTKAGINRegression(1, 1, 4, H, 2, 4, 1, 0.0, True)
        m.atom_encoder = kagnn_amd.graph_models.AtomEncoder(H, [21])
        m.bond_encoder.bond_embedding_list = torch.nn.ModuleList([torch.nn.Embedding(4, H)])
        m = m.to(DEV).train()

        def run():
            loss = torch.nn.L1Loss()(m(d).squeeze(), d.y)
            loss.backward()
            return [loss] + [p.grad for p in m.parameters() if p.grad is not None] + [b for b in m.buffers() if b.dtype.is_floating_point]
        return m, run
    return build


CASES = {}
for _shape in [(1000, 64, 64, 5), (777, 37, 70, 5), (3, 5, 1, 3), (2049, 128, 128, 8), (513, 128, 64, 8), (300, 20, 130, 12), (1500, 256, 40, 5)]:
    for _mode in ("split", "half", "fp32"):
        CASES["KANLinear %dx%d->%d G%d %s" % (*_shape, _mode)] = (_mode, _kanlinear(*_shape))
for _shape in [(1000, 64, 64, 64, 8), (333, 30, 256, 50, 4), (2100, 256, 256, 40, 8)]:
    CASES["FastKAN %dx[%d,%d,%d] G%d" % _shape] = ("split", _fastkan(*_shape))
for _fl in ("kan", "fastkan"):
    for _shape in [(5000, 60000, 64, 64), (1201, 9000, 24, 48), (4000, 50000, 128, 128)]:
        CASES["%s GIN layer n%d e%d %d->%d" % (_fl, *_shape)] = ("split", _gin(_fl, *_shape))
    CASES[f"graph-level {_fl} step"] = ("split", _graph_level(_fl))
CASES["KAN GIN layer half mode"] = ("half", _gin("kan", 5000, 60000, 64, 64))
for _a, _k, _p in [("kan", "gin", 0.0), ("kan", "gin", 0.3), ("kan", "gcn", 0.0), ("fastkan", "gin", 0.0), ("fastkan", "gcn", 0.2)]:
    CASES[f"{_a} {_k} node model dropout {_p}"] = ("split", _node_model(_a, _k, _p))
CASES["kan gin node model half mode"] = ("half", _node_model("kan", "gin", 0.0))


@pytest.mark.parametrize("case", list(CASES))
def test_blind_to_the_heap(monkeypatch, case):
    mode, build = CASES[case]
    monkeypatch.setenv("KAGNN_PRECISION", mode)
    module, run = build()
    _under_every_pattern(module, run, case)


# ---------------------------------------------------------------------------------------------------------------- guard bands
class _GuardedEmpty:
    """`torch.empty` for device tensors = a view into a buffer with GUARD bytes of 0xA5 on both sides (the band starts at the
    first byte after the tensor: a one-element overrun lands in it)."""
    GUARD = 1 << 16

    def __init__(self):
        self.real = torch.empty
        self.live = []

    def __call__(self, *size, dtype=None, device=None, pin_memory=False, **kw):
        dev = None if device is None else torch.device(device)
        if dev is None or dev.type != "cuda" or pin_memory or kw:
            return self.real(*size, dtype=dtype, device=device, pin_memory=pin_memory, **kw)
        shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(int(v) for v in size)
        dtype = dtype or torch.get_default_dtype()
        nbytes = math.prod(shape) * torch.empty(0, dtype=dtype).element_size()
        G = self.GUARD
        buf = self.real(G + nbytes + G, dtype=torch.uint8, device=dev)
        buf[:G].fill_(0xA5)
        buf[G + nbytes:].fill_(0xA5)
        self.live.append((buf, nbytes, shape, dtype))
        return buf[G:G + nbytes].view(dtype).view(shape)

    def check(self, what):
        torch.cuda.synchronize()
        for buf, nbytes, shape, dtype in self.live:
            lo, hi = buf[:self.GUARD], buf[self.GUARD + nbytes:]
            assert bool((lo == 0xA5).all()), f"{what}: a kernel wrote BEFORE a {dtype} buffer of shape {shape}"
            if not bool((hi == 0xA5).all()):
                first = int((hi != 0xA5).nonzero()[0])
                raise AssertionError(f"{what}: a kernel wrote past the end of a {dtype} buffer of shape {shape} (first byte at +{first})")
        return len(self.live)


def test_the_guard_bands_catch_an_overrun(monkeypatch):
    guarded = _GuardedEmpty()
    monkeypatch.setattr(torch, "empty", guarded)
    t = torch.empty((10, 7), dtype=torch.float32, device=DEV)
    assert t.shape == (10, 7) and t.is_contiguous() and t.data_ptr() % 512 == 0
    t.fill_(1.0)
    assert guarded.check("clean") == 1
    t.as_strided((71,), (1,)).fill_(2.0)                        # one float past the end
    with pytest.raises(AssertionError, match="past the end"):
        guarded.check("overrun")


@pytest.mark.parametrize("case", list(CASES))
def test_writes_stay_inside_their_buffers(monkeypatch, case):
    mode, build = CASES[case]
    monkeypatch.setenv("KAGNN_PRECISION", mode)
    module, run = build()
    run()                                                       # (first use: caches, packs)
    module.zero_grad(set_to_none=True)
    with torch.no_grad():
        for p in module.parameters():
            p.add_(0.0)                                         # packs are rebuilt inside guarded buffers too
    guarded = _GuardedEmpty()
    monkeypatch.setattr(torch, "empty", guarded)
    out = run()
    n = guarded.check(case)
    monkeypatch.undo()
    assert n >= 3 and all(bool(torch.isfinite(t.float()).all()) for t in out), (case, n)
