"""Does any kernel of the graph-level step read memory it did not write?  Poison the caching allocator's free blocks with a
pattern before each run (zeros / 1e30 / NaN / 0xFF bytes) and compare every leaf module's output and every gradient across
patterns: an uninitialised read shows up as a tensor that follows the pattern.  Diagnostic."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from train_determinism import make, make_batches, DEV   # noqa: E402

batches = make_batches()
m = make(); m.train()


def poison(kind):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    blocks = []
    for nbytes, count in ((512, 600), (4096, 400), (65536, 200), (1 << 20, 100), (8 << 20, 40), (256 << 20, 2)):
        for _ in range(count):
            t = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
            if kind == "zeros":
                t.zero_()
            elif kind == "ff":
                t.fill_(255)
            else:
                t.view(torch.float32).fill_({"huge": 1e30, "nan": float("nan"), "neg": -3e38}[kind])
            blocks.append(t)
    torch.cuda.synchronize()
    del blocks


def run(d, kind):
    poison(kind)
    m.zero_grad()
    with torch.no_grad():
        for p_ in m.parameters():
            p_.add_(0.0)                    # same values, new version: every cached weight pack is rebuilt under the poison
    acts, hs = {}, []
    for name, mod in m.named_modules():
        if name and len(list(mod.children())) == 0:
            hs.append(mod.register_forward_hook(lambda mod_, inp, out, name=name: acts.__setitem__(name, out.detach().clone() if torch.is_tensor(out) else None)))
    out = m(d)
    loss = torch.nn.L1Loss()(out.squeeze(), d.y)
    loss.backward()
    for h in hs:
        h.remove()
    torch.cuda.synchronize()
    return acts, out.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


def eq(a, b):
    return torch.equal(a, b) or (torch.isnan(a) == torch.isnan(b)).all() and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))


for bi, d in enumerate(batches[:2]):
    a0, o0, g0 = run(d, "zeros")
    print({k: float(v.abs().max()) for k, v in list(g0.items())[:4]})
    for kind in ("zeros", "huge", "nan", "ff", "neg"):
        a1, o1, g1 = run(d, kind)
        bad_a = [k for k in a0 if a0[k] is not None and not eq(a0[k], a1[k])]
        bad_g = [k for k in g0 if not eq(g0[k], g1[k])]
        worst = max([float((g0[k] - g1[k]).abs().max() / g0[k].abs().max().clamp_min(1e-30)) for k in bad_g] or [0.0])
        print(f"batch {bi} poison {kind}: output identical {eq(o0, o1)}; differing activations {bad_a[:8]}; differing grads ({len(bad_g)}) {bad_g[:8]} worst rel {worst:.2e}",
              flush=True)
