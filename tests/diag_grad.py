"""Diagnostic (not collected by pytest): per-parameter gradient error vs the autograd oracle and run-to-run variation."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import test_gpu_grad as T
K = int(os.environ.get("K", "3")); B = int(os.environ.get("B", "4")); ORACLE = int(os.environ.get("ORACLE", "1"))
cfg, sd, glow, x, noise = T._celeba_geometry_model(K=K, batch=B)
def run():
    for p in glow.parameters(): p.grad = None
    with torch.enable_grad():
        xd = x.to(T.DEV).requires_grad_(True)
        z, nll, _ = glow.normal_flow(xd, None, noise=noise.to(T.DEV))
        nll.mean().backward()
    return {n: p.grad.detach().cpu().clone() for n, p in glow.named_parameters() if p.grad is not None}, float(nll.mean())
runs = [run() for _ in range(4)]
print("losses", [r[1] for r in runs])
if ORACLE:
    ref, gx, lref = T.oracle_grads(cfg, sd, x, noise)
    print("oracle loss", lref)
    ref64, gx64, l64 = T.oracle_grads(cfg, {k: v.double() for k, v in sd.items()}, x.double(), noise.double())
rows = []
for n in runs[0][0]:
    g0 = runs[0][0][n]
    var = max(float((r[0][n] - g0).abs().max()) for r in runs[1:])
    sc = float(g0.abs().max()) + 1e-30
    err = float((g0 - ref64[n]).abs().max()) / (float(ref64[n].abs().max()) + 1e-30) if ORACLE else 0.0
    e32 = float((ref[n] - ref64[n]).abs().max()) / (float(ref64[n].abs().max()) + 1e-30) if ORACLE else 0.0
    rows.append((n, sc, e32, err))
layout = {i: (kind, shp) for kind, i, shp in T.O.flow_layout(cfg)}
agg = collections.OrderedDict()
for n, sc, var, err in rows:
    i = int(n.split(".")[2]); suffix = n.split(".", 3)[3]
    key = (layout[i][1], suffix)
    a = agg.setdefault(key, [0.0, 0.0, 0])
    a[0] = max(a[0], var); a[1] = max(a[1], err); a[2] += 1
for (shp, suffix), (var, err, cnt) in agg.items():
    print(f"{str(shp):18s} {suffix:22s} n={cnt:3d} oracle32-vs-64 {var:.2e}  hip-vs-64 {err:.2e}")
print("--- per-parameter rows where hip error > 5 x oracle32 error and > 1e-5")
for n, sc, e32, err in rows:
    if err > 5 * e32 and err > 1e-5:
        print(f"{n:40s} |g|max {sc:.3e} oracle32 {e32:.2e} hip {err:.2e}")
print("--- flip analysis, layer 2")
pre = []
orig = torch.relu
def rec(t):
    pre.append(t.detach().clone()); return orig(t)
T.O.torch.relu = rec
T.oracle_grads(cfg, {k: v.double() for k, v in sd.items()}, x.double(), noise.double())
T.O.torch.relu = orig
g = runs[0][0]
for li, (nm, idx) in enumerate([("f.0", 2), ("f.2", 3)]):
    e = (g[f"flow.layers.2.{nm}.actnorm.bias"].double() - ref64[f"flow.layers.2.{nm}.actnorm.bias"]).abs().flatten()
    top = torch.topk(e, 4)
    print(nm, "bias-grad abs err top4", top.values.tolist(), "channels", top.indices.tolist(), "median err", float(e.median()))
    p = pre[idx]
    for c in top.indices.tolist()[:2]:
        a = p[:, c].abs()
        print("   channel", c, "min |pre-activation| (fp64)", float(a.min()), " #|x|<1e-6:", int((a < 1e-6).sum()))
    print("   overall: #elements with |x|<1e-6:", int((p.abs() < 1e-6).sum()), "of", p.numel())
