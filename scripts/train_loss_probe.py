"""Print loss / grad-norm per optimiser step of the config-B model (diagnostic for the training loop)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, pytorch_glow_amd as G
from pytorch_glow_amd.misc import util
from pytorch_glow_amd import parallel
B = int(os.environ.get("B", "64")); steps = int(os.environ.get("STEPS", "6")); lr = float(os.environ.get("LR", "1e-3"))
glow, hps = bench.build_model(G, util, torch.device("cuda:0"), B)
x = torch.rand(B, 3, 64, 64, device="cuda")
glow.train()
with torch.no_grad():
    glow.normal_flow(x, None)
    print("nograd fwd", float(glow.normal_flow(x, None)[1].mean()))
for i in range(3):
    for p in glow.parameters(): p.grad = None
    with torch.enable_grad():
        z, nll, _ = glow.normal_flow(x, None)
        l = nll.mean(); l.backward()
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in glow.parameters() if p.grad is not None))
    print("no-opt step", i, float(l), float(gn), flush=True)
opt = torch.optim.Adam(list(glow.parameters()), lr=lr, betas=(0.9, 0.9999), eps=1e-8)
for i in range(steps):
    before = [p.detach().clone() for p in glow.parameters()]
    loss, gn = parallel.train_step(glow, opt, x, world=1, max_grad_clip=5, max_grad_norm=100)
    dmax = max(float((p.detach() - b).abs().max()) for p, b in zip(glow.parameters(), before))
    bad = [n for n, p in glow.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    with torch.no_grad():
        l2 = float(glow.normal_flow(x, None)[1].mean())
    print(i, float(loss), float(gn), "max|dp|", dmax, "post-step nograd loss", l2, bad[:3], flush=True)
