"""Diagnostic (not collected by pytest): one optimiser step at a given K/B, then HIP forward vs oracle on the updated weights."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import test_gpu_grad as T
from pytorch_glow_amd import parallel
K = int(os.environ.get("K", "32")); B = int(os.environ.get("B", "8")); lr = float(os.environ.get("LR", "1e-4"))
cfg, sd, glow, x, noise = T._celeba_geometry_model(K=K, batch=B)
opt = torch.optim.Adam(list(glow.parameters()), lr=lr, betas=(0.9, 0.9999), eps=1e-8)
xd = x.to(T.DEV)
with torch.no_grad():
    print("pre-step  hip nll", glow.normal_flow(xd, None, noise=noise.to(T.DEV))[1][:4].tolist())
    print("pre-step  ora nll", T.O.glow_forward(x, noise, sd, cfg)[1][:4].tolist())
loss0, gn = parallel.train_step(glow, opt, xd, world=1, max_grad_clip=5, max_grad_norm=100)
print("loss0", float(loss0), "gnorm", float(gn))
sd1 = {k: v.detach().cpu().clone() for k, v in glow.state_dict().items()}
with torch.no_grad():
    print("post-step hip nll", glow.normal_flow(xd, None, noise=noise.to(T.DEV))[1][:4].tolist())
    print("post-step ora nll", T.O.glow_forward(x, noise, sd1, cfg)[1][:4].tolist())
    T.O.STABLE_LOGDET = True
    print("post-step ora nll (stable logdet)", T.O.glow_forward(x, noise, sd1, cfg)[1][:4].tolist())
    # where does it go wrong: per-layer z magnitude in the oracle
    z = x + noise; obj = torch.zeros(B)
    for kind, i, shp in T.O.flow_layout(cfg):
        p = f"flow.layers.{i}."
        if kind == "squeeze": z = T.O.squeeze2d(z, 2)
        elif kind == "split": z, obj = T.O.split2d(z, obj, sd1, p, reverse=False)
        else: z, obj = T.O.flowstep(z, obj, sd1, p, cfg, reverse=False)
        if i % 8 == 0 or not torch.isfinite(obj).all():
            print(i, kind, "max|z|", float(z.abs().max()), "obj0", float(obj[0]))
        if not torch.isfinite(obj).all(): break
