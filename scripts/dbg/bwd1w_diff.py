"""Debug: per-parameter difference of a training step's gradients between the backward instance of k_cnet1w (default) and k_cnet
(debug switch 0x40000) at config-B geometry, batch 28, K = 1."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import pytorch_glow_amd as G
from pytorch_glow_amd import _lib
from oracle import glow_oracle as O
from test_gpu_grad import hps_for
K, batch = int(os.environ.get("K", "1")), 28
cfg = O.default_cfg(K=K, batch=batch)
sd = O.seeded_state_dict(cfg, seed=23, invconv_perturb=0.02, zeros_std=0.01)
g = torch.Generator().manual_seed(23)
x = torch.rand(batch, 3, 64, 64, generator=g); noise = torch.rand(batch, 3, 64, 64, generator=g) / 256
sd = O.glow_init_actnorm(x, noise, sd, cfg)
res = {}
FA = int(os.environ.get("FLAG_A", "0x40000"), 0)
for flag in (FA, 0):
    _lib.lib().glowhip_debug_force_tail_tile(flag)
    glow = G.Glow(hps_for(cfg, batch)); glow.load_state_dict(sd); glow.set_actnorm_inited(); glow = glow.to("cuda:0").train()
    with torch.enable_grad():
        xd = x.to("cuda:0").requires_grad_(True)
        z, nll, _ = glow.normal_flow(xd, None, noise=noise.to("cuda:0"))
        G.Glow.generative_loss(nll).backward()
    grads = {n: p.grad.cpu().double() for n, p in glow.named_parameters() if p.grad is not None}
    grads["dx"] = xd.grad.cpu().double()
    res[flag] = grads
    print(hex(flag), {k: v for k, v in glow.flow.plan_for(x.to("cuda:0")).launch_counts().items() if "cnet" in k})
_lib.lib().glowhip_debug_force_tail_tile(0)
for n, a in res[FA].items():
    b = res[0][n]; sc = a.abs().max().item(); err = (a - b).abs()
    if err.max().item() > 1e-4 * sc + 1e-12:
        print(f"{n:60s} shape {tuple(a.shape)} scale {sc:.3e} max err {err.max().item():.3e} rms {err.pow(2).mean().sqrt().item():.3e}")
