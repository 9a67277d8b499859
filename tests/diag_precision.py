"""Diagnostic: HIP vs fp32 oracle vs fp64 oracle on the config-B model (who is closer to the truth?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_glow_amd as G
from pytorch_glow_amd.misc import util
from oracle import glow_oracle as O
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

def make_glow(cfg, sd, batch):
    hps = util.AttrDict(dict(
        model=dict(image_shape=cfg["image_shape"], hidden_channels=cfg["hidden_channels"], K=cfg["K"], L=cfg["L"],
                   actnorm_scale=1.0, n_bits_x=8, weight_y=0.0),
        ablation=dict(learn_top=False, y_condition=False, lu_decomposition=False,
                      flow_permutation=cfg["flow_permutation"], flow_coupling=cfg["flow_coupling"]),
        optim=dict(num_batch_train=batch), dataset=dict(num_classes=1), device=dict(graph=["cuda:0"])))
    glow = G.Glow(hps)
    sd = dict(sd); sd["h_top"] = torch.zeros_like(glow.h_top)
    glow.load_state_dict(sd); glow.set_actnorm_inited()
    return glow.cuda().eval()

batch = 4
for perturb in (0.0, 0.05):
    cfg = O.default_cfg(batch=batch)
    sd = O.seeded_state_dict(cfg, seed=11, invconv_perturb=perturb)
    x = torch.rand(batch, 3, 64, 64, generator=torch.Generator().manual_seed(2384))
    noise = torch.rand(batch, 3, 64, 64, generator=torch.Generator().manual_seed(1)) / 256
    with torch.no_grad():
        sd = O.glow_init_actnorm(x, noise, sd, cfg)
        z32, nll32, _ = O.glow_forward(x, noise, sd, cfg)
        sd64 = {k: v.double() for k, v in sd.items()}
        z64, nll64, _ = O.glow_forward(x.double(), noise.double(), sd64, cfg)
    glow = make_glow(cfg, sd, batch)
    z, nll, _ = glow.normal_flow(x.cuda(), None, noise=noise.cuda())
    eps = [torch.randn(batch, *s, generator=torch.Generator().manual_seed(3 + i)) * 0.7
           for i, s in enumerate(glow.flow.split_shapes((3, 64, 64)))]
    with torch.no_grad():
        x32 = O.glow_reverse(z32, sd, cfg, eps)
        x64 = O.glow_reverse(z32.double(), sd64, cfg, [e.double() for e in eps])
    xr = glow.reverse_flow(z32.cuda(), None, eps=[e.cuda() for e in eps])
    d = lambda a, b: (a.double().cpu() - b.double().cpu()).abs().max().item()
    print(f"perturb={perturb}: |z| max {z64.abs().max():.2f} |x_dec| max {x64.abs().max():.2f}")
    print(f"  z   : hip-vs-o32 {d(z, z32):.2e}  hip-vs-o64 {d(z, z64):.2e}  o32-vs-o64 {d(z32, z64):.2e}")
    print(f"  nll : hip-vs-o32 {d(nll, nll32):.2e}  hip-vs-o64 {d(nll, nll64):.2e}  o32-vs-o64 {d(nll32, nll64):.2e}")
    print(f"  dec : hip-vs-o32 {d(xr, x32):.2e}  hip-vs-o64 {d(xr, x64):.2e}  o32-vs-o64 {d(x32, x64):.2e}")
