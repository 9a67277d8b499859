"""Debug: a graphed training step, an eager re-run on the exact-fp32 family in between, then the graph again -- stage by stage with syncs."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import pytorch_glow_amd as G
from pytorch_glow_amd import training
from oracle import glow_oracle as O
from test_gpu_grad import hps_for
DEV = "cuda:0"
torch.manual_seed(0)
cfg = O.default_cfg(image_shape=(16, 16, 3), hidden_channels=128, K=1, L=1, batch=4)
sd = O.seeded_state_dict(cfg, seed=3, zeros_std=1e-3)
hps = hps_for(cfg, 4)
hps.optim.update(optimizer="adam", optimizer_args=dict(lr=1e-4, betas=[0.9, 0.9999], eps=1e-8), lr_scheduler="noam", lr_scheduler_args=dict(warmup_steps=5, min_lr=1e-5))
hps.ablation.update(max_grad_clip=5, max_grad_norm=100)
glow = G.Glow(hps); sd["h_top"] = torch.zeros_like(glow.h_top); glow.load_state_dict(sd); glow.set_actnorm_inited(); glow = glow.to(DEV)
x = torch.rand(4, 3, 16, 16).to(DEV)
loop = training.TrainLoop(glow, hps, graph=True); loop.GRAPH_AFTER = 1
def S(tag):
    torch.cuda.synchronize(); print(tag, flush=True)
loop.step(x); S("step0 eager")
if os.environ.get("POKE", "1") == "1":
    params = dict(glow.named_parameters())
    with torch.no_grad():
        params["flow.layers.1.f.0.actnorm.logs"].add_(float(np.log(3e5)) / 3.0)
        params["flow.layers.1.f.2.actnorm.logs"].sub_(float(np.log(3e5)) / 3.0)
l, n = loop.step(x); S(f"step1 graphed={loop._graphed is not None} err={loop.graph_error} norm={n.item()}")
g = loop._graphed
print("sig before flush", g._buffer_signature() == g._buffers)
loop.flush(); S(f"flush: fallbacks={loop.range_fallbacks}")
print("valid after rerun:", g.valid(), g._buffer_signature(), g._buffers)
if os.environ.get("EAGER_AFTER"):
    loop.graph = False
l, n = loop.step(x); S(f"step2 graphed={loop._graphed is not None} same={loop._graphed is g} recaptures={loop.graph_recaptures} norm={n.item()}")
loop.flush(); S("flush2")
l, n = loop.step(x); S(f"step3 graphed={loop._graphed is not None} norm={n.item()}")
loop.flush(); S("done")
