"""Time one fwd+bwd training step of the config-B model (used under rocprofv3 for the kernel breakdown)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, pytorch_glow_amd as G
from pytorch_glow_amd.misc import util
B = int(os.environ.get("B", "64")); steps = int(os.environ.get("STEPS", "3"))
glow, hps = bench.build_model(G, util, torch.device("cuda:0"), B)
x = torch.rand(B, 3, 64, 64, device="cuda")
glow.train()
with torch.no_grad():
    glow.normal_flow(x, None)
def step():
    for p in glow.parameters(): p.grad = None
    z, nll, _ = glow.normal_flow(x, None)
    nll.mean().backward()
step(); torch.cuda.synchronize(); t = time.time()
for _ in range(steps): step()
torch.cuda.synchronize(); dt = (time.time() - t) / steps
print(f"train step (fwd+bwd) B={B}: {dt*1e3:.1f} ms = {B/dt:.0f} img/s")
