"""Host-side cost of one forward step (everything Python + the C executor do to enqueue ~217 launches), measured on an idle queue."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch, cProfile, pstats
import pytorch_glow_amd as G
from pytorch_glow_amd.misc import util
hps = util.load_profile("celeba"); hps.device.graph = ["cuda:0"]
glow = G.Glow(hps).to("cuda:0")
x = torch.rand(64, 3, 64, 64, device="cuda")
glow.train()
with torch.no_grad():
    glow.normal_flow(x, None)
glow.eval()
torch.set_grad_enabled(False)
for _ in range(5): glow.normal_flow(x, None, repack=True)
torch.cuda.synchronize()
for rp in (True, False):
    ts = []
    for _ in range(6):
        torch.cuda.synchronize()
        t = time.perf_counter(); glow.normal_flow(x, None, repack=rp); ts.append((time.perf_counter() - t) * 1e3)
    print("repack", rp, "host ms per step on an idle queue:", [round(v, 2) for v in ts])
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): glow.normal_flow(x, None, repack=True)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
