"""Debug: which k_cnet instances a config-B forward launches (run-time counters of the executor) + max-abs deviation from the exact-fp32 family."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
import pytorch_glow_amd as G
from pytorch_glow_amd.misc import util
B = int(os.environ.get("B", "64"))
hps = util.load_profile("celeba"); hps.model.K = int(os.environ.get("K", "2")); hps.optim.num_batch_train = B; hps.device.graph = ["cuda:0"]
glow = G.Glow(hps).to("cuda:0")
x = torch.rand(B, 3, 64, 64, device="cuda")
fl = int(os.environ.get("FLAGS", "0"), 0)
if fl: G.lib().glowhip_debug_force_tail_tile(fl)
glow.train()
with torch.no_grad():
    glow.normal_flow(x, None)
    glow.eval()
    plan = glow.flow.plan_for(x)
    plan.launch_counts(reset=True)
    z, nll, _ = glow.normal_flow(x, None)
    print({k: v for k, v in plan.launch_counts(reset=True).items() if "cnet" in k})
    G.lib().glowhip_debug_force_tail_tile(0x10000 | fl)
    z2, nll2, _ = glow.normal_flow(x, None)
    print({k: v for k, v in plan.launch_counts(reset=True).items() if "cnet" in k})
    print("max-abs z", (z - z2).abs().max().item(), "nll", (nll - nll2).abs().max().item())
