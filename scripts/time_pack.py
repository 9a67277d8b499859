"""Time glowhip_plan_pack(INFERENCE) of the config-B model alone, with the one-wave-per-matrix log|det| kernel and with the
workgroup-wide LU (debug switch 0x200000); run under rocprofv3 --kernel-trace --stats for the per-kernel table."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pytorch_glow_amd as G
from pytorch_glow_amd.misc import util
dev = torch.device('cuda', 0)
glow, hps = bench.build_model(G, util, dev, bench.CONFIGS['B'], 64)
glow.set_actnorm_inited(); glow.eval()
x = torch.rand(64, 3, 64, 64, device=dev)
plan = glow.flow.plan_for(x)
for flag in (0, 0x200000, 0, 0x200000):
    G.lib().glowhip_debug_force_tail_tile(flag)
    for _ in range(5): plan.pack(use=1, merge=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): plan.pack(use=1, merge=False)
    e1.record(); torch.cuda.synchronize()
    print(hex(flag), 'pack us', 1e3 * e0.elapsed_time(e1) / 50, plan.launch_counts(reset=True))
G.lib().glowhip_debug_force_tail_tile(0)
