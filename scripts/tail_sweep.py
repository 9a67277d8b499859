"""Sweep the fused tail kernel's (pixels-per-block, out-channel split) variants at the three level geometries of
config B (B=64) and print the per-launch time of each (HIP events via the plan timing hooks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_glow_amd as G
from pytorch_glow_amd import _lib

torch.manual_seed(0)
B = int(os.environ.get("B", "64"))
for (c, h, w) in [(12, 32, 32), (24, 16, 16), (48, 8, 8)]:
    st = G.FlowStep(c, 512, coupling="affine").cuda().eval()
    with torch.no_grad():
        for n_, p in st.named_parameters():
            if n_.startswith("f.4"):
                p.normal_(0, 0.002)
    x = torch.randn(B, c, h, w, device="cuda")
    plan = st._plan(x)
    res = []
    for ms in (0x200, 0x100):
        for tp in (128, 64, 32, 16):
            if (h * w) % tp or tp % w:
                continue
            G.lib().glowhip_debug_force_tail_tile(tp | ms)
            plan.timing(True)
            for _ in range(6):
                plan.encode(x, None, None, want_logdet=False)
            recs = plan.timing_read()
            plan.timing(False)
            t = sorted(ms_ for k, l, m, ms_ in recs if k == _lib.K_CONV_F4)[1:-1]
            res.append((sum(t) / len(t) * 1e3, tp, "split" if ms == 0x100 else "whole"))
    G.lib().glowhip_debug_force_tail_tile(0)
    plan.timing(True)
    for _ in range(6):
        plan.encode(x, None, None, want_logdet=False)
    recs = plan.timing_read()
    plan.timing(False)
    t = sorted(ms_ for k, l, m, ms_ in recs if k == _lib.K_CONV_F4)[1:-1]
    print(f"C={c} {h}x{w}: auto {sum(t)/len(t)*1e3:.1f} us | " + "  ".join(f"{tp}/{m}:{us:.1f}" for us, tp, m in sorted(res)))
