# debug: training tape of the taping k_cnet forward vs the per-layer forward (debug flag 0x40000000), region by region
import sys, os, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import conftest  # noqa
import pytorch_glow_amd as G
from pytorch_glow_amd import _lib
from oracle import glow_oracle as O
from test_gpu_grad import hps_for
DEV = "cuda:0"
K, batch = 2, 4
cfg = O.default_cfg(K=K, batch=batch)
sd = O.seeded_state_dict(cfg, seed=13, invconv_perturb=0.02, zeros_std=1e-5)
g = torch.Generator().manual_seed(13)
x = torch.rand(batch, 3, 64, 64, generator=g)
noise = torch.rand(batch, 3, 64, 64, generator=g) / 256
sd = O.glow_init_actnorm(x, noise, sd, cfg)
tapes = {}
for flag in (0x40000000, 0):
    _lib.lib().glowhip_debug_force_tail_tile(flag)
    glow = G.Glow(hps_for(cfg, batch)); glow.load_state_dict(sd); glow.set_actnorm_inited(); glow = glow.to(DEV).train()
    plan = glow.flow.plan_for(x.to(DEV))
    torch.cuda.empty_cache()
    z, nll, tape = plan.glow_forward_train(x.to(DEV), noise.to(DEV), None, None, 0, 8)
    torch.cuda.synchronize()
    tapes[flag] = tape.cpu().view(torch.float32).clone() if tape.numel() % 4 == 0 else None
    print(hex(flag), nll.cpu().tolist(), {k: v for k, v in plan.launch_counts().items() if "cnet" in k} if hasattr(plan, "launch_counts") else "")
a, b = tapes[0x40000000], tapes[0]
# layout (plan_train.hip tape_layout; 256-byte aligned regions)
off = 0
def take(nfl):
    global off
    o = off; off = (off + nfl + 63) // 64 * 64
    return o, nfl
C, H = 12, 32
li = 0
for lvl in range(3):
    regs = [("L%d squeeze out" % li, take(batch * C * H * H))]; li += 1
    for k in range(K):
        regs += [("L%d out" % li, take(batch * C * H * H)), ("L%d h1" % li, take(batch * 512 * H * H)), ("L%d h2" % li, take(batch * 512 * H * H)),
                 ("L%d hout" % li, take(batch * C * H * H))]; li += 1
    if lvl < 2:
        regs += [("L%d split out" % li, take(batch * (C // 2) * H * H)), ("L%d split hout" % li, take(batch * C * H * H))]; li += 1
    for name, (o, n) in regs:
        da, db = a[o:o + n], b[o:o + n]
        d = (da - db).abs()
        zf = ((da > 0) != (db > 0)).sum().item()
        print(f"{name:18s} n {n:9d} max|old| {da.abs().max().item():9.3e} maxdiff {d.max().item():9.3e} mask flips {zf}", end="")
        if zf and "h" in name.split()[-1] and n == batch * 512 * H * H:
            idx = ((da > 0) != (db > 0)).nonzero().flatten()
            ch = (idx // (H * H)) % 512
            print("  channels", sorted(set(ch.tolist()))[:10], "old", da[idx[:4]].tolist(), "new", db[idx[:4]].tolist(), end="")
        print()
    C, H = C * 2, H // 2
