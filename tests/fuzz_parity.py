"""Randomised shape sweep of the forward / inverse path against the oracle (run on the GPU box: python tests/fuzz_parity.py).
Not collected by pytest (the fixed cases in test_gpu_parity.py are the regression set); this is the wide net cast after
kernel-selection changes: image size, depth, hidden width, coupling / permutation kind and batch all vary."""
import os, sys, itertools, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import test_gpu_parity as T
from oracle import glow_oracle as O

torch.set_grad_enabled(False)
rng = random.Random(int(os.environ.get("SEED", "0")))
n_cases = int(os.environ.get("CASES", "24"))
worst = 0.0
for case in range(n_cases):
    image = rng.choice([16, 32, 64, 128] if os.environ.get("BIG") else [16, 32, 64])
    L = rng.choice([1, 2, 3]) if image > 16 else rng.choice([1, 2])
    K = rng.choice([1, 2, 3])
    hidden = rng.choice([64, 128, 256, 512])
    coup = rng.choice(["affine", "additive"])
    perm = rng.choice(["invconv", "reverse", "shuffle"])
    batch = rng.choice([1, 2, 3, 5, 16, 48]) if hidden < 512 else rng.choice([1, 2, 5])
    cfg = O.default_cfg(image_shape=(image, image, 3), hidden_channels=hidden, K=K, L=L, flow_permutation=perm,
                        flow_coupling=coup, batch=batch)
    sd = O.seeded_state_dict(cfg, seed=case, zeros_std=0.02, invconv_perturb=0.02)
    np.random.seed(case)
    glow = T.make_glow(cfg, sd, batch)
    tables = None
    if perm != "invconv":
        tables = {i: (torch.from_numpy(getattr(l, perm).indices), torch.from_numpy(getattr(l, perm).indices_inverse))
                  for i, l in enumerate(glow.flow.layers) if hasattr(l, perm)}
    g = torch.Generator().manual_seed(case)
    x = torch.rand(batch, 3, image, image, generator=g)
    noise = torch.rand(batch, 3, image, image, generator=g) / 256
    z_ref, nll_ref, _ = O.glow_forward(x, noise, sd, cfg, perm_tables=tables)
    z, nll, _ = glow.normal_flow(T.dev(x), None, noise=T.dev(noise))
    ez = (z.cpu() - z_ref).abs().max().item(); en = (nll.cpu() - nll_ref).abs().max().item()
    eps = [torch.randn(batch, *s, generator=g) * 0.7 for s in glow.flow.split_shapes((3, image, image))]
    x_ref = O.glow_reverse(z_ref, sd, cfg, eps, perm_tables=tables)
    xr = glow.reverse_flow(T.dev(z_ref), None, eps=[T.dev(e) for e in eps])
    ex = (xr.cpu() - x_ref).abs().max().item()
    desc = glow.flow.plan_for(T.dev(x)).describe()
    kinds = sorted({l.split("hidden=")[1].split(" ", 1)[1] for l in desc.splitlines() if "flowstep" in l})
    ok = max(ez, en, ex) < 1e-4 and torch.isfinite(z).all()
    worst = max(worst, ez, en, ex)
    print(f"[{case:2d}] {image}x{image} L={L} K={K} hid={hidden} {coup:8s} {perm:7s} B={batch:2d}  z {ez:.1e} nll {en:.1e} dec {ex:.1e}  "
          f"{'OK ' if ok else 'FAIL'} {kinds[0] if kinds else ''}", flush=True)
    assert ok
print("all", n_cases, "cases within 1e-4; worst", worst)
