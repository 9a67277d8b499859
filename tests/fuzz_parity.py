"""Randomised shape sweep of the forward / inverse path against the oracle: image size, depth, hidden width, coupling /
permutation kind and batch all vary with the seed.  `run_case(seed, big)` is collected by pytest through
tests/test_gpu_fused.py (seeded, parametrised, `-m gpu`); `python tests/fuzz_parity.py` casts a wider net by hand
(SEED / CASES / BIG in the environment)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from oracle import glow_oracle as O


def run_case(case, big=True, base_seed=0):
    """One random configuration, HIP path vs oracle on every element of z, nll and the decode (bar 1e-4).
    Returns (description line, worst max-abs, launch counts)."""
    import test_gpu_parity as T
    rng = random.Random(base_seed * 7919 + case)
    image = rng.choice([16, 32, 64, 128] if big else [16, 32, 64])
    L = rng.choice([1, 2, 3]) if image > 16 else rng.choice([1, 2])
    K = rng.choice([1, 2, 3])
    hidden = rng.choice([64, 128, 256, 512])
    coup = rng.choice(["affine", "additive"])
    perm = rng.choice(["invconv", "reverse", "shuffle"])
    batch = rng.choice([1, 2, 3, 5, 16, 48]) if hidden < 512 else rng.choice([1, 2, 5, 16])
    if image == 128:
        batch = min(batch, 5)
    cfg = O.default_cfg(image_shape=(image, image, 3), hidden_channels=hidden, K=K, L=L, flow_permutation=perm,
                        flow_coupling=coup, batch=batch)
    sd = O.seeded_state_dict(cfg, seed=case, zeros_std=0.02, invconv_perturb=0.02)
    np.random.seed(case)
    with torch.no_grad():
        glow = T.make_glow(cfg, sd, batch)
        tables = None
        if perm != "invconv":
            tables = {i: (torch.from_numpy(getattr(l, perm).indices), torch.from_numpy(getattr(l, perm).indices_inverse))
                      for i, l in enumerate(glow.flow.layers) if hasattr(l, perm)}
        g = torch.Generator().manual_seed(case)
        x = torch.rand(batch, 3, image, image, generator=g)
        noise = torch.rand(batch, 3, image, image, generator=g) / 256
        z_ref, nll_ref, _ = O.glow_forward(x, noise, sd, cfg, perm_tables=tables)
        plan = glow.flow.plan_for(T.dev(x))
        plan.launch_counts(reset=True)
        z, nll, _ = glow.normal_flow(T.dev(x), None, noise=T.dev(noise))
        ez = (z.cpu() - z_ref).abs().max().item(); en = (nll.cpu() - nll_ref).abs().max().item()
        eps = [torch.randn(batch, *s, generator=g) * 0.7 for s in glow.flow.split_shapes((3, image, image))]
        x_ref = O.glow_reverse(z_ref, sd, cfg, eps, perm_tables=tables)
        xr = glow.reverse_flow(T.dev(z_ref), None, eps=[T.dev(e) for e in eps])
        ex = (xr.cpu() - x_ref).abs().max().item()
        counts = plan.launch_counts()
    ok = max(ez, en, ex) < 1e-4 and bool(torch.isfinite(z).all())
    line = (f"[{case:2d}] {image}x{image} L={L} K={K} hid={hidden} {coup:8s} {perm:7s} B={batch:2d}  z {ez:.1e} nll {en:.1e} "
            f"dec {ex:.1e}  {'OK ' if ok else 'FAIL'} {counts}")
    assert ok, line
    return line, max(ez, en, ex), counts


if __name__ == "__main__":
    torch.set_grad_enabled(False)
    n_cases = int(os.environ.get("CASES", "24"))
    worst = 0.0
    for case in range(n_cases):
        line, w, _ = run_case(case, big=bool(os.environ.get("BIG")), base_seed=int(os.environ.get("SEED", "0")))
        worst = max(worst, w)
        print(line, flush=True)
    print("all", n_cases, "cases within 1e-4; worst", worst)
