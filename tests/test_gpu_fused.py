"""`-m gpu` parity of the kernels that only run at large grids: the fused f.0 + f.2 kernel (k_f02_sh, the dominant kernel of the
bench step) is selected when a launch has >= 192 workgroups, i.e. at batches the small-case parity tests never reach.  These
tests run config-B geometry (hidden 512, 64x64x3, L=3) and the 64-pixel-wide level of config D at such batches against the
oracle on EVERY element -- with K reduced so the CPU oracle stays at seconds -- and take the evidence of which kernel ran
from the executor's run-time launch counters (glowhip_plan_launch_counts), not from a static description.
Reference: network/module.py:300-319 (f), network/model.py:82-154 (FlowStep)."""
import numpy as np
import pytest
import torch

import pytorch_glow_amd as G
from oracle import glow_oracle as O
from test_gpu_parity import close, dev, make_glow

pytestmark = pytest.mark.gpu


def _case(image, L, K, hidden, batch, seed=5, coup="affine", perm="invconv"):
    cfg = O.default_cfg(image_shape=(image, image, 3), hidden_channels=hidden, K=K, L=L, flow_coupling=coup,
                        flow_permutation=perm, batch=batch)
    sd = O.seeded_state_dict(cfg, seed=seed, invconv_perturb=0.02)
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, 3, image, image, generator=g)
    noise = torch.rand(batch, 3, image, image, generator=g) / 256
    if perm == "invconv":
        sd = O.glow_init_actnorm(x, noise, sd, cfg)      # (the oracle restates the init pass for invconv models)
    np.random.seed(seed)
    glow = make_glow(cfg, sd, batch)
    tables = None
    if perm != "invconv":
        tables = {i: (torch.from_numpy(getattr(l, perm).indices), torch.from_numpy(getattr(l, perm).indices_inverse))
                  for i, l in enumerate(glow.flow.layers) if hasattr(l, perm)}
    z_ref, nll_ref, _ = O.glow_forward(x, noise, sd, cfg, perm_tables=tables)
    plan = glow.flow.plan_for(dev(x))
    plan.launch_counts(reset=True)
    z, nll, _ = glow.normal_flow(dev(x), None, noise=dev(noise))
    fwd = plan.launch_counts(reset=True)
    eps = [torch.randn(batch, *s, generator=g) * 0.7 for s in glow.flow.split_shapes((3, image, image))]
    x_ref = O.glow_reverse(z_ref, sd, cfg, eps, perm_tables=tables)
    xr = glow.reverse_flow(dev(z_ref), None, eps=[dev(e) for e in eps])
    rev = plan.launch_counts(reset=True)
    ez = close(z, z_ref, 1e-4, what="z"); en = close(nll, nll_ref, 1e-4, what="nll"); ex = close(xr, x_ref, 1e-4, what="decode")
    print(f"{image}x{image} L={L} K={K} hid={hidden} B={batch}: max-abs z {ez:.2e} nll {en:.2e} decode {ex:.2e}\n fwd {fwd}\n rev {rev}")
    return plan, fwd, rev


def test_fused_f02_runs_and_matches_oracle_config_b_geometry():
    """Config-B geometry at batch 48: levels 1 (768 workgroups) and 2 (192) take the fused kernel, forward and inverse; level 3
    (48 workgroups) the separate pair.  Full z / nll / decode vs the oracle at 1e-4."""
    K = 4
    plan, fwd, rev = _case(64, 3, K, 512, 48)
    assert fwd.get("k_f02_sh", 0) == 2 * K, fwd          # levels 1 and 2, every step
    assert fwd.get("k_first_sh", 0) == K and fwd.get("k_gemm_sh", 0) == K, fwd   # level 3
    assert rev.get("k_f02_sh", 0) == 2 * K, rev
    assert fwd.get("k_conv_direct", 0) == 0 and fwd.get("k_gemm_f32", 0) == 0, fwd
    # the batch-aware description agrees with what ran
    d48, d4 = plan.describe(48), plan.describe(4)
    assert sum("-sh-fused" in l for l in d48.splitlines()) == 2 * K
    assert sum("-sh-fused" in l for l in d4.splitlines()) == 0


def test_fused_f02_level1_only_batch16():
    """Batch 16: only level 1 (256 workgroups) reaches the fused kernel."""
    K = 3
    plan, fwd, rev = _case(64, 3, K, 512, 16, seed=6)
    assert fwd.get("k_f02_sh", 0) == K and fwd.get("k_gemm_sh", 0) == 2 * K, fwd


def test_fused_f02_w64_level():
    """The 64-pixel-wide level (config D level 1, 128x128 input): a 64-pixel tile is one image row (wshift = 6).  Batch 3 ->
    192 workgroups."""
    K = 2
    plan, fwd, rev = _case(128, 2, K, 256, 3, seed=7)
    assert fwd.get("k_f02_sh", 0) >= K, fwd              # the W=64 level is fused at this batch
    assert rev.get("k_f02_sh", 0) >= K, rev


def test_fused_f02_additive_and_shuffle():
    """Additive coupling (Cout = C/2 tail rows) and a gather permutation behind the fused kernel."""
    K = 2
    plan, fwd, rev = _case(64, 2, K, 512, 16, seed=8, coup="additive", perm="reverse")
    assert fwd.get("k_f02_sh", 0) >= K, fwd


@pytest.mark.parametrize("case", range(14))
def test_fuzz_parity_seeded(case):
    """tests/fuzz_parity.py's randomised sweep (image 16..128, L, K, hidden 64..512, coupling, permutation, batch 1..48) as
    seeded regression cases, BIG shapes included."""
    import fuzz_parity
    line, worst, counts = fuzz_parity.run_case(case, big=True)
    print(line)
    assert worst < 1e-4
