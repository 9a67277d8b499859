"""`-m gpu` parity at the shapes and batches the small-case tests never reach: config-B geometry (hidden 512, 64x64x3, L=3) and the
64-pixel-wide level of config D at batches that fill the chip, against the oracle on EVERY element -- with K reduced so the CPU
oracle stays at seconds -- on both kernel families that remain (the product path: k_cnet + finishing kernel; the exact-fp32 MFMA
kernels that serve as range fall-back and reference point), with the evidence of which kernel ran taken from the executor's
run-time launch counters (glowhip_plan_launch_counts), not from a static description.
Reference: network/module.py:300-319 (f), network/model.py:82-154 (FlowStep)."""
import numpy as np
import pytest
import torch

import pytorch_glow_amd as G
from oracle import glow_oracle as O
from test_gpu_parity import close, dev, make_glow

pytestmark = pytest.mark.gpu

EXACT_FP32 = 0x800      # glowhip_debug_force_tail_tile: the split-half path off -> every coupling network on the exact-fp32 kernels


@pytest.fixture(params=["cnet", "fp32"])
def path(request):
    """Both kernel families behind the same tests: the product default (k_cnet: f.0 + f.2 + f.4 in one kernel + finishing
    kernel) and the exact-fp32 MFMA kernels (one launch per layer), the fall-back for out-of-range batches and for shapes the
    product kernels do not take."""
    G.lib().glowhip_debug_force_tail_tile(EXACT_FP32 if request.param == "fp32" else 0)
    try:
        yield request.param
    finally:
        G.lib().glowhip_debug_force_tail_tile(0)


def ncnet(counts):
    return counts.get("k_cnet", 0) + counts.get("k_cnet+prev_finish", 0)


def nfinish(counts):
    """FlowSteps finished: by a finishing launch, or (round 6) by their own k_cnet1w launch (fused finishing)."""
    return counts.get("k_cfinish", 0) + counts.get("k_cfinish+mixer", 0) + counts.get("k_cnet(finishes the step)", 0)


FUSED_FINISH = 0x100000      # glowhip_debug_force_tail_tile: k_cnet1w finishes its step itself (off by default: measured slower)


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


FP32_KERNELS = ("k_conv_first_f32", "k_conv_wide_f32", "k_gemm_f32", "k_conv_tail_f32")


def _family_ran(path, fwd, rev, steps):
    if path == "cnet":      # one k_cnet + one finishing kernel per FlowStep, nothing of the other family
        assert ncnet(fwd) == steps and ncnet(rev) == steps, (fwd, rev)
        assert nfinish(fwd) == steps and nfinish(rev) == steps, (fwd, rev)
        assert not any(k in fwd for k in FP32_KERNELS + ("k_conv_direct",)), fwd
    else:
        assert ncnet(fwd) == 0 and ncnet(rev) == 0, (fwd, rev)
        assert fwd.get("k_gemm_f32", 0) == steps and rev.get("k_gemm_f32", 0) == steps, (fwd, rev)
        assert fwd.get("k_conv_direct", 0) == 0, fwd


def test_config_b_geometry_matches_oracle(path):
    """Config-B geometry at batch 48, full z / nll / decode vs the oracle at 1e-4, forward and inverse."""
    K = 4
    plan, fwd, rev = _case(64, 3, K, 512, 48)
    _family_ran(path, fwd, rev, 3 * K)
    if path == "cnet":
        assert sum("cnet-sh2" in l for l in plan.describe(48).splitlines()) == 3 * K


def test_level1_runs_on_the_one_wave_per_simd_kernel_and_both_kernels_match_the_oracle():
    """Round 5: at config-B geometry and batch 48 level 1 gives 384 tiles of 128 pixels, so its launches run on k_cnet1w
    (cnet1w_sh.hip: one wave per SIMD, h1 / h2 chained through the register file, k-permuted weight images) -- evidence from the
    executor's run-time counters, forward AND inverse, K launches each -- while levels 2 / 3 stay on k_cnet (64-pixel tiles; row
    split 4 at level 3).  With the debug switch 0x10000 the same launches fall back to k_cnet<512,1,128>, which reads the same
    k-permuted images: both against the oracle on every element (inside _case), and against each other well inside the 1e-4 bar."""
    K = 2
    plan, fwd, rev = _case(64, 3, K, 512, 48, seed=12)
    for counts in (fwd, rev):
        assert counts.get("variant:k_cnet1w<512,1,128>") == K, counts
        assert counts.get("variant:k_cnet<512,1,64>") == K and counts.get("variant:k_cnet<512,4,64>") == K, counts
        assert ncnet(counts) == 3 * K and nfinish(counts) == 3 * K, counts
    G.lib().glowhip_debug_force_tail_tile(0x10000)
    try:
        plan2, fwd2, rev2 = _case(64, 3, K, 512, 48, seed=12)
    finally:
        G.lib().glowhip_debug_force_tail_tile(0)
    for counts in (fwd2, rev2):
        assert counts.get("variant:k_cnet<512,1,128>") == K and "variant:k_cnet1w<512,1,128>" not in counts, counts


def test_row_split_instance_of_the_one_wave_kernel_matches_the_oracle():
    """Round 5: k_cnet1w with the h2 rows of a 128-pixel tile split over TWO workgroups (each computes all of h1, half of f.2, a
    K-half of f.4; partial sums in k_cnet's MS = 2 layout for the finishing kernel) -- built for the C = 24 levels whose 128-pixel
    tiles alone fill half the CUs (config B's level 2 at batch 64: 128 tiles).  Measured 3 % slower there than k_cnet's 64-pixel
    tiles, so it sits behind the debug switch 0x20000; this test is its parity evidence: forward and inverse at config-B geometry,
    every element against the oracle (inside _case), the instance asserted from the run-time counters; without the switch the same
    launches run on k_cnet<512,1,64>."""
    K = 2
    G.lib().glowhip_debug_force_tail_tile(0x20000)
    try:
        plan, fwd, rev = _case(64, 3, K, 512, 64, seed=14)
    finally:
        G.lib().glowhip_debug_force_tail_tile(0)
    for counts in (fwd, rev):
        assert counts.get("variant:k_cnet1w<512,1,128>") == K and counts.get("variant:k_cnet1w<512,2,128>") == K, counts
        assert counts.get("variant:k_cnet<512,4,64>") == K and ncnet(counts) == 3 * K and nfinish(counts) == 3 * K, counts
    plan2, fwd2, rev2 = _case(64, 3, K, 512, 64, seed=14)
    for counts in (fwd2, rev2):
        assert counts.get("variant:k_cnet<512,1,64>") == K and "variant:k_cnet1w<512,2,128>" not in counts, counts


def test_one_wave_kernel_at_a_16_pixel_wide_level():
    """k_cnet1w outside the BASELINE shapes: a 32x32 input, L = 1 -- C = 12 on 16x16 pixels, a 128-pixel tile is eight image rows
    (the configs run it at 32, 64 and 128 pixels per row) -- at batch 112 = 224 tiles, the smallest launch it takes; forward and
    inverse against the oracle on every element, the instance asserted from the run-time counters."""
    K = 2
    plan, fwd, rev = _case(32, 1, K, 512, 112, seed=13)
    for counts in (fwd, rev):
        assert counts.get("variant:k_cnet1w<512,1,128>") == K, counts


@pytest.mark.parametrize("image,batch,tiles", [(128, 7, 224), (256, 2, 256)])
def test_one_wave_kernel_at_the_64_and_128_pixel_wide_levels_vs_oracle(image, batch, tiles):
    """the PRODUCT forward + inverse instance of k_cnet1w at hidden 512 on the two level-1 widths of
    configs D and E it had only met through property checks -- W = 64 (128x128 input: a 128-pixel tile is two image rows) at
    batch 7 = 224 tiles, W = 128 (256x256 input: a tile is ONE image row, both halo rows belong to other tiles) at batch 2 =
    256 tiles -- against the oracle on EVERY element of z / nll / decode at 1e-4 (inside _case), the instance asserted from
    the run-time launch counters.  Reference: network/model.py:82-154, network/module.py:300-319."""
    K = 2
    assert batch * (image // 2) ** 2 // 128 == tiles
    plan, fwd, rev = _case(image, 1, K, 512, batch, seed=15)
    for counts in (fwd, rev):
        assert counts.get("variant:k_cnet1w<512,1,128>") == K, counts
        assert ncnet(counts) == K and nfinish(counts) == K, counts


@pytest.mark.parametrize("image,L,batch,K", [(64, 3, 64, 3), (64, 3, 32, 2), (128, 1, 7, 2), (256, 1, 2, 2), (32, 1, 112, 2)])
def test_fused_finishing_equals_the_finishing_kernel_bitwise(image, L, batch, K):
    """Round 6, a measured NEGATIVE kept behind the switch 0x100000: where a FlowStep runs on k_cnet1w the launch can
    FINISH the step itself -- every workgroup publishes its partial sums of f.4 (agent-scope stores), bumps the arrival counters of
    its tile and of the two neighbouring tiles of the image, and whoever completes a tile's counter runs the finishing kernel's own
    code on that tile (coupling, log-det, the next step's mixer; cnet_fin.h cfinish_chunk) -- one launch per step instead of two.
    It is slower (7.68 against 6.80 ms per config-B forward: every CU pays the finishing's latency chain twice per launch, one
    workgroup at a time), so the default keeps the finishing kernel; this test is the parity evidence of the fused form: z, nll and
    the decode are the same BITS either way, at every width the kernel runs at (32-, 64-, 128- and 16-pixel rows: four, two, one and
    eight image rows per tile -- at 128 both halo rows of a tile come from other workgroups), and ten repetitions of the fused form
    reproduce themselves (an arrival-counter race would show as a run-to-run difference).  Who finished: from the run-time counters.
    Reference: network/model.py:105-117,131-139."""
    cfg = O.default_cfg(image_shape=(image, image, 3), hidden_channels=512, K=K, L=L, batch=batch)
    sd = O.seeded_state_dict(cfg, seed=19, invconv_perturb=0.02)
    g = torch.Generator().manual_seed(19)
    x = torch.rand(batch, 3, image, image, generator=g)
    noise = torch.rand(batch, 3, image, image, generator=g) / 256
    sd = O.glow_init_actnorm(x, noise, sd, cfg)
    glow = make_glow(cfg, sd, batch)
    plan = glow.flow.plan_for(dev(x))
    eps = [dev(torch.randn(batch, *s, generator=g) * 0.7) for s in glow.flow.split_shapes((3, image, image))]

    def run():
        plan.launch_counts(reset=True)
        z, nll, _ = glow.normal_flow(dev(x), None, noise=dev(noise))
        fwd = plan.launch_counts(reset=True)
        xr = glow.reverse_flow(z, None, eps=eps)
        rev = plan.launch_counts(reset=True)
        return z.clone(), nll.clone(), xr.clone(), fwd, rev

    zu, nllu, xru, fwdu, revu = run()                  # the default: a finishing launch per step
    for counts in (fwdu, revu):
        assert "k_cnet(finishes the step)" not in counts and nfinish(counts) == L * K, counts
    G.lib().glowhip_debug_force_tail_tile(FUSED_FINISH)
    try:
        z, nll, xr, fwd, rev = run()
        for counts in (fwd, rev):
            assert counts.get("variant:k_cnet1w<512,1,128>") == K, counts
            # every step of the level that runs on k_cnet1w is finished by its own launch (the mixer-less last step of a level too)
            assert counts.get("k_cnet(finishes the step)") == K and nfinish(counts) == L * K, counts
        for rep in range(10):
            z2, nll2, xr2, _, _ = run()
            assert torch.equal(z, z2) and torch.equal(nll, nll2) and torch.equal(xr, xr2), rep
    finally:
        G.lib().glowhip_debug_force_tail_tile(0)
    assert torch.equal(z, zu) and torch.equal(nll, nllu) and torch.equal(xr, xru)
    assert torch.isfinite(nll).all() and torch.isfinite(xr).all()


def test_one_wave_per_matrix_logdet_equals_the_workgroup_lu_bitwise():
    """Round 6: the inference pack takes log|det W| of the 12 / 24 / 48-wide invconv matrices from k_step_prepare_small (lu.hip) -- one
    WAVE per matrix, the matrix in registers (lane = row, the pivot row by v_readlane) -- instead of the workgroup-wide LU with three
    barriers per pivot (74 -> 64 us per pack at config B by rocprofv3 -- the serial chain pivot search -> division -> update bounds both
    forms; the headline re-derives its weights every step).  Same algorithm and
    operation order, so the nll -- which carries 3 sum(logs) HW + log|det W| HW of all 3 K steps -- must be the same BITS as with the
    switch 0x200000 (the workgroup LU), with near-orthogonal and with badly scaled matrices (pivoting really swaps rows there), and
    within 1e-4 of the oracle's torch.det route (inside _case).  Reference: network/module.py:356-357."""
    K, batch = 3, 8
    plan, fwd, rev = _case(64, 3, K, 512, batch, seed=33)          # (oracle parity of z / nll / decode on the default path)
    cfg = O.default_cfg(image_shape=(64, 64, 3), hidden_channels=512, K=K, L=3, batch=batch)
    for perturb in (0.02, 0.6):
        sd = O.seeded_state_dict(cfg, seed=34, invconv_perturb=perturb)
        g = torch.Generator().manual_seed(34)
        x = torch.rand(batch, 3, 64, 64, generator=g)
        noise = torch.rand(batch, 3, 64, 64, generator=g) / 256
        sd = O.glow_init_actnorm(x, noise, sd, cfg)
        glow = make_glow(cfg, sd, batch)
        z, nll, _ = glow.normal_flow(dev(x), None, noise=dev(noise), repack=True)
        G.lib().glowhip_debug_force_tail_tile(0x200000)
        try:
            z2, nll2, _ = glow.normal_flow(dev(x), None, noise=dev(noise), repack=True)
        finally:
            G.lib().glowhip_debug_force_tail_tile(0)
        assert torch.isfinite(nll).all() and torch.equal(nll, nll2) and torch.equal(z, z2), perturb
        _, nll_ref, _ = O.glow_forward(x, noise, sd, cfg)
        close(nll, nll_ref, 1e-4, what=f"nll at invconv perturbation {perturb}")


def test_config_b_geometry_batch16(path):
    """Batch 16: k_cnet splits the h2 rows over 2 / 4 workgroups per tile at the levels whose pixel tiles alone would leave CUs idle."""
    K = 3
    plan, fwd, rev = _case(64, 3, K, 512, 16, seed=6)
    _family_ran(path, fwd, rev, 3 * K)


def test_w64_level(path):
    """The 64-pixel-wide level (config D level 1, 128x128 input): a 64-pixel tile is one image row (wshift = 6)."""
    K = 2
    plan, fwd, rev = _case(128, 2, K, 256, 3, seed=7)
    _family_ran(path, fwd, rev, 2 * K)


def test_additive_and_shuffle(path):
    """Additive coupling (Cout = C/2 tail rows) and a gather permutation."""
    K = 2
    plan, fwd, rev = _case(64, 2, K, 512, 16, seed=8, coup="additive", perm="reverse")
    _family_ran(path, fwd, rev, 2 * K)


@pytest.mark.parametrize("ms", [1, 2, 4])
@pytest.mark.parametrize("image,L,hidden,batch", [(64, 3, 512, 5), (64, 3, 256, 3), (32, 2, 128, 3), (128, 2, 512, 2)])
def test_cnet_every_row_split(ms, image, L, hidden, batch):
    """k_cnet with the h2 rows forced onto 1, 2 and 4 workgroups per pixel tile (the launch heuristic picks by grid size): odd
    batches (a half-filled last tile at the 8x8 level), three hidden widths, tile rows from 16 (8-wide level: two images per
    tile) down to 2 (64-wide level), halo rows between tiles summed by the finishing kernel."""
    G.lib().glowhip_debug_force_tail_tile(ms << 22)
    try:
        plan, fwd, rev = _case(image, L, 2, hidden, batch, seed=30 + ms)
    finally:
        G.lib().glowhip_debug_force_tail_tile(0)
    assert ncnet(fwd) == 2 * L, fwd
    # which instance ran, from the executor's run-time counters: at this batch every level takes 64-pixel tiles, so ms = 1 is the
    # bench's <512, 1, 64> instance (its level 2) and ms = 4 its <512, 4, 64> (level 3), each against the oracle
    if (image, L, hidden) == (64, 3, 512):
        got = {k for k in fwd if k.startswith("variant:")}
        assert f"variant:k_cnet<512,{ms},64>" in got, (ms, got)


@pytest.mark.parametrize("tile", [0x2000000, 0x4000000])
@pytest.mark.parametrize("chained", [True, False])
def test_cnet_tile_sizes_and_chaining(tile, chained):
    """128- and 64-pixel tiles forced (the heuristic picks by grid size), each with the finishing step as its own kernel (the
    default) and chained into the next FlowStep's launch (0x8000000): the two orders of evaluation must agree BIT FOR BIT -- same arithmetic, log-det terms
    summed as fixed point -- and both match the oracle."""
    outs = []
    CHAIN = 0x8000000
    for flags in [tile | (CHAIN if chained else 0), tile | (0 if chained else CHAIN)]:
        G.lib().glowhip_debug_force_tail_tile(flags)
        try:
            cfg = O.default_cfg(image_shape=(32, 32, 3), hidden_channels=128, K=3, L=2, batch=5)
            sd = O.seeded_state_dict(cfg, seed=77, invconv_perturb=0.02)
            g = torch.Generator().manual_seed(77)
            x = torch.rand(5, 3, 32, 32, generator=g); noise = torch.rand(5, 3, 32, 32, generator=g) / 256
            glow = make_glow(cfg, sd, 5)
            plan = glow.flow.plan_for(dev(x)); plan.launch_counts(reset=True)
            z, nll, _ = glow.normal_flow(dev(x), None, noise=dev(noise))
            counts = plan.launch_counts(reset=True)
            xr = glow.reverse_flow(z, None, eps=[torch.zeros(5, 6, 16, 16, device=z.device)])
            outs.append((z.clone(), nll.clone(), xr.clone(), counts))
        finally:
            G.lib().glowhip_debug_force_tail_tile(0)
    (z0, n0, x0, c0), (z1, n1, x1, c1) = outs
    assert ("k_cnet+prev_finish" in c0) != ("k_cnet+prev_finish" in c1), (c0, c1)
    assert torch.equal(z0, z1) and torch.equal(n0, n1) and torch.equal(x0, x1)
    z_ref, nll_ref, _ = O.glow_forward(x, noise, sd, cfg)
    close(z0, z_ref, 1e-4, what="z"); close(n0, nll_ref, 1e-4, what="nll")


@pytest.mark.parametrize("case", range(14))
def test_fuzz_parity_seeded(case, path):
    """tests/fuzz_parity.py's randomised sweep (image 16..128, L, K, hidden 64..512, coupling, permutation, batch 1..48) as
    seeded regression cases, BIG shapes included."""
    import fuzz_parity
    line, worst, counts = fuzz_parity.run_case(case, big=True)
    print(line)
    assert worst < 1e-4


# ------------------------------------------------------------------------------------------------ range / non-finite handling
def _range_case(scale_h):
    """One FlowStep-deep Glow whose f.0 ActNorm scale is blown up so that h1 reaches ~scale_h (the fp16 pairs of the split-half
    kernels hold |v| < 4094 with the activation pre-scale 16)."""
    cfg = O.default_cfg(image_shape=(16, 16, 3), hidden_channels=64, K=1, L=1, batch=4)
    sd = O.seeded_state_dict(cfg, seed=3, zeros_std=1e-6)
    sd["flow.layers.1.f.0.actnorm.logs"] = sd["flow.layers.1.f.0.actnorm.logs"] + float(np.log(scale_h)) / 3.0
    sd["flow.layers.1.f.2.actnorm.logs"] = sd["flow.layers.1.f.2.actnorm.logs"] - float(np.log(scale_h)) / 3.0   # keep h2 modest
    g = torch.Generator().manual_seed(9)
    x = torch.rand(4, 3, 16, 16, generator=g); noise = torch.rand(4, 3, 16, 16, generator=g) / 256
    z_ref, nll_ref, _ = O.glow_forward(x, noise, sd, cfg)
    return cfg, sd, x, noise, z_ref, nll_ref


@pytest.mark.parametrize("flags", [0, 0x800])
def test_range_overflow_is_never_a_finite_wrong_answer(flags):
    """Hidden activations of ~1e5 overflow the fp16 pairs of the product kernels but not fp32 (0x800:
    exact-fp32 kernels).  The reference stays finite.  Overflowing paths must report a NON-FINITE nll (sticky per-sample flag:
    a NaN partial sum cannot hide in the fixed-point accumulator), never a finite wrong one; the fp32 kernels must match the
    oracle; and safe=True must recover the reference's answer through the exact-fp32 re-run."""
    cfg, sd, x, noise, z_ref, nll_ref = _range_case(3e5)
    assert torch.isfinite(nll_ref).all()
    G.lib().glowhip_debug_force_tail_tile(flags)
    try:
        glow = make_glow(cfg, sd, 4)
        z, nll, _ = glow.normal_flow(dev(x), None, noise=dev(noise))
        if flags == 0x800:
            close(nll, nll_ref, 1e-4, what="nll (exact fp32 kernels)")
            return
        ok = torch.isfinite(nll).cpu()
        assert not ok.any() or (nll.cpu()[ok] - nll_ref[ok]).abs().max() < 1e-4, "finite AND wrong nll after an fp16-range overflow"
        if flags == 0:
            assert (~ok).any(), "this input is meant to overflow the split-half range"
    finally:
        G.lib().glowhip_debug_force_tail_tile(0)
    if flags != 0:
        return
    n0 = G.Glow._RANGE_FALLBACKS
    z2, nll2, _ = glow.normal_flow(dev(x), None, noise=dev(noise), safe=True)
    assert G.Glow._RANGE_FALLBACKS == n0 + 1
    close(nll2, nll_ref, 1e-4, what="nll after the exact-fp32 re-run"); close(z2, z_ref, 1e-4, what="z after the re-run")
    # and the product path is back afterwards
    cfgm, sdm, xm, noisem, zm_ref, nllm_ref = _range_case(1.0)
    glow_m = make_glow(cfgm, sdm, 4)
    zq, nllq, _ = glow_m.normal_flow(dev(xm), None, noise=dev(noisem), safe=True)
    assert G.Glow._RANGE_FALLBACKS == n0 + 1
    close(nllq, nllm_ref, 1e-4, what="nll (in range)")


def test_nan_input_gives_nan_nll_on_every_path():
    """A NaN pixel must give a NaN nll for that sample (and only that sample) -- the fixed-point accumulators cannot hold it, the
    sticky flag does."""
    cfg, sd, x, noise, _, nll_ref = _range_case(1.0)
    x = x.clone(); x[2, 1, 5, 5] = float("nan")
    for flags in (0, 0x800):
        G.lib().glowhip_debug_force_tail_tile(flags)
        try:
            glow = make_glow(cfg, sd, 4)
            _, nll, _ = glow.normal_flow(dev(x), None, noise=dev(noise))
        finally:
            G.lib().glowhip_debug_force_tail_tile(0)
        nll = nll.cpu()
        assert torch.isnan(nll[2]) and torch.isfinite(nll[[0, 1, 3]]).all(), (flags, nll)
        assert (nll[[0, 1, 3]] - nll_ref[[0, 1, 3]]).abs().max() < 1e-4


def test_nan_of_either_sign_survives_the_integer_relu():
    """k_cnet's ReLU is one integer min on the negated value (sh.h nrelu_bits), which keeps NaNs whose sign bit is set -- what gfx950
    generates; NaNs from memory are canonicalised on the way in.  A NaN pixel of EITHER sign, placed in z1 and in z2, must still give
    a NaN nll for its sample only."""
    cfg, sd, x, noise, _, nll_ref = _range_case(1.0)
    for bits in (0x7fc00000, 0xffc00000, 0x7f800001):
        for ch in (0, 1, 2):
            xx = x.clone()
            xx.view(torch.int32)[1, ch, 3, 4] = torch.tensor(bits - (1 << 32) if bits >= 1 << 31 else bits, dtype=torch.int32)
            glow = make_glow(cfg, sd, 4)
            _, nll, _ = glow.normal_flow(dev(xx), None, noise=dev(noise))
            nll = nll.cpu()
            assert torch.isnan(nll[1]) and torch.isfinite(nll[[0, 2, 3]]).all(), (hex(bits), ch, nll)


def test_kernel_family_is_a_property_of_the_plan():
    """glowhip_plan_set_family: two plans in one process on different families, no process-wide switch.  The
    exact-fp32 plan launches no split-half kernel and handles the overflowing input; the other plan keeps running k_cnet; switching a
    plan back needs no re-pack of the product images."""
    cfg, sd, x, noise, z_ref, nll_ref = _range_case(3e5)
    ga, gb = make_glow(cfg, sd, 4), make_glow(cfg, sd, 4)
    pa, pb = ga.flow.plan_for(dev(x)), gb.flow.plan_for(dev(x))
    pb.set_family(pb.FAMILY_EXACT_FP32)
    assert pa.family == pa.FAMILY_AUTO and pb.family == pb.FAMILY_EXACT_FP32
    pa.launch_counts(reset=True); pb.launch_counts(reset=True)
    _, nll_a, _ = ga.normal_flow(dev(x), None, noise=dev(noise))
    zb, nll_b, _ = gb.normal_flow(dev(x), None, noise=dev(noise))
    ca, cb = pa.launch_counts(), pb.launch_counts()
    assert ca.get("k_cnet", 0) == 1 and "k_cnet" not in cb and not any(k.endswith("_sh") or "_sh+" in k for k in cb), (ca, cb)
    assert not torch.isfinite(nll_a).all()
    close(nll_b, nll_ref, 1e-4, what="nll (exact-fp32 plan)"); close(zb, z_ref, 1e-4, what="z (exact-fp32 plan)")
    # back to the product family: same bits as a plan that never left it
    cfgm, sdm, xm, noisem, zm_ref, nllm_ref = _range_case(1.0)
    g1, g2 = make_glow(cfgm, sdm, 4), make_glow(cfgm, sdm, 4)
    p2 = g2.flow.plan_for(dev(xm))
    p2.set_family(p2.FAMILY_EXACT_FP32); g2.normal_flow(dev(xm), None, noise=dev(noisem)); p2.set_family(p2.FAMILY_AUTO)
    z1, n1, _ = g1.normal_flow(dev(xm), None, noise=dev(noisem))
    z2, n2, _ = g2.normal_flow(dev(xm), None, noise=dev(noisem))
    assert torch.equal(z1, z2) and torch.equal(n1, n2)
    close(n1, nllm_ref, 1e-4, what="nll (in range)")


def test_decode_overflow_is_flagged_and_the_checked_path_recovers():
    """Glow.reverse_flow has no nll that would show an fp16-range overflow: glowhip_plan_status returns a per-sample flag word with
    x (sticky log-det flags | a non-finite pixel).  A decode through hidden activations of ~3e5: the product path's x is flagged
    for every sample (and is non-finite there, never finite and wrong); safe=True -- what Glow.forward(reverse=True) does in eval --
    decodes again on the exact-fp32 kernels and returns the oracle's x."""
    cfg, sd, x, noise, z_ref, nll_ref = _range_case(3e5)
    x_ref = O.glow_reverse(z_ref, sd, cfg, [])
    glow = make_glow(cfg, sd, 4)
    plan = glow.flow.plan_for(dev(x))
    xd = glow.reverse_flow(dev(z_ref), None, None, eps=[])
    st = plan.status(4, xd).cpu()
    assert (st != 0).all(), st
    bad = ~torch.isfinite(xd).cpu().flatten(1).all(1)
    fin = torch.isfinite(xd).cpu()
    assert bad.all() and ((xd.cpu() - x_ref).abs()[fin] < 1e-4).all(), "a flagged sample may hold non-finite pixels, never finite wrong ones"
    n0 = G.FlowModel._RANGE_FALLBACKS
    glow.eval()
    with torch.no_grad():
        xs = glow(z=dev(z_ref), y_onehot=None, eps_std=None, reverse=True)        # the reference's call: checked by default
    assert G.FlowModel._RANGE_FALLBACKS == n0 + 1 and plan.family == plan.FAMILY_AUTO
    close(xs, x_ref, 1e-4, what="x after the exact-fp32 re-run")
    # in range: status 0, no fall-back, same bits with and without the check
    cfgm, sdm, xm, noisem, zm_ref, _ = _range_case(1.0)
    gm = make_glow(cfgm, sdm, 4)
    xa = gm.reverse_flow(dev(zm_ref), None, None, eps=[])
    assert int(gm.flow.plan_for(dev(xm)).status(4, xa).abs().sum()) == 0
    xb = gm.reverse_flow(dev(zm_ref), None, None, eps=[], safe=True)
    assert G.FlowModel._RANGE_FALLBACKS == n0 + 1 and torch.equal(xa, xb)
    close(xa, O.glow_reverse(zm_ref, sdm, cfgm, []), 1e-4, what="x (in range)")


def test_forward_captured_in_a_hip_graph_equals_the_eager_forward():
    """Glow.capture_forward: the plan's static launch list (noise draw + glowhip_plan_pack + ~kernels) as ONE hipGraph launch.  Every
    replay must equal the eager forward on the noise it drew, bit for bit; successive replays draw different noise; an in-place
    parameter update between replays is seen (the pack is inside the graph)."""
    cfg = O.default_cfg(image_shape=(32, 32, 3), hidden_channels=128, K=3, L=2, batch=6)
    sd = O.seeded_state_dict(cfg, seed=4)
    glow = make_glow(cfg, sd, 6).eval()
    g = torch.Generator().manual_seed(2)
    x = dev(torch.rand(6, 3, 32, 32, generator=g))
    with torch.no_grad():
        gf = glow.capture_forward(x, repack=True)
        z1, n1 = (t.clone() for t in gf())
        noise1 = gf.noise.clone()
        ze, ne, _ = glow.normal_flow(x, None, noise=noise1)
        assert torch.equal(z1, ze) and torch.equal(n1, ne)
        z2, n2 = (t.clone() for t in gf())
        assert not torch.equal(gf.noise, noise1) and not torch.equal(n2, n1), "every replay draws fresh dequantisation noise"
        z_ref, nll_ref, _ = O.glow_forward(x.cpu(), gf.noise.cpu(), sd, cfg)
        close(z2, z_ref, 1e-4, what="z (graph replay)"); close(n2, nll_ref, 1e-4, what="nll (graph replay)")
        # an optimiser-style in-place update: the next replay packs the new weights
        w = glow.flow.layers[1].f[2].weight
        w.mul_(1.25)
        z3, n3 = (t.clone() for t in gf())
        ze3, ne3, _ = glow.normal_flow(x, None, noise=gf.noise.clone())
        assert torch.equal(z3, ze3) and torch.equal(n3, ne3) and not torch.equal(n3, n2)
        # another batch through the same graph
        xb = dev(torch.rand(6, 3, 32, 32, generator=g))
        z4, n4 = gf(xb)
        ze4, ne4, _ = glow.normal_flow(xb, None, noise=gf.noise.clone())
        assert torch.equal(z4, ze4) and torch.equal(n4, ne4)


# ------------------------------------------------------------------------------------------------ configs D and E at full size
@pytest.mark.parametrize("name,image,L,K,batch", [("D", 128, 4, 48, 32), ("E", 256, 6, 32, 16)])
def test_full_size_properties_configs_d_e(name, image, L, K, batch):
    """BASELINE configs[3] / [4] at their TRUE K, hidden width and per-GPU batch (no CPU oracle finishes these: D is 2e11, E
    5e11 FLOP per image) through size-independent properties: finite outputs; bitwise reproducible run to run; rows of the
    full batch equal the rows of a 2-image sub-batch (other tile / row-split choices: to rounding); encode -> decode with the
    dropped halves re-derived by a second encode returns the input; kernel selection from the run-time counters (the
    product kernel k_cnet on every level with C <= 96, W up to 128; config D entirely)."""
    import bench
    cfg = bench.CONFIGS[name]
    glow, hps = bench.build_model(G, G.misc.util, torch.device("cuda:0"), cfg, batch)
    g = torch.Generator().manual_seed(21)
    x = torch.rand(batch, 3, image, image, generator=g).to("cuda:0")
    noise = (torch.rand(batch, 3, image, image, generator=g) / 256).to("cuda:0")
    glow.train()
    glow.normal_flow(x, None, noise=noise)          # data-dependent ActNorm init on the full batch
    glow.eval()
    plan = glow.flow.plan_for(x)
    plan.launch_counts(reset=True)
    z, nll, _ = glow.normal_flow(x, None, noise=noise)
    counts = plan.launch_counts(reset=True)
    assert torch.isfinite(z).all() and torch.isfinite(nll).all()
    assert counts.get("k_cnet", 0) == 4 * K and counts.get("k_conv_direct", 0) == 0, counts       # levels 1-4: C = 12, 24, 48 and 96 (f.4 in two groups)
    if name == "D":       # every level of config D on the product kernel: no round-1 pair, no fp32 kernel left
        assert not any(k.endswith("_sh") or k.endswith("_f32") for k in counts), counts
    for rep in range(2):
        z2, nll2, _ = glow.normal_flow(x, None, noise=noise, repack=(rep == 1))
        assert torch.equal(z, z2) and torch.equal(nll, nll2), "not bitwise reproducible"
    zs, ns, _ = glow.normal_flow(x[:2].contiguous(), None, noise=noise[:2].contiguous())
    close(zs, z[:2], 5e-5, what=f"{name} batch slice z"); close(ns, nll[:2], 2e-5, what=f"{name} batch slice nll")
    # inverse: decode(z, eps = 0) = the mode of the dropped halves; re-encoding it must give back z (the flow is a bijection on
    # the kept half) and a log-det consistent with the forward's
    eps0 = [torch.zeros((batch,) + s, device="cuda:0") for s in glow.flow.split_shapes((3, image, image))]
    xm = glow.reverse_flow(z, None, eps=eps0)
    assert torch.isfinite(xm).all()
    zb, _ = glow.flow.encode(xm, 0.)
    close(zb, z, 2e-3 if name == "E" else 5e-4, what=f"{name} encode(decode(z)) vs z")


def test_in_kernel_dequantisation_noise():
    """Inference without a noise tensor: the leading squeeze draws U(0, 1/2^n_bits) itself (Philox4x32-10 keyed by torch's seed).
    (1) the draw, exported as a tensor, has the right range and moments; (2) a forward with that tensor injected is BITWISE equal
    to the in-kernel draw (fp32 and uint8 inputs); (3) consecutive calls use consecutive streams; (4) torch.manual_seed replays."""
    cfg = O.default_cfg(image_shape=(32, 32, 3), hidden_channels=64, K=2, L=2, batch=8)
    sd = O.seeded_state_dict(cfg, seed=5)
    glow = make_glow(cfg, sd, 8)
    x = torch.rand(8, 3, 32, 32, generator=torch.Generator().manual_seed(1)).to("cuda:0")
    plan = glow.flow.plan_for(x)
    from pytorch_glow_amd.network import model as M
    torch.manual_seed(1234)
    M.reset_dequant_stream()
    key, call0 = M.dequant_position()
    assert (key, call0) == (1234, 0)
    z_a, nll_a, _ = glow.normal_flow(x, None)                      # call number call0
    z_b, nll_b, _ = glow.normal_flow(x, None)                      # call0 + 1
    assert not torch.equal(z_a, z_b)
    n0 = plan.dequant_noise(x.shape, 1234, call0, 8)
    n1 = plan.dequant_noise(x.shape, 1234, call0 + 1, 8)
    assert n0.min().item() >= 0.0 and n0.max().item() < 1.0 / 256 and not torch.equal(n0, n1)
    assert abs(n0.mean().item() * 512 - 1.0) < 0.02 and abs(n0.var().item() * 12 * 256 * 256 - 1.0) < 0.05
    z_i, nll_i, _ = glow.normal_flow(x, None, noise=n0)
    assert torch.equal(z_i, z_a) and torch.equal(nll_i, nll_a)
    z_j, nll_j, _ = glow.normal_flow(x, None, noise=n1)
    assert torch.equal(z_j, z_b) and torch.equal(nll_j, nll_b)
    # 8-bit input path
    xu = (x * 255).round().to(torch.uint8)
    _, c = M.dequant_position()
    assert c == call0 + 2          # (forwards with an injected noise tensor do not advance the stream)
    z_u, nll_u, _ = glow.normal_flow(xu, None)
    z_v, nll_v, _ = glow.normal_flow(xu, None, noise=plan.dequant_noise(x.shape, 1234, c, 8))
    assert torch.equal(z_u, z_v) and torch.equal(nll_u, nll_v)
    # a new seed restarts the stream; the same seed again replays after reset_dequant_stream()
    torch.manual_seed(99)
    z_c, _, _ = glow.normal_flow(x, None)
    torch.manual_seed(99)
    M.reset_dequant_stream()
    z_d, _, _ = glow.normal_flow(x, None)
    assert torch.equal(z_c, z_d)
    # ONE stream per process: a plan of another batch shape continues it instead of repeating call 0
    glow5 = make_glow(cfg, sd, 5)
    x5 = x[:5].contiguous()
    _, cq = M.dequant_position()
    z5, _, _ = glow5.normal_flow(x5, None)
    z5_v, _, _ = glow5.normal_flow(x5, None, noise=glow5.flow.plan_for(x5).dequant_noise(x5.shape, 99, cq, 8))
    assert cq == 1 and torch.equal(z5, z5_v)


def test_unused_scratch_slots_may_hold_nan():
    """The finishing step reads the neighbour tiles' halo rows with unconditional loads and discards what does not apply.  The
    workspace is uninitialised memory: poison it with NaN before every call -- a 0/1-mask multiply would leak them (0 * NaN)."""
    cfg = O.default_cfg(image_shape=(64, 64, 3), hidden_channels=128, K=2, L=3, batch=3)
    sd = O.seeded_state_dict(cfg, seed=8, zeros_std=0.02)
    glow = make_glow(cfg, sd, 3)
    g = torch.Generator().manual_seed(2)
    x = torch.rand(3, 3, 64, 64, generator=g); noise = torch.rand(3, 3, 64, 64, generator=g) / 256
    z_ref, nll_ref, _ = O.glow_forward(x, noise, sd, cfg)
    plan = glow.flow.plan_for(dev(x))
    z0, nll0, _ = glow.normal_flow(dev(x), None, noise=dev(noise))
    for fill in (float("nan"), float("inf"), -1e30):
        plan._ws.view(torch.float32).fill_(fill)
        z, nll, _ = glow.normal_flow(dev(x), None, noise=dev(noise))
        assert torch.equal(z, z0) and torch.equal(nll, nll0), fill
    close(z0, z_ref, 1e-4, what="z"); close(nll0, nll_ref, 1e-4, what="nll")
    eps = [torch.randn(3, *s, generator=g) * 0.7 for s in glow.flow.split_shapes((3, 64, 64))]
    x_ref = O.glow_reverse(z_ref, sd, cfg, eps)
    plan._ws.view(torch.float32).fill_(float("nan"))
    xr = glow.reverse_flow(dev(z_ref), None, eps=[dev(e) for e in eps])
    close(xr, x_ref, 1e-4, what="decode")


def test_captured_forward_survives_packs_of_other_use_masks():
    """the pack keeps its job tables inside `packed`; a captured forward (repack=True) launches its image
    kernels over the INFERENCE table with grids baked at capture time.  A training step, an exact-fp32 pack or the init pass
    on the same plan selects other jobs -- they must go to their own table slot, or the next replay builds its weight images
    from a table sorted for another mask.  Replays after each kind of foreign pack must still equal the eager forward bit for
    bit, also after a real parameter update (the 'optimizer step between replays' the capture advertises)."""
    cfg = O.default_cfg(image_shape=(32, 32, 3), hidden_channels=128, K=3, L=2, batch=6)
    sd = O.seeded_state_dict(cfg, seed=6)
    glow = make_glow(cfg, sd, 6).eval()
    g = torch.Generator().manual_seed(8)
    x = dev(torch.rand(6, 3, 32, 32, generator=g))
    with torch.no_grad():
        gf = glow.capture_forward(x, repack=True)
        plan = gf.plan
        for use in (plan.PACK_TRAINING, plan.PACK_INFERENCE | plan.PACK_TRAINING | plan.PACK_INVERSE, plan.PACK_TRAINING | plan.PACK_INVERSE):
            plan.pack(use, merge=False)              # a foreign mask rewrites ITS slot only
            z, n = (t.clone() for t in gf())
            ze, ne, _ = glow.normal_flow(x, None, noise=gf.noise.clone())
            assert torch.equal(z, ze) and torch.equal(n, ne), f"replay after a pack with use={use}"
    # a training step (training pack + optimiser update of every parameter) between replays
    from pytorch_glow_amd import training
    glow.train()
    opt = training.HipAdam([p for p in glow.parameters()], lr=1e-3)
    with torch.enable_grad():
        _, nll, _ = glow.normal_flow(x, None)
        G.Glow.generative_loss(nll).backward()
    opt.fused_step(5.0, 100.0)
    glow.eval()
    with torch.no_grad():
        z, n = (t.clone() for t in gf())
        ze, ne, _ = glow.normal_flow(x, None, noise=gf.noise.clone())
        assert torch.equal(z, ze) and torch.equal(n, ne), "replay after a training step"
        z_ref, nll_ref, _ = O.glow_forward(x.cpu(), gf.noise.cpu(), {k: v.detach().cpu() for k, v in glow.state_dict().items()}, cfg)
        close(z, z_ref, 1e-4, what="z (replay after a training step)"); close(n, nll_ref, 1e-4, what="nll")


def test_captured_forward_refuses_a_replay_after_a_pack_for_the_other_kernel_family():
    """A pack for the exact-fp32 family (a checked forward that fell back, a re-run of an overflowed training batch) rebuilds the
    plan's host-side job tables with other contents; the copy nodes of a captured pack would read them through the addresses of
    capture time (a replay of the training step's graph after such a re-run faulted on a host address when this was first met).  A
    captured forward therefore refuses to replay -- with an error, not a fault -- until it is captured again; packs of the same
    family, whatever their image set, are fine (test_captured_forward_survives_packs_of_other_use_masks)."""
    cfg = O.default_cfg(image_shape=(32, 32, 3), hidden_channels=128, K=2, L=2, batch=4)
    sd = O.seeded_state_dict(cfg, seed=9)
    glow = make_glow(cfg, sd, 4).eval()
    x = dev(torch.rand(4, 3, 32, 32, generator=torch.Generator().manual_seed(5)))
    with torch.no_grad():
        gf = glow.capture_forward(x, repack=True)
        z1, n1 = (t.clone() for t in gf())
        plan = gf.plan
        plan.set_family(plan.FAMILY_EXACT_FP32)
        try:
            glow.normal_flow(x, None)                     # packs for the other family
        finally:
            plan.set_family(plan.FAMILY_AUTO)
        with pytest.raises(G._lib.GlowHipError, match="other kernel family"):
            gf()
        gf2 = glow.capture_forward(x, repack=True)        # a new capture works on
        z2, n2 = gf2()
        ze, ne, _ = glow.normal_flow(x, None, noise=gf2.noise.clone())
        assert torch.equal(z2, ze) and torch.equal(n2, ne) and torch.isfinite(n2).all()


def test_eager_calls_after_a_capture_with_side_stream_pack():
    """plans with C > 128 fork part of their pack onto a side stream and consumers join through events.  Events
    last recorded INSIDE a capture cannot be waited for from an eager stream (hipErrorCapturedEvent): after
    capture_forward(repack=True) the first eager encode / decode / pack must drop the stale pending flags instead."""
    cfg = O.default_cfg(image_shape=(64, 64, 3), hidden_channels=64, K=1, L=5, batch=2)     # levels up to C = 192 on 2x2 pixels
    sd = O.seeded_state_dict(cfg, seed=12)
    glow = make_glow(cfg, sd, 2).eval()
    g = torch.Generator().manual_seed(3)
    x = dev(torch.rand(2, 3, 64, 64, generator=g))
    with torch.no_grad():
        gf = glow.capture_forward(x, repack=True)
        z1, n1 = (t.clone() for t in gf())
        ze, ne, _ = glow.normal_flow(x, None, noise=gf.noise.clone())          # eager call right after the capture
        assert torch.equal(z1, ze) and torch.equal(n1, ne)
        xr = glow.reverse_flow(ze, None, eps_std=0.0)                           # decode packs W^-1 eagerly, joins the side stream
        assert torch.isfinite(xr).all()
        z2, n2 = (t.clone() for t in gf())
        ze2, ne2, _ = glow.normal_flow(x, None, noise=gf.noise.clone())
        assert torch.equal(z2, ze2) and torch.equal(n2, ne2)
