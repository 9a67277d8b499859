"""CPU checks of the host-side mirror of the reference interface: constructors, attributes, state_dict
keys/shapes, profile loading, and that compute refuses to run without a GPU."""
import copy
import os

import numpy as np
import pytest
import torch

import pytorch_glow_amd as G
from pytorch_glow_amd.misc import ops, util
from oracle import glow_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_profiles_load_with_reference_schema():
    hps = util.load_profile("celeba")
    assert hps.model.image_shape == [64, 64, 3] and hps.model.K == 32 and hps.model.L == 3
    assert hps.model.hidden_channels == 512 and hps.ablation.flow_coupling == "affine"
    assert hps.ablation.flow_permutation == "invconv" and hps.ablation.seed == 2384
    assert hps.optim.optimizer_args.betas == [0.9, 0.9999]
    t = util.load_profile("test")
    assert t.ablation.flow_coupling == "additive" and t.optim.num_batch_train == 16
    assert util.load_profile("/nonexistent.json") is None


def test_get_devices():
    assert util.get_devices(["cpu"], verbose=False) == ["cpu"]
    n = torch.cuda.device_count()
    got = util.get_devices(["cuda:0", "cuda:1"], verbose=False)
    assert got == (["cpu"] if n == 0 else list(range(min(n, 2))))
    with pytest.raises(AssertionError):
        util.get_devices(["cpu", "cuda:0"], verbose=False)


def test_ops_helpers():
    t = torch.arange(2 * 6 * 4 * 4, dtype=torch.float32).reshape(2, 6, 4, 4)
    a, b = ops.split_channel(t, "simple")
    assert torch.equal(a, t[:, :3]) and torch.equal(b, t[:, 3:])
    c, d = ops.split_channel(t, "cross")
    assert torch.equal(c, t[:, 0::2]) and torch.equal(d, t[:, 1::2])
    assert torch.equal(ops.cat_channel(a, b), t)
    assert ops.count_pixels(t) == 16
    assert ops.reduce_sum(t, dim=[1, 2, 3]).shape == (2,)
    assert ops.reduce_mean(t, dim=[0, 2, 3], keepdim=True).shape == (1, 6, 1, 1)
    assert ops.tensor_equal(t, t + 5e-7) and not ops.tensor_equal(t, t + 1e-3) and not ops.tensor_equal(t, t[:1])
    assert torch.equal(ops.onehot(torch.tensor([1, 0]), 3), torch.tensor([[0., 1, 0], [1, 0, 0]]))


def test_state_dict_matches_reference_layout():
    hps = util.load_profile("celeba")
    glow = G.Glow(hps)
    sd = glow.state_dict()
    cfg = O.default_cfg(batch=glow.h_top.shape[0])
    ref = O.seeded_state_dict(cfg)
    assert set(sd) == set(ref)
    for k in ref:
        assert tuple(sd[k].shape) == tuple(ref[k].shape), k
    assert sum(v.numel() for k, v in sd.items() if k != "h_top") == 44_052_720
    assert glow.flow.output_shapes[0] == [-1, 12, 32, 32] and glow.flow.output_shapes[-1] == [-1, 48, 8, 8]
    assert len(glow.flow.layers) == 101 and glow.flow.K == 32 and glow.flow.L == 3
    assert tuple(glow.h_top.shape[1:]) == (96, 8, 8) and glow.batch_h_top == glow.h_top.shape[0]
    # layer kinds at the reference's indices: squeeze 0, steps 1..32, split 33
    assert isinstance(glow.flow.layers[0], G.Squeeze2d) and isinstance(glow.flow.layers[33], G.Split2d)
    assert all(isinstance(glow.flow.layers[i], G.FlowStep) for i in range(1, 33))
    # ActNorm flags are plain attributes, not state (reference module.py:29-30)
    assert not any("inited" in k for k in sd)
    glow.set_actnorm_inited()
    assert all(m.bias_inited and m.logs_inited for m in glow.modules() if isinstance(m, G.ActNorm))


def test_inits_follow_reference():
    np.random.seed(0)
    st = G.FlowStep(12, 64, permutation="invconv", coupling="affine")
    w = st.invconv.weight.detach().double()
    assert torch.allclose(w @ w.T, torch.eye(12, dtype=torch.float64), atol=1e-5)      # QR-orthogonal (module.py:341)
    assert abs(st.f[0].weight.std().item() - 0.05) < 0.01 and st.f[0].bias is None       # N(0,0.05), no bias
    assert st.f[2].weight.shape == (64, 64, 1, 1) and st.f[4].weight.shape == (12, 64, 3, 3)
    assert torch.count_nonzero(st.f[4].weight) == 0 and torch.count_nonzero(st.f[4].logs) == 0
    add = G.FlowStep(12, 64, coupling="additive")
    assert add.f[4].weight.shape[0] == 6
    p = G.Permutation2d(8)
    assert list(p.indices) == list(range(7, -1, -1)) and list(p.indices_inverse[p.indices]) == list(range(8))
    s = G.Permutation2d(8, shuffle=True)
    assert sorted(s.indices) == list(range(8)) and all(s.indices_inverse[s.indices[i]] == i for i in range(8))
    assert G.Conv2d.get_padding("SAME", 3, 1) == (1, 1) and G.Conv2d.get_padding("VALID", (3, 3), 1) == (0, 0)
    with pytest.raises(AssertionError):
        G.FlowStep(12, 8, permutation="bogus")
    with pytest.raises(AssertionError):
        G.FlowModel(in_shape=(16, 16, 2), hidden_channels=8, K=1, L=1)
    lz = G.LinearZeros(16, 16)
    assert torch.equal(lz(torch.rand(16)), torch.zeros(16))                               # reference test_module.py:22-29


def test_deepcopy_and_split_shapes():
    fm = G.FlowModel(in_shape=(64, 64, 3), hidden_channels=8, K=1, L=3)
    assert fm.split_shapes((3, 64, 64)) == [(12, 16, 16), (6, 32, 32)]                    # SURVEY R10 decode order
    assert fm._input_chw_for_latent((48, 8, 8)) == (3, 64, 64)
    fm2 = copy.deepcopy(fm)
    assert fm2._plans is not fm._plans
    assert all(torch.equal(a, b) for a, b in zip(fm.state_dict().values(), fm2.state_dict().values()))


def test_cpu_tensors_are_refused_not_silently_computed():
    x = torch.zeros(2, 4, 4, 4)
    for call in (lambda: G.ActNorm(4)(x), lambda: G.Squeeze2d()(x), lambda: G.FlowStep(4, 8)(x),
                 lambda: G.Invertible1x1Conv(4)(x), lambda: G.Conv2d(4, 8)(x), lambda: G.Split2d(4)(x),
                 lambda: G.Permutation2d(4)(x)):
        with pytest.raises(G.GlowHipError):
            call()


def test_lr_schedules_match_reference_formulas():
    """misc/lr_scheduler.py:4-96: values computed by hand from the reference's formulas."""
    from pytorch_glow_amd.misc import lr_scheduler as S
    assert S.constant(1e-3, 77) == 1e-3
    assert S.noam_decay(1e-3, 0, warmup_steps=4000) == pytest.approx(1e-3 / 4000)
    assert S.noam_decay(1e-3, 3999, warmup_steps=4000) == pytest.approx(1e-3)
    assert S.noam_decay(1e-3, 15999, warmup_steps=4000, min_lr=1e-4) == pytest.approx(5e-4)
    assert S.noam_decay(1e-3, 10 ** 9, warmup_steps=4000, min_lr=1e-4) == 1e-4          # floor after warm-up
    assert S.linear_anneal(1e-3, 50, num_train=10, warmup_steps=10) == pytest.approx(5e-4)
    assert S.linear_anneal(1e-3, 500, num_train=10, warmup_steps=10) == 1e-3
    assert S.step_anneal(1.0, 60001, anneal_rate=0.5, anneal_interval=30000) == 0.25
    assert S.cyclic_cosine_anneal(1.0, 1, t=100, m=4) == pytest.approx(1.0)
    assert S.cyclic_cosine_anneal(1.0, 1 + 25 // 2, t=100, m=4) == pytest.approx(0.5 * (np.cos(np.pi * 12 / 25) + 1))
    assert set(S.SCHEDULES) == {"constant", "noam", "linear", "step", "cyclic_cosine"}          # builder.py:14-20


def test_optimizer_and_scheduler_are_built_from_the_profile():
    from pytorch_glow_amd import training
    hps = util.load_profile("celeba")
    w = torch.nn.Parameter(torch.zeros(3))
    opt = training.build_optimizer(hps, [w])
    assert isinstance(opt, (torch.optim.Adam, training.HipAdam)) and opt.defaults["betas"] == (0.9, 0.9999) and opt.defaults["lr"] == 1e-3
    sched = training.build_scheduler(hps)
    assert sched(global_step=0) == pytest.approx(1e-3 / 4000) and sched(global_step=3999) == pytest.approx(1e-3)
    assert isinstance(training.build_optimizer(util.load_profile("test"), [w]), (torch.optim.Adamax, training.HipAdamax))
    hps.optim.optimizer = "sgd"
    with pytest.raises(KeyError):
        training.build_optimizer(hps, [w])


# ----------------------------------------------------------------------------- Builder / Inferer glue (next rows N2, N3)
def _g9_hps(batch=4):
    return util.AttrDict(dict(
        profile="g9", model=dict(image_shape=[16, 16, 3], hidden_channels=32, K=2, L=2, actnorm_scale=1.0, n_bits_x=24, weight_y=0.0),
        ablation=dict(learn_top=False, y_condition=False, lu_decomposition=False, flow_permutation="invconv",
                      flow_coupling="affine", max_grad_clip=5, max_grad_norm=100, seed=1),
        optim=dict(optimizer="adam", optimizer_args=dict(lr=1e-4, betas=[0.9, 0.9999], eps=1e-8), lr_scheduler="noam",
                   lr_scheduler_args=dict(warmup_steps=10, min_lr=1e-5), num_batch_train=batch),
        dataset=dict(num_classes=3, num_workers=0), device=dict(graph=["cuda:0"], data=["cuda:0"]),
        general=dict(result_dir=".", warm_start=True, pre_trained="", resume_run_id="", resume_step="")))


def test_reference_snapshot_loads_into_this_glow():
    """tests/golden/g9_reference_snapshot.pth was written by the REFERENCE's util.save_model: same keys, same shapes, and the
    Adam state fits an optimiser built over this package's parameters (snapshots are interchangeable)."""
    from conftest import GOLDEN, load_golden
    g = load_golden("g9_inferer")
    hps = _g9_hps()
    glow = G.Glow(hps)
    opt = torch.optim.Adam(glow.parameters(), lr=1e-4)
    state = util.load_model(GOLDEN, os.path.join(GOLDEN, "g9_reference_snapshot.pth"), glow, optimizer=opt, device="cpu")
    assert set(state) == {"step", "graph", "optimizer", "criterion", "seconds"}
    assert state["step"] == int(g["step"]) and state["seconds"] == pytest.approx(float(g["seconds"]))
    assert all(m.bias_inited and m.logs_inited for m in glow.modules() if isinstance(m, G.ActNorm))
    assert len(opt.state) == sum(1 for _ in glow.parameters())
    assert bytes(g["model_name"].astype(np.uint8)).decode() == util.get_model_name(7)
    assert bytes(g["best_name"].astype(np.uint8)).decode() == util.get_best_model_name()
    assert np.array_equal(util.make_interpolation_vector(3, step=0.5), g["interp"].numpy())


def test_snapshot_roundtrip_and_result_dirs(tmp_path):
    hps = _g9_hps()
    root = str(tmp_path)
    d0 = util.create_result_subdir(root, "exp", dict(hps))
    d1 = util.create_result_subdir(root, "exp", dict(hps))
    assert os.path.basename(d0) == "000-exp" and os.path.basename(d1) == "001-exp" and os.path.exists(os.path.join(d1, "config.json"))
    assert util.locate_result_subdir(root, 1) == d1 and util.locate_result_subdir(root, d0) == d0
    assert util.locate_result_subdir(root, 7) is None
    glow = G.Glow(hps)
    with torch.no_grad():
        for p in glow.parameters():
            p.copy_(torch.randn_like(p) * 0.1)
    opt = torch.optim.Adam(glow.parameters(), lr=1e-4)
    util.save_model(d1, 3, glow, opt, 2.0, is_best=False)
    util.save_model(d1, 12, glow, opt, 4.0, is_best=True)
    assert util.get_last_model_name(d1) == "network-snapshot-000012.pth"
    assert os.path.exists(os.path.join(d1, util.get_best_model_name()))
    other = G.Glow(hps)
    for key in (3, "best", "latest", os.path.join(d1, util.get_model_name(12))):
        st = util.load_model(d1, key, other, device="cpu")
        assert st["step"] == (3 if key == 3 else 12)
    assert all(torch.equal(a, b) for a, b in zip(glow.state_dict().values(), other.state_dict().values()))
    with pytest.raises(FileNotFoundError):
        util.load_model(d1, 99, other)
    t = torch.arange(12.).view(3, 2, 2)
    assert util.make_batch(t, 4).shape == (4, 3, 2, 2) and torch.equal(util.make_batch(t, 4)[3], t)
    util.save_deltaz(np.ones((3, 2)), os.path.join(root, "dz"))
    assert np.array_equal(util.load_deltaz(os.path.join(root, "dz", "deltaz.npy")), np.ones((3, 2)))
    assert util.load_deltaz(os.path.join(root, "missing.npy")) is None


def test_repo_written_snapshot_is_key_identical_to_the_reference_written_one(tmp_path):
    """a snapshot written by THIS package's util.save_model and the one the reference's util.save_model wrote
    (tests/golden/g9_reference_snapshot.pth, same model profile) have the same top-level keys, the same state_dict keys in
    the same order with the same shapes and dtypes, and the same optimizer-state layout -- the file format is the contract,
    the code that writes it is not."""
    from conftest import GOLDEN
    ref = torch.load(os.path.join(GOLDEN, "g9_reference_snapshot.pth"), map_location="cpu")
    glow = G.Glow(_g9_hps())
    opt = torch.optim.Adam(glow.parameters(), lr=1e-4)
    opt.load_state_dict(ref["optimizer"])            # (same parameter groups, so that the states below are comparable)
    util.save_model(str(tmp_path), int(ref["step"]), glow, opt, float(ref["seconds"]), is_best=True)
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".partial")]
    mine = torch.load(os.path.join(str(tmp_path), util.get_model_name(int(ref["step"]))), map_location="cpu")
    assert tuple(mine) == util.SNAPSHOT_KEYS and set(mine) == set(ref)
    assert list(mine["graph"]) == list(ref["graph"])
    assert all(mine["graph"][k].shape == ref["graph"][k].shape and mine["graph"][k].dtype == ref["graph"][k].dtype for k in ref["graph"])
    assert set(mine["optimizer"]) == set(ref["optimizer"])
    assert [sorted(g) for g in mine["optimizer"]["param_groups"]] == [sorted(g) for g in ref["optimizer"]["param_groups"]]
    assert {k: sorted(v) for k, v in mine["optimizer"]["state"].items()} == {k: sorted(v) for k, v in ref["optimizer"]["state"].items()}
    assert mine["criterion"] == ref["criterion"] == {}
    best = torch.load(os.path.join(str(tmp_path), util.get_best_model_name()), map_location="cpu")
    assert best["step"] == mine["step"] and list(best["graph"]) == list(mine["graph"])


def test_builder_needs_a_hip_device_and_knows_the_reference_tables():
    from pytorch_glow_amd.network import Builder
    assert set(Builder.optimizer_dict) == {"adam", "adamax"}
    assert set(Builder.lr_scheduler_dict) == {"constant", "noam", "linear", "step", "cyclic_cosine"}
    hps = _g9_hps()
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="HIP device"):
            Builder(hps).build(training=False)


def test_shard_sampler_gives_every_rank_its_slice_of_the_same_global_batches():
    """Trainer's data loading with one process per GPU: all ranks draw the SAME seeded permutation per epoch, a
    global batch is a contiguous run of it, rank r keeps positions [r*B/G, (r+1)*B/G) -- no overlap, no omission, 1/G of the
    loading per rank; a new epoch reshuffles."""
    from pytorch_glow_amd.network.trainer import _ShardSampler
    data = list(range(103))
    world, gb = 4, 16
    samplers = [_ShardSampler(data, gb, r, world, seed=7) for r in range(world)]
    epoch0 = [list(s) for s in samplers]
    per = gb // world
    nb = len(data) // gb
    assert all(len(e) == nb * per == len(s) for e, s in zip(epoch0, samplers))
    for b in range(nb):
        parts = [e[b * per:(b + 1) * per] for e in epoch0]
        batch = sum(parts, [])
        assert len(set(batch)) == gb                                   # disjoint shards
    allidx = sum(epoch0, [])
    assert len(set(allidx)) == nb * gb                                 # every sample of the kept batches exactly once
    # the union equals what ONE process with the global batch would have drawn, in the same order
    single = list(_ShardSampler(data, gb, 0, 1, seed=7))
    rebuilt = []
    for b in range(nb):
        for r in range(world):
            rebuilt += epoch0[r][b * per:(b + 1) * per]
    assert rebuilt == single
    epoch1 = [list(s) for s in samplers]
    assert epoch1 != epoch0 and len(set(sum(epoch1, []))) == nb * gb


def test_dequant_stream_is_keyed_by_seed_and_rank_and_counts_per_process(monkeypatch):
    """network/model.py dequant_position: one stream per process -- the key folds the data-parallel rank into torch's seed, the
    call number counts forwards across ALL plans, a new seed value restarts it, reset_dequant_stream() replays."""
    import torch
    import torch.distributed as dist
    from pytorch_glow_amd.network import model as M
    torch.manual_seed(11)
    M.reset_dequant_stream()
    k0, c0 = M.dequant_position(advance=True)
    k1, c1 = M.dequant_position(advance=True)
    assert (k0, c0, c1) == (11, 0, 1) and k1 == k0
    assert M.dequant_position() == (11, 2)                              # peeking does not advance
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_rank", lambda: 3)
    k3, c3 = M.dequant_position()
    assert k3 != k0 and c3 == 2 and k3 == (11 + 0x9E3779B97F4A7C15 * 3) % 2 ** 64
    monkeypatch.undo()
    torch.manual_seed(12)
    assert M.dequant_position() == (12, 0)                              # a new seed restarts the stream
    torch.manual_seed(11)
    M.dequant_position(advance=True)
    M.reset_dequant_stream()
    assert M.dequant_position() == (11, 0)


def test_multi_rank_range_check_is_resolved_at_a_fixed_lag():
    """with more than one rank the deferred range check of step N must be resolved at the same point of every
    rank's collective sequence -- when MAX_LAG checks are pending -- and never earlier because the rank's own event has already
    passed (a rank that re-ran the skipped batch a step before its peers paired the re-run's bucket all-reduces with another
    batch's).  One rank keeps the opportunistic early look.  Host logic only: the events and norms are stand-ins."""
    from pytorch_glow_amd import training

    class Passed:
        def query(self):
            return True

    class Opt:
        undone = 0

        def undo_step(self):
            self.undone += 1

    def loop(world):
        t = training.TrainLoop.__new__(training.TrainLoop)
        t.world, t._pending, t._rerun, t.range_fallbacks, t.optimizer = world, [], None, 0, Opt()
        return t

    bad = lambda: ("batch", torch.tensor([float("nan")]), Passed(), 1e-4)
    ok = lambda: ("batch", torch.tensor([1.0]), Passed(), 1e-4)
    assert training.TrainLoop.MAX_LAG == 2
    one, two = loop(1), loop(2)
    for t in (one, two):
        t._pending.append(bad())
        t._check_previous()
    assert one.range_fallbacks == 1 and not one._pending               # one rank: looked at as soon as it has landed
    assert two.range_fallbacks == 0 and len(two._pending) == 1         # two ranks: not yet, whatever the event says
    two._pending.append(ok())
    two._check_previous()                                              # MAX_LAG pending: the oldest is resolved now, and only it
    assert two.range_fallbacks == 1 and two.optimizer.undone == 1 and len(two._pending) == 1 and len(two._rerun) == 1
    two._check_previous(drain=True)                                    # flush(): everything
    assert not two._pending and two.range_fallbacks == 1


def test_bucket_overlap_report_arithmetic():
    """`parallel.bucket_overlap_report` (the N-rank bench line's `gradient_bucket_overlap`): per step the time the bucket all-reduces
    took on the side stream, the part still running after the backward sweep had finished on the main stream, the fraction hidden.
    Stand-in events on a common clock: host logic only."""
    from pytorch_glow_amd import parallel

    class Ev:
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t

    # step 1: sweep done at t = 10; buckets [2, 5], [6, 9], [9.5, 12] -> 8.5 ms of collectives, 2 ms of them after the sweep
    # step 2: everything over before the sweep ends -> nothing exposed
    recs = [{"sweep_done": Ev(10.0), "spans": [(Ev(2.0), Ev(5.0), 100), (Ev(6.0), Ev(9.0), 200), (Ev(9.5), Ev(12.0), 300)]},
            {"sweep_done": Ev(30.0), "spans": [(Ev(21.0), Ev(24.0), 100), (Ev(25.0), Ev(28.0), 200), (Ev(28.0), Ev(29.5), 300)]}]
    r = parallel.bucket_overlap_report(recs)
    assert r["steps"] == 2 and r["buckets_per_step"] == 3 and r["bytes_per_step"] == 600
    assert r["allreduce_ms_per_step"] == pytest.approx((8.5 + 7.5) / 2) and r["exposed_ms_per_step"] == pytest.approx(1.0)
    assert r["hidden_fraction"] == pytest.approx(1.0 - 1.0 / 8.0)
    assert parallel.bucket_overlap_report([]) is None and parallel.bucket_overlap_report([{"sweep_done": Ev(0), "spans": []}]) is None
