"""FlowStep / FlowModel / Glow with the reference's surface (corenel/pytorch-glow network/model.py),
executed as flow plans on libglowhip: one C call per encode / decode / Glow forward.

state_dict keys and shapes equal the reference's, so its snapshots load unchanged.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .._lib import require_device_tensor
from .._plan import PlanCache
from ..misc import util
from . import module
from .module import _deepcopy_without_plans, _logdet_arg


def _uninited_actnorms(mod):
    return [m for m in mod.modules() if isinstance(m, module.ActNorm) and not (m.bias_inited and m.logs_inited)]


def _all_actnorms(mod):
    return [m for m in mod.modules() if isinstance(m, module.ActNorm)]


def _maybe_data_dependent_init(owner, plan, x, noise, actnorm_scale):
    """First training-mode forward: set every ActNorm of the plan from this batch
    (reference network/module.py:45-46,66-67; trainer.py:112-115)."""
    if not owner.training or getattr(owner, "_actnorms_all_inited", None) == module.ActNorm.RESET_EPOCH[0]:
        return
    pending = _uninited_actnorms(owner)
    if not pending:
        # (the walk over ~6 400 sub-modules costs 1.5 ms: remembered until any ActNorm flag anywhere is cleared again)
        owner._actnorms_all_inited = module.ActNorm.RESET_EPOCH[0]
        return
    if len(pending) != len(_all_actnorms(owner)):
        raise _lib.GlowHipError("partially initialised ActNorm layers: call set_actnorm_inited() or reset all flags")
    plan.actnorm_init(x, noise, actnorm_scale)
    for m in pending:
        m.bias_inited = True
        m.logs_inited = True
    owner._actnorms_all_inited = module.ActNorm.RESET_EPOCH[0]


class FlowStep(nn.Module):
    """One step of flow: ActNorm -> permutation (invconv | reverse | shuffle) -> coupling (additive | affine).
    Reference network/model.py:10-173."""

    flow_permutation_list = ['invconv', 'reverse', 'shuffle']
    flow_coupling_list = ['additive', 'affine']
    glowhip_kind = _lib.LAYER_FLOWSTEP

    def __init__(self, in_channels, hidden_channels, permutation='invconv', coupling='additive',
                 actnorm_scale=1., lu_decomposition=False):
        super().__init__()
        assert permutation in self.flow_permutation_list, 'Unsupported flow permutation: {}'.format(permutation)
        assert coupling in self.flow_coupling_list, 'Unsupported flow coupling: {}'.format(coupling)
        self.permutation = permutation
        self.coupling = coupling
        self.in_channels = in_channels
        self.hidden_channels = hidden_channels
        self.actnorm_scale = actnorm_scale
        self.actnorm = module.ActNorm(num_channels=in_channels, scale=actnorm_scale)
        if permutation == 'invconv':
            self.invconv = module.Invertible1x1Conv(num_channels=in_channels, lu_decomposition=lu_decomposition)
        elif permutation == 'reverse':
            self.reverse = module.Permutation2d(num_channels=in_channels, shuffle=False)
        else:
            self.shuffle = module.Permutation2d(num_channels=in_channels, shuffle=True)
        if coupling == 'additive':
            self.f = module.f(in_channels // 2, hidden_channels, in_channels // 2)
        else:
            self.f = module.f(in_channels // 2, hidden_channels, in_channels)
        self._plans = PlanCache()

    def _plan(self, x):
        return self._plans.get([self], tuple(x.shape[1:]), x.device)

    def normal_flow(self, x, logdet=None):
        x = require_device_tensor(x, "FlowStep input")
        plan = self._plan(x)
        _maybe_data_dependent_init(self, plan, x, None, self.actnorm_scale)
        ld = _logdet_arg(logdet, x.shape[0], x.device)
        return plan.encode(x, None, ld, want_logdet=ld is not None)

    def reverse_flow(self, x, logdet=None):
        x = require_device_tensor(x, "FlowStep input")
        ld = _logdet_arg(logdet, x.shape[0], x.device)
        return self._plan(x).decode(x, [], ld, want_logdet=ld is not None)

    def forward(self, x, logdet=None, reverse=False):
        assert x.shape[1] % 2 == 0
        return self.reverse_flow(x, logdet) if reverse else self.normal_flow(x, logdet)

    def __deepcopy__(self, memo):
        return _deepcopy_without_plans(self, memo)


class FlowModel(nn.Module):
    """Multi-scale flow: [Squeeze2d, FlowStep x K, Split2d] x (L-1) + [Squeeze2d, FlowStep x K].
    Reference network/model.py:176-314.  ``in_shape`` is (H, W, C)."""

    def __init__(self, in_shape, hidden_channels, K, L, permutation='invconv', coupling='additive',
                 actnorm_scale=1., lu_decomposition=False):
        super().__init__()
        self.K = K
        self.L = L
        self.actnorm_scale = actnorm_scale
        assert len(in_shape) == 3
        assert in_shape[2] == 1 or in_shape[2] == 3
        nh, nw, nc = in_shape
        self.in_chw = (nc, nh, nw)
        self.layers = nn.ModuleList()
        self.output_shapes = []
        for i in range(L):
            self.layers.append(module.Squeeze2d(factor=2))
            nc, nh, nw = nc * 4, nh // 2, nw // 2
            self.output_shapes.append([-1, nc, nh, nw])
            for _ in range(K):
                self.layers.append(FlowStep(in_channels=nc, hidden_channels=hidden_channels, permutation=permutation,
                                            coupling=coupling, actnorm_scale=actnorm_scale,
                                            lu_decomposition=lu_decomposition))
                self.output_shapes.append([-1, nc, nh, nw])
            if i < L - 1:
                self.layers.append(module.Split2d(num_channels=nc))
                nc = nc // 2
                self.output_shapes.append([-1, nc, nh, nw])
        self._plans = PlanCache()

    def plan_for(self, x_or_chw, device=None):
        if isinstance(x_or_chw, torch.Tensor):
            return self._plans.get(list(self.layers), tuple(x_or_chw.shape[1:]), x_or_chw.device)
        return self._plans.get(list(self.layers), tuple(x_or_chw), device)

    def pop_grad_buckets(self):
        """[(flat gradient bucket, ready event)] of the backward that just ran (FlowPlan.glow_backward), in the order the buckets
        become final -- for `parallel.allreduce_buckets`; None when no HIP backward ran since the last call."""
        for plan in self._plans._plans.values():
            b = getattr(plan, "last_grad_buckets", None)
            if b is not None:
                plan.last_grad_buckets = None
                return b
        return None

    def invalidate_packed(self):
        """Call after writing parameters through ``.data`` (which torch's version counters do not see)."""
        self._plans.invalidate()

    def encode(self, z, logdet=0.):
        z = require_device_tensor(z, "FlowModel input")
        plan = self.plan_for(z)
        _maybe_data_dependent_init(self, plan, z, None, self.actnorm_scale)
        ld = _logdet_arg(logdet, z.shape[0], z.device)
        return plan.encode(z, None, ld, want_logdet=ld is not None)

    _RANGE_FALLBACKS = 0

    def decode(self, z, eps_std=None, eps=None, safe=False):
        """``eps``: optional list of injected draws, one per Split2d in decode order (deepest first).  ``safe``: see
        Glow.reverse_flow."""
        z = require_device_tensor(z, "FlowModel latent")
        n = z.shape[0]
        c, h, w = z.shape[1:]
        # input CHW of the plan from the latent shape: invert the squeeze/split bookkeeping
        plan = self._plans.get(list(self.layers), self._input_chw_for_latent((c, h, w)), z.device)
        if eps is None:
            eps = self.draw_eps(n, plan, eps_std, z.device)
        eps = [require_device_tensor(e, "eps") for e in eps]
        x, _ = plan.decode(z, eps, None, want_logdet=False)
        if safe and bool(plan.status(n, x).any()):       # out of the fp16 pairs' range somewhere: the exact-fp32 kernels, same draws
            FlowModel._RANGE_FALLBACKS += 1
            prev = plan.family
            plan.set_family(plan.FAMILY_EXACT_FP32)
            try:
                x, _ = plan.decode(z, eps, None, want_logdet=False)
            finally:
                plan.set_family(prev)
        return x

    def _input_chw_for_latent(self, chw):
        c, h, w = chw
        for i in range(self.L):
            if i > 0:
                c = c * 2
            c, h, w = c // 4, h * 2, w * 2
        return (c, h, w)

    def split_shapes(self, in_chw):
        """(C,H,W) of the z2 half dropped at each Split2d, in DECODE order (deepest first)."""
        c, h, w = in_chw
        shapes = []
        for i in range(self.L):
            c, h, w = c * 4, h // 2, w // 2
            if i < self.L - 1:
                c = c // 2
                shapes.append((c, h, w))
        return shapes[::-1]

    def draw_eps(self, n, plan, eps_std, device):
        std = eps_std or 1.  # reference network/module.py:419
        return [torch.randn((n,) + s, dtype=torch.float32, device=device) * std for s in self.split_shapes(plan.in_chw)]

    def forward(self, z, logdet=0., eps_std=None, reverse=False):
        if not reverse:
            return self.encode(z, logdet)
        return self.decode(z, eps_std)

    def __deepcopy__(self, memo):
        return _deepcopy_without_plans(self, memo)


class _GlowTrainFn(torch.autograd.Function):
    """Glow.normal_flow as one autograd node: forward records the HIP activation tape, backward runs the HIP reverse
    sweep (glowhip_glow_backward) and hands the parameter gradients to autograd."""

    @staticmethod
    def forward(ctx, plan, x, noise, n_bits, mean, logs, *params):
        # mean / logs: the top prior's parameters (N, Cz, H, W) when it is LEARNED (ablation.learn_top, network/model.py:375-376),
        # differentiable inputs of this node; None for the reference profiles' fixed N(0, 1) prior
        stride = 0 if mean is None else mean.stride(0)
        z, nll, tape = plan.glow_forward_train(x, noise, mean, logs, stride, n_bits)
        ctx.plan, ctx.tape, ctx.x_in = plan, tape, x
        ctx.want_gx = x.requires_grad
        ctx.prior = None if mean is None else (mean, logs, z)
        return z, nll

    @staticmethod
    def backward(ctx, gz, gnll):
        plan = ctx.plan
        n = ctx.x_in.shape[0]
        gnll = torch.zeros(n, device=ctx.x_in.device) if gnll is None else gnll.contiguous().float()
        gz = None if gz is None else gz.contiguous().float()
        mean, logs, z = ctx.prior if ctx.prior is not None else (None, None, None)
        grads, gx = plan.glow_backward(ctx.x_in, ctx.tape, gnll, gz, mean, logs, 0 if mean is None else mean.stride(0),
                                       want_grad_x=ctx.want_gx)
        ctx.tape = None
        gmean = glogs = None
        if mean is not None:
            # nll = -(... + logp(z | mean, logs)) / (ln 2 CHW), logp = sum -0.5 (ln 2 pi + 2 logs + (z - mean)^2 exp(-2 logs))
            # (GaussianDiag, network/module.py:400-467): the prior's own parameters get their gradient here, in a handful of
            # elementwise launches on the top latent; d nll / d z -- through the flow -- is the sweep's (it takes mean / logs)
            coef = (-gnll / (float(np.log(2.0)) * ctx.x_in[0].numel())).view(n, 1, 1, 1)
            d = (z - mean) * torch.exp(-2.0 * logs)
            gmean = coef * d
            glogs = coef * (d * (z - mean) - 1.0)
        return (None, gx, None, None, gmean, glogs) + tuple(grads)


# ---- the dequantisation stream of the inference path (network/model.py:421: z = x + U(0, 1/2^n_bits), drawn inside the leading
# squeeze kernel by Philox4x32-10).  ONE stream per process: key = torch's seed with the data-parallel rank folded in (ranks
# seeded alike still draw different noise), position = a process-wide count of such forwards (plans of different batch shapes
# do not repeat one another's draws).  A new torch.manual_seed value restarts it at 0; re-seeding with the SAME value cannot
# be seen from here -- call reset_dequant_stream() to replay.  (The reference consumes torch's CPU generator instead; the
# draws differ, their distribution does not.  Training mode and explicit `noise=` are unaffected.)
_DEQUANT_STREAM = {"seed": None, "calls": 0}


def reset_dequant_stream():
    _DEQUANT_STREAM["seed"], _DEQUANT_STREAM["calls"] = None, 0


def dequant_position(advance=False):
    """(key, call) the next in-kernel draw uses; advance=True consumes it."""
    import torch.distributed as dist
    seed = torch.initial_seed()
    st = _DEQUANT_STREAM
    if st["seed"] != seed:
        st["seed"], st["calls"] = seed, 0
    call = st["calls"]
    if advance:
        st["calls"] += 1
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    return (seed + 0x9E3779B97F4A7C15 * rank) & (2 ** 64 - 1), call


class GraphedForward:
    """`Glow.normal_flow` (eval, no grad; network/model.py:409-452) for one batch shape, captured in a hipGraph.

    Static buffers: ``x`` (copy the batch in, or pass one to __call__), ``noise``, ``z``, ``nll``.  What a replay runs: the
    dequantisation draw U(0, 2^-n_bits) with torch's Philox generator (registered with the graph, so every replay advances the
    stream like an eager draw), optionally `glowhip_plan_pack`, then the plan's launch list.  Parameters are read through their
    live addresses, so updates between replays are seen (with ``repack=True``); re-allocating a parameter or changing the
    kernel family needs a new capture.  No range fall-back inside a graph: look at ``nll`` (non-finite = flagged)."""

    def __init__(self, glow, x, repack=True):
        assert not glow.training, "capture_forward is the inference path: call glow.eval() first"
        x = require_device_tensor(x, "Glow input")
        self.glow, self.repack = glow, repack
        self.n_bits = glow.hps.model.n_bits_x
        self.plan = plan = glow.flow.plan_for(x)
        mean, logs = glow.prior(None)
        assert mean is None, "capture_forward: learn_top priors are not captured"
        self.x = x.clone()
        self.noise = torch.empty_like(self.x)
        n = x.shape[0]
        self.z = torch.empty((n,) + plan.out_chw, dtype=torch.float32, device=x.device)
        self.nll = torch.empty(n, dtype=torch.float32, device=x.device)
        self._obj = torch.empty(n, dtype=torch.float32, device=x.device)
        plan.set_dequant_rng(0, False)
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.no_grad(), torch.cuda.stream(side):          # warm-up on the capture stream: workspace, job tables, lazy inits
            for _ in range(2):
                self._body()
        torch.cuda.current_stream(x.device).wait_stream(side)
        torch.cuda.synchronize(x.device)
        plan.pack_sync()            # (repack=False: nothing of an earlier pack is left for the captured calls to join)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph, stream=side):
            self._body()
        self._version = plan._version_signature()
        self._pack_epoch = getattr(plan, "_pack_epoch", 0)

    def _body(self):
        self.noise.uniform_(0, 1. / 2 ** self.n_bits)
        self.plan.glow_forward(self.x, self.noise, None, None, 0, self.n_bits, repack=self.repack, out=(self.z, self.nll, self._obj))

    def __call__(self, x=None):
        """Replay; returns the static (z, nll) (valid until the next replay)."""
        if x is not None and x is not self.x:
            self.x.copy_(x)
        if not self.plan.still_valid():
            raise _lib.GlowHipError("capture_forward: a parameter was re-allocated since the capture -- capture again")
        if not self.repack and self.plan._version_signature() != self._version:
            raise _lib.GlowHipError("capture_forward(repack=False): the parameters changed since the capture")
        if self.repack and getattr(self.plan, "_pack_epoch", 0) != self._pack_epoch:
            # (an eager pack for the OTHER kernel family rebuilds the plan's host-side job tables; the captured pack's copy nodes read
            # them through the addresses of capture time -- a replay after a re-run on the exact-fp32 family faulted on a host
            # address when the training step's graph first met this, training.GraphedTrainStep)
            raise _lib.GlowHipError("capture_forward: the plan was packed for the other kernel family since the capture -- capture again")
        self.graph.replay()
        return self.z, self.nll


class _GradBinding:
    """Token for "the parameters' .grad are this plan's persistent gradient views" (`Glow.loss_and_grads`): the optimiser keeps its
    chunk table while it is shown the same token object."""
    __slots__ = ("grads",)

    def __init__(self, grads):
        self.grads = grads


class Glow(nn.Module):
    """Glow (reference network/model.py:317-550): dequantisation noise, flow encode, top prior, nll in bits/dim."""

    bce_criterion = nn.BCEWithLogitsLoss()
    ce_criterion = nn.CrossEntropyLoss()

    def __init__(self, hps):
        super().__init__()
        self.hps = hps
        self.flow = FlowModel(in_shape=hps.model.image_shape, hidden_channels=hps.model.hidden_channels,
                              K=hps.model.K, L=hps.model.L, permutation=hps.ablation.flow_permutation,
                              coupling=hps.ablation.flow_coupling, actnorm_scale=hps.model.actnorm_scale,
                              lu_decomposition=hps.ablation.lu_decomposition)
        if hps.ablation.learn_top:
            nc = self.flow.output_shapes[-1][1]
            self.learn_top = module.Conv2dZeros(in_channels=2 * nc, out_channels=2 * nc)
        if hps.ablation.y_condition:
            raise NotImplementedError("class-conditional Glow (y_condition) is outside the flow hot path "
                                      "(off in every reference profile)")
        num_device = len(util.get_devices(self.hps.device.graph, verbose=False))
        assert hps.optim.num_batch_train % num_device == 0
        self.register_parameter('h_top', nn.Parameter(torch.zeros([hps.optim.num_batch_train // num_device,
                                                                   self.flow.output_shapes[-1][1] * 2,
                                                                   self.flow.output_shapes[-1][2],
                                                                   self.flow.output_shapes[-1][3]])))

    @property
    def batch_h_top(self):
        return self.h_top.shape[0]

    def prior(self, y_onehot=None):
        """Top prior parameters (mean, logs).  The reference asserts h_top == 0 with a host sync on every
        call (network/model.py:373); the sync is not reproduced.  Returns (None, None) for the all-zero
        prior so the kernels skip the loads."""
        if not self.hps.ablation.learn_top:
            return None, None
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.learn_top.parameters()):
            # training: h_top is all zeros (the reference asserts it), so the Conv2dZeros output is (0 + bias) exp(3 logs) in every
            # pixel -- written with torch ops, which makes learn_top.bias / .logs differentiable (its weight multiplies zeros: no
            # gradient, as under the reference's autograd)
            lt = self.learn_top
            h = (lt.bias.view(1, -1, 1, 1) * torch.exp(lt.logs.view(1, -1, 1, 1) * 3.0)).expand(tuple(self.h_top.shape)).contiguous()
        else:
            h = self.learn_top(self.h_top.detach())
        nc = h.shape[1]
        return h[:, :nc // 2, ...], h[:, nc // 2:, ...]

    def normal_flow(self, x, y_onehot=None, noise=None, repack=False, safe=False):
        """z = x + U(0, 1/2^n_bits); objective = -ln(n_bins)*CHW + logdet + logp(z); nll = -objective/(ln2*CHW).
        ``noise`` (optional, beyond the reference signature) injects the dequantisation draw.

        Range: the product kernels carry the coupling networks' activations as fp16 pairs (csrc/sh.h): a hidden activation
        beyond 4094 becomes inf there where the fp32 reference stays finite.  That never yields a finite wrong answer -- a
        sticky per-sample flag makes the nll NaN -- and ``safe=True`` (inference path; costs one host sync) re-runs such a batch
        on the exact-fp32 MFMA kernels, so the result is the reference's up to fp32 rounding for any input the reference
        handles."""
        x = require_device_tensor(x, "Glow input", allow_uint8=True)   # uint8 = pixels as loaded, scaled by 1/255 in-kernel
        if x.dtype == torch.uint8 and (self.training or torch.is_grad_enabled()):
            x = x.float() / 255.0   # the ActNorm init pass and the training step take fp32; inference reads the bytes itself
        n_bits = self.hps.model.n_bits_x
        plan = self.flow.plan_for(x)
        in_kernel_rng = False
        if noise is None:
            if self.training or torch.is_grad_enabled():
                noise = torch.empty(x.shape, dtype=torch.float32, device=x.device).uniform_(0, 1. / 2 ** n_bits)
            else:     # inference: the leading squeeze draws it (Philox keyed by torch's seed; no noise tensor, no RNG launch)
                in_kernel_rng = True
                rng_key, rng_call = dequant_position(advance=True)      # (key, call number) this forward draws with
                plan.set_dequant_stream(rng_key, rng_call)
        else:
            noise = require_device_tensor(noise, "noise")
            assert noise.shape == x.shape
        if not in_kernel_rng:
            plan.set_dequant_rng(0, False)
        _maybe_data_dependent_init(self.flow, plan, x, noise, self.flow.actnorm_scale)
        mean, logs = self.prior(y_onehot)
        stride = 0
        if mean is not None:
            assert mean.shape[0] == x.shape[0], "batch must equal h_top's batch when learn_top is on"
            stride = mean.stride(0)
            assert mean[0].is_contiguous() and logs[0].is_contiguous() and logs.stride(0) == stride
        params = plan.trainable_parameters() if torch.is_grad_enabled() else ()     # (a walk over ~1 060 tensors: skipped on the inference path)
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            # training step: one autograd node over the whole flow (HIP forward with tape + HIP backward)
            z, nll = _GlowTrainFn.apply(plan, x, noise, n_bits, mean, logs, *params)
            return z, nll, None
        z, nll, _ = plan.glow_forward(x, noise, mean, logs, stride, n_bits, repack=repack)
        if safe and not bool(torch.isfinite(nll).all()):
            if in_kernel_rng:     # the re-run sees the same dequantisation draw as the flagged run
                noise = plan.dequant_noise(x.shape, rng_key, rng_call, n_bits)
            z, nll = self._forward_exact_fp32(plan, x.float() / 255.0 if x.dtype == torch.uint8 else x, noise, mean, logs, stride, n_bits)
        return z, nll, None

    def loss_and_grads(self, x, noise=None, force_pack=False):
        """mean(nll) of the batch (`generative_loss`) and its gradient for every trainable parameter, WITHOUT an autograd graph:
        the HIP forward with tape and the HIP reverse sweep are called directly, and the gradients land in the plan's persistent
        flat buckets, whose views are the parameters' ``.grad`` (assigned once).  This is `Trainer`'s step (network/trainer.py:123-133:
        forward, `loss.backward()`) minus ~10 ms of per-step host work in autograd's bookkeeping for ~1 060 parameter tensors --
        eight data-parallel ranks share one host.  Same kernels, same bits as ``normal_flow(x)`` + ``loss.backward()``
        (tests/test_gpu_grad.py).  Returns the loss (a device scalar); gradient buckets: ``flow.pop_grad_buckets()``."""
        assert self.training, "loss_and_grads is the training step: call glow.train() first"
        x = require_device_tensor(x, "Glow input", allow_uint8=True)
        if x.dtype == torch.uint8:
            x = x.float() / 255.0
        n_bits = self.hps.model.n_bits_x
        plan = self.flow.plan_for(x)
        if noise is None:
            noise = torch.empty(x.shape, dtype=torch.float32, device=x.device).uniform_(0, 1. / 2 ** n_bits)
        else:
            noise = require_device_tensor(noise, "noise")
        plan.set_dequant_rng(0, False)
        _maybe_data_dependent_init(self.flow, plan, x, noise, self.flow.actnorm_scale)
        if self.hps.ablation.learn_top:
            raise NotImplementedError("loss_and_grads: a learned top prior goes through the autograd route (normal_flow + backward)")
        with torch.no_grad():
            z, nll, tape = plan.glow_forward_train(x, noise, None, None, 0, n_bits, force_pack=force_pack)
            loss = self.generative_loss(nll)
            n = x.shape[0]
            gn = getattr(plan, "_mean_grad", None)
            if gn is None or gn.numel() != n:
                gn = plan._mean_grad = torch.full((n,), 1.0 / n, dtype=torch.float32, device=x.device)      # d mean(nll) / d nll
            grads, _ = plan.glow_backward(x, tape, gn, None, None, None, 0, want_grad_x=False, persistent=True)
        # the views ARE the parameters' gradients: bound once per plan -- and again whenever somebody took them away in between
        # (optimizer.zero_grad() sets .grad to None, an autograd backward assigns its own tensors): ~1 060 identity tests, 60 us.  A
        # re-bind hands out a NEW token, so the optimiser rebuilds its chunk table from the gradients that are there now
        params = plan.trainable_parameters()
        bound = getattr(plan, "_pgrad_bound", None)
        if bound is None or bound.grads is not grads or any(p.grad is not g for p, g in zip(params, grads)):
            for p, g in zip(params, grads):
                p.grad = g
            plan._pgrad_bound = _GradBinding(grads)
        self._train_plan = plan        # (parallel.train_step: the optimiser's table stays valid while this plan's buckets are the gradients)
        return loss

    def capture_forward(self, x, repack=True):
        """The inference forward of this batch shape as ONE hipGraph launch (`GraphedForward`): the flow plan's launch list is
        static (~220 kernels for celeba64), so it is captured once and replayed -- host cost per step ~10 us instead of ~1 ms of
        launches from Python / C.  ``repack=True`` keeps `glowhip_plan_pack` inside the graph: every replay re-derives the
        weight images from the LIVE parameters, as a forward after an optimiser step must."""
        return GraphedForward(self, x, repack=repack)

    _RANGE_FALLBACKS = 0   # how often safe=True had to re-run on the exact-fp32 kernels (diagnostics / tests)

    def _forward_exact_fp32(self, plan, x, noise, mean, logs, stride, n_bits):
        """The same forward on the exact-fp32 MFMA kernels (v_mfma_f32_32x32x2_f32): no fp16 range limit.  The kernel family is
        a property of the plan (glowhip_plan_set_family), set for this one call: nothing process-wide is touched, the product
        kernels' weight images stay valid for the next call."""
        Glow._RANGE_FALLBACKS += 1
        prev = plan.family
        plan.set_family(plan.FAMILY_EXACT_FP32)
        try:
            z, nll, _ = plan.glow_forward(x, noise, mean, logs, stride, n_bits)
        finally:
            plan.set_family(prev)
        return z, nll

    def reverse_flow(self, z, y_onehot=None, eps_std=None, eps=None, safe=False):
        """``safe=True``: the decode's range status (glowhip_plan_status: sticky log-det flags | a non-finite pixel) is read back
        -- one host sync -- and a flagged batch is decoded again on the exact-fp32 kernels with the same eps draws."""
        with torch.no_grad():
            if z is None:
                mean, logs = self.prior(y_onehot)
                if mean is None:
                    c2, h, w = self.h_top.shape[1:]
                    mean = logs = torch.zeros((self.batch_h_top, c2 // 2, h, w), device=self.h_top.device)
                z = module.GaussianDiag.sample(mean, logs, eps_std)
            return self.flow.decode(z, eps_std=eps_std, eps=eps, safe=safe)

    # Range policy of ``forward`` -- the call the reference's Trainer / Inferer make (network/trainer.py:113,123,163,171,
    # inferer.py:55,81,98,133).  Under torch.no_grad() in eval mode (validation, sampling, Inferer) the checked path is the default:
    # the result is the reference's for every input the reference handles, at the price of one host sync per call -- which those
    # callers pay anyway when they take the result to the host.  ``Glow.range_check = False`` (class or instance) switches it off;
    # benchmarks call normal_flow / reverse_flow directly (safe=False).  Training steps are checked by training.TrainLoop.
    range_check = True

    def forward(self, x=None, y_onehot=None, z=None, eps_std=None, reverse=False):
        safe = bool(self.range_check) and not self.training and not torch.is_grad_enabled()
        if not reverse:
            return self.normal_flow(x, y_onehot, safe=safe)
        return self.reverse_flow(z, y_onehot, eps_std, safe=safe)

    @staticmethod
    def generative_loss(nll):
        return torch.mean(nll)

    @staticmethod
    def single_class_loss(y_logits, y):
        if y_logits is None:
            return 0
        return Glow.ce_criterion(y_logits, y.long())

    @staticmethod
    def multi_class_loss(y_logits, y_onehot):
        if y_logits is None:
            return 0
        return Glow.bce_criterion(y_logits, y_onehot.float())

    def actnorm_inited(self):
        """True when every ActNorm already holds its data-dependent statistics (nothing left for a first batch to do)."""
        return not _uninited_actnorms(self)

    def set_actnorm_inited(self, inited=True):
        for name, m in self.named_modules():
            if m.__class__.__name__.find("ActNorm") >= 0:
                m.bias_inited = inited
                m.logs_inited = inited

