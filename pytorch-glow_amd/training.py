"""The reference's training step around the HIP flow (network/builder.py:9-20,108-126 and network/trainer.py:85-150),
one process per GPU.

Only the per-step arithmetic is here -- schedule -> forward (with tape) -> HIP backward -> RCCL gradient average ->
clip by value / by norm -> optimiser -- because that is what bounds training throughput.  The reference's logging,
snapshotting, sampling and dataset code around the loop is out of scope.  The optimisers themselves
are ``torch.optim.Adam`` / ``Adamax`` exactly as in the reference (builder.py:10-13).
"""
from functools import partial
from typing import Iterable, Optional

import time

import torch

from . import parallel
from .misc import lr_scheduler

import ctypes

from . import _lib


class _HipOptimizer(torch.optim.Optimizer):
    """torch.optim.Adam / Adamax with the whole step -- and the reference loop's two gradient clippings in front of it
    (network/trainer.py:142-150) -- as TWO HIP launches over all parameter tensors (csrc/optim.hip) instead of several foreach
    kernels per operation over ~1 060 tensors.  Same hyper-parameters, same per-parameter state names (``step``, ``exp_avg``,
    ``exp_avg_sq`` / ``exp_inf``), so ``state_dict()`` round-trips with torch's classes and the optimiser state of a
    reference snapshot (misc/util.py:309-322) loads.  The state tensors are views into two flat buffers.

    ``step()`` alone is torch's semantics; ``fused_step(max_grad_clip, max_grad_norm)`` also clips (in place, like
    ``clip_grad_value_`` / ``clip_grad_norm_``) and returns the total gradient norm as a device tensor -- no host sync."""

    KIND = 0
    SECOND = "exp_avg_sq"
    CHUNK = 1 << 16

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, **unsupported):
        for k, v in unsupported.items():
            if v not in (False, None, 0):
                raise ValueError(f"{type(self).__name__}: option {k}={v!r} is not supported on the HIP optimiser")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._flat = None
        self._steps = 0

    def _ensure_state(self):
        if self._flat is not None:
            return
        ps = [p for g in self.param_groups for p in g["params"]]
        if not ps:
            return
        dev = ps[0].device
        if dev.type != "cuda" or any(p.device != dev or p.dtype != torch.float32 or not p.is_contiguous() for p in ps):
            raise _lib.GlowHipError("the HIP optimiser needs contiguous fp32 parameters on one HIP device")
        total = sum(p.numel() for p in ps)
        m, v = torch.zeros(total, device=dev), torch.zeros(total, device=dev)
        off = 0
        for p in ps:
            n = p.numel()
            st = self.state[p]
            for name, buf in (("exp_avg", m), (self.SECOND, v)):
                view = buf[off:off + n].view_as(p)
                if name in st:                      # state loaded before the first step (load_state_dict)
                    view.copy_(st[name])
                st[name] = view
            st.setdefault("step", torch.tensor(float(self._steps)))
            off += n
        self._flat = (m, v)
        self._partial = None
        self._table_key = None       # the cached chunk table points into the previous state buffers
        self._keep = None

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = [float(s["step"]) for s in self.state.values() if "step" in s]
        self._steps = int(max(steps)) if steps else 0
        self._flat = None            # re-home the loaded tensors in the flat buffers at the next step
        self._ensure_state()

    @torch.no_grad()
    def fused_step(self, max_grad_clip=0.0, max_grad_norm=0.0, skip_nonfinite=False, grads_token=None, hyper_dev=None, norm_out=None):
        """``skip_nonfinite``: a NaN / inf total gradient norm leaves parameters and state untouched ON THE DEVICE (no host sync);
        the caller finds out from the returned norm when it next looks (``undo_step`` then takes the step count back).
        ``hyper_dev`` (3 doubles on the device, `hyper_values`) / ``norm_out`` (1 float on the device): the step as it sits in a
        captured graph (`GraphedTrainStep`) -- learning rate and bias corrections are read from device memory by the update kernel
        (glowhip_optim_step_dev), the norm lands in the caller's static word."""
        self._ensure_state()
        group0 = self.param_groups[0]
        dev = group0["params"][0].device
        for g in self.param_groups:
            if (g["betas"], g["eps"], g["weight_decay"]) != (group0["betas"], group0["eps"], group0["weight_decay"]) or g["lr"] != group0["lr"]:
                raise _lib.GlowHipError("the HIP optimiser takes one set of hyper-parameters for all parameter groups")
        # The chunk table holds parameter, gradient and state addresses.  The caching allocator usually hands the gradients the same
        # addresses again, so the table is keyed by (parameter address, gradient address) of every parameter -- two data_ptr() calls
        # each -- and rebuilt (with the full checks) only when one of them moved; the state addresses change only with the state
        # buffers, which resets the key (a reloaded state or a re-allocated parameter must not reuse the table).
        plist = getattr(self, "_plist", None)
        if plist is None:
            plist = self._plist = [p for g in self.param_groups for p in g["params"]]
        # ``grads_token``: an object that stands for "the gradients are the plan's persistent buckets" (Glow.loss_and_grads): as long
        # as the caller shows the same token, parameter, gradient and state addresses cannot have moved and the key below -- four
        # attribute reads for each of ~1 060 parameters, 1.2 ms -- is not recomputed
        fast = grads_token is not None and grads_token is getattr(self, "_table_token", None) and self._table_key is not None
        # (dtype and contiguity are part of the key: a gradient that comes back at the same address as a different view must not ride
        # on the checks the cached table was built under -- ADVICE r4)
        key = self._table_key if fast else tuple((p.data_ptr(), p.grad.data_ptr(), p.grad.dtype, p.grad.is_contiguous()) if p.grad is not None else (0, 0, None, True) for p in plist)
        self._steps += 1
        self._publish_step()
        norm = norm_out if norm_out is not None else torch.zeros(1, device=dev)
        if getattr(self, "_table_key", None) == key:
            table, n_chunks = self._keep, self._n_chunks
        else:
            chunks = []
            for p in plist:
                if p.grad is None:
                    continue
                gr = p.grad
                if not gr.is_contiguous() or gr.dtype != torch.float32:
                    raise _lib.GlowHipError("gradients must be contiguous fp32")
                st = self.state[p]
                n, base = p.numel(), (p.data_ptr(), gr.data_ptr(), st["exp_avg"].data_ptr(), st[self.SECOND].data_ptr())
                for o in range(0, n, self.CHUNK):
                    chunks.append((base[0] + 4 * o, base[1] + 4 * o, base[2] + 4 * o, base[3] + 4 * o, min(self.CHUNK, n - o)))
            n_chunks = len(chunks)
            if n_chunks:
                arr = (_lib.OptimChunk * n_chunks)(*[_lib.OptimChunk(*c, 0) for c in chunks])
                host = torch.frombuffer(bytearray(ctypes.string_at(arr, ctypes.sizeof(arr))), dtype=torch.uint8)
                table = host.to(dev, non_blocking=False)
            else:
                table = None
            self._table_key, self._n_chunks = key, n_chunks
        self._table_token = grads_token
        if not n_chunks:
            return norm[0]
        if self._partial is None or self._partial.numel() < n_chunks:
            self._partial = torch.empty(n_chunks, dtype=torch.float64, device=dev)
        if hyper_dev is not None:
            _lib.check(_lib.lib().glowhip_optim_step_dev(
                _lib.ptr(table), n_chunks, self.KIND, _lib.ptr(hyper_dev), float(group0["betas"][0]), float(group0["betas"][1]),
                float(group0["eps"]), float(group0["weight_decay"]), float(max_grad_clip or 0.0), float(max_grad_norm or 0.0),
                _lib.ptr(self._partial), _lib.ptr(norm), int(bool(skip_nonfinite)), _lib.stream_ptr(dev)))
        else:
            _lib.check(_lib.lib().glowhip_optim_step(
                _lib.ptr(table), n_chunks, self.KIND, float(group0["lr"]), float(group0["betas"][0]), float(group0["betas"][1]),
                float(group0["eps"]), float(group0["weight_decay"]), self._steps, float(max_grad_clip or 0.0), float(max_grad_norm or 0.0),
                _lib.ptr(self._partial), _lib.ptr(norm), int(bool(skip_nonfinite)), _lib.stream_ptr(dev)))
        self._keep = table      # alive until the stream has consumed it (the next step replaces it)
        # the kernel wrote the parameters behind torch's back: bump their version counters, which is what tells the flow plans to
        # re-derive their packed weight images (one call for the whole list)
        torch.autograd.graph.increment_version([p for p in plist if p.grad is not None])
        return norm[0]

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self.fused_step(0.0, 0.0)
        return loss

    def _publish_step(self):
        """`step` of every parameter's state (torch's optimisers keep one tensor per parameter; the value is the same for all of
        them here, so ONE tensor object is shared -- a thousand torch.tensor() calls per step were 1.5 ms of host time)."""
        t = torch.tensor(float(self._steps))
        for st in self.state.values():
            if "step" in st:
                st["step"] = t

    def state_dict(self):
        """torch's optimisers increment `step` IN PLACE per parameter after a load_state_dict: whoever receives this state must get
        one tensor object per parameter, not the shared one."""
        for st in self.state.values():
            if "step" in st:
                st["step"] = torch.tensor(float(self._steps))
        return super().state_dict()

    def hyper_values(self, lr, step):
        """{lr, 1 - beta1^step, 1 - beta2^step} as glowhip_optim_step computes them from its arguments (the learning rate passes
        through a float; the bias corrections are python-double arithmetic, as torch's)."""
        b1, b2 = self.param_groups[0]["betas"]
        return (ctypes.c_float(float(lr)).value, 1.0 - float(b1) ** step, 1.0 - float(b2) ** step)

    def undo_step(self):
        """Take back the count of a step the device skipped (fused_step(skip_nonfinite=True) with a non-finite norm)."""
        self._steps = max(self._steps - 1, 0)
        self._publish_step()


class HipAdam(_HipOptimizer):
    KIND, SECOND = 0, "exp_avg_sq"


class HipAdamax(_HipOptimizer):
    KIND, SECOND = 1, "exp_inf"

    def __init__(self, params, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, **kw):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **kw)


# hps.optim.optimizer -> class (builder.py:10-13).  The HIP classes are drop-in for torch.optim.Adam / Adamax (same arguments,
# same state_dict); GLOWHIP_TORCH_OPTIM=1 selects torch's own for A/B comparisons.
import os as _os
OPTIMIZERS = ({"adam": torch.optim.Adam, "adamax": torch.optim.Adamax} if _os.environ.get("GLOWHIP_TORCH_OPTIM") == "1"
              else {"adam": HipAdam, "adamax": HipAdamax})


def build_optimizer(hps, params: Iterable[torch.nn.Parameter]) -> torch.optim.Optimizer:
    """``hps.optim.optimizer`` / ``optimizer_args`` -> optimiser (builder.py:108-113)."""
    name = hps.optim.optimizer.lower()
    if name not in OPTIMIZERS:
        raise KeyError(f"unknown optimizer {name!r}; the reference knows {sorted(OPTIMIZERS)}")
    args = dict(hps.optim.optimizer_args)
    if "betas" in args:
        args["betas"] = tuple(args["betas"])
    return OPTIMIZERS[name](list(params), **args)


def build_scheduler(hps):
    """``hps.optim.lr_scheduler`` / ``lr_scheduler_args`` -> ``f(global_step) -> lr`` (builder.py:115-126: the base lr is the
    optimiser's, the remaining arguments come from the profile)."""
    name = hps.optim.lr_scheduler.lower()
    if name not in lr_scheduler.SCHEDULES:
        raise KeyError(f"unknown lr_scheduler {name!r}; the reference knows {sorted(lr_scheduler.SCHEDULES)}")
    args = dict(hps.optim.lr_scheduler_args)
    args.setdefault("base_lr", hps.optim.optimizer_args["lr"])     # builder.py:118-119: only when the profile does not set it
    return partial(lr_scheduler.SCHEDULES[name], **args)


class GraphedTrainStep:
    """One training step of one rank -- dequantisation draw, HIP forward with tape, HIP reverse sweep, both clippings, the Adam /
    Adamax update (reference network/trainer.py:123-150) -- captured in ONE hipGraph: the step's ~1 300 launches cost the host one
    graph launch instead of ~3 ms of Python + launch calls (and the ROCm runtime's helper thread nothing), which is what eight
    data-parallel ranks sharing a host want.  Single-rank steps only (the gradient exchange of a multi-rank step stays eager), no
    learned top prior (as `Glow.loss_and_grads`).

    Static buffers: ``x`` (the batch is copied in), the noise, ``loss``, ``norm``, and ``hyper`` = {lr, 1 - beta1^step, 1 - beta2^step}
    on the device: kernel arguments are frozen in a graph, so the update kernel reads the three values that change from step to step
    from memory (glowhip_optim_step_dev) and __call__ uploads them before each replay.  `glowhip_plan_pack` is inside the graph:
    every replay re-derives the weight images from the live parameters, as the forward after an update must.  Capture executes
    nothing, but it needs the lazy allocations of an eager step behind it: capture after at least one eager step of the same batch
    shape (`TrainLoop(graph=True)` does).  Same kernels, same bits as the eager step (tests/test_gpu_grad.py)."""

    def __init__(self, glow, optimizer, x, max_grad_clip=0.0, max_grad_norm=0.0, skip_nonfinite=False):
        if not (x.is_cuda and hasattr(optimizer, "fused_step") and hasattr(glow, "loss_and_grads")) or glow.hps.ablation.learn_top:
            raise _lib.GlowHipError("GraphedTrainStep: the HIP optimisers on a device batch, no learned top prior")
        self.glow, self.optimizer = glow, optimizer
        self.clip, self.max_norm, self.skip = max_grad_clip, max_grad_norm, skip_nonfinite
        dev = x.device
        self.x = x.clone()
        self.noise = torch.empty(x.shape, dtype=torch.float32, device=dev)
        self.n_bits = glow.hps.model.n_bits_x
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.hyper = torch.zeros(3, dtype=torch.float64, device=dev)
        # A pinned-source H2D copy reads the host words when the STREAM gets to it, not at enqueue time, and the host may run several
        # steps ahead of the device: one pinned slot per step in flight, each with the event recorded behind its copy, and a slot is
        # written again only once that event has passed (ADVICE r5 -- with a single slot, step N could read step N + 1's learning rate
        # and bias corrections)
        self._hyper_host = [torch.zeros(3, dtype=torch.float64, pin_memory=True) for _ in range(self.HYPER_SLOTS)]
        self._hyper_done = [None] * self.HYPER_SLOTS
        self._hyper_next = 0
        self._params = [p for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        glow.flow.plan_for(x).pack_sync()
        steps = optimizer._steps
        self.graph = torch.cuda.CUDAGraph()
        try:
            # (thread_local: a DataLoader's pinning thread may touch the device while this thread captures)
            with torch.no_grad(), torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
                self._body()
        except BaseException:
            optimizer._table_token = None     # (a table half-built under a failed capture must not be trusted by the eager steps that follow)
            raise
        finally:
            # the captured call counted a step on the host; nothing ran -- also when the capture failed half-way and TrainLoop carries
            # on eagerly: the bias corrections of every later step hang on this count
            optimizer._steps = steps
            optimizer._publish_step()
        self.plan = glow.flow.plan_for(x)
        self._buffers = self._buffer_signature()

    HYPER_SLOTS = 4

    def _upload_hyper(self, values):
        i = self._hyper_next
        self._hyper_next = (i + 1) % self.HYPER_SLOTS
        ev = self._hyper_done[i]
        if ev is not None:
            while not ev.query():         # the copy that last read this slot has not run yet: sleep, do not spin (see _check_previous)
                time.sleep(2e-4)
        host = self._hyper_host[i]
        host[0], host[1], host[2] = values
        self.hyper.copy_(host, non_blocking=True)
        if ev is None:
            ev = self._hyper_done[i] = torch.cuda.Event()
        ev.record()

    def _buffer_signature(self):
        """Addresses of everything the captured launches reach through the plan and the optimiser and that an EAGER call in between
        may re-allocate (the workspaces grow with the batch size and differ by kernel family; the optimiser's table and scratch
        follow the gradients): a replay over a moved buffer would write through a stale pointer."""
        plan, opt = self.plan, self.optimizer
        ptr = lambda t: t.data_ptr() if isinstance(t, torch.Tensor) else 0
        pg = getattr(plan, "_pgrad", None)
        return (ptr(plan.packed), ptr(getattr(plan, "_ws", None)), ptr(getattr(plan, "_tws", None)),
                tuple(ptr(f) for f in pg[0]) if pg else (), ptr(getattr(opt, "_partial", None)), ptr(getattr(opt, "_keep", None)),
                # an eager pack of this plan for the OTHER kernel family since the capture (a re-run of an overflowed batch): it rebuilt
                # the host-side job tables the captured pack's copy nodes read (measured: a replay after such a re-run faulted on a
                # host address); packs of the same family keep them in place
                getattr(plan, "_pack_epoch", 0),
                # an eager backward with NON-persistent gradients (normal_flow + loss.backward() on this plan) since the capture: it
                # rewrote the gradient job tables with pointers to its own temporary buffers
                getattr(plan, "_grad_table_epoch", 0))

    def valid(self):
        """False once a parameter or one of the buffers above moved: capture again (after an eager step)."""
        return self.plan.still_valid() and self._buffer_signature() == self._buffers

    def _body(self):
        self.noise.uniform_(0, 1. / 2 ** self.n_bits)
        loss = self.glow.loss_and_grads(self.x, noise=self.noise, force_pack=True)
        if hasattr(self.glow.flow, "pop_grad_buckets"):
            self.glow.flow.pop_grad_buckets()            # (one rank: nothing to exchange)
        token = getattr(getattr(self.glow, "_train_plan", None), "_pgrad_bound", None)
        self.optimizer.fused_step(self.clip, self.max_norm, skip_nonfinite=self.skip, grads_token=token, hyper_dev=self.hyper, norm_out=self.norm)
        self.loss.copy_(loss.detach())

    def __call__(self, x, lr):
        """Replay with this batch and learning rate; returns (loss, gradient norm) as fresh device scalars."""
        if not self.valid():
            raise _lib.GlowHipError("GraphedTrainStep: a parameter or a workspace was re-allocated since the capture -- capture again")
        if x is not self.x:
            self.x.copy_(x, non_blocking=True)
        opt = self.optimizer
        opt._steps += 1
        opt._publish_step()
        self._upload_hyper(opt.hyper_values(lr, opt._steps))
        self.graph.replay()
        torch.autograd.graph.increment_version(self._params)      # (the parameters changed behind torch's back, as after fused_step)
        out = torch.stack((self.loss, self.norm[0]))              # (the static words are overwritten by the next replay)
        return out[0], out[1]


class TrainLoop:
    """The state the reference's ``Trainer`` carries from step to step, minus its I/O: model, optimiser, schedule, step
    counter, clipping thresholds (trainer.py:44-60).  ``step(x_local)`` runs one iteration on this rank's shard."""

    GRAPH_AFTER = 3      # graph=True: the first steps run eagerly (data-dependent init, lazy allocations), then the step is captured

    def __init__(self, glow, hps, rank: int = 0, world: int = 1, optimizer: Optional[torch.optim.Optimizer] = None,
                 range_check: bool = True, graph: bool = False):
        """``range_check``: the training forward carries h1 as fp16 pairs through f.2 (|v| < 65504; the reference's fp32 has no
        such limit).  With the check on, a step whose gradient norm comes out non-finite is SKIPPED on the device (csrc/optim.hip:
        parameters and optimiser state untouched, no host sync); the host looks at the norm one step later -- when that step's
        work is already queued, so nothing stalls -- and runs the skipped batch again with the plan on the exact-fp32 family
        (the batch order changes by one; a norm that is non-finite there too is the model's own divergence and is left to the
        caller, counted in ``diverged_steps``)."""
        self.glow, self.hps, self.rank, self.world = glow, hps, rank, world
        self.range_check = range_check
        # ``graph``: from step GRAPH_AFTER on a single-rank step is ONE hipGraph launch (`GraphedTrainStep`); a batch of another
        # shape, a re-run on the exact-fp32 family and multi-rank steps run eagerly
        self.graph = graph
        self._graphed = None
        self.graph_error = None
        self.graph_recaptures = 0
        self.range_fallbacks = 0
        self.diverged_steps = 0
        self.reruns = []             # (global step at the time, loss, grad norm) of every batch that was run again (range check)
        self.last_rerun = None
        self._rerun = None
        self._pending = []           # checks not resolved yet, oldest first (at most MAX_LAG)
        self._host = None
        self.optimizer = optimizer or build_optimizer(hps, glow.parameters())
        self.scheduler = build_scheduler(hps)
        self.max_grad_clip = hps.ablation.get("max_grad_clip", 0)      # trainer.py:58-59
        self.max_grad_norm = hps.ablation.get("max_grad_norm", 0)
        self.global_step = 0
        self.lr = None

    def step(self, x_local: torch.Tensor):
        if self.global_step == 0 and not self.glow.actnorm_inited():
            # data-dependent ActNorm init on rank 0's batch, broadcast to the others (trainer.py:112-115)
            self.glow.train()
            parallel.data_dependent_init(self.glow, x_local, rank=self.rank, world=self.world)
        if not self.glow.training:      # (nn.Module.train() walks all ~6 400 sub-modules: 4.5 ms of host time when called every step)
            self.glow.train()
        self.lr = self.scheduler(global_step=self.global_step)
        for group in self.optimizer.param_groups:           # trainer.py:89-91
            group["lr"] = self.lr
        checked = self.range_check and hasattr(self.optimizer, "undo_step") and x_local.is_cuda
        self._rerun = None
        graphed = self._graph_for(x_local, checked)
        if graphed is not None:
            if checked:
                self._check_previous()       # (before this step's bias corrections are computed from the step count)
            loss, grad_norm = graphed(x_local, self.lr)
        else:
            loss, grad_norm = parallel.train_step(self.glow, self.optimizer, x_local, world=self.world,
                                                  max_grad_clip=self.max_grad_clip, max_grad_norm=self.max_grad_norm,
                                                  skip_nonfinite=checked, before_update=self._check_previous if checked else None)
        self.global_step += 1
        if checked:
            self._pending.append(self._stash(x_local, grad_norm, self.lr))
            reruns, self._rerun = self._rerun, None
            for pending in reruns or ():
                self._run_again(pending)
        return loss, grad_norm

    def _graph_for(self, x_local, checked):
        """The captured step for this batch, or None (eager): graph=True, one rank, a device batch, the HIP optimiser, past the
        eager warm-up steps; a failed capture is remembered (``graph_error``) and the loop stays eager."""
        if not self.graph or self.world != 1 or not x_local.is_cuda or self.global_step < self.GRAPH_AFTER or self.graph_error:
            return None
        if not hasattr(self.optimizer, "fused_step") or self.glow.hps.ablation.learn_top:
            return None
        g = self._graphed
        if g is not None and (g.x.shape != x_local.shape or g.x.dtype != x_local.dtype or g.skip != checked):
            # another batch shape: its lazy allocations (workspaces, tape, gradient tables) need an eager step behind them before a
            # capture, as the first one had -- this step runs eagerly, the next one captures
            self._graphed = None
            self.graph_recaptures += 1
            return None
        if g is not None and not g.valid():
            # an eager call in between (a re-run on the other kernel family, a forward of a larger batch) rebuilt the plan's job tables
            # or moved a workspace: this step runs eagerly -- it brings the plan back to the training step's state -- and the next one
            # captures again
            self._graphed = None
            self.graph_recaptures += 1
            return None
        if g is None:
            try:
                g = self._graphed = GraphedTrainStep(self.glow, self.optimizer, x_local, self.max_grad_clip, self.max_grad_norm, checked)
            except Exception as e:      # capture is an optimisation of the host side only
                self.graph_error = f"{type(e).__name__}: {str(e)[:300]}"
                return None
        return g

    # ---- deferred range check (no host sync on the step's own work)
    def _stash(self, x_local, grad_norm, lr):
        if self._host is None:       # pinned words, used in turn (one more than checks can be pending)
            self._host = [torch.zeros(1, pin_memory=True) for _ in range(self.MAX_LAG + 1)]
        host = self._host[self.global_step % (self.MAX_LAG + 1)]
        host.copy_(grad_norm.detach().reshape(1), non_blocking=True)
        ev = torch.cuda.Event(blocking=True)       # (a forced wait yields the core instead of spinning: eight ranks share one host)
        ev.record()
        return x_local, host, ev, lr

    MAX_LAG = 2      # a check is forced (host sync) once it is this many steps old

    def _check_previous(self, drain=False):
        """Runs between this step's gradient exchange and its optimiser step (`parallel.train_step`).  NON-BLOCKING: a norm that has
        not landed yet is looked at one step later (the host may run up to MAX_LAG steps ahead of the device -- with a blocking wait
        here it could never be more than one step ahead, and every step's host time included the device's previous step: 12 of
        the 25 ms `host_enqueue_ms_per_step` of round 4); only a check MAX_LAG steps old is waited for.  If the device skipped an
        update, its count is taken back before this step's bias corrections are computed from it, and the batch is queued to run
        again right after this step."""
        while self._pending:
            pending = self._pending[0]
            # world > 1: the check of step N is resolved at a FIXED lag (when MAX_LAG checks are pending), never earlier because its
            # event happens to have passed -- every rank must take a skipped step back, and issue the re-run's bucket all-reduces, at the
            # same point of its collective sequence (with the opportunistic look, rank A re-ran at N + 1 and rank B at N + 2,
            # pairing the re-run's all-reduce with another batch's).  The norm is the all-reduced one, identical on all ranks.
            if self.world > 1 and not drain and len(self._pending) < self.MAX_LAG:
                break
            if not pending[2].query():
                if not drain and len(self._pending) < self.MAX_LAG:
                    break
                # a forced wait: poll and SLEEP.  hipEventSynchronize spins on this runtime whatever the event's flags say (measured:
                # main thread + one runtime thread at 100 % for the whole wait, 52 ms of CPU per 29 ms step) -- a rank that is two
                # steps ahead of its GPU has nothing to do and must not take two of the host's cores to do it
                while not pending[2].query():
                    time.sleep(2e-4)
            self._pending.pop(0)
            if bool(torch.isfinite(pending[1]).all()):
                continue
            self.optimizer.undo_step()
            self.range_fallbacks += 1
            self._rerun = (self._rerun or []) + [pending]

    def _run_again(self, pending):
        """The skipped batch on the exact-fp32 family, with the learning rate of the step it belonged to.  Its loss and norm
        replace nothing that was already returned (that step reported NaN): they are kept in ``reruns`` / ``last_rerun``.
        Accepted ordering: the re-run of batch N is applied AFTER step N + 1 has updated the parameters -- the price of a
        range check without a host sync on the step's own work; an overflow is a once-in-a-run event and the two updates commute to
        first order in the learning rate.  If step N + 1 is skipped on the device as well, its count is only taken back at N + 2, so
        this re-run's bias correction uses s + 2 instead of s + 1: a relative change of the step size of O(beta^s), nothing after
        warm-up.  The float() conversions below are host syncs on this rare path only."""
        x_local, _, _, lr = pending
        plan = self.glow.flow.plan_for(x_local)
        prev = plan.family
        plan.set_family(plan.FAMILY_EXACT_FP32)
        for group in self.optimizer.param_groups:
            group["lr"] = lr
        try:
            loss, grad_norm = parallel.train_step(self.glow, self.optimizer, x_local, world=self.world,
                                                  max_grad_clip=self.max_grad_clip, max_grad_norm=self.max_grad_norm,
                                                  skip_nonfinite=True)
        finally:
            plan.set_family(prev)
            for group in self.optimizer.param_groups:
                group["lr"] = self.lr
        if not bool(torch.isfinite(grad_norm).all()):      # (a sync, on a path that only runs after an overflow)
            self.optimizer.undo_step()
            self.diverged_steps += 1
        self.last_rerun = (loss, grad_norm)
        self.reruns.append((self.global_step, float(loss), float(grad_norm)))

    def flush(self):
        """Resolve the check of the last step (call before reading parameters for a snapshot, and at the end of training)."""
        self._rerun = None
        self._check_previous(drain=True)
        reruns, self._rerun = self._rerun, None
        for pending in reruns or ():
            self._run_again(pending)
