"""Data parallelism for the flow path, MI355X-style: ONE PROCESS PER GPU, parameters resident on every rank,
the minibatch sharded on dim 0, collectives over RCCL (torch.distributed backend "nccl"; "gloo" in the CPU
tests).  Replaces the reference's single-process torch.nn.DataParallel (network/trainer.py:117-123), which
re-broadcasts all 176 MB of parameters and gathers activations on every forward.

What the forward+logdet path actually exchanges:
  * step 0: data-dependent ActNorm init computed on rank 0's shard (= the reference's "first B/G samples on the
    un-replicated model", trainer.py:112-115), then ONE flat broadcast of all parameters;
  * every step: one scalar all-reduce (sum of the per-sample nll) -- images are independent units, so there is
    no data-path collective.
Training adds ONE more collective per step: the gradients (44 M fp32 = 176 MB for the celeba64 model) are averaged
with a single all-reduce over a flat buffer -- on xGMI (point-to-point links, ring collectives are per-link bound)
one large transfer beats ~1000 per-parameter ones; the reference's DataParallel instead reduces to GPU 0 and
re-broadcasts all parameters on the next forward.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous equal shards of a global batch; the reference requires divisibility (model.py:351)."""
    assert world >= 1 and 0 <= rank < world
    assert n % world == 0, f"global batch {n} not divisible by {world} ranks"
    per = n // world
    return rank * per, (rank + 1) * per


def shard_batch(x: torch.Tensor, world: int, rank: int) -> torch.Tensor:
    lo, hi = shard_bounds(x.shape[0], world, rank)
    return x[lo:hi].contiguous()


def broadcast_parameters(module: torch.nn.Module, src: int = 0, world: Optional[int] = None) -> None:
    """Make every rank's parameters equal to rank `src`'s with ONE collective on a flat buffer (one large
    transfer per xGMI link instead of ~1000 small ones)."""
    world = dist.get_world_size() if world is None else world
    if world <= 1:
        return
    params = [p for p in module.parameters()]
    if not params:
        return
    with torch.no_grad():
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        dist.broadcast(flat, src=src)
        off = 0
        for p in params:
            n = p.numel()
            p.copy_(flat[off:off + n].view_as(p))
            off += n


STEP0 = {}      # timing of the last step-0 exchange on this rank: {"init_ms", "wait_ms", "broadcast_ms"} (bench.py reports it)


def data_dependent_init(glow, x_local: torch.Tensor, rank: int, world: int,
                        init_fn: Optional[Callable] = None) -> None:
    """Step-0 exchange: rank 0 initialises every ActNorm from ITS shard, everyone receives the result.
    `init_fn(glow, x)` defaults to one training-mode forward (which performs the init on the HIP path).
    The other ranks have nothing to do while rank 0 runs the init pass (66 ms at config B, 0.3 s at config E, once): they wait in a
    barrier of their own first, so that the wait is visible as such (`STEP0["wait_ms"]`) and the parameter broadcast behind it is
    timed on its own.  What bounds that wait is the timeout of the process group the CALLER created: bench.py creates it with 30
    minutes; a launcher that leaves the backend default (10 minutes for RCCL) is still two orders of magnitude above the init pass.
    On device tensors the barrier names its device (an RCCL barrier on a group that is not bound to one otherwise guesses it)."""
    import time
    sync = (lambda: torch.cuda.synchronize(x_local.device)) if x_local.is_cuda else (lambda: None)
    t0 = time.perf_counter()
    if rank == 0:
        if init_fn is None:
            was_training = glow.training
            glow.train()
            glow.normal_flow(x_local, None)
            glow.train(was_training)
        else:
            init_fn(glow, x_local)
    sync()
    t1 = time.perf_counter()
    if world > 1:
        if x_local.is_cuda and dist.get_backend() == "nccl":
            dist.barrier(device_ids=[x_local.device.index])
        else:
            dist.barrier()
    t2 = time.perf_counter()
    broadcast_parameters(glow, src=0, world=world)
    sync()
    t3 = time.perf_counter()
    STEP0.update(init_ms=round(1e3 * (t1 - t0), 2), wait_ms=round(1e3 * (t2 - t1), 2), broadcast_ms=round(1e3 * (t3 - t2), 2))
    glow.set_actnorm_inited(True)


def reduce_loss(nll: torch.Tensor, world: int) -> torch.Tensor:
    """Sum of the per-sample nll over the GLOBAL batch: local sum, then one scalar all-reduce."""
    s = nll.sum()
    if world > 1:
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return s


def gather_nll(nll: torch.Tensor, world: int) -> torch.Tensor:
    """Per-sample nll of the global batch in rank order (what DataParallel's gather returned)."""
    if world <= 1:
        return nll
    out = [torch.empty_like(nll) for _ in range(world)]
    dist.all_gather(out, nll.contiguous())
    return torch.cat(out)


def allreduce_gradients(module: torch.nn.Module, world: Optional[int] = None, average: bool = True) -> None:
    """Average (or sum) the gradients of all parameters over the ranks with ONE all-reduce on a flat buffer.
    Parameters without a gradient on this rank (e.g. ``h_top``, which the reference detaches, model.py:372) are
    skipped on every rank alike, so the buffers line up."""
    world = dist.get_world_size() if world is None else world
    if world <= 1:
        return
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= world
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


_SIDE_STREAMS = {}
# Diagnostics (bench.py --gpus N): a list here makes `allreduce_buckets` stamp its collectives with timing events -- one record per
# call: {"sweep_done": event on the caller's stream behind the backward sweep, "spans": [(start, end, bytes)] on the side stream}.
BUCKET_TIMING: Optional[list] = None
# Run the exchange also in a ONE-rank process group (a sum over the one rank there is): lets a one-GPU box execute the RCCL path.
FORCE_EXCHANGE = False


def _exchange_off(world: int) -> bool:
    return world <= 1 and not (FORCE_EXCHANGE and dist.is_available() and dist.is_initialized())


def allreduce_buckets(buckets, world: Optional[int] = None, average: bool = True) -> None:
    """All-reduce flat gradient buckets IN PLACE, each as soon as its `ready` event has passed, on a side stream -- so the
    collective of the level the backward sweep has already left runs over xGMI while the sweep is still computing the levels
    below it (`FlowPlan.last_grad_buckets`, csrc glowhip_plan_backward_marks).  buckets: [(flat tensor, event or None)] in the
    order they become ready.  The current stream waits for the last collective before returning to the caller's work."""
    world = dist.get_world_size() if world is None else world
    if _exchange_off(world) or not buckets:
        return
    if not buckets[0][0].is_cuda:                     # CPU / gloo (tests): same collectives, no streams
        for flat, _ in buckets:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            if average:
                flat /= world
        return
    dev = buckets[0][0].device
    side = _SIDE_STREAMS.get(dev)
    if side is None:
        side = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)
    cur = torch.cuda.current_stream(dev)
    rec = None
    if BUCKET_TIMING is not None:
        rec = {"sweep_done": torch.cuda.Event(enable_timing=True), "spans": []}
        rec["sweep_done"].record(cur)                 # (the whole sweep is already enqueued on the caller's stream)
        BUCKET_TIMING.append(rec)
    with torch.cuda.stream(side):
        for flat, ready in buckets:
            if ready is not None:
                side.wait_event(ready)
            flat.record_stream(side)
            if rec is not None:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(side)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)       # RCCL: ordered after the side stream's wait
            if average and world > 1:
                flat.div_(world)
            if rec is not None:
                t1.record(side)
                rec["spans"].append((t0, t1, flat.numel() * flat.element_size()))
    done = torch.cuda.Event()
    done.record(side)
    cur.wait_event(done)


def train_step(glow, optimizer, x_local: torch.Tensor, world: int = 1, max_grad_clip: float = 0.0,
               max_grad_norm: float = 0.0, skip_nonfinite: bool = False,
               before_update: Optional[Callable[[], None]] = None, direct: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """One data-parallel training step of the reference's loop (network/trainer.py:123-150) on this rank's shard:
    forward (HIP, with tape) -> loss = mean(nll) -> backward (HIP reverse sweep) -> gradient all-reduce (RCCL) ->
    clip_grad_value_ / clip_grad_norm_ -> optimizer.step().  Returns (global mean loss, gradient norm).
    ``skip_nonfinite`` (HIP optimisers): a NaN / inf gradient norm skips the update on the device (training.TrainLoop's range check).
    ``direct`` (default): forward and backward through `Glow.loss_and_grads` instead of an autograd graph (False: the reference's
    `loss.backward()` route, kept for comparison and for models the direct call does not take).
    ``before_update``: called after forward, backward and the gradient exchange are enqueued and before the optimiser step is --
    the place where TrainLoop looks at the PREVIOUS step's norm (the device is still a whole forward + backward behind the host).
    The local loss is mean over the LOCAL shard; averaging the gradients over ranks makes it the global mean."""
    direct = direct and hasattr(glow, "loss_and_grads") and x_local.is_cuda and not glow.hps.ablation.learn_top
    if direct:
        # HIP forward + reverse sweep called directly (Glow.loss_and_grads): same kernels and bits as the autograd route below,
        # gradients in the plan's persistent buckets (the parameters' .grad are views into them: nothing to zero, nothing to
        # re-assign) -- the per-step host work of autograd over ~1 060 parameter tensors is what eight ranks on one host cannot afford
        loss = glow.loss_and_grads(x_local)
    else:
        optimizer.zero_grad(set_to_none=True)
        with torch.enable_grad():
            z, nll, _ = glow.normal_flow(x_local, None)
            loss = glow.generative_loss(nll)
            loss.backward()
    buckets = glow.flow.pop_grad_buckets() if hasattr(glow, "flow") and hasattr(glow.flow, "pop_grad_buckets") else None
    if buckets is not None:
        allreduce_buckets(buckets, world)       # per level, overlapped with the rest of the sweep (already enqueued)
    else:
        allreduce_gradients(glow, world)
    if before_update is not None:
        before_update()
    if hasattr(optimizer, "fused_step"):     # training.HipAdam / HipAdamax: both clippings + the update in two HIP launches
        token = getattr(getattr(glow, "_train_plan", None), "_pgrad_bound", None) if direct else None      # persistent gradients: the chunk table stays valid
        grad_norm = optimizer.fused_step(max_grad_clip, max_grad_norm, skip_nonfinite=skip_nonfinite, grads_token=token)
    else:
        params = [p for p in glow.parameters() if p.grad is not None]
        if max_grad_clip and max_grad_clip > 0:
            torch.nn.utils.clip_grad_value_(params, max_grad_clip)
        if max_grad_norm and max_grad_norm > 0:
            grad_norm = torch.nn.utils.clip_grad_norm_(params, max_grad_norm)
        else:
            grad_norm = torch.zeros((), device=x_local.device)
        optimizer.step()
    loss = loss.detach()
    if world > 1:
        dist.all_reduce(loss, op=dist.ReduceOp.SUM)
        loss /= world
    return loss, grad_norm


# ------------------------------------------------------------------------------------------------ diagnostics (bench.py --gpus N)
def bucket_overlap_report(records) -> Optional[dict]:
    """What `BUCKET_TIMING` collected, after a device sync: per step the time the bucket all-reduces took on the side stream, the
    part of it that was still running after the backward sweep had finished on the main stream (`exposed`), and the fraction
    hidden under the sweep = 1 - exposed / total."""
    steps = []
    for rec in records or ():
        if not rec["spans"]:
            continue
        total = sum(a.elapsed_time(b) for a, b, _ in rec["spans"])
        exposed = max(0.0, rec["sweep_done"].elapsed_time(rec["spans"][-1][1]))
        steps.append((total, min(exposed, total), sum(n for _, _, n in rec["spans"])))
    if not steps:
        return None
    total = sum(t for t, _, _ in steps) / len(steps)
    exposed = sum(e for _, e, _ in steps) / len(steps)
    return {"steps": len(steps), "buckets_per_step": len(records[0]["spans"]), "bytes_per_step": steps[0][2],
            "allreduce_ms_per_step": round(total, 3), "exposed_ms_per_step": round(exposed, 3),
            "hidden_fraction": round(1.0 - exposed / total, 4) if total > 0 else None}


def timed_allreduce(numel: int, device, reps: int = 5, warmup: int = 2) -> dict:
    """One flat fp32 all-reduce of `numel` elements (the training step's whole gradient: 44.05 M = 176 MB at config B), timed with
    events on the stream it runs on; algorithmic and bus bandwidth as nccl-tests define them (bus = alg x 2 (n-1) / n: what each
    rank's links carry in a ring -- xGMI is point-to-point, so a ring all-reduce is bound by ONE ~153 GB/s link per direction)."""
    world = dist.get_world_size()
    flat = torch.ones(numel, dtype=torch.float32, device=device)
    cuda = flat.is_cuda
    for _ in range(warmup):
        dist.all_reduce(flat)
    if cuda:
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    import time
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(flat)
    if cuda:
        e1.record()
        torch.cuda.synchronize(device)
        ms = e0.elapsed_time(e1) / reps
    else:
        ms = 1e3 * (time.perf_counter() - t0) / reps
    nbytes = numel * 4
    alg = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {"bytes": nbytes, "world": world, "allreduce_ms": round(ms, 3), "algbw_GBps": round(alg, 1),
            "busbw_GBps": round(alg * 2 * (world - 1) / world, 1) if world > 1 else 0.0,
            "xgmi_link_peak_GBps": 153.0, "reps": reps}


def gather_scalars(value: float, device) -> list:
    """One float from every rank, in rank order (per-rank step times for the bench line)."""
    world = dist.get_world_size()
    mine = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]
