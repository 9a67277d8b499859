"""Data parallelism for the flow path, MI355X-style: ONE PROCESS PER GPU, parameters resident on every rank,
the minibatch sharded on dim 0, collectives over RCCL (torch.distributed backend "nccl"; "gloo" in the CPU
tests).  Replaces the reference's single-process torch.nn.DataParallel (network/trainer.py:117-123), which
re-broadcasts all 176 MB of parameters and gathers activations on every forward.

What the forward+logdet path actually exchanges:
  * step 0: data-dependent ActNorm init computed on rank 0's shard (= the reference's "first B/G samples on the
    un-replicated model", trainer.py:112-115), then ONE flat broadcast of all parameters;
  * every step: one scalar all-reduce (sum of the per-sample nll) -- images are independent units, so there is
    no data-path collective.
Gradient all-reduce for training belongs to the backward path (next scope row) and is not here yet.
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


def data_dependent_init(glow, x_local: torch.Tensor, rank: int, world: int,
                        init_fn: Optional[Callable] = None) -> None:
    """Step-0 exchange: rank 0 initialises every ActNorm from ITS shard, everyone receives the result.
    `init_fn(glow, x)` defaults to one training-mode forward (which performs the init on the HIP path)."""
    if rank == 0:
        if init_fn is None:
            was_training = glow.training
            glow.train()
            glow.normal_flow(x_local, None)
            glow.train(was_training)
        else:
            init_fn(glow, x_local)
    broadcast_parameters(glow, src=0, world=world)
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
