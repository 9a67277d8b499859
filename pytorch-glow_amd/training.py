"""The reference's training step around the HIP flow (network/builder.py:9-20,108-126 and network/trainer.py:85-150),
one process per GPU.

Only the per-step arithmetic is here -- schedule -> forward (with tape) -> HIP backward -> RCCL gradient average ->
clip by value / by norm -> optimiser -- because that is what bounds training throughput.  The reference's logging,
snapshotting, sampling and dataset code around the loop is out of scope.  The optimisers themselves
are ``torch.optim.Adam`` / ``Adamax`` exactly as in the reference (builder.py:10-13).
"""
from functools import partial
from typing import Iterable, Optional

import torch

from . import parallel
from .misc import lr_scheduler

OPTIMIZERS = {"adam": torch.optim.Adam, "adamax": torch.optim.Adamax}


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


class TrainLoop:
    """The state the reference's ``Trainer`` carries from step to step, minus its I/O: model, optimiser, schedule, step
    counter, clipping thresholds (trainer.py:44-60).  ``step(x_local)`` runs one iteration on this rank's shard."""

    def __init__(self, glow, hps, rank: int = 0, world: int = 1, optimizer: Optional[torch.optim.Optimizer] = None):
        self.glow, self.hps, self.rank, self.world = glow, hps, rank, world
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
        self.glow.train()
        self.lr = self.scheduler(global_step=self.global_step)
        for group in self.optimizer.param_groups:           # trainer.py:89-91
            group["lr"] = self.lr
        loss, grad_norm = parallel.train_step(self.glow, self.optimizer, x_local, world=self.world,
                                              max_grad_clip=self.max_grad_clip, max_grad_norm=self.max_grad_norm)
        self.global_step += 1
        return loss, grad_norm
