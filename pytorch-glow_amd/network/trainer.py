"""`Trainer` with the reference's constructor and ``.train()`` (network/trainer.py:14-179), so ``train.py:36-49`` runs as written:

    state = Builder(hps).build()
    trainer = Trainer(hps=hps, dataset=dataset, **state)
    trainer.train()

What is kept: the epoch / batch loop over a ``DataLoader`` of ``{'x', 'y_onehot', ...}`` items, learning-rate schedule written into
the optimiser before every step, data-dependent ActNorm init on the first batch, loss = mean(nll) (+ the classification term's
hook), clip by value / by norm, optimiser step, snapshots in the reference's format every ``interval_snapshot`` steps, the
reconstruction / sampling calls every ``interval_valid`` / ``interval_sample`` steps, scalars to TensorBoard.
What changes: the step itself is `training.TrainLoop` -- HIP forward with tape, HIP backward, one flat RCCL all-reduce, fused
clip + Adam -- and ``devices`` are RANKS (one process per GPU, `parallel`), not `DataParallel` replicas; ``tensorboardX`` and
``tqdm`` are optional (absent: scalars are kept in ``self.scalars`` and written to all_scalars.json at the end)."""
import json
import os
import time

import torch
from torch.utils.data import DataLoader

from .. import parallel, training
from ..misc import ops, util
from .model import Glow


class _ScalarLog:
    """Stand-in for tensorboardX.SummaryWriter when that package is not installed."""

    def __init__(self, log_dir):
        self.log_dir, self.scalars, self.images = log_dir, {}, 0

    def add_scalar(self, tag, value, step):
        self.scalars.setdefault(tag, []).append((int(step), float(value)))

    def add_image(self, tag, img, step):
        self.images += 1

    def export_scalars_to_json(self, path):
        with open(path, "w") as f:
            json.dump(self.scalars, f)

    def close(self):
        pass


def _writer(log_dir):
    try:
        from tensorboardX import SummaryWriter
        return SummaryWriter(log_dir=log_dir)
    except ImportError:
        return _ScalarLog(log_dir)


def _shared_seed(seed, world):
    """rank 0's `seed` on every rank of the default process group (the value itself in a single-process run)."""
    import torch.distributed as dist
    if world <= 1 or not (dist.is_available() and dist.is_initialized()):
        assert world <= 1, "a multi-rank Trainer without a profile seed needs the process group to agree on one"
        return int(seed)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([int(seed)], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=0)
    return int(t.item())


class _ShardSampler(torch.utils.data.Sampler):
    """Indices of rank `rank`'s shard of every global batch: one seeded permutation per epoch (the same on every rank), cut
    into global batches of `global_batch`, of which this rank yields positions [rank*per, (rank+1)*per)."""

    def __init__(self, dataset, global_batch, rank, world, seed=None):
        assert global_batch % world == 0, f"global batch {global_batch} not divisible by {world} ranks"
        if seed is None:
            # no profile seed: follow torch's seed as DataLoader(shuffle=True) does (trainer.py:36-41).  The ranks of a
            # data-parallel run do NOT seed alike by themselves (nothing on this path calls util.manual_seed), and ranks that
            # shuffle differently would silently overlap / omit samples: rank 0's draw is THE seed.
            seed = _shared_seed(torch.initial_seed() % (2 ** 31), world)
        self.n, self.gb, self.rank, self.world, self.seed, self.epoch = len(dataset), global_batch, rank, world, seed, 0
        self.per = global_batch // world

    def __len__(self):
        return (self.n // self.gb) * self.per

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.seed * 1000003 + self.epoch)
        self.epoch += 1
        perm = torch.randperm(self.n, generator=g)
        nb = self.n // self.gb
        shard = perm[:nb * self.gb].view(nb, self.gb)[:, self.rank * self.per:(self.rank + 1) * self.per]
        return iter(shard.reshape(-1).tolist())


class Trainer:
    def __init__(self, hps, result_subdir, step, graph, optimizer, scheduler, devices, dataset, data_device,
                 rank=0, world=1):
        self.hps = hps
        self.result_subdir = result_subdir
        self.start_time = time.time()
        self.step = step
        self.graph = graph
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.devices = devices
        self.rank, self.world = rank, world
        self.data_device = data_device
        self.batch_size = self.hps.optim.num_batch_train
        self.num_classes = self.hps.dataset.num_classes
        # One process per GPU: every rank loads ONLY its shard of each global batch.  All ranks shuffle with the same seeded
        # generator (profile seed), so the global batch of a step is the same set of samples on every rank and rank r keeps
        # positions [r*B/G, (r+1)*B/G) of it -- no overlap, no omission, no G-fold loading.
        self.data_loader = DataLoader(dataset, batch_size=self.batch_size // world, num_workers=self.hps.dataset.num_workers,
                                      sampler=_ShardSampler(dataset, self.batch_size, rank, world,
                                                            seed=(int(self.hps.ablation.seed) if getattr(self.hps.ablation, "seed", None)
                                                                  is not None else None)),
                                      drop_last=True)
        self.num_epochs = (self.hps.optim.num_epochs + len(self.data_loader) - 1) // len(self.data_loader)
        self.y_condition = self.hps.ablation.y_condition
        if self.y_condition:
            raise NotImplementedError("class-conditional training (y_condition) is outside the flow hot path")
        self.max_grad_clip = self.hps.ablation.max_grad_clip
        self.max_grad_norm = self.hps.ablation.max_grad_norm
        self.writer = _writer(self.result_subdir) if rank == 0 else _ScalarLog(self.result_subdir)   # rank 0 alone writes logs
        self.interval_scalar = self.hps.optim.interval_scalar
        self.interval_snapshot = self.hps.optim.interval_snapshot
        self.interval_valid = self.hps.optim.interval_valid
        self.interval_sample = self.hps.optim.interval_sample
        self.num_sample = self.hps.optim.num_sample
        # the per-step arithmetic (trainer.py:88-150)
        # (a single-rank run replays its step as one hipGraph launch after the first eager steps -- training.GraphedTrainStep; the
        # environment variable GLOWHIP_TRAIN_GRAPH=0 keeps every step eager; a failed capture falls back to the eager step by itself)
        self.loop = training.TrainLoop(graph, hps, rank=rank, world=world, optimizer=optimizer,
                                       graph=world == 1 and os.environ.get("GLOWHIP_TRAIN_GRAPH", "1") != "0")
        self.loop.scheduler = scheduler or self.loop.scheduler
        self.loop.global_step = step
        self.last_loss = None

    def _device(self):
        return next(self.graph.parameters()).device

    def train(self, max_steps=None):
        """The reference's loop.  ``max_steps`` (beyond its signature) stops early -- for tests and smoke runs."""
        self.graph.train()
        done = 0
        try:
            from tqdm import tqdm
        except ImportError:
            tqdm = lambda it: it
        for epoch in range(self.num_epochs):
            print('[Trainer] Epoch ({}/{})'.format(epoch, self.num_epochs))
            for idx, batch in enumerate(tqdm(self.data_loader)):
                for i in batch:
                    batch[i] = batch[i].to(self._device())       # the flow runs on this rank's GPU only
                x = batch['x']                                   # already this rank's shard (_ShardSampler)
                loss, grad_norm = self.loop.step(x.float().contiguous())
                lr = self.loop.lr
                self.last_loss = loss
                log = self.step % self.interval_scalar == 0 and self.step > 0
                if log:
                    self.writer.add_scalar('lr/lr', lr, self.step)
                    self.writer.add_scalar('loss/generative_loss', loss, self.step)
                    if self.max_grad_norm is not None and self.max_grad_norm > 0:
                        self.writer.add_scalar("grad_norm/grad_norm", grad_norm, self.step)
                if self.step % self.interval_snapshot == 0 and self.step > 0:
                    self.loop.flush()                            # (collective when it re-runs a step: every rank, not just 0)
                if self.step % self.interval_snapshot == 0 and self.step > 0 and self.rank == 0:
                    util.save_model(result_subdir=self.result_subdir, step=self.step, graph=self.graph, optimizer=self.optimizer,
                                    seconds=time.time() - self.start_time, is_best=True)
                if self.step % self.interval_valid == 0 and self.step > 0 and self.rank == 0:
                    with torch.no_grad():
                        self.graph.eval()
                        z, _, _ = self.graph(x=x, y_onehot=None)
                        img = self.graph(z=z, y_onehot=None, reverse=True)
                        self.graph.train()
                    for i in range(min(self.num_sample, img.shape[0])):
                        self.writer.add_image("reconstructed/{}".format(i), ops.cat_channel(img[i], x[i]), self.step)
                if self.step % self.interval_sample == 0 and self.step > 0 and self.rank == 0:
                    with torch.no_grad():
                        self.graph.eval()
                        img = self.graph(z=None, y_onehot=None, eps_std=0.5, reverse=True)
                        self.graph.train()
                    for i in range(min(self.num_sample, img.shape[0])):
                        self.writer.add_image("sample/{}".format(i), img[i], self.step)
                self.step += 1
                done += 1
                if max_steps is not None and done >= max_steps:
                    break
            if max_steps is not None and done >= max_steps:
                break
        self.loop.flush()
        if self.rank == 0:
            self.writer.export_scalars_to_json(os.path.join(self.result_subdir, "all_scalars.json"))
        self.writer.close()
