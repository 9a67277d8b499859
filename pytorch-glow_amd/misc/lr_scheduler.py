"""Learning-rate schedules of the reference (misc/lr_scheduler.py:4-96), as a table of pure functions of
``(base_lr, global_step, **args)``; `network/builder.py:14-20` binds them by the profile's ``optim.lr_scheduler`` name.

Why this is on the hot path's side of the fence: the celeba profile trains with ``noam`` warm-up; without it the first
Adam step (each weight moves by ~lr in the gradient's sign direction) sends the 96-layer flow to nll ~ 1e5 bits/dim --
measured here on both the oracle and the HIP path -- so the training bench has to use the schedule to be a valid step.
"""
import math


def constant(base_lr, global_step):
    return base_lr


def noam_decay(base_lr, global_step, warmup_steps=4000, min_lr=1e-4):
    # "Attention is all you need" 5.3: linear warm-up to base_lr over warmup_steps, then ~ 1/sqrt(step), floored at min_lr
    n = global_step + 1.0
    lr = base_lr * math.sqrt(warmup_steps) * min(n ** -0.5, n * float(warmup_steps) ** -1.5)
    return max(min_lr, lr) if global_step >= warmup_steps else lr


def linear_anneal(base_lr, global_step, num_train, warmup_steps=10):
    return base_lr * min(1.0, global_step / (num_train * warmup_steps))


def step_anneal(base_lr, global_step, anneal_rate=0.98, anneal_interval=30000):
    return base_lr * anneal_rate ** (global_step // anneal_interval)


def cyclic_cosine_anneal(base_lr, global_step, t, m):
    # snapshot ensembles, section 3: m cosine cycles over t steps
    period = t // m
    return 0.5 * base_lr * (math.cos(math.pi * ((global_step - 1) % period) / period) + 1.0)


SCHEDULES = {"constant": constant, "noam": noam_decay, "linear": linear_anneal, "step": step_anneal,
             "cyclic_cosine": cyclic_cosine_anneal}
