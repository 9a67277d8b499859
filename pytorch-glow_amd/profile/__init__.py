"""Built-in experiment profiles in the reference's JSON schema (sections general / dataset / optim / model /
ablation / device, consumed as ``hps.<section>.<key>``).

The two configurations the reference ships are expressed here as data: ``celeba`` (64x64x3, L=3, K=32, hidden 512,
affine coupling, invertible 1x1 conv -- the benchmark model) and ``test`` (the same network with additive coupling,
batch 16, adamax).  ``load_profile`` in ``misc/util.py`` accepts either a path to any JSON file of this schema
(e.g. the reference's own ``profile/celeba.json``) or one of these names; ``write_json`` materialises a built-in
profile as a file for tools that want one.
"""
import copy
import json

_FLOW = dict(image_shape=[64, 64, 3], hidden_channels=512, K=32, L=3, actnorm_scale=1.0, n_bits_x=8,
             weight_y=0.0, anchor_size=32)

_CELEBA = {
    "profile": "celeba_64x64_8bit",
    "model": _FLOW,
    "ablation": dict(flow_permutation="invconv", flow_coupling="affine", lu_decomposition=False, learn_top=False,
                     y_condition=False, y_criterion="multi_classes", seed=2384, max_grad_clip=5, max_grad_norm=100),
    "optim": dict(optimizer="adam", optimizer_args=dict(lr=1e-3, betas=[0.9, 0.9999], eps=1e-8, weight_decay=0),
                  lr_scheduler="noam", lr_scheduler_args=dict(warmup_steps=4000, min_lr=1e-4),
                  num_batch_train=50, num_batch_test=50, num_batch_init=256, num_epochs=1000000, num_train=50000,
                  num_test=-1, num_sample=4, interval_scalar=10, interval_snapshot=5000, interval_valid=10,
                  interval_sample=10, gradient_checkpointing=True),
    "dataset": dict(problem="celeba", root="/Data/CelebA", num_classes=40, num_workers=8, argument="standard"),
    "general": dict(verbose=False, result_dir="/Data/glow", warm_start=False, pre_trained="", resume_run_id=1,
                    resume_step="latest"),
    "device": dict(graph=["cuda:0", "cuda:1"], data=["cuda:0"]),
}

# the reference's second profile, as differences from the first
_TEST_OVERRIDES = {
    "profile": "celebahq_256x256_5bit",
    "ablation": dict(flow_coupling="additive", y_criterion="", seed=0),
    "optim": dict(optimizer="adamax", num_batch_train=16, num_sample=1, interval_scalar=50, interval_snapshot=50,
                  interval_valid=50, interval_sample=50),
    "dataset": dict(num_classes=1),
    "general": dict(result_dir=".", warm_start=True, resume_run_id=0),
    "device": dict(data=["cpu"]),
}


def _merged(base, over):
    out = copy.deepcopy(base)
    for k, v in over.items():
        if isinstance(v, dict):
            out[k].update(copy.deepcopy(v))
        else:
            out[k] = v
    return out


def builtin(name):
    """A fresh plain-dict copy of a built-in profile ('celeba' or 'test')."""
    if name == "celeba":
        return copy.deepcopy(_CELEBA)
    if name == "test":
        d = _merged(_CELEBA, _TEST_OVERRIDES)
        d["optim"]["optimizer_args"]["betas"] = [0.9, 0.99]
        d["optim"]["lr_scheduler_args"] = dict(warmup_steps=4000)
        return d
    raise KeyError(f"unknown built-in profile {name!r} (have: celeba, test)")


def write_json(name, path):
    with open(path, "w") as f:
        json.dump(builtin(name), f, indent=2)
