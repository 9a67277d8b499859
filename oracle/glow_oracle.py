"""CPU oracle for the Glow flow hot path -- TEST INFRASTRUCTURE ONLY.

This file is a functional, out-of-place, pure-PyTorch (CPU, fp32) restatement of the
algorithm implemented by corenel/pytorch-glow's `network/module.py`, `network/model.py`
and `misc/ops.py`.  It exists to CHECK the HIP path, never to be it:

  * only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
    may import it;
  * the product package (`pytorch-glow_amd/`) never imports anything from `oracle/`.

Parity status: PINNED.  `tests/golden/make_golden.py` imports the real reference in the
build container (it is Python) and records input/output vectors for every function
below; `tests/test_oracle_golden.py` replays them against this file (max-abs <= 2e-6 on
activations, rtol 1e-6 on log-determinants).

The oracle works on a flat ``state_dict`` (same keys/shapes as the reference's
``Glow.state_dict()``) plus a small ``cfg`` dict, instead of on
``nn.Module`` objects, so the same weights can be fed to the HIP path and to the oracle.

All citations are ``file:line`` relative to the reference checkout.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

LOG_2PI = float(np.log(2 * np.pi))  # network/module.py:405
LOGSCALE_FACTOR = 3.0               # network/module.py:10, :266


# ----------------------------------------------------------------------------- misc/ops.py
def reduce_sum_dims(t: torch.Tensor, dims: Sequence[int], keepdim: bool = False) -> torch.Tensor:
    """misc/ops.py:40-73 -- sum one dimension at a time in ascending order, then squeeze."""
    dims = sorted(dims)
    for d in dims:
        t = t.sum(dim=d, keepdim=True)
    if not keepdim:
        for cnt, d in enumerate(dims):
            t = t.squeeze(d - cnt)
    return t


def reduce_mean_dims(t: torch.Tensor, dims: Sequence[int], keepdim: bool = False) -> torch.Tensor:
    """misc/ops.py:4-37 -- mean one dimension at a time in ascending order."""
    dims = sorted(dims)
    for d in dims:
        t = t.mean(dim=d, keepdim=True)
    if not keepdim:
        for cnt, d in enumerate(dims):
            t = t.squeeze(d - cnt)
    return t


def split_channel(t: torch.Tensor, kind: str = "simple") -> Tuple[torch.Tensor, torch.Tensor]:
    """misc/ops.py:95-113 -- 'simple' halves, 'cross' even/odd channels (copies, not views)."""
    assert t.dim() == 4 and kind in ("simple", "cross")
    c = t.shape[1]
    if kind == "simple":
        return t[:, : c // 2].clone(), t[:, c // 2:].clone()
    return t[:, 0::2].clone(), t[:, 1::2].clone()


def cat_channel(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """misc/ops.py:116-127"""
    return torch.cat((a, b), dim=1)


def count_pixels(t: torch.Tensor) -> int:
    """misc/ops.py:130-140"""
    assert t.dim() == 4
    return int(t.shape[2] * t.shape[3])


# ----------------------------------------------------------------------------- squeeze (R11)
def squeeze2d(x: torch.Tensor, factor: int = 2) -> torch.Tensor:
    """network/module.py:573-592 -- out[n, c*f*f + i*f + j, h, w] = x[n, c, h*f+i, w*f+j]."""
    if factor == 1:
        return x
    n, c, h, w = x.shape
    assert h % factor == 0 and w % factor == 0
    x = x.reshape(n, c, h // factor, factor, w // factor, factor)
    x = x.permute(0, 1, 3, 5, 2, 4).contiguous()
    return x.reshape(n, c * factor * factor, h // factor, w // factor)


def unsqueeze2d(x: torch.Tensor, factor: int = 2) -> torch.Tensor:
    """network/module.py:551-570 -- inverse of :func:`squeeze2d`."""
    if factor == 1:
        return x
    n, c, h, w = x.shape
    f2 = factor * factor
    assert c >= f2 and c % f2 == 0
    x = x.reshape(n, c // f2, factor, factor, h, w)
    x = x.permute(0, 1, 4, 2, 5, 3).contiguous()
    return x.reshape(n, c // f2, h * factor, w * factor)


# ----------------------------------------------------------------------------- ActNorm (R3, R4)
def actnorm_init(x: torch.Tensor, scale: float = 1.0, batch_variance: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Data-dependent init, network/module.py:86-120.

    bias = -mean_{N,H,W}(x);  logs = log(scale / (sqrt(mean((x+bias)^2)) + 1e-6)) / 3, the mean over (N,H,W) per channel --
    or, with batch_variance=True (:109-110), over every dimension: one value, copied into all channels (`logs.data.copy_`).
    Returns (bias, logs), each (1, C, 1, 1).
    """
    bias = -1.0 * reduce_mean_dims(x, [0, 2, 3], keepdim=True)
    xc = x + bias
    var = torch.mean(xc ** 2) if batch_variance else reduce_mean_dims(xc ** 2, [0, 2, 3], keepdim=True)
    logs = torch.log(scale / (torch.sqrt(var) + 1e-6)) / LOGSCALE_FACTOR
    return bias, logs.expand_as(bias).clone()


def actnorm(x, bias, logs, logdet=None, reverse: bool = False):
    """network/module.py:122-149 (center :34-50, scale :52-84).

    fwd: y = (x + bias) * exp(3 logs), logdet += 3*sum(logs)*H*W
    rev: y = x * exp(-3 logs) - bias,  logdet -= 3*sum(logs)*H*W
    """
    assert x.dim() == 4 and x.shape[1] == bias.shape[1]
    l3 = logs * LOGSCALE_FACTOR
    if not reverse:
        y = (x + bias) * torch.exp(l3)
    else:
        y = x * torch.exp(-l3) - bias
    if logdet is not None:
        dlogdet = torch.sum(l3) * count_pixels(x)
        if reverse:
            dlogdet = dlogdet * -1
        logdet = logdet + dlogdet
    return y, logdet


# ----------------------------------------------------------------------------- invconv (R5) / permutation (R14)
STABLE_LOGDET = False  # tests only: see invconv_dlogdet


def invconv_dlogdet(weight: torch.Tensor, pixels: int) -> torch.Tensor:
    """network/module.py:356-357 -- log|det W| * H*W (torch.det = LU with partial pivoting).

    Reference quirk: `torch.det` forms the PRODUCT of the LU pivots in fp32; at C=384 (config E, deepest level)
    that product underflows to 0 for a perfectly conditioned orthogonal W and the reference's log-det is -inf.
    With ``STABLE_LOGDET`` the oracle evaluates sum(log|pivot|) in fp64 instead (what the HIP LU kernel does), so
    tests can still check the large-C levels; the default restates the reference exactly."""
    if STABLE_LOGDET:
        return torch.linalg.slogdet(weight.double())[1].to(weight.dtype) * pixels
    return torch.log(torch.abs(torch.det(weight))) * pixels


def invconv(x, weight, logdet=None, reverse: bool = False):
    """network/module.py:344-369 -- z[n,o,h,w] = sum_i W[o,i] x[n,i,h,w]; reverse uses W^-1."""
    c = weight.shape[0]
    dlogdet = invconv_dlogdet(weight, count_pixels(x))
    if not reverse:
        z = F.conv2d(x, weight.view(c, c, 1, 1))
        if logdet is not None:
            logdet = logdet + dlogdet
    else:
        z = F.conv2d(x, weight.inverse().view(c, c, 1, 1))
        if logdet is not None:
            logdet = logdet - dlogdet
    return z, logdet


def permutation_indices(num_channels: int, shuffle: bool, rng: Optional[np.random.RandomState] = None):
    """network/module.py:383-390 -- reversed order, optionally shuffled; plus the inverse table."""
    idx = np.arange(num_channels - 1, -1, -1, dtype=np.int64)
    if shuffle:
        (rng or np.random).shuffle(idx)
    inv = np.zeros(num_channels, dtype=np.int64)
    for i in range(num_channels):
        inv[idx[i]] = i
    return idx, inv


def permute2d(x, indices, indices_inverse, reverse: bool = False):
    """network/module.py:392-397 -- fixed channel gather."""
    assert x.dim() == 4
    sel = indices_inverse if reverse else indices
    return x[:, torch.as_tensor(np.asarray(sel), dtype=torch.long)]


# ----------------------------------------------------------------------------- convolutions (R6, R7, R8)
def conv2d_actnorm(x, weight, an_bias, an_logs):
    """`Conv2d` wrapper, network/module.py:188-260: bias-free 'SAME' conv then its own ActNorm
    (logdet=None)."""
    k = weight.shape[2]
    y = F.conv2d(x, weight, None, stride=1, padding=(k - 1) // 2)
    y, _ = actnorm(y, an_bias, an_logs, None, reverse=False)
    return y


def conv2d_zeros(x, weight, bias, logs):
    """`Conv2dZeros`, network/module.py:263-297: y = (conv(x) + b) * exp(3*logs), logs (Cout,1,1)."""
    k = weight.shape[2]
    y = F.conv2d(x, weight, bias, stride=1, padding=(k - 1) // 2)
    return y * torch.exp(logs * LOGSCALE_FACTOR)


def coupling_net(z1, sd: Dict[str, torch.Tensor], prefix: str):
    """`f()`, network/module.py:300-319: Conv2d 3x3 -> ReLU -> Conv2d 1x1 -> ReLU -> Conv2dZeros 3x3.
    ``prefix`` addresses ``{prefix}0.weight`` ... ``{prefix}4.logs``."""
    h = conv2d_actnorm(z1, sd[prefix + "0.weight"], sd[prefix + "0.actnorm.bias"], sd[prefix + "0.actnorm.logs"])
    h = torch.relu(h)
    h = conv2d_actnorm(h, sd[prefix + "2.weight"], sd[prefix + "2.actnorm.bias"], sd[prefix + "2.actnorm.logs"])
    h = torch.relu(h)
    return conv2d_zeros(h, sd[prefix + "4.weight"], sd[prefix + "4.bias"], sd[prefix + "4.logs"])


# ----------------------------------------------------------------------------- FlowStep (R9)
def flowstep(x, logdet, sd, prefix: str, permutation: str = "invconv", coupling: str = "additive",
             reverse: bool = False, perm_tables=None):
    """network/model.py:82-154.  ``prefix`` e.g. ``'flow.layers.1.'``.
    ``perm_tables`` = (indices, indices_inverse) for 'reverse'/'shuffle'."""
    assert x.shape[1] % 2 == 0  # model.py:169
    an_b, an_l = sd[prefix + "actnorm.bias"], sd[prefix + "actnorm.logs"]
    if not reverse:
        z, logdet = actnorm(x, an_b, an_l, logdet, reverse=False)
        if permutation == "invconv":
            z, logdet = invconv(z, sd[prefix + "invconv.weight"], logdet, reverse=False)
        else:
            z = permute2d(z, perm_tables[0], perm_tables[1], reverse=False)
        z1, z2 = split_channel(z, "simple")
        if coupling == "additive":
            z2 = z2 + coupling_net(z1, sd, prefix + "f.")
        else:
            h = coupling_net(z1, sd, prefix + "f.")
            shift, scale = split_channel(h, "cross")
            scale = torch.sigmoid(scale + 2.0)
            z2 = (z2 + shift) * scale
            logdet = reduce_sum_dims(torch.log(scale), [1, 2, 3]) + logdet
        return cat_channel(z1, z2), logdet
    # reverse, model.py:119-154
    z1, z2 = split_channel(x, "simple")
    if coupling == "additive":
        z2 = z2 - coupling_net(z1, sd, prefix + "f.")
    else:
        h = coupling_net(z1, sd, prefix + "f.")
        shift, scale = split_channel(h, "cross")
        scale = torch.sigmoid(scale + 2.0)
        z2 = z2 / scale - shift
        logdet = -reduce_sum_dims(torch.log(scale), [1, 2, 3]) + logdet
    z = cat_channel(z1, z2)
    if permutation == "invconv":
        z, logdet = invconv(z, sd[prefix + "invconv.weight"], logdet, reverse=True)
    else:
        z = permute2d(z, perm_tables[0], perm_tables[1], reverse=True)
    z, logdet = actnorm(z, an_b, an_l, logdet, reverse=True)
    return z, logdet


# ----------------------------------------------------------------------------- Gaussian prior / Split2d (R10)
def gaussian_logps(mean, logs, x):
    """network/module.py:437-451"""
    return -0.5 * (LOG_2PI + 2.0 * logs + ((x - mean) ** 2) / torch.exp(2.0 * logs))


def gaussian_logp(mean, logs, x):
    """network/module.py:453-467 -- summed over C,H,W one dim at a time."""
    return reduce_sum_dims(gaussian_logps(mean, logs, x), [1, 2, 3])


def effective_eps_std(eps_std):
    """network/module.py:419 -- ``eps_std or 1.``: None AND 0 both mean 1.0."""
    return eps_std or 1.0


def gaussian_sample(mean, logs, eps):
    """network/module.py:470-483 with the N(0, eps_std) draw ``eps`` injected (already scaled)."""
    return mean + torch.exp(logs) * eps


def split2d(x, logdet, sd, prefix: str, reverse: bool = False, eps=None):
    """network/module.py:511-536.  fwd returns (z1, logdet + logp(z2 | prior(z1))) -- z2 is dropped.
    rev: z2 = mean + exp(logs)*eps (``eps`` = the injected draw, std already applied), cat."""
    w, b, l = sd[prefix + "conv2d_zeros.weight"], sd[prefix + "conv2d_zeros.bias"], sd[prefix + "conv2d_zeros.logs"]
    if not reverse:
        z1, z2 = split_channel(x, "simple")
        mean, logs = split_channel(conv2d_zeros(z1, w, b, l), "cross")
        logdet = gaussian_logp(mean, logs, z2) + logdet
        return z1, logdet
    z1 = x
    mean, logs = split_channel(conv2d_zeros(z1, w, b, l), "cross")
    z2 = gaussian_sample(mean, logs, eps)
    return cat_channel(z1, z2), logdet


# ----------------------------------------------------------------------------- FlowModel (R12)
def flow_layout(cfg) -> List[Tuple[str, int, Tuple[int, int, int]]]:
    """network/model.py:230-261 -- ordered list of (kind, layer_index, (C,H,W) of the OUTPUT)."""
    nh, nw, nc = cfg["image_shape"]
    assert nc in (1, 3)
    out, idx = [], 0
    for i in range(cfg["L"]):
        nc, nh, nw = nc * 4, nh // 2, nw // 2
        out.append(("squeeze", idx, (nc, nh, nw))); idx += 1
        for _ in range(cfg["K"]):
            out.append(("step", idx, (nc, nh, nw))); idx += 1
        if i < cfg["L"] - 1:
            nc = nc // 2
            out.append(("split", idx, (nc, nh, nw))); idx += 1
    return out


def flow_encode(x, logdet, sd, cfg, prefix: str = "flow.layers.", perm_tables=None):
    """network/model.py:263-276"""
    z = x
    for kind, i, _ in flow_layout(cfg):
        if kind == "squeeze":
            z = squeeze2d(z, 2)
        elif kind == "step":
            z, logdet = flowstep(z, logdet, sd, f"{prefix}{i}.", cfg["flow_permutation"], cfg["flow_coupling"],
                                 reverse=False, perm_tables=None if perm_tables is None else perm_tables[i])
        else:
            z, logdet = split2d(z, logdet, sd, f"{prefix}{i}.", reverse=False)
    return z, logdet


def flow_decode(z, sd, cfg, eps_list: Sequence[torch.Tensor], prefix: str = "flow.layers.", perm_tables=None):
    """network/model.py:278-294.  ``eps_list`` holds one injected draw per Split2d in DECODE order
    (deepest split first); each already carries its std."""
    eps_iter = iter(eps_list)
    for kind, i, _ in reversed(flow_layout(cfg)):
        if kind == "squeeze":
            z = unsqueeze2d(z, 2)
        elif kind == "step":
            z, _ = flowstep(z, 0.0, sd, f"{prefix}{i}.", cfg["flow_permutation"], cfg["flow_coupling"],
                            reverse=True, perm_tables=None if perm_tables is None else perm_tables[i])
        else:
            z, _ = split2d(z, 0.0, sd, f"{prefix}{i}.", reverse=True, eps=next(eps_iter))
    return z


# ----------------------------------------------------------------------------- Glow (R13)
def glow_forward(x, noise, sd, cfg, perm_tables=None):
    """`Glow.normal_flow`, network/model.py:409-452, with the dequantisation noise injected.

    z = x + noise; objective = -ln(n_bins)*CHW; encode; objective += logp(z | h_top prior);
    nll = -objective / (ln2 * CHW).  Returns (z, nll, objective)."""
    assert not cfg.get("y_condition", False)
    n_bins = 2 ** cfg["n_bits_x"]
    z = x + noise
    factor = x.shape[1] * count_pixels(x)
    objective = torch.zeros_like(x[:, 0, 0, 0])
    objective = objective + float(-np.log(n_bins)) * factor
    z, objective = flow_encode(z, objective, sd, cfg, perm_tables=perm_tables)
    h = sd["h_top"][: z.shape[0]].detach()                              # model.py:362-379 (h_top == 0, detached)
    if cfg.get("learn_top", False):                                    # :375-376: h = learn_top(h), a Conv2dZeros (module.py:263-297)
        h = conv2d_zeros(h, sd["learn_top.weight"], sd["learn_top.bias"], sd["learn_top.logs"])
    mean, logs = split_channel(h, "simple")
    objective = objective + gaussian_logp(mean, logs, z)
    nll = (-objective) / float(np.log(2.0) * factor)
    return z, nll, objective


def glow_reverse(z, sd, cfg, eps_list, perm_tables=None):
    """`Glow.reverse_flow`, network/model.py:454-471 with `z` given (top sample injected by caller)."""
    with torch.no_grad():
        return flow_decode(z, sd, cfg, eps_list, perm_tables=perm_tables)


# ----------------------------------------------------------------------------- seeded weights + data-dependent init
def default_cfg(image_shape=(64, 64, 3), hidden_channels=512, K=32, L=3, flow_permutation="invconv",
                flow_coupling="affine", actnorm_scale=1.0, n_bits_x=8, batch=64):
    return dict(image_shape=list(image_shape), hidden_channels=hidden_channels, K=K, L=L,
                flow_permutation=flow_permutation, flow_coupling=flow_coupling,
                actnorm_scale=actnorm_scale, n_bits_x=n_bits_x, batch=batch,
                learn_top=False, y_condition=False)


def seeded_state_dict(cfg, seed: int = 2384, zeros_std: float = 0.002, invconv_perturb: float = 0.0):
    """Build-side seeded weights: N(0,0.05) convs (module.py:237), QR-orthogonal
    invconv (module.py:341) [+ optional perturbation], Conv2dZeros ~ N(0, zeros_std) so the tail
    is exercised (the reference zero-inits them, module.py:282-283), ActNorm zero (filled by
    :func:`glow_init_actnorm`).  Keys/shapes follow the reference state_dict."""
    g = torch.Generator().manual_seed(seed)
    rs = np.random.RandomState(seed)
    hid = cfg["hidden_channels"]
    sd: Dict[str, torch.Tensor] = {}
    last = None
    for kind, i, (c, h, w) in flow_layout(cfg):
        p = f"flow.layers.{i}."
        if kind == "step":
            sd[p + "actnorm.bias"] = torch.zeros(1, c, 1, 1)
            sd[p + "actnorm.logs"] = torch.zeros(1, c, 1, 1)
            if cfg["flow_permutation"] == "invconv":
                q = np.linalg.qr(rs.randn(c, c))[0].astype("float32")
                if invconv_perturb:
                    q = q + invconv_perturb * rs.randn(c, c).astype("float32")
                sd[p + "invconv.weight"] = torch.from_numpy(q)
            cout = c if cfg["flow_coupling"] == "affine" else c // 2
            sd[p + "f.0.weight"] = torch.randn(hid, c // 2, 3, 3, generator=g) * 0.05
            sd[p + "f.0.actnorm.bias"] = torch.zeros(1, hid, 1, 1)
            sd[p + "f.0.actnorm.logs"] = torch.zeros(1, hid, 1, 1)
            sd[p + "f.2.weight"] = torch.randn(hid, hid, 1, 1, generator=g) * 0.05
            sd[p + "f.2.actnorm.bias"] = torch.zeros(1, hid, 1, 1)
            sd[p + "f.2.actnorm.logs"] = torch.zeros(1, hid, 1, 1)
            sd[p + "f.4.weight"] = torch.randn(cout, hid, 3, 3, generator=g) * zeros_std
            sd[p + "f.4.bias"] = torch.randn(cout, generator=g) * zeros_std
            sd[p + "f.4.logs"] = torch.randn(cout, 1, 1, generator=g) * zeros_std
        elif kind == "split":
            cin = c  # output channels of split == C/2 of its input
            sd[p + "conv2d_zeros.weight"] = torch.randn(2 * cin, cin, 3, 3, generator=g) * zeros_std
            sd[p + "conv2d_zeros.bias"] = torch.randn(2 * cin, generator=g) * zeros_std
            sd[p + "conv2d_zeros.logs"] = torch.randn(2 * cin, 1, 1, generator=g) * zeros_std
        last = (c, h, w)
    c, h, w = last
    sd["h_top"] = torch.zeros(cfg["batch"], 2 * c, h, w)
    return sd


def glow_init_actnorm(x, noise, sd, cfg, perm_tables=None) -> Dict[str, torch.Tensor]:
    """The first training-mode forward (network/trainer.py:112-115, module.py:45-46,66-67):
    every ActNorm's bias/logs are set from the activations that reach it.  Returns a NEW dict.
    ``perm_tables`` = {layer index: (indices, indices_inverse)} for the 'reverse' / 'shuffle' permutations
    (network/module.py:372-397), as glow_forward takes them."""
    sd = dict(sd)
    z = x + noise
    for kind, i, _ in flow_layout(cfg):
        p = f"flow.layers.{i}."
        if kind == "squeeze":
            z = squeeze2d(z, 2)
        elif kind == "split":
            z, _ = split2d(z, 0.0, sd, p, reverse=False)
        else:
            b, l = actnorm_init(z, cfg["actnorm_scale"])
            sd[p + "actnorm.bias"], sd[p + "actnorm.logs"] = b, l
            y, _ = actnorm(z, b, l, None)
            if cfg["flow_permutation"] == "invconv":
                y, _ = invconv(y, sd[p + "invconv.weight"], None)
            else:
                y = permute2d(y, perm_tables[i][0], perm_tables[i][1], reverse=False)
            z1, z2 = split_channel(y, "simple")
            h = F.conv2d(z1, sd[p + "f.0.weight"], None, padding=1)
            b0, l0 = actnorm_init(h, 1.0)
            sd[p + "f.0.actnorm.bias"], sd[p + "f.0.actnorm.logs"] = b0, l0
            h = torch.relu(actnorm(h, b0, l0)[0])
            h = F.conv2d(h, sd[p + "f.2.weight"], None)
            b2, l2 = actnorm_init(h, 1.0)
            sd[p + "f.2.actnorm.bias"], sd[p + "f.2.actnorm.logs"] = b2, l2
            h = torch.relu(actnorm(h, b2, l2)[0])
            h = conv2d_zeros(h, sd[p + "f.4.weight"], sd[p + "f.4.bias"], sd[p + "f.4.logs"])
            if cfg["flow_coupling"] == "additive":
                z2 = z2 + h
            else:
                shift, scale = split_channel(h, "cross")
                z2 = (z2 + shift) * torch.sigmoid(scale + 2.0)
            z = cat_channel(z1, z2)
    return sd


def flop_per_image(cfg) -> float:
    """Contraction FLOPs (2/MAC) of one forward, SURVEY.md 8d formula."""
    h = cfg["hidden_channels"]
    total = 0.0
    for kind, i, (c, hh, ww) in flow_layout(cfg):
        p = hh * ww
        if kind == "step":
            cout = c if cfg["flow_coupling"] == "affine" else c // 2
            total += 2.0 * p * (9 * (c // 2) * h + h * h + 9 * h * cout + c * c)
        elif kind == "split":
            total += 2.0 * p * 9 * c * (2 * c)
    return total


# ----------------------------------------------------------------------------- Inferer (next row N3)
def attribute_delta(zs, ys, batch_size: int, per_batch=None):
    """`Inferer.compute_attribute_delta`, network/inferer.py:104-153, on latents already computed: `zs` (S, C, H, W) and
    `ys` (S, classes) in DATA-LOADER ORDER, consumed in batches of `batch_size` (drop_last).  Per class: mean latent of the
    samples that have the attribute minus the mean of those that do not (empty sets count as one sample of zeros, :146-147).
    `per_batch` = how many samples of each batch enter the sums: the reference loops `for i in range(len(batch))` over the
    batch DICT (:131), i.e. 2 for {'x', 'y_onehot'}; None = all of them."""
    zs = np.asarray(zs, dtype=np.float64)
    ys = np.asarray(ys)
    S, K = ys.shape
    pos = np.zeros((K,) + zs.shape[1:]); neg = np.zeros_like(pos)
    n_pos = np.zeros(K); n_neg = np.zeros(K)
    for b0 in range(0, S - batch_size + 1, batch_size):
        take = batch_size if per_batch is None else per_batch
        for i in range(b0, b0 + take):
            for c in range(K):
                if ys[i, c] > 0:
                    pos[c] += zs[i]; n_pos[c] += 1
                else:
                    neg[c] += zs[i]; n_neg[c] += 1
    out = np.zeros_like(pos)
    for c in range(K):
        out[c] = pos[c] / max(1.0, n_pos[c]) - neg[c] / max(1.0, n_neg[c])
    return out
