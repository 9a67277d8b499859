"""Flow layers with the reference's names, constructor arguments, attributes and state_dict keys
(corenel/pytorch-glow network/module.py), computing through libglowhip (HIP, gfx950) only.

Every ``forward`` takes fp32 CUDA(HIP) tensors and raises on anything else: there is no PyTorch/CPU
fallback.  torch is used for memory, parameters and RNG draws.

Differences from the reference that are deliberate:
  * nothing mutates its inputs (F7); outputs are new tensors;
  * ``logdet`` given as a python number comes back as an (N,) tensor (the reference returns a 0-dim
    tensor when no data-dependent term is added);
  * forward passes are inference passes: outputs carry no autograd graph (backward kernels are the next
    scope row).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .._lib import check, lib, ptr, require_device_tensor, stream_ptr
from .._plan import PlanCache


def _logdet_arg(logdet, n, device):
    """None -> None; number -> (N,) tensor; tensor -> fp32 (N,) tensor on device."""
    if logdet is None:
        return None
    if isinstance(logdet, torch.Tensor):
        t = logdet.detach().to(device=device, dtype=torch.float32)
        if t.dim() == 0:
            t = t.expand(n)
        assert t.shape == (n,), f"logdet must have shape ({n},), got {tuple(t.shape)}"
        return t.contiguous()
    return torch.full((n,), float(logdet), dtype=torch.float32, device=device)


class ActNorm(nn.Module):
    """Activation normalisation (reference network/module.py:9-149).

    forward : y = (x + bias) * exp(3*logs),  logdet += 3*sum(logs)*H*W
    reverse : y = x * exp(-3*logs) - bias,   logdet -= 3*sum(logs)*H*W
    The first training-mode forward sets bias/logs from the batch (data-dependent init, :86-120).
    """

    # `bias_inited` / `logs_inited` are plain attributes in the reference (module.py:30-31).  Here every RESET of one bumps a class-wide
    # epoch, which is what lets FlowModel cache "every ActNorm is initialised" (a walk over ~6 400 sub-modules per training forward
    # otherwise) without missing a flag somebody clears by hand.
    RESET_EPOCH = [0]

    def _get_bias_inited(self):
        return self.__dict__.get("_bias_inited", False)

    def _set_bias_inited(self, v):
        self.__dict__["_bias_inited"] = bool(v)
        if not v:
            ActNorm.RESET_EPOCH[0] += 1

    def _get_logs_inited(self):
        return self.__dict__.get("_logs_inited", False)

    def _set_logs_inited(self, v):
        self.__dict__["_logs_inited"] = bool(v)
        if not v:
            ActNorm.RESET_EPOCH[0] += 1

    bias_inited = property(_get_bias_inited, _set_bias_inited)
    logs_inited = property(_get_logs_inited, _set_logs_inited)

    def __init__(self, num_channels, scale=1., logscale_factor=3., batch_variance=False):
        super().__init__()
        assert logscale_factor == 3., "the HIP kernels hard-code logscale_factor=3 (the only value the reference uses)"
        self.num_channels = num_channels
        self.scale = scale
        self.logscale_factor = logscale_factor
        self.batch_variance = batch_variance
        self.bias_inited = False
        self.logs_inited = False
        self.register_parameter('bias', nn.Parameter(torch.zeros(1, self.num_channels, 1, 1)))
        self.register_parameter('logs', nn.Parameter(torch.zeros(1, self.num_channels, 1, 1)))

    def initialize_parameters(self, x):
        """bias = -mean(x), logs = log(scale/(std+1e-6))/3 over (N,H,W); training mode only (:93,:109).  batch_variance=True
        (:109-110): the second moment is pooled over the channels as well -- one log-scale, copied into every channel."""
        if not self.training:
            return
        n, c, h, w = x.shape
        init = lib().glowhip_actnorm_init_batch_variance if self.batch_variance else lib().glowhip_actnorm_init
        check(init(ptr(x), c * h * w, n, c, h * w, float(self.scale), ptr(self.bias.data), ptr(self.logs.data), stream_ptr(x.device)))
        self.bias_inited = True
        self.logs_inited = True

    def forward(self, x, logdet=None, reverse=False):
        assert len(x.shape) == 4
        assert x.shape[1] == self.num_channels, \
            'Input shape should be NxCxHxW, however channels are {} instead of {}'.format(x.shape[1], self.num_channels)
        assert x.device == self.bias.device and x.device == self.logs.device, \
            'Expect input device {} instead of {}'.format(self.bias.device, x.device)
        x = require_device_tensor(x, "ActNorm input")
        if not (self.bias_inited and self.logs_inited):
            self.initialize_parameters(x)
        n, c, h, w = x.shape
        ld_in = _logdet_arg(logdet, n, x.device)
        ld_out = torch.empty(n, dtype=torch.float32, device=x.device) if ld_in is not None else None
        y = torch.empty_like(x)
        check(lib().glowhip_actnorm(ptr(x), ptr(y), ptr(self.bias), ptr(self.logs), n, c, h * w, int(bool(reverse)),
                                    ptr(ld_in), ptr(ld_out), stream_ptr(x.device)))
        return y, ld_out


class LinearZeros(nn.Linear):
    """Zero-initialised linear layer with a learned log-scale (reference network/module.py:152-185).
    Only used by the class-conditional branch (y_condition), which is outside the flow hot path, so it
    stays a plain torch layer."""

    def __init__(self, in_features, out_features, bias=True, logscale_factor=3.):
        super().__init__(in_features, out_features, bias)
        self.logscale_factor = logscale_factor
        self.weight.data.zero_()
        self.bias.data.zero_()
        self.register_parameter('logs', nn.Parameter(torch.zeros(out_features)))

    def forward(self, x):
        return super().forward(x) * torch.exp(self.logs * self.logscale_factor)


def _conv_call(x, weight, bias, post_bias, post_logs, relu):
    x = require_device_tensor(x, "conv input")
    n, cin, h, w = x.shape
    cout, cin_w, kh, kw = weight.shape
    assert cin == cin_w, f"input has {cin} channels, weight expects {cin_w}"
    assert kh == kw and kh in (1, 3), "HIP convolution supports 1x1 and 3x3 kernels"
    y = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
    check(lib().glowhip_conv2d(ptr(x), cin * h * w, ptr(weight), ptr(bias), ptr(y), n, cin, h, w, cout, kh,
                               ptr(post_bias), ptr(post_logs), int(relu), stream_ptr(x.device)))
    return y


class Conv2d(nn.Conv2d):
    """'SAME' convolution followed by its own ActNorm (reference network/module.py:188-260):
    no conv bias when do_actnorm, weight ~ N(0, 0.05)."""

    @staticmethod
    def get_padding(padding_type, kernel_size, stride):
        assert padding_type in ['SAME', 'VALID'], "Unsupported padding type: {}".format(padding_type)
        if isinstance(kernel_size, int):
            kernel_size = [kernel_size, kernel_size]
        if padding_type == 'SAME':
            assert stride == 1, "'SAME' padding only supports stride=1"
            return tuple((k - 1) // 2 for k in kernel_size)
        return tuple(0 for _ in kernel_size)

    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), stride=1, padding_type='SAME',
                 do_weightnorm=False, do_actnorm=True, dilation=1, groups=1):
        padding = self.get_padding(padding_type, kernel_size, stride)
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                         bias=(not do_actnorm))
        if padding_type != 'SAME' or dilation != 1 or groups != 1:
            raise NotImplementedError("the HIP convolution implements what the flow uses: SAME, dilation 1, groups 1")
        self.do_weight_norm = do_weightnorm
        self.do_actnorm = do_actnorm
        self.weight.data.normal_(mean=0.0, std=0.05)
        if self.do_actnorm:
            self.actnorm = ActNorm(out_channels)
        else:
            self.bias.data.zero_()

    def forward(self, x, relu=False):
        if not self.do_actnorm:
            return _conv_call(x, self.weight, self.bias, None, None, relu)
        an = self.actnorm
        if self.training and not (an.bias_inited and an.logs_inited):
            raw = _conv_call(x, self.weight, None, None, None, False)
            an.initialize_parameters(raw)
        return _conv_call(x, self.weight, None, an.bias, an.logs, relu)


class Conv2dZeros(nn.Conv2d):
    """Zero-initialised 'SAME' convolution with bias and per-channel log-scale
    (reference network/module.py:263-297): y = (conv(x) + b) * exp(3*logs)."""

    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), stride=1, padding_type='SAME',
                 logscale_factor=3, dilation=1, groups=1, bias=True):
        padding = Conv2d.get_padding(padding_type, kernel_size, stride)
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        if padding_type != 'SAME' or dilation != 1 or groups != 1 or not bias or logscale_factor != 3:
            raise NotImplementedError("the HIP convolution implements what the flow uses")
        self.logscale_factor = logscale_factor
        self.bias.data.zero_()
        self.weight.data.zero_()
        self.register_parameter("logs", nn.Parameter(torch.zeros(out_channels, 1, 1)))

    def forward(self, x):
        return _conv_call(x, self.weight, self.bias, None, self.logs, False)


class CouplingNet(nn.Sequential):
    """The 3-layer coupling CNN ``f`` (reference network/module.py:300-319); state_dict keys 0/2/4 as the
    reference's nn.Sequential.  The ReLUs are fused into the convolution epilogues."""

    def forward(self, x):
        h = self[0](x, relu=True)
        h = self[2](h, relu=True)
        return self[4](h)


def f(in_channels, hidden_channels, out_channels):
    """Conv2d 3x3 -> ReLU -> Conv2d 1x1 -> ReLU -> Conv2dZeros 3x3."""
    return CouplingNet(
        Conv2d(in_channels, hidden_channels),
        nn.ReLU(inplace=True),
        Conv2d(hidden_channels, hidden_channels, kernel_size=1),
        nn.ReLU(inplace=True),
        Conv2dZeros(hidden_channels, out_channels)
    )


class Invertible1x1Conv(nn.Module):
    """Invertible 1x1 convolution (reference network/module.py:322-369).  log|det W| and W^-1 come from an
    in-kernel LU; like the reference they are recomputed on every call."""

    def __init__(self, num_channels, lu_decomposition=False):
        super().__init__()
        self.num_channels = num_channels
        self.lu_decomposition = lu_decomposition
        if self.lu_decomposition:
            raise NotImplementedError()
        w_init = np.linalg.qr(np.random.randn(num_channels, num_channels))[0].astype('float32')
        self.register_parameter('weight', nn.Parameter(torch.Tensor(w_init)))

    def forward(self, x, logdet=None, reverse=False):
        x = require_device_tensor(x, "Invertible1x1Conv input")
        n, c, h, w = x.shape
        assert c == self.num_channels
        dev = x.device
        s = stream_ptr(dev)
        aux = torch.empty(c * c + 1, dtype=torch.float32, device=dev)
        winv, lad = aux[:c * c], aux[c * c:]
        scratch = torch.empty(int(lib().glowhip_invconv_scratch_bytes(c)), dtype=torch.uint8, device=dev)
        check(lib().glowhip_invconv_prepare(ptr(self.weight), c, ptr(winv), ptr(lad), ptr(scratch), s))
        ld_in = _logdet_arg(logdet, n, dev)
        ld_out = torch.empty(n, dtype=torch.float32, device=dev) if ld_in is not None else None
        z = torch.empty_like(x)
        m = winv if reverse else self.weight
        check(lib().glowhip_invconv(ptr(x), ptr(z), ptr(m), ptr(lad), n, c, h * w, int(bool(reverse)), ptr(ld_in),
                                    ptr(ld_out), s))
        return z, ld_out


class Permutation2d(nn.Module):
    """Fixed channel permutation: reversed order or a random shuffle (reference network/module.py:372-397)."""

    def __init__(self, num_channels, shuffle=False):
        super().__init__()
        self.num_channels = num_channels
        self.indices = np.arange(self.num_channels - 1, -1, -1, dtype=np.int64)
        if shuffle:
            np.random.shuffle(self.indices)
        self.indices_inverse = np.zeros(self.num_channels, dtype=np.int64)
        for i in range(self.num_channels):
            self.indices_inverse[self.indices[i]] = i
        self._tables = {}

    def device_tables(self, device):
        key = str(device)
        if key not in self._tables:
            self._tables[key] = (torch.from_numpy(self.indices.astype(np.int32)).to(device),
                                 torch.from_numpy(self.indices_inverse.astype(np.int32)).to(device))
        return self._tables[key]

    def forward(self, x, reverse=False):
        assert len(x.shape) == 4
        x = require_device_tensor(x, "Permutation2d input")
        n, c, h, w = x.shape
        idx, inv = self.device_tables(x.device)
        y = torch.empty_like(x)
        check(lib().glowhip_permute_channels(ptr(x), ptr(y), ptr(inv if reverse else idx), n, c, h * w,
                                             stream_ptr(x.device)))
        return y


class GaussianDiag:
    """Diagonal Gaussian helpers (reference network/module.py:400-483)."""

    log_2pi = float(np.log(2 * np.pi))

    @staticmethod
    def eps(shape_tensor, eps_std=None):
        """N(0, eps_std) draw shaped like ``shape_tensor``; ``eps_std or 1.`` as the reference (:419), so
        eps_std=0 means 1."""
        eps_std = eps_std or 1.
        return torch.randn_like(shape_tensor) * eps_std

    @staticmethod
    def flatten_sum(tensor):
        assert len(tensor.shape) == 4
        return tensor.sum(dim=[1, 2, 3])

    @staticmethod
    def logps(mean, logs, x):
        """Element-wise log-density (API helper; the reduction-fused HIP kernel is :meth:`logp`)."""
        return -0.5 * (GaussianDiag.log_2pi + 2. * logs + ((x - mean) ** 2) / torch.exp(2. * logs))

    @staticmethod
    def logp(mean, logs, x):
        """sum_{C,H,W} logps -> (N,), one fused HIP reduction."""
        x = require_device_tensor(x, "GaussianDiag.logp input")
        n, c, h, w = x.shape
        mean = None if mean is None else require_device_tensor(mean.expand_as(x), "mean")
        logs = None if logs is None else require_device_tensor(logs.expand_as(x), "logs")
        out = torch.empty(n, dtype=torch.float32, device=x.device)
        scratch = torch.empty(2 * n, dtype=torch.int64, device=x.device)   # accumulator + sticky non-finite flag per sample
        check(lib().glowhip_gaussian_logp(ptr(x), c * h * w, ptr(mean), ptr(logs), c * h * w, n, c, h * w, None, ptr(out),
                                          ptr(scratch), stream_ptr(x.device)))
        return out

    @staticmethod
    def sample(mean, logs, eps_std=None):
        eps = GaussianDiag.eps(mean, eps_std)
        return mean + torch.exp(logs) * eps


class Split2d(nn.Module):
    """Multi-scale split (reference network/module.py:486-536).  forward scores the second half of the
    channels under a prior predicted from the first half and returns the first half; reverse samples it."""

    glowhip_kind = _lib.LAYER_SPLIT2D

    def __init__(self, num_channels):
        super().__init__()
        self.num_channels = num_channels
        self.conv2d_zeros = Conv2dZeros(num_channels // 2, num_channels)
        self._plans = PlanCache()

    def prior(self, z):
        h = self.conv2d_zeros(z)
        return h[:, 0::2, ...], h[:, 1::2, ...]

    def forward(self, x, logdet=0., reverse=False, eps_std=None, eps=None):
        """``eps`` (optional, beyond the reference signature): inject the N(0,1)*eps_std draw."""
        x = require_device_tensor(x, "Split2d input")
        n, c, h, w = x.shape
        if not reverse:
            assert c == self.num_channels
            plan = self._plans.get([self], (c, h, w), x.device)
            return plan.encode(x, None, _logdet_arg(logdet, n, x.device), want_logdet=True)
        assert c == self.num_channels // 2
        plan = self._plans.get([self], (2 * c, h, w), x.device)
        if eps is None:
            eps = GaussianDiag.eps(x, eps_std)
        z, _ = plan.decode(x, [require_device_tensor(eps, "eps")], None, want_logdet=False)
        return z, logdet

    def __deepcopy__(self, memo):
        return _deepcopy_without_plans(self, memo)


class Squeeze2d(nn.Module):
    """Space-to-depth by ``factor`` (reference network/module.py:539-612)."""

    glowhip_kind = _lib.LAYER_SQUEEZE

    def __init__(self, factor=2):
        super().__init__()
        self.factor = factor

    @staticmethod
    def _run(x, factor, reverse):
        assert factor >= 1
        if factor == 1:
            return x
        x = require_device_tensor(x, "Squeeze2d input")
        n, c, h, w = x.shape
        f2 = factor * factor
        if reverse:
            assert c >= f2 and c % f2 == 0
            y = torch.empty((n, c // f2, h * factor, w * factor), dtype=torch.float32, device=x.device)
        else:
            assert h % factor == 0 and w % factor == 0
            y = torch.empty((n, c * f2, h // factor, w // factor), dtype=torch.float32, device=x.device)
        check(lib().glowhip_squeeze2d(ptr(x), ptr(y), n, c, h, w, factor, int(reverse), stream_ptr(x.device)))
        return y

    @staticmethod
    def unsqueeze(x, factor=2):
        return Squeeze2d._run(x, factor, True)

    @staticmethod
    def squeeze(x, factor=2):
        return Squeeze2d._run(x, factor, False)

    def forward(self, x, logdet=None, reverse=False):
        return (self.unsqueeze(x, self.factor) if reverse else self.squeeze(x, self.factor)), logdet


def _deepcopy_without_plans(module, memo):
    """copy.deepcopy support: C plan handles are not copyable; the copy gets an empty cache."""
    import copy
    cls = module.__class__
    new = cls.__new__(cls)
    memo[id(module)] = new
    for k, v in module.__dict__.items():
        new.__dict__[k] = PlanCache() if isinstance(v, PlanCache) else copy.deepcopy(v, memo)
    return new
