"""Tensor helpers with the reference's names and argument meaning (misc/ops.py of corenel/pytorch-glow).

They are address arithmetic and bookkeeping on torch tensors (views, a cat, a dim-list reduction) used by
callers of the flow path; the flow kernels themselves fold these index maps into their loads
 and never call them.
"""
import torch


def _dims(dim):
    return sorted([dim] if isinstance(dim, int) else list(dim))


def reduce_mean(tensor, dim=None, keepdim=False, out=None):
    """Mean over a dimension list (reference misc/ops.py:4-37)."""
    res = torch.mean(tensor) if dim is None else tensor.mean(dim=_dims(dim), keepdim=keepdim)
    if out is not None:
        out.copy_(res)
    return res


def reduce_sum(tensor, dim=None, keepdim=False, out=None):
    """Sum over a dimension list (reference misc/ops.py:40-73)."""
    res = torch.sum(tensor) if dim is None else tensor.sum(dim=_dims(dim), keepdim=keepdim)
    if out is not None:
        out.copy_(res)
    return res


def tensor_equal(a, b, eps=1e-6):
    """Same shape and max-abs difference <= eps (reference misc/ops.py:76-92)."""
    if a.shape != b.shape:
        return False
    return 0 <= float(torch.max(torch.abs(a - b))) <= eps


def split_channel(tensor, split_type='simple'):
    """'simple': first/second half of the channels; 'cross': even/odd channels (views).
    Reference misc/ops.py:95-113."""
    assert len(tensor.shape) == 4
    assert split_type in ['simple', 'cross']
    nc = tensor.shape[1]
    if split_type == 'simple':
        return tensor[:, :nc // 2, ...], tensor[:, nc // 2:, ...]
    return tensor[:, 0::2, ...], tensor[:, 1::2, ...]


def cat_channel(a, b):
    """Concatenate on the channel axis (reference misc/ops.py:116-127)."""
    return torch.cat((a, b), dim=1)


def count_pixels(tensor):
    """H * W of an NCHW tensor (reference misc/ops.py:130-140)."""
    assert len(tensor.shape) == 4
    return int(tensor.shape[2] * tensor.shape[3])


def onehot(y, num_classes):
    """One-hot labels (reference misc/ops.py:143-160; label path, outside the flow hot path)."""
    assert len(y.shape) in [1, 2], "Label y should be 1D or 2D vector"
    y_onehot = torch.zeros(y.shape[0], num_classes, device=y.device)
    idx = y.unsqueeze(-1) if len(y.shape) == 1 else y
    return y_onehot.scatter_(1, idx, 1)
