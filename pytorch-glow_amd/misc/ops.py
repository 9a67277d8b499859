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
    """True when the two tensors have one shape and no element differs by more than ``eps`` (reference misc/ops.py:76-92; a NaN
    anywhere makes it False, as there)."""
    if tuple(a.shape) != tuple(b.shape):
        return False
    if a.numel() == 0:
        return True
    worst = (a - b).abs().max().item()
    return worst <= eps          # (a NaN difference compares False)


def split_channel(tensor, split_type='simple'):
    """Two channel halves of an NCHW tensor as views (reference misc/ops.py:95-113): 'simple' = the first C/2 channels and the
    rest, 'cross' = the even-numbered and the odd-numbered channels."""
    assert len(tensor.shape) == 4
    assert split_type in ['simple', 'cross']
    if split_type == 'cross':
        return tensor[:, 0::2], tensor[:, 1::2]
    half = tensor.shape[1] // 2
    return tensor.narrow(1, 0, half), tensor.narrow(1, half, tensor.shape[1] - half)


def cat_channel(a, b):
    """Inverse of the 'simple' split: the two tensors joined along dim 1 (reference misc/ops.py:116-127)."""
    return torch.cat([a, b], 1)


def count_pixels(tensor):
    """Spatial size H * W of an NCHW tensor (reference misc/ops.py:130-140)."""
    assert len(tensor.shape) == 4
    _, _, height, width = tensor.shape
    return int(height) * int(width)


def onehot(y, num_classes):
    """Class indices (B,) or (B, 1) -> one-hot rows (B, num_classes), float32 on y's device (reference misc/ops.py:143-160; the
    label path, outside the flow hot path)."""
    assert len(y.shape) in [1, 2], "Label y should be 1D or 2D vector"
    index = y.reshape(y.shape[0], -1).long()
    rows = torch.zeros((y.shape[0], num_classes), dtype=torch.float32, device=y.device)
    rows.scatter_(1, index, 1.0)
    return rows
