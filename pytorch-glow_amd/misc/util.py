"""The slice of the reference's misc/util.py the flow path needs: profile loading (attribute-style
dict, since `easydict` is not a dependency here), device-list parsing and seeding.
Reference: misc/util.py:18-29 (load_profile), :34-75 (get_devices), :515-524 (manual_seed)."""
import json
import os
import random
import re

import numpy as np
import torch


class AttrDict(dict):
    """dict with attribute access, nested dicts converted recursively (stands in for EasyDict)."""

    def __init__(self, d=None, **kwargs):
        super().__init__()
        for k, v in dict(d or {}, **kwargs).items():
            self[k] = v

    def __setitem__(self, key, value):
        if isinstance(value, dict) and not isinstance(value, AttrDict):
            value = AttrDict(value)
        elif isinstance(value, (list, tuple)):
            value = type(value)(AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v for v in value)
        super().__setitem__(key, value)

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(key) from e

    __setattr__ = __setitem__


def load_profile(filepath):
    """JSON profile (reference schema) -> attribute dict; None if the file does not exist (as the reference).
    Also accepts the name of a built-in profile: 'celeba' or 'test' (see pytorch-glow_amd/profile)."""
    if os.path.exists(filepath):
        with open(filepath) as f:
            return AttrDict(json.load(f))
    if filepath in ("celeba", "test"):
        from .. import profile
        return AttrDict(profile.builtin(filepath))
    return None


def get_devices(devices, verbose=True):
    """Usable devices among those a profile lists: ['cpu'] or a list of CUDA ordinals.
    'cuda:N' entries beyond torch.cuda.device_count() are dropped; none left -> ['cpu']."""
    def parse(device):
        origin = str(device)
        if isinstance(device, str) and re.search(r'cuda:(\d+)', device):
            device = int(re.findall(r'cuda:(\d+)', device)[0])
        if isinstance(device, int) and 0 <= device <= torch.cuda.device_count() - 1:
            return device
        if verbose:
            print('[Builder] Incorrect device "{}"'.format(origin))
        return None

    use_cpu = any(isinstance(d, str) and d.find('cpu') >= 0 for d in devices)
    use_cuda = any(isinstance(d, int) or d.find('cuda') >= 0 for d in devices)
    assert not (use_cpu and use_cuda), 'CPU and GPU cannot be mixed.'
    if use_cuda:
        devices = [d for d in (parse(d) for d in devices) if d is not None]
        if len(devices) == 0:
            if verbose:
                print('[Builder] No available GPU found, use CPU only')
            devices = ['cpu']
    return devices


def manual_seed(seed):
    """Seed python, numpy and torch (all devices)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
