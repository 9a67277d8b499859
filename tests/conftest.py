"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"`  : oracle vs golden vectors, host logic, C-ABI load/export checks (no GPU needed).
`-m gpu`        : parity tests proper -- HIP path (through the C-ABI) vs oracle / golden vectors.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _usable_cores(cap=16):
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


# the GPU box reports 256 hardware threads; the oracle (torch CPU) must not oversubscribe its cgroup
torch.set_num_threads(_usable_cores())


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    """Load tests/golden/<name>.npz as {key: torch.Tensor | np.ndarray(int)}."""
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        out = {}
        for k in z.files:
            a = z[k]
            out[k] = torch.from_numpy(a.copy()) if a.dtype.kind == "f" else a.copy()
        return out


def sub(d, prefix):
    """Entries of d whose key starts with prefix, prefix stripped."""
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture(autouse=True)
def _inference_mode_by_default():
    """Parity tests exercise the inference path; gradient tests switch autograd on explicitly."""
    with torch.no_grad():
        yield


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get
