"""pytorch-glow_amd: MI355X-native Glow flow engine (HIP kernels for gfx950 behind the reference's
nn.Module surface).  Import name: ``pytorch_glow_amd`` (see ../pytorch_glow_amd.py).

Layout
  csrc/        hand-written HIP kernels + the C ABI (include/glowhip.h) -> libglowhip.so
  _lib.py      ctypes binding, build helper, error translation (no fallback path)
  _plan.py     FlowPlan: a whole flow stack as one C call
  network/     FlowStep / FlowModel / Glow and the flow layers (reference surface)
  misc/        ops + util helpers with the reference's names
  parallel.py  one-process-per-GPU data parallelism over RCCL
  profile/     built-in profiles 'celeba' / 'test' in the reference's JSON schema
"""
from . import _lib  # noqa: F401
from ._lib import GlowHipError, build, lib  # noqa: F401
from ._plan import FlowPlan  # noqa: F401
from . import misc, network  # noqa: F401
from .network import (ActNorm, Conv2d, Conv2dZeros, FlowModel, FlowStep, GaussianDiag, Glow,  # noqa: F401
                      Invertible1x1Conv, LinearZeros, Permutation2d, Split2d, Squeeze2d, f)

__version__ = "0.1.1"
