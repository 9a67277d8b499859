"""Flow layers and models (host-side mirror of the reference's ``network`` package for the flow hot path).
Builder / Trainer / Inferer of the reference are orchestration around this path and are not part of it."""
from . import model as _model
from . import module as _module

_LAYERS = ("ActNorm", "LinearZeros", "Conv2d", "Conv2dZeros", "CouplingNet", "f", "Invertible1x1Conv",
           "Permutation2d", "GaussianDiag", "Split2d", "Squeeze2d")
_MODELS = ("FlowStep", "FlowModel", "Glow")

for _n in _LAYERS:
    globals()[_n] = getattr(_module, _n)
for _n in _MODELS:
    globals()[_n] = getattr(_model, _n)

__all__ = _MODELS + _LAYERS
