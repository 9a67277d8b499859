"""Flow layers and models (host-side mirror of the reference's ``network`` package for the flow hot path), plus the two
callers either side of it: `Builder` (model / optimiser / schedule / snapshot glue), `Trainer` (the reference's training loop
around `training.TrainLoop`) and `Inferer` (the inverse-path application) -- the names ``network/__init__.py:1-16`` exports."""
from . import model as _model
from . import module as _module
from .builder import Builder
from .inferer import Inferer
from .trainer import Trainer

_LAYERS = ("ActNorm", "LinearZeros", "Conv2d", "Conv2dZeros", "CouplingNet", "f", "Invertible1x1Conv",
           "Permutation2d", "GaussianDiag", "Split2d", "Squeeze2d")
_MODELS = ("FlowStep", "FlowModel", "Glow")

for _n in _LAYERS:
    globals()[_n] = getattr(_module, _n)
for _n in _MODELS:
    globals()[_n] = getattr(_model, _n)

__all__ = _MODELS + _LAYERS + ("Builder", "Trainer", "Inferer")
