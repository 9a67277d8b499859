from .model import FlowStep, FlowModel, Glow
from .module import (ActNorm, LinearZeros, Conv2d, Conv2dZeros, CouplingNet,
                     f, Invertible1x1Conv, Permutation2d, GaussianDiag,
                     Split2d, Squeeze2d)

__all__ = (
    'FlowStep', 'FlowModel', 'Glow',
    'ActNorm', 'LinearZeros', 'Conv2d', 'Conv2dZeros', 'CouplingNet',
    'f', 'Invertible1x1Conv', 'Permutation2d', 'GaussianDiag',
    'Split2d', 'Squeeze2d',
)
