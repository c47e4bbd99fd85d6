"""Dropout layers whose mask can be replayed (reference: advchain/common/layers.py:5-63).  Model-side
helpers: they let the user's network stay deterministic across the solver's inner loop."""
import torch
import torch.nn as nn
from torch.nn import functional as F


class _FixableDropout(nn.Module):
    _fn = None

    def __init__(self, p: float = 0.5, inplace=False, lazy_load: bool = False, training=True):
        super().__init__()
        if p < 0 or p > 1:
            raise ValueError("dropout probability has to be between 0 and 1, but got {}".format(p))
        self.p = p
        self.inplace = inplace
        self.seed = None
        self.lazy_load = lazy_load
        self.training = training

    def forward(self, X):
        # replay the stored seed only in training mode with lazy_load set (layers.py:20-33)
        if self.training and self.lazy_load and self.seed is not None:
            seed = self.seed
        else:
            seed = torch.seed()
        self.seed = seed
        torch.manual_seed(seed)
        return type(self)._fn(X, p=self.p, training=self.training, inplace=self.inplace)


class Fixable2DDropout(_FixableDropout):
    _fn = staticmethod(F.dropout2d)


class Fixable3DDropout(_FixableDropout):
    _fn = staticmethod(F.dropout3d)
