"""Model-state context managers used by the solver (reference: advchain/common/utils.py:114-173) and
the chain-sampling helper of the README usage (utils.py:180-212)."""
import contextlib
import random

import numpy as np
import torch

from .layers import Fixable2DDropout, Fixable3DDropout


def _toggle_fixable_dropout(model):
    for _, module in model.named_modules():
        if isinstance(module, (Fixable2DDropout, Fixable3DDropout)):
            module.lazy_load = not module.lazy_load


@contextlib.contextmanager
def _disable_tracking_bn_stats(model):
    """BatchNorm running statistics are not updated inside the block; Fixable*Dropout.lazy_load is
    flipped on entry and on exit (utils.py:114-147)."""
    saved = {}
    for name, module in model.named_modules():
        if isinstance(module, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
            saved[name] = module.track_running_stats
            module.track_running_stats = False
    _toggle_fixable_dropout(model)
    yield
    for name, module in model.named_modules():
        if name in saved:
            module.track_running_stats = saved[name]
    _toggle_fixable_dropout(model)


@contextlib.contextmanager
def _fix_dropout(model):
    """Flip Fixable*Dropout.lazy_load inside the block (utils.py:149-173)."""
    _toggle_fixable_dropout(model)
    yield
    _toggle_fixable_dropout(model)


def set_grad(module, requires_grad=False):
    for p in module.parameters():
        p.requires_grad = requires_grad


def random_chain(alist, max_length=None, size_list=None):
    """Random sub-chain in random order (utils.py:180-212; the reference relies on the 2-argument
    ``random.shuffle`` removed in Python 3.11 and on an undefined name for 1-element lists -- both
    are made well-defined here)."""
    length = len(alist)
    assert length >= 1, "input list must contains at least one element"
    max_length = length if max_length is None else min(max_length, length)
    if length == 1:
        return [alist[0]] if size_list is None else ([alist[0]], [size_list[0]])
    sub_len = np.random.randint(low=1, high=max_length + 1)
    order = list(range(length))
    random.shuffle(order)
    chain = [alist[i] for i in order][:sub_len]
    if size_list is not None and len(size_list) >= 0:
        return chain, [size_list[i] for i in order][:sub_len]
    return chain
