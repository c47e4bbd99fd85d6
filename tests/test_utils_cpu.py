"""Model-side helpers of the reference's README usage (common/utils.py:114-212, common/layers.py:5-63): host logic only."""
import random

import numpy as np
import torch

from advchain_amd.common.layers import Fixable2DDropout, Fixable3DDropout
from advchain_amd.common.utils import _disable_tracking_bn_stats, _fix_dropout, random_chain, set_grad


def test_fixable_dropout_replays_its_mask_only_when_asked():
    x = torch.ones(2, 8, 5, 5)
    drop = Fixable2DDropout(p=0.5, lazy_load=True)
    a = drop(x)          # first call: draws and stores a seed
    b = drop(x)          # lazy_load: same channels dropped again
    assert torch.equal(a, b)
    drop.lazy_load = False
    outs = [drop(x) for _ in range(6)]
    assert any(not torch.equal(a, o) for o in outs)       # fresh masks
    drop.eval()
    assert torch.equal(drop(x), x)
    assert Fixable3DDropout(p=0.3, lazy_load=True)(torch.ones(1, 4, 3, 3, 3)).shape == (1, 4, 3, 3, 3)


def test_bn_tracking_and_dropout_toggles_are_restored():
    model = torch.nn.Sequential(torch.nn.Conv2d(1, 4, 3, 1, 1), torch.nn.BatchNorm2d(4), Fixable2DDropout(0.2))
    bn, drop = model[1], model[2]
    assert bn.track_running_stats and not drop.lazy_load
    with _disable_tracking_bn_stats(model):
        assert not bn.track_running_stats and drop.lazy_load
        before = bn.running_mean.clone()
        model(torch.rand(2, 1, 6, 6))
        assert torch.equal(bn.running_mean, before)        # statistics frozen inside the block
    assert bn.track_running_stats and not drop.lazy_load
    with _fix_dropout(model):
        assert drop.lazy_load
    assert not drop.lazy_load
    set_grad(model, False)
    assert all(not p.requires_grad for p in model.parameters())


def test_random_chain_is_a_shuffled_prefix_with_matching_sizes():
    random.seed(0)
    np.random.seed(0)
    items, sizes = ["noise", "bias", "morph", "affine"], [1, 2, 3, 4]
    seen = set()
    for _ in range(50):
        chain, sz = random_chain(items, max_length=3, size_list=sizes)
        assert 1 <= len(chain) <= 3 and len(set(chain)) == len(chain)
        assert sz == [sizes[items.index(c)] for c in chain]
        seen.add(tuple(chain))
    assert len(seen) > 5
    assert random_chain(["only"]) == ["only"]
    assert random_chain(["only"], size_list=[7]) == (["only"], [7])
