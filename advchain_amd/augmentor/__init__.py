"""Drop-in surface of ``advchain.augmentor`` (reference: advchain/augmentor/__init__.py:1-7)."""
from .adv_transformation_base import AdvTransformBase  # noqa: F401
from .adv_noise import AdvNoise  # noqa: F401
from .adv_bias import AdvBias  # noqa: F401
from .adv_morph import AdvMorph, get_base_grid  # noqa: F401
from .adv_affine import AdvAffine  # noqa: F401
from .adv_compose_solver import ComposeAdversarialTransformSolver  # noqa: F401

__all__ = ["AdvTransformBase", "AdvNoise", "AdvBias", "AdvMorph", "AdvAffine",
           "ComposeAdversarialTransformSolver", "get_base_grid"]
