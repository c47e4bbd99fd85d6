"""Drop-in alias: ``from advchain.augmentor import *`` resolves to the MI355X-native package
``advchain_amd`` (same class names, constructor arguments and methods as cherise215/advchain)."""
import importlib
import sys

import advchain_amd as _impl

for _name in ("augmentor", "common", "common.loss", "common.utils", "common.layers",
              "augmentor.adv_transformation_base", "augmentor.adv_noise", "augmentor.adv_bias",
              "augmentor.adv_morph", "augmentor.adv_affine", "augmentor.adv_compose_solver"):
    sys.modules[__name__ + "." + _name] = importlib.import_module("advchain_amd." + _name)

augmentor = sys.modules[__name__ + ".augmentor"]
common = sys.modules[__name__ + ".common"]
__version__ = _impl.__version__
