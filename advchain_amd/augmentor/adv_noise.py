"""AdvNoise: additive adversarial noise (reference: advchain/augmentor/adv_noise.py:10-117)."""
import logging

import torch

from .. import ops
from .adv_transformation_base import AdvTransformBase, _LazyDiff

logger = logging.getLogger(__name__)


class AdvNoise(AdvTransformBase):
    """x + epsilon * delta with delta unit-L2 per sample (xi * delta in power-iteration training)."""

    def __init__(self, spatial_dims=2, config_dict={'epsilon': 0.1, 'xi': 1e-6, 'data_size': [10, 1, 8, 8]},
                 power_iteration=False, ignore_values=None, use_gpu=True, debug=False, device=torch.device("cuda")):
        super(AdvNoise, self).__init__(spatial_dims=spatial_dims, config_dict=config_dict, use_gpu=use_gpu,
                                       debug=debug, device=device)
        self.power_iteration = power_iteration
        self.ignore_values = ignore_values

    def init_config(self, config_dict):
        self.epsilon = config_dict['epsilon']
        self.xi = config_dict['xi']
        self.data_size = config_dict['data_size']

    def init_parameters(self):
        # adv_noise.py:41-49
        noise = self.unit_normalize(torch.randn(*self.data_size, device=self.device, dtype=torch.float32))
        self.param = noise
        return noise

    def train(self):
        # adv_noise.py:108-114
        self.is_training = True
        if self.param is None:
            self.init_parameters()
        if self.power_iteration:
            self.param = self.unit_normalize(self.param)
        self.param = torch.nn.Parameter(self.param, requires_grad=True)

    def forward(self, data, **kwargs):
        # adv_noise.py:67-90
        if self.param is None:
            self.init_parameters()
        scale = self.xi if (self.power_iteration and self.is_training) else self.epsilon
        out = ops.axpy(data, self.param, scale)
        if self.ignore_values is not None:
            keep = (abs(data - self.ignore_values) < 1e-8).detach()
            out = torch.where(keep, torch.full_like(out, float(self.ignore_values)), out)
        self.diff = _LazyDiff(lambda o=out.detach(), d=data.detach(): o - d)
        return out

    def optimize_parameters(self, step_size=None):
        # adv_noise.py:51-64 : unit-normalised ascent step, fused into one reduction + one axpy launch
        if step_size is None:
            step_size = self.step_size
        grad = self.param.grad
        if self.power_iteration:
            self.param = ops.normalized_axpy(None, grad, 1.0, gate=self._gate, old=self.param)
        else:
            self.param = ops.normalized_axpy(self.param, grad, step_size, gate=self._gate, old=self.param)
        return self.param

    def rescale_parameters(self):
        # adv_noise.py:92-94
        self.param = self.unit_normalize(self.param.detach(), p_type='l2')

    def backward(self, data, **kwargs):
        return data

    def predict_forward(self, data, **kwargs):
        return data

    def predict_backward(self, data, **kwargs):
        return data

    def get_name(self):
        return 'noise'
