"""AdvBias: adversarial multiplicative bias field (reference: advchain/augmentor/adv_bias.py:50-380).

The reference synthesises the field with a dense ``conv_transpose`` against a (10s+3)^2 / (4s-3)^3
B-spline window, crops it and upsamples it.  Here the same linear map is kept as ONE small banded matrix
per axis (:mod:`advchain_amd.bands`) and the HIP kernel evaluates control points -> exp -> clip -> multiply
per voxel in a single pass (12 B/voxel of HBM traffic instead of a dense convolution)."""
import logging

import numpy as np
import torch

from .. import bands, ops
from .adv_transformation_base import AdvTransformBase

logger = logging.getLogger(__name__)


class AdvBias(AdvTransformBase):
    """Adv Bias."""

    def __init__(self, spatial_dims=2,
                 config_dict={'epsilon': 0.3, 'control_point_spacing': [64, 64], 'downscale': 2,
                              'data_size': [2, 1, 128, 128], 'interpolation_order': 3, 'init_mode': 'random',
                              'space': 'log'},
                 power_iteration=False, ignore_values=None, use_gpu=True, debug=False, device=torch.device("cuda")):
        super(AdvBias, self).__init__(spatial_dims=spatial_dims, config_dict=config_dict, use_gpu=use_gpu,
                                      debug=debug, device=device)
        self.param = None
        self.power_iteration = power_iteration
        self.ignore_values = ignore_values

    def init_config(self, config_dict):
        # adv_bias.py:84-102
        self.epsilon = config_dict['epsilon']
        self.xi = 1e-6
        self.data_size = config_dict['data_size']
        self.downscale = config_dict['downscale']
        assert self.downscale <= min(self.data_size[2:]), 'downscale factor is too  large'
        self.control_point_spacing = [i // self.downscale for i in config_dict['control_point_spacing']]
        if sum(self.control_point_spacing) > sum([48] * len(self.control_point_spacing)):
            logging.warning('control point spacing may be too large, please increase the downscale factor.')
        self.interpolation_order = config_dict['interpolation_order']
        self.space = config_dict['space']
        self.init_mode = config_dict['init_mode']

    def init_parameters(self):
        # adv_bias.py:104-128
        self.init_config(self.config_dict)
        self._dim = len(self.control_point_spacing)
        self.spacing = self.control_point_spacing
        self._dtype = torch.float32
        self.batch_size = self.data_size[0]
        self._image_size = np.array(self.data_size[2:])
        assert self.spatial_dims == self._dim, f'image dimension must be {self.spatial_dims} as specified in spatial_dims'
        self.magnitude = self.epsilon
        assert 0 <= self.magnitude < 1, 'please set magnitude witihin [0,1)'
        self.order = self.interpolation_order
        self.use_log = True if self.space == 'log' else False
        self.param, self.interp_kernel = self.init_control_points_config()
        return self.param

    def init_control_points_config(self, init_mode=None):
        """Control-point lattice, crop arithmetic (adv_bias.py:202-277, Q6) and the band tables."""
        mode = self.init_mode if init_mode is None else init_mode
        stride = np.array(self.spacing)
        low = self._image_size / (1.0 * self.downscale)
        cp_grid = np.ceil(np.divide(low, stride)).astype(dtype=int)
        inner_image_size = np.multiply(stride, cp_grid) - (stride - 1)
        cp_grid = cp_grid + 2
        image_size_diff = inner_image_size - low
        image_size_diff_floor = np.floor((np.abs(image_size_diff) / 2)) * np.sign(image_size_diff)
        self._crop_start = (image_size_diff_floor + np.remainder(image_size_diff, 2) * np.sign(image_size_diff)).astype(int)
        self._crop_end = image_size_diff_floor.astype(int)
        self.cp_grid = [self.batch_size, 1] + cp_grid.tolist()
        self._stride = stride.astype(int).tolist()
        self.low = -np.inf
        self.high = np.inf
        if mode == 'gaussian':
            self.param = torch.ones(*self.cp_grid, dtype=self._dtype, device=self.device).normal_(mean=0, std=0.5)
        elif mode == 'random':
            if self.use_log:
                self.low = np.log(1 - self.magnitude)
                self.high = np.log(1 + self.magnitude)
            else:
                self.low = -self.magnitude
                self.high = self.magnitude
            self.param = torch.rand(*self.cp_grid, dtype=self._dtype, device=self.device) * (self.high - self.low) + self.low
        elif mode == 'identity':
            self.param = torch.zeros(*self.cp_grid, dtype=self._dtype, device=self.device)
        else:
            raise NotImplementedError
        self._tables = self._build_tables()
        self._bias_field = None
        return self.param, None

    def _build_tables(self):
        """Per-axis (image_size x n_control_points) map = linear-upsample o B-spline-synthesis-and-crop."""
        key = ("bias", self._dim, tuple(int(c) for c in self.cp_grid[2:]), tuple(int(s) for s in self._stride), int(self.order),
               tuple(int(c) for c in self._crop_start), tuple(int(c) for c in self._crop_end),
               tuple(int(s) for s in self._image_size))
        return bands.cached_tables(key, self.device, self._build_tables_uncached)

    def _build_tables_uncached(self):
        variant = '2d' if self._dim == 2 else '3d'
        mats, lows = [], []
        for a in range(self._dim):
            W = bands.bspline_synthesis_matrix(self.cp_grid[2 + a], self._stride[a], self.order, variant,
                                               self._crop_start[a], self._crop_end[a])
            lows.append(W.shape[0])
            mats.append(W)
        img = [int(s) for s in self._image_size]
        factors = [img[a] / lows[a] for a in range(self._dim)]
        if any(f > 1 for f in factors):  # adv_bias.py:316-327
            for a in range(self._dim):
                if self._dim == 2:
                    U = bands.linear_upsample_matrix(lows[a], img[a])          # Upsample(size=...)
                else:
                    if int(np.floor(lows[a] * float(factors[a]))) != img[a]:
                        raise NotImplementedError('bias field: scale_factor upsample does not reproduce the image size')
                    U = bands.linear_upsample_matrix(lows[a], img[a], float(factors[a]))  # Upsample(scale_factor=...)
                mats[a] = U @ mats[a]
        else:
            for a in range(self._dim):
                if lows[a] != img[a]:
                    raise NotImplementedError('bias field: low-res grid larger than the image is not supported')
        return bands.BandTables(mats, self.device)

    # the reference keeps ``bias_field`` from the last forward (clipped field, expanded over channels)
    @property
    def bias_field(self):
        if self._bias_field is None and self.param is not None:
            self._bias_field = ops.bias_field_only(self.param, self._tables, self.magnitude, self.use_log, 1.0)
        return self._bias_field

    @bias_field.setter
    def bias_field(self, v):
        self._bias_field = v

    def train(self):
        # adv_bias.py:130-134
        self.is_training = True
        if self.power_iteration:
            self.param = self.unit_normalize(self.param.data)
        self.param = torch.nn.Parameter(self.param.data, requires_grad=True)

    def rescale_parameters(self):
        # adv_bias.py:136-137
        self.param = torch.clamp(self.param, self.low, self.high)

    def optimize_parameters(self, step_size=0.3):
        # adv_bias.py:139-148
        grad = self.param.grad
        if self.power_iteration:
            self.param = ops.normalized_axpy(None, grad, 1.0, gate=self._gate, old=self.param)
        else:
            self.param = ops.normalized_axpy(self.param, grad, step_size, gate=self._gate, old=self.param)
        return self.param

    def forward(self, data, **kwargs):
        # adv_bias.py:152-188 -- one fused kernel: field synthesis + exp + clip + multiply
        if self.param is None:
            self.init_parameters()
        cp_scale = self.xi if (self.power_iteration and self.is_training) else 1.0
        out, field = ops.bias_apply(self.param, data, self._tables, self.magnitude, self.use_log, cp_scale)
        if field.size(1) < data.size(1):
            field = field.expand(data.size())
        self._bias_field = field
        self.diff = field
        if self.ignore_values is not None:
            if isinstance(self.ignore_values, float):
                keep = (abs(data - self.ignore_values) < 1e-8).detach()
                out = torch.where(keep, torch.full_like(out, self.ignore_values), out)
            else:
                raise UnboundLocalError('ignore values must be in float type (adv_bias.py:176-184)')
        return out

    def compute_smoothed_bias(self, cpoint=None, interpolation_kernel=None, padding=None, stride=None):
        """Un-clipped field exp(L) (or 1+L) for the given control points (adv_bias.py:279-335)."""
        if cpoint is None:
            cpoint = self.param
        ones = torch.ones((cpoint.shape[0], 1) + tuple(self.data_size[2:]), device=cpoint.device, dtype=torch.float32)
        out, _ = ops.bias_apply(cpoint, ones, self._tables, 3.0e38, self.use_log, 1.0)
        return out

    def clip_bias(self, bias_field, magnitude=None):
        # adv_bias.py:337-356
        if magnitude is None:
            magnitude = self.magnitude
        assert magnitude >= 0
        return 1 + torch.clamp(bias_field - 1, -magnitude, magnitude)

    def backward(self, data, **kwargs):
        return data

    def predict_forward(self, data, **kwargs):
        return data

    def predict_backward(self, data, **kwargs):
        return data

    def get_name(self):
        return 'bias'

    def is_geometric(self):
        return 0
