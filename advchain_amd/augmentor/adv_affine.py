"""AdvAffine: adversarial affine transformation (reference: advchain/augmentor/adv_affine.py:13-330).

Parameters (N,5) 2D / (N,9) 3D -> Hardtanh -> matrix (and its inverse) in one tiny HIP launch
(:func:`ops.affine_theta`), then a warp whose sampling grid ``theta * (x,y[,z],1)`` is evaluated in
registers (:func:`ops.affine_warp`) instead of materialising ``F.affine_grid``."""
import logging

import torch

from .. import ops
from .adv_transformation_base import AdvTransformBase, _LazyDiff

logger = logging.getLogger(__name__)

_CFG_2D = ('rot', 'scale_x', 'scale_y', 'shift_x', 'shift_y')
_CFG_3D = ('rot_x', 'rot_y', 'rot_z', 'scale_x', 'scale_y', 'scale_z', 'shift_x', 'shift_y', 'shift_z')


class AdvAffine(AdvTransformBase):
    """Adv Affine."""

    def __init__(self, spatial_dims=2,
                 config_dict={'rot': 30.0 / 180.0, 'scale_x': 0.2, 'scale_y': 0.2, 'shift_x': 0.1, 'shift_y': 0.1,
                              'data_size': [1, 1, 8, 8], 'forward_interp': 'bilinear',
                              'backward_interp': 'bilinear'},
                 image_padding_mode="zeros", power_iteration=False, use_gpu=True, debug=False,
                 device=torch.device("cuda")):
        super(AdvAffine, self).__init__(spatial_dims=spatial_dims, config_dict=config_dict, use_gpu=use_gpu,
                                        debug=debug, device=device)
        self.power_iteration = power_iteration
        self.image_padding_mode = image_padding_mode
        # Q10: reset after init_config; init_parameters() re-reads the config
        self.forward_interp = 'bilinear'
        self.backward_interp = 'bilinear'
        self.affine_matrix = None
        self._inverse_of = None

    def init_config(self, config_dict):
        # adv_affine.py:73-105
        self.translation_x = config_dict['shift_x']
        self.translation_y = config_dict['shift_y']
        self.scale_x = config_dict['scale_x']
        self.scale_y = config_dict['scale_y']
        if self.spatial_dims == 2:
            self.rot_ratio = config_dict['rot']
        if self.spatial_dims == 3:
            self.rot_x = config_dict['rot_x']
            self.rot_y = config_dict['rot_y']
            self.rot_z = config_dict['rot_z']
            self.scale_z = config_dict['scale_z']
            self.translation_z = config_dict['shift_z']
        self.xi = 1e-6
        self.data_size = config_dict['data_size']
        if 'forward_interp' in config_dict:
            self.forward_interp = config_dict['forward_interp']
        if 'backward_interp' in config_dict:
            self.backward_interp = config_dict['backward_interp']

    def _cfg_vector(self):
        if self.spatial_dims == 2:
            return [self.rot_ratio, self.scale_x, self.scale_y, self.translation_x, self.translation_y]
        return [self.rot_x, self.rot_y, self.rot_z, self.scale_x, self.scale_y, self.scale_z,
                self.translation_x, self.translation_y, self.translation_z]

    def init_parameters(self):
        # adv_affine.py:108-119
        self.init_config(self.config_dict)
        self.batch_size = self.data_size[0]
        self.param = self.draw_random_affine_tensor_list(batch_size=self.batch_size)
        return self.param

    def draw_random_affine_tensor_list(self, batch_size, identity_init=False):
        # adv_affine.py:166-180
        num_params = 5 if self.spatial_dims == 2 else 9
        if identity_init:
            return torch.zeros(batch_size, num_params, device=self.device, dtype=torch.float32)
        t = 2 * torch.rand(batch_size, num_params, dtype=torch.float32, device=self.device) - 1
        return torch.nn.Hardtanh()(t)

    def gen_batch_affine_matrix(self, affine_tensors):
        """(N,5|9) -> (N,d,d+1) (adv_affine.py:210-273); also caches the analytic inverse."""
        theta, theta_inv = ops.affine_theta(affine_tensors, self._cfg_vector(), 1.0, self.spatial_dims)
        self._inverse_of = (theta, theta_inv)
        return theta

    def get_inverse_matrix(self, affine_matrix):
        # adv_affine.py:316-324
        if self._inverse_of is not None and self._inverse_of[0] is affine_matrix:
            return self._inverse_of[1]
        d = self.spatial_dims
        homo = torch.eye(d + 1, device=affine_matrix.device, dtype=torch.float32).repeat(affine_matrix.size(0), 1, 1)
        homo[:, :d] = affine_matrix
        return homo.inverse()[:, :d, :]

    def transform(self, data, affine_matrix, interp=None, padding_mode=None, _ride=None, _ride_nonzero=False):
        # adv_affine.py:289-314 (Q9: a caller-supplied padding_mode is replaced by the constructor's)
        if padding_mode is not None:
            padding_mode = self.image_padding_mode
        if interp is None:
            interp = self.forward_interp
        if _ride is not None:    # (solver-internal, see _ride_ok: the validity mask through the same launch)
            return ops.affine_warp(data, affine_matrix, interp, padding_mode, ride=_ride, ride_nonzero=_ride_nonzero)
        if padding_mode == "lowest":
            flat = data.reshape(data.size(0), -1)
            self.padding_values = torch.min(flat, dim=1, keepdim=True).values.detach().clone()
            return ops.affine_warp(data - self.padding_values, affine_matrix, interp, 'zeros') + self.padding_values
        if isinstance(padding_mode, (float, int)):
            self.padding_values = padding_mode
            return ops.affine_warp(data - padding_mode, affine_matrix, interp, 'zeros') + padding_mode
        return ops.affine_warp(data, affine_matrix, interp, padding_mode)

    def _ride_ok(self, interp=None, padding_mode=None):
        """See AdvMorph._ride_ok."""
        pad = self.image_padding_mode
        return (interp is None and padding_mode is None and isinstance(pad, str) and pad != 'lowest'
                and self.forward_interp in ops.RIDE_INTERPS and self.backward_interp in ops.RIDE_INTERPS)

    def forward(self, data, interp=None, padding_mode=None, _ride=None):
        # adv_affine.py:121-146
        if padding_mode is None:
            padding_mode = self.image_padding_mode
        if self.param is None:
            self.init_parameters()
        if interp is None:
            interp = self.forward_interp
        pscale = self.xi if (self.power_iteration and self.is_training) else 1.0
        theta, theta_inv = ops.affine_theta(self.param, self._cfg_vector(), pscale, self.spatial_dims)
        self.affine_matrix = theta
        self._inverse_of = (theta, theta_inv)
        out = self.transform(data, theta, interp=interp, padding_mode=padding_mode, _ride=_ride)
        if _ride is not None:    # (the reference's last forward() of a step is the mask's: adv_compose_solver.py:262-264)
            out, rout = out
            self.diff = _LazyDiff(lambda o=rout, d=_ride: d - o)
            return out, rout
        self.diff = _LazyDiff(lambda o=out.detach(), d=data.detach(): d - o)
        return out

    def backward(self, data, interp=None, padding_mode=None, _ride=None, _ride_nonzero=False):
        # adv_affine.py:154-164
        assert self.param is not None, 'play forward before backward'
        inverse_matrix = self.get_inverse_matrix(self.affine_matrix)
        if interp is None:
            interp = self.backward_interp
        if padding_mode is None:
            padding_mode = self.image_padding_mode
        return self.transform(data, inverse_matrix, interp=interp, padding_mode=padding_mode, _ride=_ride,
                              _ride_nonzero=_ride_nonzero)

    def predict_forward(self, data, interp=None, padding_mode=None):
        return self.forward(data, interp=interp, padding_mode=padding_mode)

    def predict_backward(self, data, interp=None, padding_mode=None, _ride=None, _ride_nonzero=False):
        return self.backward(data, interp=interp, padding_mode=padding_mode, _ride=_ride, _ride_nonzero=_ride_nonzero)

    def train(self):
        # adv_affine.py:204-208
        self.is_training = True
        if self.power_iteration:
            self.param = self.param.sign()
        self.param = torch.nn.Parameter(self.param, requires_grad=True)

    def optimize_parameters(self, step_size=None):
        # adv_affine.py:182-198 : param + step * sign(grad)   (N x 5|9 values: host-level bookkeeping)
        try:
            grad = self.param.grad
            if not isinstance(grad, torch.Tensor):
                raise TypeError('no gradient')
            if self.power_iteration:
                self.param = ops.sign_axpy(None, grad, 1.0, gate=self._gate, old=self.param)
            else:
                self.param = ops.sign_axpy(self.param, grad, step_size, gate=self._gate, old=self.param)     # one launch
        except Exception:
            logging.warning('fail to optimize')
        return self.param

    def rescale_parameters(self):
        return self.param

    def get_name(self):
        return 'affine'

    def is_geometric(self):
        return 1
