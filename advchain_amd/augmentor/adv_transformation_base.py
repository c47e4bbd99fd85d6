"""Plug-in base class of the adversarial transforms (same protocol as the reference's
``advchain/augmentor/adv_transformation_base.py:5-189``: third-party subclasses written against the
reference keep working inside :class:`ComposeAdversarialTransformSolver`)."""
import torch

from .. import ops


class _LazyDiff(object):
    """Descriptor-free lazy value: ``thunk`` is evaluated on first read (keeps full-resolution
    bookkeeping tensors such as ``transform.diff`` off the hot path)."""

    def __init__(self, thunk):
        self._thunk, self._val = thunk, None

    def get(self):
        if self._thunk is not None:
            self._val, self._thunk = self._thunk(), None
        return self._val


class AdvTransformBase(object):
    """Adv Transformer base (reference: adv_transformation_base.py:5-189)."""

    def __init__(self, spatial_dims=2, config_dict={"data_size": [1, 1, 1, 1]}, use_gpu=True,
                 device=torch.device("cuda"), debug=False):
        self.spatial_dims = spatial_dims
        assert self.spatial_dims == 2 or self.spatial_dims == 3, 'only support 2D/3D'
        self.config_dict = config_dict
        data_dim = len(config_dict["data_size"])
        assert data_dim == self.spatial_dims + 2, \
            f"check data size in the config file, should be {self.spatial_dims + 2}D, but got {data_dim}D"
        self.param = None
        self.is_training = False
        self.use_gpu = use_gpu
        self.device = device if self.use_gpu else torch.device('cpu')
        self.debug = debug
        self._diff = None
        self._gate = None     # solver hook: a device scalar (the step's loss); a non-finite gate voids optimize_parameters()
        self.init_config(self.config_dict)
        self.step_size = 1  # step size for optimizing data augmentation

    # ``diff`` is evaluated lazily (the reference materialises it on every forward)
    @property
    def diff(self):
        d = self._diff
        return d.get() if isinstance(d, _LazyDiff) else d

    @diff.setter
    def diff(self, value):
        self._diff = value

    def init_config(self, config_dict=None):
        raise NotImplementedError

    def init_parameters(self):
        raise NotImplementedError

    def set_parameters(self, param):
        self.param = param.detach().clone()

    def get_parameters(self):
        return self.param

    def set_step_size(self, step_size=1):
        self.step_size = step_size

    def get_step_size(self):
        return self.step_size

    def train(self):
        if self.param is None:
            self.init_parameters()
        self.is_training = True
        self.param = self.param.detach().clone()
        self.param.requires_grad = True

    def eval(self):
        if self.is_training:
            try:
                self.param.requires_grad = False
            except Exception:
                self.param = self.param.detach()
            self.is_training = False

    def rescale_parameters(self, param=None):
        if param is None:
            param = self.param
        self.param = param.renorm(p=2, dim=0, maxnorm=self.epsilon)
        return self.param

    def optimize_parameters(self, step_size=None):
        raise NotImplementedError

    def forward(self, data, **kwargs):
        raise NotImplementedError

    def backward(self, data, **kwargs):
        raise NotImplementedError

    def predict_forward(self, data, **kwargs):
        raise NotImplementedError

    def predict_backward(self, data, **kwargs):
        raise NotImplementedError

    def unit_normalize(self, d, p_type='l2'):
        """Per-sample normalisation (adv_transformation_base.py:129-156).  'l2' on a GPU tensor without
        autograd runs the fused HIP reduction; the differentiable / exotic variants use tensor ops."""
        if p_type == 'l2' and d.is_cuda and d.dtype == torch.float32 and not (d.requires_grad and torch.is_grad_enabled()):
            return ops.normalized_axpy(None, d, 1.0).view(d.size())
        old_size = d.size()
        flat = d.reshape(d.size(0), -1)
        if p_type == 'l1':
            out = flat.div(flat.norm(p=1, dim=1, keepdim=True).expand_as(flat))
        elif p_type == 'infinity':
            out = flat / (1e-20 + torch.max(flat, 1, keepdim=True)[0].expand_as(flat))
        elif p_type == 'l2':
            out = flat / (torch.norm(flat, dim=1, keepdim=True) + 1e-20)
        else:
            out = flat
        return out.view(old_size)

    def rescale_intensity(self, data, new_min=0, new_max=1, eps=1e-20):
        bs, c = data.size(0), data.size(1)
        flat = data.reshape(bs * c, -1)
        old_max = torch.max(flat, dim=1, keepdim=True).values
        old_min = torch.min(flat, dim=1, keepdim=True).values
        out = (flat - old_min + eps) / (old_max - old_min + eps) * (new_max - new_min) + new_min
        return out.view(data.size())

    def get_name(self):
        raise NotImplementedError

    def is_geometric(self):
        return 0
