"""AdvMorph: adversarial diffeomorphic deformation (reference: advchain/augmentor/adv_morph.py:204-564).

Pipeline of ``DemonsCompose`` (adv_morph.py:454-491): Gaussian-smooth the low-res velocity, upsample,
integrate by scaling-and-squaring (8+ self-compositions), compose with the identity, smooth again,
clamp, then warp with ``grid_sample``.  All of it runs in the HIP kernels of ``advchain_amd/csrc``
through :func:`advchain_amd.ops.demons_field` (hand-written adjoint) and :func:`ops.grid_sample`; the
identity grid is never materialised and the final clamp is applied by the sampler on load."""
import logging

import torch

from .. import bands, ops
from .adv_transformation_base import AdvTransformBase, _LazyDiff

logger = logging.getLogger(__name__)


def get_base_grid(batch_size, image_height, image_width, image_depth=None, device=torch.device('cuda')):
    """Identity sampling grid (N,d,...) in channel order (x,y[,z]) (adv_morph.py:14-55).  Utility only: the
    kernels evaluate the same linspace formula in registers."""
    sizes = [image_height, image_width] + ([] if image_depth is None else [image_depth])
    axes = [torch.linspace(-1, 1, s, device=device) for s in sizes]
    mesh = torch.meshgrid(axes, indexing='ij')
    chans = [m.unsqueeze(0).unsqueeze(0).repeat(batch_size, 1, *([1] * len(sizes))) for m in reversed(mesh)]
    return torch.cat(chans, dim=1)


class AdvMorph(AdvTransformBase):
    """Adv Morph."""

    def __init__(self, spatial_dims=2,
                 config_dict={'epsilon': 1.5, 'data_size': [10, 1, 8, 8], 'vector_size': [4, 4],
                              'forward_interp': 'bilinear', 'backward_interp': 'bilinear'},
                 power_iteration=False, device=torch.device("cuda"), image_padding_mode="zeros",
                 use_gpu: bool = True, debug: bool = False):
        super(AdvMorph, self).__init__(spatial_dims=spatial_dims, config_dict=config_dict, use_gpu=use_gpu,
                                       debug=debug, device=device)
        self.align_corners = True
        self.sigma = 1
        self.gaussian_ks = 5      # the reference overrides it to 9 = 2*int(4*sigma+0.5)+1 (Q3)
        self.smooth_iter = 1
        self.num_steps = 8
        # Q10: reset AFTER init_config read the dict; init_parameters() re-reads the config
        self.forward_interp = 'bilinear'
        self.backward_interp = 'bilinear'
        self.integration_type = 'ss'
        self.param = None
        self.power_iteration = power_iteration
        self.image_padding_mode = image_padding_mode
        self._tables = None
        self._base_grid = None
        self._field_cache = {}
        self._share_fields = False
        self._displacement = None
        self.process_group = None  # set by the solver when the batch is sharded (whole-batch norm, Q2)

    def init_config(self, config_dict):
        # adv_morph.py:247-258
        self.epsilon = config_dict['epsilon']
        self.xi = 0.5
        self.data_size = config_dict['data_size']
        self.vector_size = config_dict['vector_size']
        if 'forward_interp' in config_dict:
            self.forward_interp = config_dict['forward_interp']
        if 'backward_interp' in config_dict:
            self.backward_interp = config_dict['backward_interp']

    def init_parameters(self):
        # adv_morph.py:260-283
        self.init_config(self.config_dict)
        if self.spatial_dims not in (2, 3):
            raise NotImplementedError('only 2D and 3D are supported')
        self._base_grid = None
        self._field_cache = {}
        self._tables = bands.upsample_tables(list(self.vector_size), list(self.data_size[2:]), self.device)
        vector = self.init_velocity(self.data_size[0], *self.vector_size)
        self.param = vector
        return vector

    @property
    def base_grid(self):
        if self._base_grid is None:
            self._base_grid = get_base_grid(self.data_size[0], *self.data_size[2:], device=self.device)
        return self._base_grid

    def init_velocity(self, batch_size, height, width, depth=None, use_zero=False):
        # adv_morph.py:349-375
        shape = (batch_size, self.spatial_dims, height, width) + (() if self.spatial_dims == 2 else (depth,))
        if use_zero:
            velocity = torch.zeros(*shape, device=self.device)
        else:
            velocity = torch.rand(*shape, device=self.device) * 2 - 1
        return self.unit_normalize(velocity)

    # ------------------------------------------------------------------------------------ fields
    def _scale(self):
        return self.xi if (self.power_iteration and self.is_training) else self.epsilon

    def _reduce_sumsq(self):
        if self.process_group is None:
            return None
        import torch.distributed as dist

        def red(t):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.process_group)
            return t
        return red

    def _chain_opts(self, positions_only=False):
        """The attributes DemonsCompose reads (adv_morph.py:236-242,461-471) as the operator's options; None = the
        reference defaults (num_steps=8, smooth_iter=1, sigma=1, scaling and squaring)."""
        n, it = int(self.num_steps), int(self.smooth_iter)
        if n < 1 or it < 1:
            raise NotImplementedError('num_steps and smooth_iter must be at least 1 (got %d, %d)' % (n, it))
        taps = self._taps()
        if (n, it, float(self.sigma), taps) == (8, 1, 1.0, 9) and not positions_only:
            return None
        return (n, it, float(self.sigma), bool(positions_only), taps)

    def _taps(self):
        """Window length of the Gaussian: `gaussian_ks` unless the rule 2 * int(4 sigma + 0.5) + 1 is larger
        (adv_morph.py:393-398; 9 with the defaults, Q3)."""
        return bands.gaussian_taps(self.sigma, self.gaussian_ks)

    def _nine_taps(self):
        """The reference's window rule (adv_morph.py:393-398) gives the 9 taps of the fused kernels."""
        return self._taps() == 9

    def _field(self, sign):
        """Un-clamped sampling grid for sign*scale*param, shared between the data / prediction / mask paths of
        one solver step (the reference recomputes the identical field 4x per step)."""
        if self._tables is None:
            self._tables = bands.upsample_tables(list(self.vector_size), list(self.data_size[2:]), self.device)
        p = self.param
        scale = sign * self._scale()
        want_grad = torch.is_grad_enabled() and p.requires_grad
        opts = self._chain_opts()
        if not self._nine_taps() or self.integration_type != 'ss':      # another Gaussian window / Euler steps: the general route of DemonsCompose (no pairing, no bound)
            return self.DemonsCompose(duv=scale * p) if want_grad else self.DemonsCompose(duv=scale * p.detach()).detach()
        if not self._share_fields:
            return ops.demons_field(p, scale, self._tables, self.spatial_dims == 3, self._reduce_sumsq(), opts)
        key = (scale, opts)
        hit = self._field_cache.get(key)
        if hit is not None and hit[0] is p and hit[1] == p._version and (hit[3] or not want_grad):
            q = hit[2]
            return q if want_grad else q.detach()
        if ops.PAIR_FIELDS:
            # a solver step warps forward with field(+s) and back with field(-s): integrate both as one batch
            s = self._scale()
            qp, qm = ops.demons_field_pair(p, s, self._tables, self.spatial_dims == 3, self._reduce_sumsq(), opts)
            self._field_cache[(s, opts)] = (p, p._version, qp, want_grad)
            self._field_cache[(-s, opts)] = (p, p._version, qm, want_grad)
            return self._field_cache[key][2]
        q = ops.demons_field(p, scale, self._tables, self.spatial_dims == 3, self._reduce_sumsq(), opts)
        self._field_cache[key] = (p, p._version, q, want_grad)
        return q

    def _begin_shared_fields(self):
        """Solver hook: within one solver step every call with the same parameters reuses the field."""
        self._share_fields = True
        self._field_cache = {}

    def _end_shared_fields(self):
        self._share_fields = False
        self._field_cache = {}

    def DemonsCompose(self, duv, init_deformation_dxy=None, smooth=True):
        """Clamped sampling grid (N,d,...) for an explicit low-res velocity (adv_morph.py:454-491).

        The reference's own (and only) call is ``DemonsCompose(duv=duv, init_deformation_dxy=self.base_grid,
        smooth=True)`` (adv_morph.py:299-303,322-324,342-343): composition with the IDENTITY grid followed by the
        smoothing -- the fused HIP chain (``ops.demons_field``), taken whenever ``init_deformation_dxy`` is None, the
        identity grid itself or a tensor equal to it, and ``smooth`` is true.  Any other initial deformation and
        ``smooth=False`` take the same chain up to the sampling positions and then the reference's own steps on them
        (adv_morph.py:474-490), each a HIP operator: ``grid_sample(init, positions, border)``, ``G * (. - id) + id``.
        ``num_steps``, ``smooth_iter``, ``sigma`` and ``gaussian_ks`` are honoured on both routes; a window other than 9
        taps (max(gaussian_ks, 2 int(4 sigma + 0.5) + 1), adv_morph.py:393-398: sigma outside [0.875, 1.125) or a
        gaussian_ks above 9) takes the general route with the plain K-tap Gaussian."""
        if self._tables is None:
            self._tables = bands.upsample_tables(list(self.vector_size), list(self.data_size[2:]), self.device)
        identity = init_deformation_dxy is None or init_deformation_dxy is self._base_grid
        if not identity:
            base = self.base_grid
            if tuple(init_deformation_dxy.shape) != tuple(base.shape):
                raise ValueError('DemonsCompose: init_deformation_dxy must have the shape of the sampling grid %s, got %s'
                                 % (tuple(base.shape), tuple(init_deformation_dxy.shape)))
            identity = (not init_deformation_dxy.requires_grad) and torch.equal(init_deformation_dxy.to(base.device), base)
        if identity and smooth and self._nine_taps() and self.integration_type == 'ss':
            q = ops.demons_field(duv, 1.0, self._tables, self.spatial_dims == 3, self._reduce_sumsq(), self._chain_opts())
            return torch.clamp(q, -1, 1)
        # general route: positions = integrated offsets + identity (adv_morph.py:464-472), then adv_morph.py:474-490
        if self.integration_type != 'ss':
            pos = self._euler_positions(duv)
        else:
            pos = ops.demons_field(duv, 1.0, self._tables, self.spatial_dims == 3, self._reduce_sumsq(),
                                   self._chain_opts(positions_only=True))
        init = self.base_grid if init_deformation_dxy is None else init_deformation_dxy.to(self.device)
        comp = ops.grid_sample(init.contiguous(), pos, 'bilinear', 'border')          # applyComposition{2,3}D
        if smooth:
            comp = ops.axpy(ops.gauss_smooth(ops.axpy(comp, self.base_grid, -1.0), self.sigma, self._taps()), self.base_grid, 1.0)
        return torch.clamp(comp, -1, 1)

    def _euler_positions(self, duv):
        """The `else` branch of vectorFieldExponentiation2D (adv_morph.py:136-141; any integration_type other than 'ss'):
        phi_0 = id + u / 2^n composed n times with the running field, returned as (phi_n - phi_0) + id.  Every step is the
        general HIP sampler (the field as a 2-channel image, border padding).  3D: the reference's own loop cannot run
        (adv_morph.py:171 calls range() on a float) -- the same TypeError is raised here."""
        if self.spatial_dims == 3:
            raise TypeError("'float' object cannot be interpreted as an integer")    # range(2.0 ** nb_steps), adv_morph.py:171
        n, it = int(self.num_steps), int(self.smooth_iter)
        u = duv
        for _ in range(it):
            u = ops.gauss_smooth(u, self.sigma, self._taps())
        phi0 = ops.upsample_field(u, self._tables, 1.0 / (2.0 ** n))
        phi = phi0
        for _ in range(n):
            phi = ops.grid_sample(phi0, phi, 'bilinear', 'border')                  # applyComposition2D(interval_phi, phi)
        return ops.axpy(ops.axpy(phi, phi0, -1.0), self.base_grid, 1.0)

    def get_deformation_displacement_field(self, duv=None):
        # adv_morph.py:339-347
        if duv is None:
            duv = self.param
        dxy = self.DemonsCompose(duv=duv)
        perm = (0, 2, 3, 1) if self.spatial_dims == 2 else (0, 2, 3, 4, 1)
        return dxy, dxy.permute(*perm) - self.base_grid.permute(*perm)

    @property
    def displacement(self):
        d = self._displacement
        return d.get() if isinstance(d, _LazyDiff) else d

    # ------------------------------------------------------------------------------------ warps
    def transform(self, data, deformation_dxy, interp=None, padding_mode=None, _clamp=False, _ride=None, _ride_nonzero=False):
        """Warp with a dense sampling grid (N,d,...) (adv_morph.py:524-558).  _ride (solver-internal, see _ride_ok): a
        one-channel tensor warped through the same grid by the same launch -> (out, ride_out)."""
        if padding_mode is None:
            padding_mode = self.image_padding_mode
        if interp is None:
            interp = self.forward_interp
        if _ride is not None:
            return ops.grid_sample(data, deformation_dxy, interp, padding_mode, _clamp, ride=_ride, ride_nonzero=_ride_nonzero)
        if padding_mode == "lowest":
            flat = data.reshape(data.size(0), -1)
            self.padding_values = torch.min(flat, dim=1, keepdim=True).values.detach().clone()
            out = ops.grid_sample(data - self.padding_values, deformation_dxy, interp, 'zeros', _clamp)
            return out + self.padding_values
        if isinstance(padding_mode, (float, int)):
            self.padding_values = padding_mode
            out = ops.grid_sample(data - padding_mode, deformation_dxy, interp, 'zeros', _clamp)
            return out + padding_mode
        return ops.grid_sample(data, deformation_dxy, interp, padding_mode, _clamp)

    def _ride_ok(self, interp=None, padding_mode=None):
        """Can the solver's validity mask ride along with the data through this transform's warps (one launch for both)?
        The reference warps the mask by calling forward / backward a second time with the defaults
        (adv_compose_solver.py:262-268): same field, the transform's own interpolation and padding."""
        pad = self.image_padding_mode
        return (interp is None and padding_mode is None and isinstance(pad, str) and pad != 'lowest'
                and self.forward_interp in ops.RIDE_INTERPS and self.backward_interp in ops.RIDE_INTERPS)

    def forward(self, data, interp=None, padding_mode=None, _ride=None):
        # adv_morph.py:285-311
        if self.param is None:
            self.param = self.init_parameters()
        if interp is None:
            interp = self.forward_interp
        q = self._field(+1.0)
        out = self.transform(data, q, interp=interp, padding_mode=padding_mode, _clamp=True, _ride=_ride)
        rout = None
        if _ride is not None:
            out, rout = out
        # detached captures: a closure over `out` / `q` themselves would keep the whole DemonsCompose graph (n+1
        # full-resolution fields) alive until the next forward
        if _ride is not None:   # (the reference's last forward() of a step is the mask's: adv_compose_solver.py:262-264)
            self.diff = _LazyDiff(lambda o=rout, d=_ride: o - d)
        else:
            self.diff = _LazyDiff(lambda o=out.detach(), d=data.detach(): o - d)
        perm = (0, 2, 3, 1) if self.spatial_dims == 2 else (0, 2, 3, 4, 1)
        self._displacement = _LazyDiff(
            lambda q=q.detach(): torch.clamp(q, -1, 1).permute(*perm) - self.base_grid.permute(*perm))
        return out if _ride is None else (out, rout)

    def backward(self, data, interp=None, padding_mode=None, _ride=None, _ride_nonzero=False):
        # adv_morph.py:313-331
        if interp is None:
            interp = self.backward_interp
        q = self._field(-1.0)
        return self.transform(data, q, interp=interp, padding_mode=padding_mode, _clamp=True, _ride=_ride,
                              _ride_nonzero=_ride_nonzero)

    def predict_forward(self, data, interp=None, padding_mode=None):
        return self.forward(data, interp=interp, padding_mode=padding_mode)

    def predict_backward(self, data, interp=None, padding_mode=None, _ride=None, _ride_nonzero=False):
        return self.backward(data, interp=interp, padding_mode=padding_mode, _ride=_ride, _ride_nonzero=_ride_nonzero)

    # ------------------------------------------------------------------------------------ lifecycle
    def train(self):
        # adv_morph.py:493-499
        self.is_training = True
        if self.param is None:
            self.init_parameters()
        if self.power_iteration:
            self.param = self.unit_normalize(self.param)
        self.param = torch.nn.Parameter(self.param, requires_grad=True)

    def eval(self):
        super(AdvMorph, self).eval()
        self._field_cache = {}

    def set_parameters(self, param):
        super(AdvMorph, self).set_parameters(param)
        self._field_cache = {}

    def optimize_parameters(self, step_size=None):
        # adv_morph.py:501-516
        try:
            grad = self.param.grad
            if self.power_iteration:
                self.param = ops.normalized_axpy(None, grad, 1.0, gate=self._gate, old=self.param)
            else:
                self.param = ops.normalized_axpy(self.param, grad, step_size, gate=self._gate, old=self.param)
        except Exception:
            logging.warning('fail to optimize.This may due to the strength of deformation is too strong, '
                            'that the structure cannot be well preserved. Try use smaller epsilon')
        self._field_cache = {}
        return self.param

    def rescale_parameters(self, param=None):
        # adv_morph.py:518-522
        if param is None:
            param = self.param
        self.param = self.unit_normalize(param.detach())
        return self.param

    def get_name(self):
        return 'morph'

    def is_geometric(self):
        return 1
