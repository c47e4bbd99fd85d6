"""Segmentation-consistency loss (reference: advchain/common/loss.py:8-249).

'mse', 'contour' and 'kl' all run in the fused HIP kernels (:func:`advchain_amd.ops.consistency_sums`:
softmax + mask + squared error + KL sum + 3^d edge stencils in two launches forward, one backward).

Batch sharding: ``global_batch`` overrides N in every normaliser -- mse ~ 1/(N^2 K V^2),
contour ~ 1/(N V) -- so that the per-shard values SUM to the whole-batch loss and the mse:contour mix
(hence the ascent direction) does not depend on the shard size."""
import warnings

import torch

from .. import ops

MAX_CLASSES = 16      # kMaxK of csrc/loss.hip: the fused kernels keep one softmax row per voxel in registers


def _check_operands(output, reference):
    """The restrictions of the fused kernels, stated where the user meets them (INTEGRATION.md "Known deviations"): fp32
    ROCm tensors (no CPU path -- ops raises), at most MAX_CLASSES channels, gradient w.r.t. the prediction only."""
    if output.size(1) > MAX_CLASSES:
        raise NotImplementedError('the fused consistency kernels take at most %d classes, got %d'
                                  % (MAX_CLASSES, output.size(1)))
    if torch.is_grad_enabled() and isinstance(reference, torch.Tensor) and reference.requires_grad:
        warnings.warn('advchain_amd: the consistency loss is differentiated w.r.t. the prediction only; the reference '
                      'is treated as a constant (detach it to silence this warning)', stacklevel=3)


def _pooled(x, scale):
    pool = torch.nn.AvgPool2d if x.dim() == 4 else torch.nn.AvgPool3d
    return pool(2 ** scale)(x)


def _single_channel_mask(mask, K):
    """The solver's validity mask has K identical channels (Q12); stencils and MSE only need what the
    kernel reads: 1 channel, or K distinct ones."""
    if mask is None:
        return None
    if mask.shape[1] not in (1, K):
        raise ValueError('mask must have 1 or %d channels' % K)
    if mask.shape[1] == K and K > 1 and mask.stride(1) == 0:
        return mask[:, :1]          # expanded view of a 1-channel mask
    return mask


def calc_segmentation_consistency(output, reference, divergence_types=['kl', 'contour'],
                                  divergence_weights=[1.0, 0.5], class_weights=None, scales=[0],
                                  mask=None, is_gt=False, global_batch=None):
    """Difference between two predictions (logits), same signature as the reference (loss.py:8-87)."""
    if class_weights is not None:
        raise NotImplementedError
    spatial_dims = output.dim() - 2
    assert spatial_dims == 2 or spatial_dims == 3, 'only support 2d or 3d segmentation'
    assert output.dim() == reference.dim(), 'output and reference must have the same rank'
    K = reference.size(1)
    _check_operands(output, reference)
    dist = 0.
    for scale in scales:
        out_s, ref_s = (output, reference) if scale == 0 else (_pooled(output, scale), _pooled(reference, scale))
        N = out_s.shape[0]
        V = 1
        for s in out_s.shape[2:]:
            V *= s
        Ng = N if global_batch is None else int(global_batch)
        w_mse = sum(w for t, w in zip(divergence_types, divergence_weights) if t == 'mse')
        w_cnt = sum(w for t, w in zip(divergence_types, divergence_weights) if t == 'contour')
        for t in divergence_types:
            if t not in ('kl', 'mse', 'contour'):
                raise NotImplementedError
        has_mse = 'mse' in divergence_types
        has_cnt = 'contour' in divergence_types and K > 1
        has_kl = 'kl' in divergence_types
        m = _single_channel_mask(mask if scale == 0 else (None if mask is None else mask), K)
        if has_mse or has_cnt or has_kl:
            # 'mse': MSELoss(mean) over N*K*V elements, divided again by numel(mask)/K (loss.py:62-64, Q13): N*V for the
            # default all-ones / K-channel mask, N*V/K for a caller's 1-channel mask (an expanded view counts as K)
            mask_ch = K if mask is None else mask.shape[1]
            c_mse = (w_mse / (float(Ng) * K * V * (float(Ng) * mask_ch * V / K))) if has_mse else 0.0
            # 'contour': mean over classes 1..K-1 of  2D 0.5*(MSE_x + MSE_y) | 3D 1/3*(2*MSE_A + MSE_B)  (loss.py:74-79,211-219, Q14)
            if has_cnt:
                if spatial_dims == 2:
                    c_a = c_b = w_cnt * 0.5 / (float(Ng) * V * (K - 1))
                else:
                    c_a = w_cnt * (2.0 / 3.0) / (float(Ng) * V * (K - 1))
                    c_b = w_cnt * (1.0 / 3.0) / (float(Ng) * V * (K - 1))
            else:
                c_a = c_b = 0.0
            # 'kl': mean over the N*V voxels of sum_k m_k p_k (log p_k - log q_k)  (loss.py:244-248)
            w_kl = sum(w for t, w in zip(divergence_types, divergence_weights) if t == 'kl')
            c_kl = (w_kl / (float(Ng) * V)) if has_kl else 0.0
            coef = [(2 ** scale) * c for c in (c_mse, c_a, c_b, c_kl)]
            val, _ = ops.consistency_sums(out_s, ref_s, m, coef, ref_is_prob=is_gt, want_edges=has_cnt)
            dist = val if isinstance(dist, float) and dist == 0. else dist + val     # (0. + x is x: no launch for it)
    # (x / 1.0 is x: the reference's division by the number of scales is skipped for its only call, scales = [0])
    return dist if len(scales) == 1 else dist / (1.0 * len(scales))


def kl_divergence(reference, pred, mask=None, is_gt=False, global_batch=None):
    """KL(P||Q) of two logit maps (loss.py:223-249): the 'kl' term of the fused kernels on its own."""
    K = pred.size(1)
    _check_operands(pred, reference)
    V = 1
    for s in pred.shape[2:]:
        V *= s
    Ng = pred.shape[0] if global_batch is None else int(global_batch)
    val, _ = ops.consistency_sums(pred, reference, _single_channel_mask(mask, K), [0.0, 0.0, 0.0, 1.0 / (float(Ng) * V)],
                                  ref_is_prob=is_gt, want_edges=False)
    return val


def calc_segmentation_mse_consistency(input, target):
    return calc_segmentation_consistency(output=input, reference=target, divergence_types=['mse'],
                                         divergence_weights=[1.0], class_weights=None, mask=None)


def calc_segmentation_kl_consistency(input, target):
    return calc_segmentation_consistency(output=input, reference=target, divergence_types=['kl'],
                                         divergence_weights=[1.0], class_weights=None, mask=None)
