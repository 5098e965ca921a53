"""VolSDFLoss with the reference's constructor, attributes and output dict (volsdf/model/loss.py:15-114).

One fused HIP launch (svs_loss) evaluates every term AND the gradient of the total with respect to the model
outputs; the gradients are kept in `self.last_grads` for the training backward (they are exactly what
`loss.backward()` would deliver to rgb_values / grad_theta / weights / depth_values).
"""
import numpy as np
import torch
from torch import nn

from svs_hip import ops


def anneal_linearly(t, val0, val1):
    if t >= 1:
        return val1
    if t <= 0:
        return val0
    return val0 + (val1 - val0) * np.minimum(t, 1.)


class VolSDFLoss(nn.Module):
    def __init__(self, rgb_loss, eikonal_weight, rgb_weight=1., mvs_weight=0., sparse_weight=0., anneal_rgb=0, gce=1,
                 confi=0):
        super().__init__()
        if rgb_loss not in ("torch.nn.L1Loss",):
            raise NotImplementedError("the fused loss kernel implements rgb_loss = torch.nn.L1Loss (config/vol/dtu.yaml:20)")
        self.eikonal_weight, self.rgb_weight, self.mvs_weight = eikonal_weight, rgb_weight, mvs_weight
        self.sparse_weight, self.gce, self.anneal_rgb, self.confi = sparse_weight, gce, anneal_rgb, confi
        self.iter_step = 0
        self.last_grads = None

    def set_stg(self, stg):
        self.iter_step = 0
        if stg >= 1:
            self.anneal_rgb = 0
            self.sparse_weight = 0
            raise NotImplementedError

    def anneal_state(self):
        """(annealed, anneal_sparse) of the current iteration (loss.py:88-90,103-105): while `iter_step < anneal_rgb` the
        rgb term uses `rgb_smooth` on rays without MVS support and the sparsity weight decays linearly to zero."""
        annealed = self.sparse_weight > 0 and self.anneal_rgb > 0 and self.iter_step < self.anneal_rgb
        return annealed, (anneal_linearly(self.iter_step / self.anneal_rgb, 1.0, 0.) if annealed else 0.0)

    def forward(self, model_outputs, ground_truth, norm=None, advance=True, anneal_dev=None, grad_theta_out=None):
        """anneal_dev: optional device float32[2] holding anneal_state() -- the kernels then read the annealing from it
        (a captured launch sequence stays valid while the iteration count advances; trainer.TrainStep)."""
        dev = model_outputs['rgb_values'].device
        annealed, anneal_sparse = self.anneal_state()
        has_mvs = 'pi' in model_outputs
        target = ground_truth['rgb_smooth'] if annealed else ground_truth['rgb']
        losses, grads = ops.loss_fwd_bwd(
            model_outputs['rgb_values'], target.to(dev), model_outputs['weights'], model_outputs.get(
                'depth_values_all', model_outputs['depth_values']),
            grad_theta=model_outputs.get('grad_theta'), pi=model_outputs.get('pi'), pj=model_outputs.get('pj'),
            rgb_weight=self.rgb_weight, eikonal_weight=self.eikonal_weight,
            mvs_weight=self.mvs_weight if has_mvs else 0.0, sparse_weight=self.sparse_weight, gce=float(self.gce),
            confi=float(self.confi), annealed=annealed and has_mvs, anneal_sparse=float(anneal_sparse), norm=norm,
            anneal_dev=anneal_dev, grad_theta_out=grad_theta_out)
        self.last_grads = grads
        if advance:
            self.iter_step += 1
        total = losses[4]
        # (the depth the loss read -- and whose gradient grads['depth_values'] is -- is depth_values_all for the fg + bg model)
        depth_key = 'depth_values_all' if 'depth_values_all' in model_outputs else 'depth_values'
        diff = [(k, depth_key if k == 'depth_values' else k) for k in ('rgb_values', 'grad_theta', 'weights', 'depth_values')]
        diff = [(gk, ok) for gk, ok in diff
                if ok in model_outputs and torch.is_tensor(model_outputs[ok]) and model_outputs[ok].requires_grad]
        if diff:
            # connect the fused loss to autograd so that the reference's `loss.backward()` (vsdf.py:215) works
            total = _FusedLossFunction.apply(losses, grads, [gk for gk, _ in diff], *[model_outputs[ok] for _, ok in diff])
        return {'rgb_loss': losses[0], 'eikonal_loss': losses[1], 'mvs_loss': losses[2], 'sparse_loss': losses[3],
                'loss': total}


class _FusedLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, losses, grads, names, *tensors):
        ctx.grads = [grads[n].reshape(t.shape) for n, t in zip(names, tensors)]
        return losses[4].clone()

    @staticmethod
    def backward(ctx, g):
        return (None, None, None, *[g * x for x in ctx.grads])
