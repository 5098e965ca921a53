"""Fused training step around the HIP path: flat parameter / gradient buffers, one fused clip+guard+Adam
launch, and the single RCCL all-reduce of the flat gradient for ray-sharded data parallelism.

`FusedAdam` is a drop-in for the `torch.optim.Adam(model.parameters(), lr)` + `clip_grad_norm_` +
`on_after_backward` sequence of VolOpt.train_step (volsdf/vsdf.py:214-219); `TrainStep` is the whole
train_step (vsdf.py:196-235) without the dataset / logging plumbing, used by bench.py and the tests.
"""
import ctypes

import torch

from . import lib as _lib
from . import ops
from .ops import _ptr, _stream


class FlatParams:
    """Re-homes every parameter of a module into one flat float32 buffer (and its .grad into another), so that the
    optimiser and the gradient all-reduce are single launches / single collectives."""

    def __init__(self, params):
        self.params = [p for p in params]
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, device=dev)
        self.grad = torch.zeros(n, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view(p.shape)
            p.grad = self.grad[off:off + k].view(p.shape)
            off += k
        self.n = n

    def views(self, buf):
        out, off = [], 0
        for p in self.params:
            out.append(buf[off:off + p.numel()].view(p.shape))
            off += p.numel()
        return out


class FusedAdam:
    """clip_grad_norm_(max_norm) + NaN/Inf guard + Adam in one launch pair (csrc/svs_optim.hip)."""

    def __init__(self, params, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0):
        self.fp = params if isinstance(params, FlatParams) else FlatParams(list(params))
        self.lr, self.betas, self.eps, self.max_norm = lr, betas, eps, max_norm
        dev = self.fp.flat.device
        self.exp_avg = torch.zeros_like(self.fp.flat)
        self.exp_avg_sq = torch.zeros_like(self.fp.flat)
        L = _lib.load()
        self.ws = torch.empty(L.svs_adam_workspace_bytes() // 4, dtype=torch.int32, device=dev)
        self.info = torch.zeros(2, device=dev)
        self.step_count = 0

    def zero_grad(self, set_to_none=False):
        self.fp.grad.zero_()

    def step(self):
        L = _lib.load()
        self.step_count += 1
        _lib.check(L.svs_clip_guard_adam(_ptr(self.fp.flat), _ptr(self.fp.grad), _ptr(self.exp_avg), _ptr(self.exp_avg_sq),
                                         self.fp.n, self.step_count, float(self.max_norm), float(self.lr),
                                         float(self.betas[0]), float(self.betas[1]), float(self.eps), _ptr(self.ws),
                                         _ptr(self.info), _stream()), "svs_clip_guard_adam")

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])


def shard_rays(uv, rank, world):
    """Contiguous ray shard of this rank: uv (1,R,2) -> (1,R/world,2).  R must divide evenly (2048 rays / 8 GPUs)."""
    R = uv.shape[1]
    if R % world:
        raise ValueError(f"{R} rays do not shard over {world} ranks")
    k = R // world
    return uv[:, rank * k:(rank + 1) * k]


def allreduce_flat_grad(flat_grad, world):
    """The one collective of a data-parallel step: sum the flat float32 gradient over ranks (RCCL over xGMI on the GPU
    box, gloo in the CPU tests).  The loss of each rank is already divided by the GLOBAL ray count."""
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


class TrainStep:
    """VolOpt.train_step (volsdf/vsdf.py:196-235) for one batch, on the HIP path end to end:
    forward -> MVS prior lookup -> fused loss (+ output gradients) -> compositing / MLP backward ->
    [gradient all-reduce] -> fused clip + guard + Adam."""

    def __init__(self, model, loss, lr=5e-4, grad_clip=True, world=1, rank=0):
        self.model, self.loss = model, loss
        self.fp = FlatParams(model._flat_param_list())
        self.opt = FusedAdam(self.fp, lr=lr, max_norm=1.0 if grad_clip else 0.0)
        self.world, self.rank = world, rank
        self.grad_views = self.fp.views(self.fp.grad)
        # the unpack kernels write straight into the flat gradient buffer: (grad_v, grad_g, grad_b) views per layer
        it = iter(self.grad_views)
        self.grad_out = []
        for net, n in ((model.implicit_network, 9), (model.rendering_network, 5)):
            group = []
            for _ in range(n):
                gv = next(it)
                gg = next(it) if net.weight_norm else None
                group.append((gv, gg, next(it)))
            self.grad_out.append(group)
        self.beta_grad = next(it)

    def __call__(self, model_input, ground_truth, mvs=None, fast=1):
        """mvs: optional dict(views=[...], same_view=int, img_res=(H,W), inverse_depth=bool) for cost_mapping."""
        m = self.model
        m.train()
        keep = {}
        out = m._forward_impl(model_input, fast, keep)
        if mvs is not None:
            out['pj'], out['pi'], _ = ops.cost_lookup(mvs["views"], mvs["same_view"], mvs["img_res"], cam=keep["cam_loc"],
                                                      dirs=keep["ray_dirs"], z=keep["z_vals"],
                                                      inverse_depth=mvs.get("inverse_depth", False))
        loss_out = self.loss(out, ground_truth)
        g = self.loss.last_grads
        scale = 1.0 / self.world       # each rank's means are over its own shard
        sc = (lambda t: t if (scale == 1.0 or t is None) else t * scale)
        _, _, d_beta = m.backward_from_output_grads(keep, sc(g["rgb_values"]), sc(g["weights"]), sc(g["depth_values"]),
                                                    sc(g["grad_theta"]), out=self.grad_out)
        self.beta_grad.copy_(d_beta.reshape(()))
        allreduce_flat_grad(self.fp.grad, self.world)
        self.opt.step()
        m.invalidate_packed()          # the fused kernel bypasses torch's version counters
        return loss_out, out
