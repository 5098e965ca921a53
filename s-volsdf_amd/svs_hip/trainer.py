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
    import torch.distributed as dist
    if world > 1 or (dist.is_available() and dist.is_initialized()):
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


class _GroupedOutputs(dict):
    """Model outputs of a step that ran as ray groups: per-ray tensors are concatenated in ray order on first access
    (the step itself never needs the merged tensors; logging does, every 50 steps)."""

    def __init__(self, results):
        super().__init__()
        self._parts = [r[1] for r in results]

    def __missing__(self, key):
        vals = [p[key] for p in self._parts]
        self[key] = torch.cat(vals, 0) if torch.is_tensor(vals[0]) and vals[0].dim() > 0 else vals[0]
        return self[key]

    def __contains__(self, key):
        return key in self._parts[0]

    def keys(self):
        return self._parts[0].keys()


class _GroupedLosses(dict):
    """Loss terms of a step that ran as ray groups: each group's terms are already normalised by the whole batch, so a
    term is the sum over the groups -- formed on first access (logging reads them every 50 steps; summed eagerly they
    were ten 5-us launches on the main stream between Adam and the next step's first kernel)."""

    def __init__(self, results):
        super().__init__()
        self._parts = [r[0] for r in results]

    def __missing__(self, key):
        vals = [p[key] for p in self._parts]
        self[key] = torch.stack(vals).sum(0) if torch.is_tensor(vals[0]) else sum(vals)
        return self[key]

    def __contains__(self, key):
        return key in self._parts[0]

    def keys(self):
        return self._parts[0].keys()

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self._parts[0])

    def get(self, key, default=None):
        return self[key] if key in self else default


class TrainStep:
    """VolOpt.train_step (volsdf/vsdf.py:196-235) for one batch, on the HIP path end to end:
    forward -> MVS prior lookup -> fused loss (+ output gradients) -> compositing / MLP backward ->
    [gradient all-reduce] -> fused clip + guard + Adam.

    Ray groups (groups="auto").  Every stage before the weight-gradient reduction is local to a ray, so the batch can
    be processed as ray groups on concurrent HIP streams: the first group sized so that each of its fused-MLP launches
    fills the 256 CUs a whole number of times (1024 rays x 100 points = 800 workgroups = 3.125 rounds otherwise: the last
    round of every launch runs on 32 CUs), the rest on a second stream whose launches overlap with the first group's.
    Results do not depend on the grouping: train mode samples with fast = 1 (one sampler iteration, no batch-global
    convergence decision), every ray keeps its own random draws, the loss means are over the whole batch; only the
    float-atomic summation order of the weight gradients varies, as it does between any two runs.  Measured on MI355X
    (tools/ab_groups.py, interleaved A/B in one process): 5.05 ms/step ungrouped, 4.81 ms "auto", 4.78 ms two halves,
    4.94 ms three streams, 5.19 ms four (more launches than the host thread can enqueue ahead)."""

    def __init__(self, model, loss, lr=5e-4, grad_clip=True, world=1, rank=0, groups=None):
        from .train import MlpBackward, TrainStreams, WGradAccum
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
        dev = self.fp.flat.device
        self.is_bg = hasattr(model, "bg_implicit_network")      # VolSDFNetworkBG: fg + inverted-sphere background
        if self.is_bg:
            from .train import BgBackward
            self.bg_grad_out = [[(next(it), next(it)) for _ in range(n)] for n in (9, 2)]
            self.bg_bwd = BgBackward(dev)
        self.groups = groups
        self.tstreams = TrainStreams(dev)
        self.accum = WGradAccum(dev)
        self.prep = None
        self._new_bwd = lambda: MlpBackward(dev, self.tstreams, self.accum)
        self.bwd = [self._new_bwd()]                    # one scratch set and one stream per concurrent ray group
        self.sides = []
        self.d_beta = torch.zeros(8, device=dev)

    def samples_per_ray(self):
        rs = self.model.ray_sampler
        return rs.N_samples + rs.N_samples_extra + 2 - (1 if self.is_bg else 0)

    def check_batch(self, R):
        """The fused MLP kernels work on 32-point wave tiles and the ray samples of a batch (R x S points) must end on a
        tile boundary, where the eikonal points start: R x S % 32 == 0, i.e. R % 16 == 0 for the DTU model (S = 98) and
        R % 32 == 0 for the fg + background model (S = 97).  The reference has no such constraint; its configurations
        use 512 / 1024 / 2048 rays."""
        S = self.samples_per_ray()
        if R <= 0 or (R * S) % 32:
            import math
            raise ValueError(f"train.num_pixels = {R}: rays x samples ({R} x {S}) must be a multiple of 32 on the fused "
                             f"path -- use a multiple of {32 // math.gcd(S, 32)} rays")

    @staticmethod
    def split_rays(R, S, n_cu=256, wg_points=128):
        """[(lo,hi)] ray ranges: the first group's (S+2) points per ray (ray samples + 2 eikonal points) fill a whole
        number of rounds of n_cu workgroups; group sizes keep rays*S a multiple of 32."""
        per_ray = S + 2
        rounds = (R * per_ray) // (n_cu * wg_points)
        if rounds < 1:
            return [(0, R)]
        r1 = (rounds * n_cu * wg_points) // per_ray
        while r1 > 0 and ((r1 * S) % 32 or ((R - r1) * S) % 32):
            r1 -= 1
        if r1 <= 0 or r1 >= R:
            return [(0, R)]
        return [(0, r1), (r1, R)]

    def __call__(self, model_input, ground_truth, mvs=None, fast=1):
        """mvs: optional dict(views=[...], same_view=int, img_res=(H,W), inverse_depth=bool) for cost_mapping."""
        return self._step(model_input, ground_truth, mvs, fast)

    def _step(self, model_input, ground_truth, mvs=None, fast=1):
        from .train import finalize
        m = self.model
        m.train()
        uv = model_input["uv"]
        R = uv.shape[1]
        dev = uv.device
        S = self.samples_per_ray()
        self.check_batch(R)
        groups = self.split_rays(R, S) if self.groups == "auto" else (self.groups or [(0, R)])
        if len(groups) > self.d_beta.numel():
            raise ValueError("at most %d ray groups" % self.d_beta.numel())
        while len(self.bwd) < len(groups):
            self.bwd.append(self._new_bwd())
            self.sides.append(torch.cuda.Stream(device=dev))
        rng = m.draw_train_rng(R, dev)
        sdf_p, rgb_p = m.mlp_params()
        main = torch.cuda.current_stream()
        # Only the SDF forward streams are packed on the main stream (the sampler needs them first).  The radiance forward
        # stream and the weight streams of the BACKWARD kernels are packed on their own stream meanwhile; the forward
        # waits for the first event (recorded long before it gets there), the backward launches for the second.
        if self.prep is None:
            self.prep = torch.cuda.Stream(device=dev)
        self.prep.wait_stream(main)                      # parameters of this step are final, last step's readers are done
        pk = m.packed_mlp(rgb=False)                     # pack once, before the streams fork
        if self.is_bg:
            m.packed_bg()                                # (the group streams are ordered behind `fork`, not behind each other)
        with torch.cuda.stream(self.prep):
            m.rendering_network.pack_into(pk)
            rgb_packed = torch.cuda.Event(); rgb_packed.record(self.prep)
            self.tstreams.pack(sdf_p, rgb_p)
            if self.is_bg:
                bg_sdf_wb, bg_rgb_wb = m.bg_params()
                self.bg_bwd.pack(bg_sdf_wb, bg_rgb_wb)
            packed = torch.cuda.Event(); packed.record(self.prep)
        if self.is_bg:
            self.bg_bwd.zero()
        self.accum.zero()
        self.d_beta.zero_()
        main.wait_event(rgb_packed)
        fork = torch.cuda.Event(); fork.record(main)
        gt_rgb, gt_smooth = ground_truth["rgb"].reshape(-1, 3), ground_truth["rgb_smooth"].reshape(-1, 3)
        results, joins, holds = [], [], []
        for gi, (lo, hi) in enumerate(groups):
            stream = main if gi == 0 else self.sides[gi - 1]
            with torch.cuda.stream(stream):
                if gi:
                    stream.wait_event(fork)
                inp = dict(model_input)
                inp["uv"] = uv[:, lo:hi].contiguous()
                keep = {}
                out = m._forward_impl(inp, fast, keep, rng=m.slice_rng(rng, lo, hi))
                if mvs is not None:
                    out['pj'], out['pi'], _ = ops.cost_lookup(mvs["views"], mvs["same_view"], mvs["img_res"],
                                                              cam=keep["cam_loc"], dirs=keep["ray_dirs"], z=keep["z_vals"],
                                                              inverse_depth=mvs.get("inverse_depth", False))
                gt = {"rgb": gt_rgb[lo:hi], "rgb_smooth": gt_smooth[lo:hi]}
                lo_out = self.loss(out, gt, norm=(R * self.world, 2 * R * self.world), advance=(gi == len(groups) - 1))
                g = self.loss.last_grads
                stream.wait_event(packed)
                if self.is_bg:
                    d_sdf, d_rgb, d_bo, d_brgb, d_beta = ops.composite_bg_bwd(
                        keep["z_vals"], keep["z_max"], keep["sdf"], keep["rgb_flat"], keep["depth_scale"], m.density.beta,
                        m.density.beta_min_value, keep["z_bg"], keep["bg_out0"], keep["bg_rgb"], g["rgb_values"],
                        g["weights"], g["depth_values"])
                    self.bg_bwd.accumulate(keep, d_brgb, d_bo, slot=gi)
                else:
                    gw = m.white_bkgd_weight_grad(g["rgb_values"], g["weights"], keep["z_vals"].shape[1])
                    d_sdf, d_rgb, d_beta = ops.composite_bwd(keep["z_vals"], keep["sdf"], keep["rgb_flat"], keep["depth_scale"],
                                                             m.density.beta, m.density.beta_min_value, g["rgb_values"],
                                                             gw, g["depth_values"])
                self.d_beta[gi:gi + 1].copy_(d_beta)
                self.bwd[gi].accumulate(keep, d_rgb, d_sdf, g["grad_theta"])
                results.append((lo_out, out))
                holds.append((keep, g, d_sdf, d_rgb, inp, gt))
                if gi:
                    ev = torch.cuda.Event(); ev.record(stream); joins.append(ev)
        for ev in joins:
            main.wait_event(ev)
        finalize(self.accum, sdf_p, rgb_p, out=self.grad_out)
        if self.is_bg:
            self.bg_bwd.finalize(bg_sdf_wb, bg_rgb_wb, out=self.bg_grad_out)
        self.beta_grad.copy_(self.d_beta.sum())
        allreduce_flat_grad(self.fp.grad, self.world)
        self.opt.step()
        m.invalidate_packed()          # the fused kernel bypasses torch's version counters
        self._hold = holds
        self._results = results
        if len(results) == 1:
            return results[0]
        return _GroupedLosses(results), _GroupedOutputs(results)
