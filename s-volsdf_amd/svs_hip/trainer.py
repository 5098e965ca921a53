"""Fused training step around the HIP path: flat parameter / gradient buffers, one fused clip+guard+Adam
launch, and the single RCCL all-reduce of the flat gradient for ray-sharded data parallelism.

`FusedAdam` is a drop-in for the `torch.optim.Adam(model.parameters(), lr)` + `clip_grad_norm_` +
`on_after_backward` sequence of VolOpt.train_step (volsdf/vsdf.py:214-219); `TrainStep` is the whole
train_step (vsdf.py:196-235) without the dataset / logging plumbing, used by bench.py and the tests.
"""
import ctypes

import os

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
        self.step_dev = None          # device step counter while the step is replayed from a hipGraph (set by the capturer)

    def zero_grad(self, set_to_none=False):
        self.fp.grad.zero_()

    def step(self):
        L = _lib.load()
        self.step_count += 1
        _lib.check(L.svs_clip_guard_adam(_ptr(self.fp.flat), _ptr(self.fp.grad), _ptr(self.exp_avg), _ptr(self.exp_avg_sq),
                                         self.fp.n, self.step_count, _ptr(self.step_dev) if self.step_dev is not None else None,
                                         float(self.max_norm), float(self.lr), float(self.betas[0]), float(self.betas[1]),
                                         float(self.eps), _ptr(self.ws), _ptr(self.info), _stream()), "svs_clip_guard_adam")

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


def data_parallel(world):
    import torch.distributed as dist
    return world > 1 or (dist.is_available() and dist.is_initialized())


def grad_buckets(n_sdf, n):
    """The flat gradient as two contiguous buckets in the order they COMPLETE inside a step: [n_sdf, n) -- the radiance
    network, density.beta and (VolSDFNetworkBG) the background networks, whose weight gradients are final once the radiance
    GEMM launch / the background backward has retired, ~0.3 ms (1024 rays) before the step ends -- then [0, n_sdf): the SDF
    network, whose GEMM launch is the last of the step.  (The flat order is model._flat_param_list(): SDF layers, radiance
    layers, beta, background networks -- the reference's parameter order, which the checkpoints' Adam state follows.)"""
    return [(n_sdf, n), (0, n_sdf)]


def allreduce_range(flat_grad, lo, hi, async_op=False):
    """all-reduce (sum) of flat_grad[lo:hi] in place; -> the Work handle when async_op (its .wait() orders the CURRENT stream
    behind the collective).  A sum all-reduce acts element by element, so reducing a buffer in contiguous pieces gives every
    rank the same values as reducing it whole -- bit for bit with two ranks (a + b commutes), and with a ring of more ranks up
    to the order in which the ranks' contributions meet, which depends on where an element falls in the collective's chunks:
    the replicas stay identical among themselves either way (tests/test_dist_gloo.py::test_bucketed_allreduce_equals_single_world2)."""
    import torch.distributed as dist
    return dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, async_op=async_op)


# Experiment, OFF by default: in captured sequences the prior look-up as a branch beside the fused SDF / radiance launches and
# lin8's first-row gradient beside the SDF weight-gradient launch (51 us of small launches off a 256-ray step's critical
# chain).  Measured with launch plans, A/B twice on one box: 1.315 / 1.311 against 1.319 / 1.311 ms (DTU model, 256 rays), 1.55
# against 1.495 with the background model -- the branches' stream crossings and the fifth busy stream cost what they save.
_PLAN_BRANCHES = os.environ.get("SVS_PLAN_BRANCHES", "0") == "1"
_SMALL_GROUP_INLINE = os.environ.get("SVS_SMALL_GROUP_INLINE", "0") == "1"     # A/B: the small ray group's radiance weight gradients in line
# Two ray groups: the second (small) group only runs its sweeps; its weight-gradient jobs ride along in the first group's two
# launches (train.MlpBackward.accumulate, defer_wgrad / extra).  The small group's own launches cost a ring fill and a
# workgroup per CU each for 5 % of the points: 0.46 + 0.2 ms of kernel time per step beside the large group's (round 4).
_FOLD_WGRAD = os.environ.get("SVS_FOLD_WGRAD", "1") == "1"
_DP_BUCKETS = os.environ.get("SVS_DP_BUCKETS", "1") == "1"        # A/B: 0 = one all-reduce of the whole flat gradient at the end


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


class _ValidRays(dict):
    """Model outputs of a step whose batch was padded to a multiple of the kernels' ray granularity: per-ray tensors are cut
    back to the caller's rays on access (grad_theta: its two per-ray halves)."""

    def __init__(self, outputs, n_valid, n_padded):
        super().__init__()
        self._o, self._v, self._p = outputs, n_valid, n_padded

    def __missing__(self, key):
        t = self._o[key]
        if torch.is_tensor(t) and t.dim() > 0:
            if key == "grad_theta" and t.shape[0] == 2 * self._p:
                t = torch.cat([t[:self._v], t[self._p:self._p + self._v]], 0)
            elif t.shape[0] == self._p:
                t = t[:self._v]
            elif t.shape[0] % self._p == 0 and t.shape[0] > self._p:       # flattened (rays x samples, ...) tensors
                t = t.reshape(self._p, -1, *t.shape[1:])[:self._v].reshape(-1, *t.shape[1:])
        self[key] = t
        return t

    def __contains__(self, key):
        return key in self._o

    def keys(self):
        return self._o.keys()


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


class _Scratch:
    """Everything a step launches into besides the model's own buffers: packed training weight streams, weight-gradient
    accumulators, per-group backward scratch and the streams the groups run on.  The eager path owns one set; every
    captured graph owns its own (a graph replays into the addresses it was captured with, so its scratch must never be
    re-allocated by a step of another shape)."""

    def __init__(self, dev, is_bg):
        from .train import BgBackward, MlpBackward, TrainStreams, WGradAccum
        self.dev = dev
        self.tstreams = TrainStreams(dev)
        self.accum = WGradAccum(dev)
        self.prep = torch.cuda.Stream(device=dev)
        self._new_bwd = lambda: MlpBackward(dev, self.tstreams, self.accum)
        self.bwd = [self._new_bwd()]                    # one scratch set and one stream per concurrent ray group
        self.sides = []
        self.d_beta = torch.zeros(8, device=dev)
        self.bg_bwd = BgBackward(dev) if is_bg else None
        self._bg_streams = {}

    def lookup_stream(self):
        """stream of the prior look-up when it runs as a branch of a captured sequence"""
        if getattr(self, "_lookup", None) is None:
            self._lookup = torch.cuda.Stream(device=self.dev)
        return self._lookup

    def bg_stream(self, gi):
        """stream of ray group gi's background-network backward (runs beside the fg backward)"""
        if gi not in self._bg_streams:
            self._bg_streams[gi] = torch.cuda.Stream(device=self.dev)
        return self._bg_streams[gi]

    def for_groups(self, n):
        if n > self.d_beta.numel():
            raise ValueError("at most %d ray groups" % self.d_beta.numel())
        while len(self.bwd) < n:
            self.bwd.append(self._new_bwd())
            self.sides.append(torch.cuda.Stream(device=self.dev))


class _LaunchPlan:
    """The launch sequence of a captured step as a plan of the library (csrc/svs_plan.hip): the capture's nodes and edges
    read once, then enqueued per step by one call -- plain launches on the step's stream topology, no hipGraphLaunch, no
    interpreter between the launches (the call releases the GIL)."""

    def __init__(self, graph, side_streams=()):
        import ctypes
        from .lib import check, load
        self._lib, self._check = load(), check
        self.graph = graph                               # the kernel arguments live in the graph's nodes
        self.side_streams = list(side_streams)           # torch streams the side chains run on (kept alive here)
        arr = (ctypes.c_void_p * max(1, len(self.side_streams)))(*[s.cuda_stream for s in self.side_streams])
        handle = ctypes.c_void_p()
        check(self._lib.svs_plan_build(ctypes.c_void_p(int(graph.raw_cuda_graph())), arr, len(self.side_streams),
                                       ctypes.byref(handle)), "svs_plan_build")
        self.handle = handle
        counts = (ctypes.c_int * 8)()
        check(self._lib.svs_plan_info(handle, counts), "svs_plan_info")
        self.info = dict(zip(("nodes", "kernels", "copies", "memsets", "empty", "streams", "events", "entry_streams"),
                             list(counts)))

    def describe(self):
        """one line per node, in issue order (stream, kernel name and launch shape, events waited for / recorded)"""
        import ctypes
        buf = ctypes.create_string_buffer(1 << 18)
        self._check(self._lib.svs_plan_describe(self.handle, buf, len(buf)), "svs_plan_describe")
        return buf.value.decode()

    def run(self):
        self._check(self._lib.svs_plan_run(self.handle, torch.cuda.current_stream().cuda_stream), "svs_plan_run")

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                self._lib.svs_plan_destroy(h)
            except Exception:
                pass


class _CapturedStep:
    """One captured launch sequence (hipGraph) of the device part of a step, with the static tensors it reads."""

    def __init__(self):
        self.graph = None
        self.plan = None            # graph == "plan": the capture replayed as eager launches by the library
        self.static = {}            # name -> persistent device tensor (inputs, random draws, step-varying scalars)
        self.scratch = None
        self.result = None          # what the eager step would have returned: tensors inside the graph's pool
        self.hold = None
        self.calls = 0


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
    4.94 ms three streams, 5.19 ms four (more launches than the host thread can enqueue ahead).

    Whether the split pays depends on how the runtime maps the step's streams onto hardware queues (measured with
    GPU_MAX_HW_QUEUES = 2 ... 8: 3.0 ms/step at the default 4, 3.7 - 4.3 ms at any other value, against 3.1 ms unsplit at
    every value: with a queue of its own the small group's launches run truly concurrently and cost the large group a
    fourth round).  groups="auto" therefore MEASURES, once the process has warmed up (the first steps of a process are
    slow and noisy: allocator growth, clocks): from step TUNE_START on, 2 * TUNE_STEPS steps alternate between the split
    and the unsplit schedule (the first TUNE_SKIP of each untimed, whole steps timed with events on the step's stream), and
    the unsplit schedule is kept only if its median beats the split one by more than half the spread of the samples
    (`schedule`); until then the step runs split.

    Captured steps (graph=True or SVS_TRAIN_GRAPH=1; off by default: eager launches are faster on this ROCm stack,
    DESIGN.md section 5).  The launch sequence contains no host
    decision, so after one eager step per configuration (ray count, ray groups, model, MVS prior on / off) it can be captured
    once into a hipGraph and replayed: per step the host uploads the inputs (pixels, targets, camera, random draws and
    three scalars: rendered-view index, annealing state) into the graph's static tensors and launches the graph; the
    all-reduce and the fused optimiser launch stay outside it.  What a replay returns are views into the graph's memory:
    valid until the next step."""

    def __init__(self, model, loss, lr=5e-4, grad_clip=True, world=1, rank=0, groups=None, graph=None, shard_draws=False):
        import os
        self.model, self.loss = model, loss
        # world > 1 and shard_draws: every rank makes the train-mode draws of the WHOLE batch (world x local rays; the
        # ranks' host generators are in the same state) and keeps the rows of its own rays -- the sharded step then sees
        # exactly the draws the single-GPU step of the same batch sees.  Off: each rank draws for
        # its own rays (independent shards, bench.py's weak scaling).
        self.shard_draws = bool(shard_draws)
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
        n_sdf = sum(t.numel() for grp in self.grad_out[0] for t in grp if t is not None)
        self._buckets = grad_buckets(n_sdf, self.fp.n)
        self._early_work = None
        self._comm = None                                   # stream the early bucket is unpacked and reduced on
        dev = self.fp.flat.device
        self.is_bg = hasattr(model, "bg_implicit_network")      # VolSDFNetworkBG: fg + inverted-sphere background
        if self.is_bg:
            self.bg_grad_out = [[(next(it), next(it)) for _ in range(n)] for n in (9, 2)]
        # SVS_DETERMINISTIC=1 (svs_hip.lib.deterministic): the library sums the weight gradients in one fixed order, and the
        # step runs as ONE ray group on ONE stream from eager launches -- no measured choice between schedules, no launches
        # on concurrent streams adding into one accumulator: the same inputs give the same bits, run after run
        self.deterministic = _lib.deterministic()
        if self.deterministic:
            groups, graph = None, False
        self.groups = groups
        self.schedule = {}                              # groups == "auto": ray count -> dict(choice, ms_split, ms_whole)
        self._tune = {}
        self._force_groups = None
        self.scratch = _Scratch(dev, self.is_bg)
        if graph is None:
            graph = os.environ.get("SVS_TRAIN_GRAPH", "auto")
            graph = {"0": False, "off": False, "1": True}.get(graph, graph)
        if graph not in (False, True, "linear", "plan", "auto"):
            raise ValueError(f"graph / SVS_TRAIN_GRAPH = {graph!r}: one of 0 | 1 | linear | plan | auto")
        # False | True (hipGraph replay, the step's stream topology) | "linear" (hipGraph, one chain) | "plan" (the capture
        # read into a launch plan and enqueued as plain launches by the library: csrc/svs_plan.hip) | "auto" (plans for
        # single-group batches, eager launches otherwise)
        self.graph = graph
        self._captured = {}
        self._graph_pool = None

    # compatibility with code that reached into the former attributes
    @property
    def bwd(self):
        return self.scratch.bwd

    @property
    def accum(self):
        return self.scratch.accum

    def takes_host_inputs(self, R):
        """True when a batch of R rays runs from a captured sequence: the step then stages pixels, camera and targets itself
        (one transfer from its ring of pinned buffers), so a caller holding HOST tensors hands them over as they are."""
        if not self.graph:
            return False
        return self.graph != "auto" or self._auto_plans(R)

    def inputs_changed(self):
        """Tell the captured steps that a DEVICE-resident input (uv, intrinsics, pose, target colours) was modified in place by
        something torch does not see -- a kernel of this library or any other raw-pointer writer leaves `tensor._version`
        untouched.  A captured step copies a device input into its static buffer only when it is a new tensor object or
        its version moved (the bench's resident batch is copied once, not every step); host inputs are always staged anew,
        new device tensors and in-place torch ops are noticed by themselves.  Eager steps read the inputs directly."""
        for cs in self._captured.values():
            cs.static.pop("_seen", None)

    def _auto_plans(self, R):
        """graph == "auto": launch plans where the step is short enough for the host to matter -- batches of less than two
        rounds of 256 workgroups x 128 points (< 656 rays of the DTU model; config 4 over 8 GPUs runs 256 per GPU).  Measured
        (DESIGN.md section 5): the planned step is 4-15 % faster at 128 / 256 rays and keeps `VolOpt.run` at the bare step's
        time at 512 (2.25 against 2.6-3.3 ms: the eager step's 1.5 ms of enqueueing and the DataLoader share one interpreter);
        from 1024 rays on the device bounds the step either way and the eager launches are 1.5-2 % ahead."""
        return R * (self.samples_per_ray() + 2) < 2 * 256 * 128

    def samples_per_ray(self):
        rs = self.model.ray_sampler
        return rs.N_samples + rs.N_samples_extra + 2 - (1 if self.is_bg else 0)

    def ray_multiple(self):
        """The fused MLP kernels work on 32-point wave tiles and the ray samples of a launch (R x S points) must end on a
        tile boundary, where the eikonal points start: R x S % 32 == 0, i.e. R % 16 == 0 for the DTU model (S = 98) and
        R % 32 == 0 for the fg + background model (S = 97)."""
        import math
        return 32 // math.gcd(self.samples_per_ray(), 32)

    def check_batch(self, R):
        """Any ray count > 0 is accepted (as by the reference).  A count that is not a multiple of ray_multiple() is padded
        up to one by repeating the last ray (`_pad_batch`); the padding rays run through the kernels, are left out of the
        loss and contribute no gradient."""
        if R <= 0:
            raise ValueError(f"train.num_pixels = {R}")

    def _pad_batch(self, model_input, ground_truth):
        """-> (model_input, ground_truth, n_valid): inputs padded to a multiple of ray_multiple() rays, n_valid = the true
        ray count (== the padded count when nothing was added)."""
        R = model_input["uv"].shape[1]
        m = self.ray_multiple()
        pad = (-R) % m
        if pad == 0:
            return model_input, ground_truth, R
        rep = lambda t: torch.cat([t, t[:, -1:].expand(t.shape[0], pad, *t.shape[2:])], 1)
        mi = dict(model_input)
        mi["uv"] = rep(model_input["uv"])
        gt = {k: (rep(v) if torch.is_tensor(v) and v.dim() == 3 and v.shape[1] == R else v) for k, v in ground_truth.items()}
        return mi, gt, R

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
        model_input, ground_truth, n_valid = self._pad_batch(model_input, ground_truth)
        self._n_valid = n_valid
        out = self._step(model_input, ground_truth, mvs, fast)
        if n_valid == model_input["uv"].shape[1]:
            return out
        losses, outputs = out
        return losses, _ValidRays(outputs, n_valid, model_input["uv"].shape[1])

    TUNE_START, TUNE_STEPS, TUNE_SKIP = 24, 12, 2

    def _groups_for(self, R):
        groups = self._groups_raw(R)
        # a padded batch (up to ray_multiple() - 1 repeated rays at the end): no ray group may consist of padding only
        # (its loss would be a mean over zero rays); such a batch runs as one group
        if len(groups) > 1 and groups[-1][0] >= getattr(self, "_n_valid", R):
            return [(0, R)]
        return groups

    def _groups_raw(self, R):
        if self._force_groups is not None:
            return self._force_groups
        if self.groups != "auto":
            return self.groups or [(0, R)]
        sched = self.schedule.get(R)
        if sched is not None and sched["choice"] == "whole":
            return [(0, R)]
        return self.split_rays(R, self.samples_per_ray())

    def _tune_begin(self, R):
        """groups == "auto", eager steps: which schedule does this step run, and is it timed?  -> (state, mode, timed)"""
        if self.groups != "auto" or R in self.schedule:
            return None
        split = self.split_rays(R, self.samples_per_ray())
        if len(split) == 1:
            self.schedule[R] = dict(choice="whole", ms_split=None, ms_whole=None)
            return None
        st = self._tune.setdefault(R, dict(i=0, ev=[]))
        i = st["i"] - self.TUNE_START
        if i < 0:
            return st, None, None                             # not yet: split, untimed
        # the two schedules alternate step by step (clock drift, a preview render or a logging stall then hit both alike)
        mode = "split" if i % 2 == 0 else "whole"
        timed = i >= 2 * self.TUNE_SKIP
        self._force_groups = split if mode == "split" else [(0, R)]
        ev0 = None
        if timed:
            ev0 = torch.cuda.Event(enable_timing=True); ev0.record()
        return st, mode, ev0

    def _tune_end(self, R, tune):
        st, mode, ev0 = tune
        self._force_groups = None
        if ev0 is not None:
            ev1 = torch.cuda.Event(enable_timing=True); ev1.record()
            st["ev"].append((mode, ev0, ev1))
        st["i"] += 1
        if st["i"] == self.TUNE_START + 2 * self.TUNE_STEPS:
            torch.cuda.synchronize()
            ms = {"split": [], "whole": []}
            for mode, a, b in st["ev"]:
                ms[mode].append(a.elapsed_time(b))
            med = {k: sorted(v)[len(v) // 2] for k, v in ms.items()}
            spread = max(max(v) - min(v) for v in (sorted(x)[1:-1] or x for x in ms.values()))
            # within the noise of the samples the split schedule stays (it is the better one wherever a difference shows)
            choice = "whole" if med["whole"] < med["split"] - 0.5 * spread else "split"
            self.schedule[R] = dict(choice=choice, ms_split=med["split"], ms_whole=med["whole"], spread=spread)
            del self._tune[R]

    def _step(self, model_input, ground_truth, mvs=None, fast=1):
        m = self.model
        m.train()
        uv = model_input["uv"]
        R = uv.shape[1]
        self.check_batch(R)
        captured = bool(self.graph)
        if self.graph == "auto":
            captured = self._auto_plans(R)
        if captured:
            out = self._step_captured(model_input, ground_truth, mvs, fast)
            if out is not None:
                return self._finish(out)
        # (an eager step works on device tensors; a caller of the captured path may hand over the DataLoader's host tensors)
        dev = self.fp.flat.device
        if not uv.is_cuda:
            model_input = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in model_input.items()}
            uv = model_input["uv"]
        if not ground_truth["rgb"].is_cuda:
            ground_truth = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in ground_truth.items()}
        tune = None if captured else self._tune_begin(R)
        try:
            n_valid = getattr(self, "_n_valid", R)
            if self.world > 1 and self.shard_draws:
                rng = m.slice_rng(m.draw_train_rng(n_valid * self.world, uv.device, stream=self.scratch.prep),
                                  self.rank * n_valid, (self.rank + 1) * n_valid)
            else:
                rng = m.draw_train_rng(n_valid, uv.device, stream=self.scratch.prep)  # uploads on the (idle) pack stream
            if n_valid < R:                      # padded batch: the random stream is consumed as for the caller's rays
                from volsdf.model.network import pad_rng
                rng = pad_rng(rng, R)
            self._draws_done()
            gt = {"rgb": ground_truth["rgb"].reshape(-1, 3), "rgb_smooth": ground_truth["rgb_smooth"].reshape(-1, 3)}
            results, holds = self._device_step(self.scratch, model_input, gt, mvs, fast, rng, dyn=None, serial=self.deterministic)
            self._hold = holds
            out = self._finish(results)
        except BaseException:
            self._force_groups = None            # a failed step (e.g. an OOM the caller catches) must not pin the schedule
            raise
        if tune is not None:
            self._tune_end(R, tune)
        return out

    after_draws = None      # optional callable, invoked once per step right after the step's random draws have been made

    def _draws_done(self):
        """The step has consumed the CPU generator (sampler jitter, eikonal points, ...): from here on it only enqueues
        launches.  A caller that feeds the step from the reference's DataLoader loop starts drawing the NEXT batch now
        (volsdf/vsdf.py::VolOpt.run) -- the generator is used by one thread at a time, in the reference's order."""
        cb, self.after_draws = self.after_draws, None
        if cb is not None:
            cb()

    def _loss_on_valid(self, out, g_gt, v, Rg, norm, anneal_dev):
        """The loss of a ray group whose last Rg - v rays are padding: evaluated on the first v rays (and their 2 eikonal
        points each: rows [0,v) and [Rg, Rg+v) of grad_theta, network.py:258-266), gradients zero-padded to the group."""
        per_ray = ("rgb_values", "weights", "depth_values", "depth_values_all", "pi", "pj")
        ov = {k: (t[:v] if k in per_ray and torch.is_tensor(t) else t) for k, t in out.items()}
        gt_full = out.get("grad_theta")
        if gt_full is not None:
            ov["grad_theta"] = torch.cat([gt_full[:v], gt_full[Rg:Rg + v]], 0)
        gv = {k: t[:v] for k, t in g_gt.items()}
        lo_out = self.loss(ov, gv, norm=norm, advance=False, anneal_dev=anneal_dev)
        # (zero-padded with torch.cat: a slice assignment of contiguous rows is a device-to-device copy, which a captured
        # sequence cannot carry into a launch plan -- csrc/svs_plan.hip)
        g = {}
        for k, t in self.loss.last_grads.items():
            if t is None:
                g[k] = None
            elif k == "grad_theta":
                z = t.new_zeros((Rg - v,) + tuple(t.shape[1:]))
                g[k] = torch.cat([t[:v], z, t[v:], z], 0)
            else:
                g[k] = torch.cat([t, t.new_zeros((Rg - v,) + tuple(t.shape[1:]))], 0)
        return lo_out, g

    def _finish(self, results):
        """What follows the gradient: the one collective of a data-parallel step, the fused optimiser, host counters."""
        early, self._early_work = getattr(self, "_early_work", None), None
        if early is not None:
            # the early bucket (radiance / beta / background networks) is being reduced on the comm stream since its GEMM
            # launch retired; the SDF bucket is complete now; the optimiser waits for both
            lo, hi = self._buckets[1]
            allreduce_range(self.fp.grad, lo, hi)
            early.wait()
        else:
            allreduce_flat_grad(self.fp.grad, self.world)
        self.opt.step()
        self.model.invalidate_packed()          # the fused kernel bypasses torch's version counters
        self.loss.iter_step += 1
        self._results = results
        if len(results) == 1:
            return results[0]
        return _GroupedLosses(results), _GroupedOutputs(results)

    # ---- the device part of a step: everything between the uploaded inputs and the flat gradient ---------------------------
    def _device_step(self, sc, model_input, gt, mvs, fast, rng, dyn, serial=False):
        """Launches forward, prior lookup, loss and backward of every ray group and leaves d loss / d parameters (this
        rank's share, before the all-reduce) in the flat gradient.  No host synchronisation, no host decision that
        depends on device data: the sequence can be captured.  dyn: None, or dict(same_view=int32[1], anneal=float32[2])
        device tensors that carry the step-varying scalars of a captured sequence."""
        from .train import finalize
        m = self.model
        uv = model_input["uv"]
        R = uv.shape[1]
        dev = uv.device
        groups = [(0, R)] if serial else self._groups_for(R)
        sc.for_groups(len(groups))
        sdf_p, rgb_p = m.mlp_params()
        main = torch.cuda.current_stream()
        prep = main if serial else sc.prep
        # Only the SDF forward streams are packed on the main stream (the sampler needs them first).  The radiance forward
        # stream and the weight streams of the BACKWARD kernels are packed on their own stream meanwhile; the forward
        # waits for the first event (recorded long before it gets there), the backward launches for the second.
        # Stream topology (also what a capture records): every side stream forks from `main` or from a group stream and
        # joins `main` DIRECTLY.  A stream forked from a forked stream that joins its parent again makes
        # hipStreamEndCapture crash (ROCm 7.0 runtime of this torch build; tools/repro_hip_capture_nested_fork.py), so the radiance
        # weight-gradient stream of a group hands its completion event up to here instead of joining the group stream.
        if not serial:
            prep.wait_stream(main)                       # parameters of this step are final, last step's readers are done
        pk = m.packed_mlp(rgb=False)                     # pack once, before the streams fork
        with torch.cuda.stream(prep):
            bg_packed = None
            if self.is_bg:
                # the background networks' forward streams too (two rownorm + pack pairs: 30 us that sat on the origin
                # stream in front of the sampler); awaited in front of their first launch (`_before_bg`)
                m.packed_bg()
                bg_packed = torch.cuda.Event(); bg_packed.record(prep)
            m.rendering_network.pack_into(pk)
            rgb_packed = torch.cuda.Event(); rgb_packed.record(prep)
            sc.tstreams.pack(sdf_p, rgb_p)
            if self.is_bg:
                bg_sdf_wb, bg_rgb_wb = m.bg_params()
                sc.bg_bwd.pack(bg_sdf_wb, bg_rgb_wb)
            # the accumulators are zeroed here too: nothing adds into them before a stream has waited for `packed`
            if self.is_bg:
                sc.bg_bwd.zero()
            sc.accum.zero()
            packed = torch.cuda.Event(); packed.record(prep)
        # (the radiance forward stream is needed by the radiance launch only: every group's stream waits for `rgb_packed`
        # right in front of that launch -- the model calls the hook -- and the sampler and the fused SDF launch start ~20 us
        # earlier than when the origin stream waited here, in front of the fork)
        fork = torch.cuda.Event(); fork.record(main)
        results, joins, holds = [None] * len(groups), [], [None] * len(groups)     # (in group order whatever the enqueue order)
        # d loss / d beta of a group: one group writes it straight into the flat gradient, several into slots that are summed
        beta_out = (lambda gi: self.beta_grad.view(1)) if len(groups) == 1 else (lambda gi: sc.d_beta[gi:gi + 1])
        fold = _FOLD_WGRAD and len(groups) == 2 and not serial
        folded = None
        # (folded: the small group is enqueued FIRST, on its side stream, so that the events the large group's weight-gradient
        # launches wait for exist when those launches are enqueued)
        for gi, (lo, hi) in (list(enumerate(groups))[::-1] if fold else enumerate(groups)):
            stream = main if gi == 0 else sc.sides[gi - 1]
            with torch.cuda.stream(stream):
                if gi:
                    stream.wait_event(fork)
                inp = dict(model_input)
                inp["uv"] = uv[:, lo:hi].contiguous()
                inp["_skip_xyz"] = True              # the prior lookup below works from (cam, dirs, z): no (R,S,3) point list
                inp["_before_rgb"] = lambda stream=stream: stream.wait_event(rgb_packed)
                if bg_packed is not None:
                    inp["_before_bg"] = lambda: torch.cuda.current_stream().wait_event(bg_packed)
                keep = {}
                # (a capture tolerates the background forward's side stream only below the ORIGIN stream: the note above)
                m._side_ok_in_capture = gi == 0 and not serial
                # (experiment, SVS_PLAN_BRANCHES=1: the prior look-up -- it needs the sample depths only -- as a branch of a
                # captured sequence; the model calls the hook right after its sampler)
                looked_up = {}
                if mvs is not None and dyn is not None and m._side_ok_in_capture and _PLAN_BRANCHES:
                    def after_sampling(cam_loc, ray_dirs, z_vals, stream=stream):
                        ev = torch.cuda.Event(); ev.record(stream)
                        ls = sc.lookup_stream()
                        with torch.cuda.stream(ls):
                            ls.wait_event(ev)
                            looked_up["res"] = ops.cost_lookup(mvs["views"], mvs["same_view"], mvs["img_res"], cam=cam_loc,
                                                               dirs=ray_dirs, z=z_vals,
                                                               inverse_depth=mvs.get("inverse_depth", False),
                                                               same_view_dev=dyn["same_view"])
                            looked_up["join"] = torch.cuda.Event(); looked_up["join"].record(ls)
                    inp["_after_sampling"] = after_sampling
                out = m._forward_impl(inp, fast, keep, rng=m.slice_rng(rng, lo, hi))
                m._side_ok_in_capture = False
                if looked_up:
                    stream.wait_event(looked_up["join"])
                    out['pj'], out['pi'], _ = looked_up["res"]
                elif mvs is not None:
                    out['pj'], out['pi'], _ = ops.cost_lookup(mvs["views"], mvs["same_view"], mvs["img_res"],
                                                              cam=keep["cam_loc"], dirs=keep["ray_dirs"], z=keep["z_vals"],
                                                              inverse_depth=mvs.get("inverse_depth", False),
                                                              same_view_dev=dyn["same_view"] if dyn else None)
                g_gt = {"rgb": gt["rgb"][lo:hi], "rgb_smooth": gt["rgb_smooth"][lo:hi]}
                n_valid = getattr(self, "_n_valid", R)
                valid_g = max(0, min(hi, n_valid) - lo)          # rays of this group that are not padding
                if valid_g == hi - lo:
                    lo_out = self.loss(out, g_gt, norm=loss_norm(n_valid, self.world), advance=False,
                                       anneal_dev=dyn["anneal"] if dyn else None,
                                       grad_theta_out=sc.bwd[gi].grad_extra_out(keep["src"].n, keep["rgb_flat"].shape[0]))
                    g = self.loss.last_grads
                else:
                    lo_out, g = self._loss_on_valid(out, g_gt, valid_g, hi - lo, loss_norm(n_valid, self.world),
                                                    dyn["anneal"] if dyn else None)
                stream.wait_event(packed)
                if self.is_bg:
                    # the loss read depth_values_all (fg + bg, loss.py:72-73): its gradient enters as such
                    d_sdf, d_rgb, d_bo, d_brgb, d_beta = ops.composite_bg_bwd(
                        keep["z_vals"], keep["z_max"], keep["sdf"], keep["rgb_flat"], keep["depth_scale"], m.density.beta,
                        m.density.beta_min_value, keep["z_bg"], keep["bg_out0"], keep["bg_rgb"], g["rgb_values"],
                        g["weights"], None, d_depth_values_all=g["depth_values"], bg_depth=keep["bg_depth"],
                        d_sdf_out=sc.bwd[gi].sdf_grad_out(keep["src"].n, keep["rgb_flat"].shape[0]),
                        d_beta_out=beta_out(gi))
                    # The background networks' backward (radiance backward, pass B, weight gradients: three launches that
                    # depend on compositing's backward only) runs BESIDE the fg backward on a stream of its own -- at 256
                    # rays per GPU (config 4 over 8 GPUs) its 64 workgroups and the fg sweeps' 200 fit the chip together.
                    # Like the radiance weight-gradient stream it joins the ORIGIN stream, not its parent group stream.
                    if serial or os.environ.get("SVS_BG_SIDE", "1") == "0":
                        sc.bg_bwd.accumulate(keep, d_brgb, d_bo, slot=gi)
                    else:
                        # (a stream of its own: re-using the stream of the group's background FORWARD was measured slower at
                        # 1024 rays, 4.87 against 4.58 ms)
                        bs = sc.bg_stream(gi)
                        ev = torch.cuda.Event(); ev.record(stream)
                        with torch.cuda.stream(bs):
                            bs.wait_event(ev)
                            sc.bg_bwd.accumulate(keep, d_brgb, d_bo, slot=gi)
                            evj = torch.cuda.Event(); evj.record(bs); joins.append(evj)
                else:
                    gw = m.white_bkgd_weight_grad(g["rgb_values"], g["weights"], keep["z_vals"].shape[1])
                    d_sdf, d_rgb, d_beta = ops.composite_bwd(
                        keep["z_vals"], keep["sdf"], keep["rgb_flat"], keep["depth_scale"], m.density.beta,
                        m.density.beta_min_value, g["rgb_values"], gw, g["depth_values"],
                        d_sdf_out=sc.bwd[gi].sdf_grad_out(keep["src"].n, keep["rgb_flat"].shape[0]),
                        d_beta_out=beta_out(gi))
                if d_beta.data_ptr() != beta_out(gi).data_ptr():
                    beta_out(gi).copy_(d_beta)
                if fold and gi == 1:
                    folded = sc.bwd[gi].accumulate(keep, d_rgb, d_sdf, g["grad_theta"], wait=False, side=False, defer_wgrad=True)
                else:
                    joins.append(sc.bwd[gi].accumulate(keep, d_rgb, d_sdf, g["grad_theta"], wait=False,
                                                       side=not serial and not (gi and _SMALL_GROUP_INLINE),
                                                       extra=folded if (fold and gi == 0) else None))
                results[gi] = (lo_out, out)
                holds[gi] = (keep, g, d_sdf, d_rgb, inp, g_gt, folded)
                if gi:
                    ev = torch.cuda.Event(); ev.record(stream); joins.append(ev)
        # Data-parallel eager steps (SVS_DP_BUCKETS=0: one collective at the end): everything but the SDF network's gradients
        # is final once the radiance GEMM launch (and the background backward) has retired -- `joins` holds exactly those
        # events plus the small group's -- while pass A / pass B / the SDF GEMM still run on `main`.  A comm stream waits for
        # them, unpacks the radiance (and background) gradients into the flat gradient and starts the all-reduce of that
        # bucket; `_finish` reduces the SDF bucket and waits for both.  Captured sequences keep the single collective (the
        # capture ends at the flat gradient; a collective inside a launch plan is not something the plan builder replays).
        early = (dyn is None and not serial and _DP_BUCKETS and data_parallel(self.world)
                 and not torch.cuda.is_current_stream_capturing())
        if early:
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=dev)
            comm = self._comm
            with torch.cuda.stream(comm):
                for ev in joins:
                    comm.wait_event(ev)
                finalize(sc.accum, sdf_p, rgb_p, out=self.grad_out, nets=(1,))
                if self.is_bg:
                    sc.bg_bwd.finalize(bg_sdf_wb, bg_rgb_wb, out=self.bg_grad_out)
                if len(groups) > 1:
                    torch.sum(sc.d_beta[:len(groups)], dim=0, keepdim=True, out=self.beta_grad.view(1))
                lo, hi = self._buckets[0]
                self._early_work = allreduce_range(self.fp.grad, lo, hi, async_op=True)
        for ev in joins:
            main.wait_event(ev)
        finalize(sc.accum, sdf_p, rgb_p, out=self.grad_out, nets=(0,) if early else (0, 1))
        if self.is_bg and not early:
            sc.bg_bwd.finalize(bg_sdf_wb, bg_rgb_wb, out=self.bg_grad_out)
        if len(groups) == 1 or early:
            pass                                         # (written in place: d_beta_out above / summed on the comm stream)
        else:
            torch.sum(sc.d_beta[:len(groups)], dim=0, keepdim=True, out=self.beta_grad.view(1))
        return results, holds

    # ---- captured steps -----------------------------------------------------------------------------------------------------
    def _capture_key(self, model_input, mvs, fast):
        R = model_input["uv"].shape[1]
        mk = None
        if mvs is not None:
            # everything ops.cost_lookup hands to the kernel BY VALUE is baked into the capture: the cost / z range
            # addresses and shapes, and the camera parameters of every view (an MVS re-run can return a re-used address
            # with different cameras)
            def view_key(v):
                ptrs = tuple((int(v[k].data_ptr()), tuple(v[k].shape)) for k in ("cost", "z_mvs") if torch.is_tensor(v.get(k)))
                cams = []
                for k in sorted(v):
                    if k in ("cost", "z_mvs"):
                        continue
                    x = v[k]
                    if torch.is_tensor(x):
                        cams.append((k, int(x.data_ptr()), x._version, tuple(x.shape)))
                    else:
                        cams.append((k, repr(x)))
                return ptrs, tuple(cams)
            mk = (len(mvs["views"]), tuple(mvs["img_res"]), bool(mvs.get("inverse_depth", False)),
                  tuple(view_key(v) for v in mvs["views"]))
        return (R, getattr(self, "_n_valid", R), tuple(self._groups_for(R)), fast, mk, str(self.fp.flat.device), self.graph)

    def _upload(self, cs, model_input, ground_truth, mvs):
        """Host -> static tensors of a captured step, on the current stream (ordered before the replay).  Everything
        except the random draws -- pixels, camera, target colours, the two annealing scalars, the rendered-view index --
        has a fixed place in ONE static device buffer and travels in one transfer from a 4-deep ring of pinned staging
        buffers (the host waits for the transfer made FOUR steps ago, i.e. never in practice: with one staging buffer it
        waited for the previous step's, which sits behind that step's kernels -- host and GPU took turns).  An input that
        already lives on the device is copied into its place by a device copy behind the transfer."""
        st = cs.static
        dev = self.fp.flat.device
        annealed, anneal_sparse = self.loss.anneal_state()
        target = ground_truth["rgb_smooth"] if annealed else ground_truth["rgb"]
        origin = {k: model_input[k] for k in ("uv", "intrinsics", "pose")}
        origin["target"] = target
        pieces = [(k, model_input[k]) for k in ("uv", "intrinsics", "pose")] + [("target", target.reshape(-1, 3))]
        if "_all" not in st:
            off = 4                                          # words 0..1: annealing state, word 2: rendered-view index (int32)
            st["_layout"] = {}
            for k, src in pieces:
                st["_layout"][k] = (off, tuple(src.shape))
                off += (src.numel() + 3) // 4 * 4            # 16-byte aligned pieces
            st["_all"] = torch.zeros(off, dtype=torch.float32, device=dev)
            st["_ring"] = [dict(pin=torch.zeros(off, dtype=torch.float32).pin_memory(), ev=None) for _ in range(4)]
            st["_i"] = 0
            for k, (o, shape) in st["_layout"].items():
                n = 1
                for d in shape:
                    n *= d
                st[k] = st["_all"][o:o + n].view(shape)
            st["anneal"] = st["_all"][0:2]
            st["same_view"] = st["_all"][2:3].view(torch.int32)
            st["rng"] = {}
        slot = st["_ring"][st["_i"] % 4]
        st["_i"] += 1
        if slot["ev"] is not None:
            slot["ev"].synchronize()
        pin = slot["pin"]
        pin[0] = 1.0 if annealed else 0.0
        pin[1] = float(anneal_sparse)
        pin[2:3].view(torch.int32)[0] = int(mvs["same_view"]) if mvs is not None else -1
        on_device, host = [], False
        seen = st.setdefault("_seen", {})
        for k, src in pieces:
            o, shape = st["_layout"][k]
            if tuple(src.shape) != shape:
                raise ValueError(f"captured step: input {k} changed shape {shape} -> {tuple(src.shape)}")
            if src.is_cuda:
                # a device tensor the caller hands over unchanged step after step is in place: the SAME tensor object (kept
                # referenced here, so its storage cannot have been handed to another tensor) at the same version
                tag = (origin[k], origin[k]._version)
                old = seen.get(k)
                if old is None or old[0] is not tag[0] or old[1] != tag[1]:
                    on_device.append((k, src, tag))
            else:
                pin[o:o + src.numel()].copy_(src.reshape(-1))
                host = True
                seen.pop(k, None)
        head = (float(pin[0]), float(pin[1]), int(pin[2:3].view(torch.int32)[0]))
        if host:
            # (pieces that live on the device keep their place in the static buffer: the transfer writes their region of the
            # staging buffer -- stale -- over them, so they are copied again behind it)
            if any(src.is_cuda for _, src in pieces):
                on_device = [(k, src, (origin[k], origin[k]._version)) for k, src in pieces if src.is_cuda]
            ops.stage_in(st["_all"], pin)
        elif st.get("_head") != head:
            ops.stage_in(st["_all"][:4], pin[:4])                  # the three scalars only
        if host or st.get("_head") != head:
            slot["ev"] = torch.cuda.Event()
            slot["ev"].record()
            st["_head"] = head
        for k, src, tag in on_device:
            st[k].copy_(src, non_blocking=True)
            seen[k] = tag
        n_valid, n_pad = getattr(self, "_n_valid", model_input["uv"].shape[1]), model_input["uv"].shape[1]
        m = self.model
        sharded = self.world > 1 and self.shard_draws
        if sharded or n_valid < n_pad:
            # (as the eager step: a data-parallel rank draws for the whole batch and keeps its rays' rows; a padded batch
            # consumes the random stream of the caller's rays)
            if sharded:
                drawn = m.slice_rng(m.draw_train_rng(n_valid * self.world, dev), self.rank * n_valid, (self.rank + 1) * n_valid)
            else:
                drawn = m.draw_train_rng(n_valid, dev)
            if n_valid < n_pad:
                from volsdf.model.network import pad_rng
                drawn = pad_rng(drawn, n_pad)
            for k, v in drawn.items():
                if k.startswith("_"):
                    continue
                if k not in st["rng"]:
                    st["rng"][k] = torch.empty_like(v)
                st["rng"][k].copy_(v, non_blocking=True)
        else:
            m.draw_train_rng(n_pad, dev, out=st["rng"])

    def _step_captured(self, model_input, ground_truth, mvs, fast):
        """-> results of the step (replayed from its graph), or None when this call has to run eagerly: the first step of
        a configuration runs eagerly (it also performs the one-time kernel attribute set-up), the second is captured."""
        key = self._capture_key(model_input, mvs, fast)
        cs = self._captured.get(key)
        if cs is None:
            if len(self._captured) >= 4:                 # a few configurations at most (stages, render previews)
                # the evicted graph's result tensors may still be the caller's: it is destroyed one step later
                self._evicted = self._captured.pop(next(iter(self._captured)))
            cs = self._captured[key] = _CapturedStep()
        cs.calls += 1
        if cs.calls == 1:
            return None
        m = self.model
        self._upload(cs, model_input, ground_truth, mvs)
        self._draws_done()
        if cs.graph is None:
            st = cs.static
            cs.scratch = _Scratch(self.fp.flat.device, self.is_bg)
            inp = dict(model_input)
            inp.update(uv=st["uv"], intrinsics=st["intrinsics"], pose=st["pose"])
            gt = {"rgb": st["target"], "rgb_smooth": st["target"]}
            dyn = dict(same_view=st["same_view"], anneal=st["anneal"])
            # one eager pass over the capture's own scratch first: what a step allocates once and keeps (the backward's blocks,
            # zero-initialised: ~1 GB per 256 rays) must exist BEFORE the recording -- allocated inside it, the zero fills would
            # be recorded as launches and repeated by every replay (that, not the replay mechanism, was what made the captured
            # step of round 3 slower than the eager one)
            # -- on the stream the recording will run on: the model keeps per-stream workspaces (the sampler's, the side
            # streams of the background networks), which would otherwise be created, and zero-filled, inside the recording
            if getattr(self, "_capture_stream", None) is None:
                self._capture_stream = torch.cuda.Stream(device=self.fp.flat.device)
            cap = self._capture_stream
            cap.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cap):
                self._device_step(cs.scratch, inp, gt, mvs, fast, st["rng"], dyn, serial=self.graph == "linear")
            torch.cuda.current_stream().wait_stream(cap)
            m.invalidate_packed()                        # the capture must contain the weight packing
            plan_mode = self.graph in ("plan", "auto")
            # (keep_graph: the capture stays a hipGraph_t that svs_plan_build can read; it is never instantiated)
            graph = torch.cuda.CUDAGraph(keep_graph=True) if plan_mode else torch.cuda.CUDAGraph()
            if self._graph_pool is None:
                self._graph_pool = torch.cuda.graph_pool_handle()
            # (thread_local: a helper thread that prepares the next batch meanwhile -- VolOpt.run -- does not disturb the capture)
            # No cyclic garbage collection while the recording runs: a collection that finds an earlier TrainStep's
            # capture (a graph + its memory pool) would free device memory in the middle of this one, which the runtime
            # refuses -- from a destructor, i.e. the process aborts.  (torch.cuda.graph collects once on entry.)
            import gc
            gc_was_on = gc.isenabled()
            gc.disable()
            try:
                with torch.cuda.graph(graph, pool=self._graph_pool, stream=cap, capture_error_mode="thread_local"):
                    cs.result, cs.hold = self._device_step(cs.scratch, inp, gt, mvs, fast, st["rng"], dyn,
                                                           serial=self.graph == "linear")
            finally:
                if gc_was_on:
                    gc.enable()
            cs.graph = graph
            if plan_mode:
                # the side chains run on streams of the capture's own scratch (torch pool streams, as in the eager schedule)
                sc = cs.scratch
                side = ([sc.prep] + list(sc.sides) + [b._side for b in sc.bwd] + list(getattr(m, "_bg_streams", {}).values())
                        + list(sc._bg_streams.values()))
                try:
                    cs.plan = _LaunchPlan(graph, [x for x in side if x is not None])
                except _lib.SvsError as e:
                    # "auto" never costs a run: a sequence the plan builder refuses (a node type it cannot replay) is
                    # launched as the graph it is (hipGraphLaunch: same results, slower above ~500 rays)
                    if self.graph != "auto":
                        raise
                    import warnings
                    warnings.warn(f"launch plan refused, this configuration replays its hipGraph instead: {e}")
                    cs.plan = None
                    graph.instantiate()
                    if os.environ.get("SVS_PLAN_DEBUG") == "1":
                        import sys
                        print(f"launch plan refused: {e}", file=sys.stderr)
                if cs.plan is not None and os.environ.get("SVS_PLAN_DEBUG") == "1":
                    import sys
                    print(cs.plan.info, file=sys.stderr)
                    print(cs.plan.describe(), file=sys.stderr)
        if cs.plan is not None:
            cs.plan.run()
        else:
            cs.graph.replay()
        return cs.result


def loss_norm(n_rays_local, world):
    """Denominators (rays, eikonal points) of the loss means of ONE rank when a batch of world x n_rays_local rays is
    sharded over `world` ranks (and possibly processed as ray groups inside a rank): every rank's terms are normalised by
    the GLOBAL counts, so that the all-reduced (summed) gradient is the gradient of the reference's batch mean
    (volsdf/model/loss.py:43-46,50,67,78 take `.mean()` over all rays of the batch; the reference samples
    2 eikonal points per ray, network.py:258-266)."""
    return n_rays_local * world, 2 * n_rays_local * world
