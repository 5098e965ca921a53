"""`VolOpt` with the reference's call surface (volsdf/vsdf.py:18-463): constructor keywords `args, batch_size,
is_continue, timestamp, checkpoint, scan`; attributes `trains_i`, `train_dataset`, `stg`, `loss`, `plots_dir`,
`iter_step`; methods `gen_dataset(stg)`, `get_mvs_input(outs)`, `run(opt_stepN) -> epoch`, `render_mvs(id_k, epoch) ->
(depth (1,H,W) on the device, None)`, `train_step`, `render_step`, `cost_mapping`, `save_checkpoints`,
`load_from_dir` -- what runner.py:164-243 drives.

Everything per ray runs on the HIP path: `train_step` is one `svs_hip.trainer.TrainStep` (forward, MVS prior lookup,
fused loss, hand-written backward, fused clip + NaN guard + Adam), `render_step` is `svs_hip.renderer.render_image`
(8000-ray launches, per-chunk convergence on the device, no per-chunk device-to-host copies), `cost_mapping` is the
lookup kernel.  Around it stays host plumbing only: experiment folders, checkpoints in the reference's layout and key
names, the DataLoader, optional TensorBoard / plot hooks.  The dataset class is resolved from `train.dataset_class`
like the reference does (its `volsdf/datasets/` is outside this path: keep the reference checkout importable, see
INTEGRATION.md), or injected with the `dataset_class=` keyword.
"""
import copy
import itertools
import os
from datetime import datetime

import torch

import volsdf.utils.general as utils
from svs_hip import ops, renderer
from svs_hip.trainer import TrainStep, shard_rays
from volsdf.utils.conf import Conf, attr_view, to_plain


class _NoWriter:
    def add_scalar(self, *a, **k):
        pass

    def add_images(self, *a, **k):
        pass


def _summary_writer(log_dir):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir=log_dir)
    except Exception:                                    # tensorboard is optional host tooling
        return _NoWriter()


class AdamStateView:
    """`torch.optim.Adam`-layout `state_dict()` / `load_state_dict()` over the fused optimiser's flat moment buffers, so
    that OptimizerParameters/*.pth written by either implementation resumes in the other (vsdf.py:143-145,181-195).

    torch numbers optimiser state by the position of a parameter in `model.parameters()` -- for a weight-normed layer
    that is bias, weight_g, weight_v (registration order) -- while the flat buffers follow the kernels' order
    (weight_v, weight_g, bias).  The two are matched by parameter identity; shapes are checked before anything is
    copied."""

    def __init__(self, fused, model):
        self.fused = fused
        slot = {id(p): i for i, p in enumerate(fused.fp.params)}
        self._torch_order = [slot[id(p)] for p in model.parameters()]        # torch index -> flat-buffer view index
        if sorted(self._torch_order) != list(range(len(fused.fp.params))):
            raise ValueError("the fused optimiser does not cover exactly model.parameters()")

    def zero_grad(self, set_to_none=False):
        self.fused.zero_grad()

    def _moments(self):
        f = self.fused
        m, v = f.fp.views(f.exp_avg), f.fp.views(f.exp_avg_sq)
        return [(m[k], v[k]) for k in self._torch_order]

    def state_dict(self):
        f = self.fused
        state = {}
        if f.step_count > 0:
            for i, (m, v) in enumerate(self._moments()):
                state[i] = {"step": torch.tensor(float(f.step_count)), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
        group = dict(lr=f.lr, betas=tuple(f.betas), eps=f.eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                     capturable=False, differentiable=False, fused=None, params=list(range(len(self._torch_order))))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        f = self.fused
        moments = self._moments()
        for i, st in sd["state"].items():
            if not 0 <= int(i) < len(moments):
                raise ValueError(f"optimizer state for parameter {i}: the model has {len(moments)} parameters")
            for key, dst in zip(("exp_avg", "exp_avg_sq"), moments[int(i)]):
                if tuple(st[key].shape) != tuple(dst.shape):
                    raise ValueError(f"optimizer state {i}.{key} has shape {tuple(st[key].shape)}, parameter {i} of "
                                     f"model.parameters() has {tuple(dst.shape)}")
        steps = {int(s["step"]) for s in sd["state"].values()}
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ: not a state the fused Adam can represent")
        f.step_count = steps.pop() if steps else 0
        f.exp_avg.zero_(); f.exp_avg_sq.zero_()
        for i, st in sd["state"].items():
            m, v = moments[int(i)]
            m.copy_(st["exp_avg"]); v.copy_(st["exp_avg_sq"])
        if sd.get("param_groups"):
            f.lr = float(sd["param_groups"][0].get("lr", f.lr))


def init_data_parallel():
    """-> (world, rank, local_rank).  Data-parallel optimisation of ONE scan (BASELINE config 4: a 2048-ray batch sharded
    over the 8 GPUs of a node): launched as one process per GPU (`python -m torch.distributed.run --nproc-per-node N
    runner.py ...`), every process builds the same VolOpt; WORLD_SIZE / RANK / LOCAL_RANK come from the launcher.  Binds
    the process to its GPU BEFORE anything touches a device and joins the RCCL process group (backend "nccl"; gloo when
    there is no GPU, which is what the CPU tests use).  SVS_FORCE_DIST=1 joins a one-rank group (the RCCL path on a single
    GPU).  Without a launcher: (1, 0, 0) and no process group."""
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world <= 1 and os.environ.get("SVS_FORCE_DIST", "0") != "1":
        return 1, 0, 0
    has_gpu = torch.cuda.device_count() > 0
    if has_gpu and os.environ.get("SVS_DIST_SHARE_GPU", "0") == "1":
        # validation aid (tools/dev/dp_two_ranks.py): several ranks on ONE GPU, collectives over gloo -- exercises the whole
        # data-parallel path on a single-GPU box; RCCL itself refuses two ranks on one device
        local %= torch.cuda.device_count()
    if has_gpu:
        if local >= torch.cuda.device_count():
            raise RuntimeError(f"LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) are visible: launch at most "
                               f"one rank per GPU")
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(os.environ.get("SVS_DIST_BACKEND", "nccl" if has_gpu else "gloo"), rank=rank, world_size=world)
    return world, rank, local


def sync_host_rng(world):
    """Every rank continues from rank 0's host generators (torch CPU, Python `random`, numpy): the ranks then draw the same
    view, the same pixels and the same sampler / eikonal numbers at every step, of which each keeps its rays' share."""
    if world <= 1:
        return
    import random
    import numpy as np
    import torch.distributed as dist
    box = [(torch.get_rng_state(), random.getstate(), np.random.get_state()) if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    t, r, n = box[0]
    torch.set_rng_state(t)
    random.setstate(r)
    np.random.set_state(n)


_RESAMPLE_IS_REFERENCE = {}
_PREFIX_CHECK = {}


def randperm_prefix_matches_torch(n=442368, k=1024):
    """One self-check per process and size: `svs_randperm_prefix` rests on internals of THIS torch -- the serialised layout of
    the CPU generator (mt19937, 5056 bytes), ATen's forward Fisher-Yates `randperm` and its n < 2^32 / 20 branch (tested against
    torch 2.10; the reference pins 1.9).  On a private generator it must return torch.randperm(n)[:k] AND leave the generator
    where torch leaves it (state bytes and the next draw); otherwise the dataset's own method is used (warned once)."""
    hit = _PREFIX_CHECK.get((n, k))
    if hit is None:
        hit = False
        try:
            from svs_hip import lib as _lib
            g = torch.Generator()
            g.manual_seed(0x5EED5)
            torch.randint(0, 10, (3,), generator=g)                 # (not at a state boundary of the twister)
            state = g.get_state().clone()
            want = torch.randperm(n, generator=g)[:k]
            out = torch.empty(k, dtype=torch.int64)
            rc = _lib.load().svs_randperm_prefix(state.data_ptr(), state.numel(), n, k, out.data_ptr())
            if rc == 0 and torch.equal(out, want) and torch.equal(state, g.get_state()):
                g2 = torch.Generator()
                g2.set_state(state)
                hit = bool(torch.equal(torch.rand(5, generator=g2), torch.rand(5, generator=g)))
        except Exception:
            hit = False
        if not hit:
            import warnings
            warnings.warn("svs_randperm_prefix does not reproduce this torch's randperm (generator layout or shuffle changed): "
                          "pixel batches are drawn by the dataset's own change_sampling_idx")
        _PREFIX_CHECK[(n, k)] = hit
    return hit



def change_sampling_idx(dataset, sampling_size):
    """`dataset.change_sampling_idx(sampling_size)` (volsdf/vsdf.py:234 -> volsdf/datasets/scene_dataset.py:275-279).  Where the
    dataset's method is the reference's -- `self.sampling_idx = torch.randperm(self.total_pixels)[:sampling_size]`, recognised
    by its source text -- the same indices are drawn by `svs_randperm_prefix`: the first sampling_size iterations of torch's
    shuffle, the generator advanced over the rest (same batches, same generator state afterwards; 0.15 ms instead of the
    2-4 ms a shuffle of all 442 368 pixels of a 576 x 768 image takes -- more than a 256-ray step on the GPU).  Any other
    dataset class, sampling_size == -1 or SVS_FAST_RESAMPLE=0: the dataset's own method."""
    cls = type(dataset)
    ok = _RESAMPLE_IS_REFERENCE.get(cls)
    if ok is None:
        ok = False
        if os.environ.get("SVS_FAST_RESAMPLE", "1") != "0":
            try:
                import inspect
                src = "".join(inspect.getsource(cls.change_sampling_idx).split())
                ok = ("torch.randperm(self.total_pixels)[:sampling_size]" in src and src.count("torch.rand") == 1
                      and "self.sampling_idx=" in src)
            except Exception:
                ok = False
        _RESAMPLE_IS_REFERENCE[cls] = ok
    n = getattr(dataset, "total_pixels", None)
    try:
        n, sampling_size = int(n), int(sampling_size)
    except (TypeError, ValueError):
        ok = False
    if not ok or sampling_size == -1 or not 0 <= sampling_size <= n or not 1 <= n < (1 << 32) // 20 or \
            not randperm_prefix_matches_torch(n, sampling_size):
        return dataset.change_sampling_idx(sampling_size)
    from svs_hip import lib as _lib
    state = torch.get_rng_state()
    out = torch.empty(sampling_size, dtype=torch.int64)
    _lib.check(_lib.load().svs_randperm_prefix(state.data_ptr(), state.numel(), n, sampling_size, out.data_ptr()),
               "svs_randperm_prefix")
    torch.set_rng_state(state)
    dataset.sampling_idx = out


class VolOpt():
    def __init__(self, **kwargs):
        torch.set_default_dtype(torch.float32)
        torch.set_num_threads(1)
        # one process per GPU; a single process when no launcher set WORLD_SIZE (init_data_parallel)
        self.world, self.rank, self.local_rank = init_data_parallel()

        # get configs
        self.hparams = attr_view(copy.deepcopy(kwargs['args']))
        self.conf = Conf(to_plain(self.hparams['vol']))
        self.batch_size = kwargs['batch_size']
        if self.batch_size != 1:
            raise NotImplementedError("the fused step takes one view per batch (runner.py:165 passes batch_size=1)")
        self.exps_folder_name = self.hparams.exps_folder

        root = './'
        self.expname = self.conf.get_string('train.expname')
        kwargs_scan_id = int(kwargs['scan'][4:])
        scan_id = kwargs_scan_id if kwargs_scan_id != -1 else self.conf.get_int('dataset.scan_id', default=-1)
        self.scan_id = scan_id
        if scan_id != -1:
            self.expname = self.expname + '_{0}'.format(scan_id)

        is_continue, timestamp = kwargs['is_continue'], kwargs['timestamp']
        if is_continue and timestamp == 'latest':
            runs = os.path.join(root, self.exps_folder_name, self.expname)
            stamps = sorted(os.listdir(runs)) if os.path.exists(runs) else []
            is_continue, timestamp = (True, stamps[-1]) if stamps else (False, None)

        # experiment / checkpoint folders (the reference's layout)
        self.expdir = os.path.join(root, self.exps_folder_name, self.expname)
        self.timestamp = '{:%Y_%m_%d_%H_%M_%S}'.format(datetime.now())
        if self.world > 1:                               # one run folder: rank 0's time stamp
            import torch.distributed as dist
            box = [self.timestamp]
            dist.broadcast_object_list(box, src=0)
            self.timestamp = box[0]
        self.plots_dir = os.path.join(self.expdir, self.timestamp, 'plots')
        self.checkpoints_path = os.path.join(self.expdir, self.timestamp, 'checkpoints')
        self.model_params_subdir = "ModelParameters"
        self.optimizer_params_subdir = "OptimizerParameters"
        for d in (self.plots_dir, os.path.join(self.checkpoints_path, self.model_params_subdir),
                  os.path.join(self.checkpoints_path, self.optimizer_params_subdir)):
            os.makedirs(d, exist_ok=True)
        if self.rank == 0:
            self._save_run_config(os.path.join(self.expdir, self.timestamp, 'run.yaml'), kwargs['args'])

        # dataset config
        dataset_conf = dict(self.conf.get_config('dataset'))
        if kwargs_scan_id != -1:
            dataset_conf['scan_id'] = kwargs_scan_id
        dataset_conf['data_dir_root'] = self.hparams.data_dir_root
        assert [self.hparams.max_h, self.hparams.max_w] == list(dataset_conf['img_res'])
        self._dataset_class = kwargs.get('dataset_class') or utils.get_class(self.conf.get_string('train.dataset_class'))
        # opt-in: train batches drawn on the device instead of the reference's DataLoader loop (svs_hip/batches.py)
        self._device_batches = bool(kwargs.get('device_batches', os.environ.get('SVS_DEVICE_BATCHES', '0') == '1'))
        # the reference's DataLoader loop with the next batch prepared by a helper thread while the current step is being
        # enqueued: same batches, same random streams (_epoch_overlapped; pinned by test_overlapped_loader_draws_the_same_
        # batches).  Default since round 4 (4.7-5.3 -> 4.0-4.15 ms per step end to end: the dataset's ~2.6 ms of host work per
        # step -- torch.randperm over all pixels, the full pixel grid per item -- no longer sits in front of the step's
        # 1.7 ms of enqueueing); `overlap_loader=False` / SVS_OVERLAP_LOADER=0 gives the strictly sequential loop.
        self._overlap_loader = bool(kwargs.get('overlap_loader', os.environ.get('SVS_OVERLAP_LOADER', '1') == '1'))
        self._cached_items = bool(kwargs.get('cached_items', os.environ.get('SVS_CACHED_ITEMS', '1') == '1'))
        self._loader_pool = None

        # generate dataset
        self.data_confs = [copy.deepcopy(dataset_conf) for _ in range(3)]
        self.gen_dataset(stg=2)          # full resolution
        self.gen_plot_dataset()
        self.stg = 2
        self.ds_len = len(self.train_dataset)

        # model, loss, fused optimiser step
        self.model = utils.get_class(self.conf.get_string('train.model_class'))(conf=self.conf.get_config('model'))
        self.model.cuda()
        self.loss = utils.get_class(self.conf.get_string('train.loss_class'))(**self.conf.get_config('loss'))
        self.lr = self.conf.get_float('train.learning_rate')
        # data parallel: this rank's share of every batch, one all-reduce of the flat gradient per step (trainer.py)
        self.step_fn = TrainStep(self.model, self.loss, lr=self.lr, grad_clip=bool(self.hparams.grad_clip), groups="auto",
                                 world=self.world, rank=self.rank, shard_draws=True, graph=self._launch_mode())
        self.optimizer = AdamStateView(self.step_fn.opt, self.model)

        # load ckpt
        self.start_epoch = 0
        self.iter_step, self.total_step = 0, 0
        ckpt_dir = self.conf.get_string('train.ckpt_dir', '')
        if is_continue:
            self.load_from_dir(dir=os.path.join(self.expdir, timestamp), checkpoint=kwargs['checkpoint'])
        elif ckpt_dir != '':
            self.load_from_dir(dir=ckpt_dir, checkpoint='latest')

        # some parameters
        self.num_pixels = self.conf.get_int('train.num_pixels')
        self.step_fn.check_batch(self.num_pixels)
        self.plot_freq = self.conf.get_int('train.plot_freq')
        self.render_freq = self.conf.get_int('train.render_freq')
        self.checkpoint_freq = self.conf.get_int('train.checkpoint_freq', default=100)
        self.split_n_pixels = self.conf.get_int('train.split_n_pixels', default=10000)
        self.plot_conf = self.conf.get_config('plot')

        # logs, plots and checkpoints are rank 0's (all ranks hold identical replicas)
        self.writer = _summary_writer(os.path.join(self.plots_dir, 'logs')) if self.rank == 0 else _NoWriter()
        self.model.hparams = self.hparams
        self.loss.hparams = self.hparams
        sync_host_rng(self.world)

    @staticmethod
    def _save_run_config(path, args):
        try:
            from omegaconf import OmegaConf
            with open(path, "w") as f:
                OmegaConf.save(args, f)
        except Exception:
            import yaml
            with open(path, "w") as f:
                yaml.safe_dump(to_plain(args), f)

    # ---- checkpoints: the reference's files and keys (vsdf.py:128-195) ---------------------------------------------------
    def load_from_dir(self, dir, checkpoint='latest'):
        old = os.path.join(dir, 'checkpoints')
        saved = torch.load(os.path.join(old, 'ModelParameters', str(checkpoint) + ".pth"), map_location="cuda")
        self.model.load_state_dict(saved["model_state_dict"])
        self.model.invalidate_packed()
        self.start_epoch = saved['epoch']
        self.iter_step = saved['iter_step']
        data = torch.load(os.path.join(old, 'OptimizerParameters', str(checkpoint) + ".pth"), map_location="cuda")
        self.optimizer.load_state_dict(data["optimizer_state_dict"])

    def save_checkpoints(self, epoch, latest_only=False):
        if self.rank != 0:
            return 0
        for name in ("latest",) + (() if latest_only else (str(epoch),)):
            torch.save({"epoch": epoch, "model_state_dict": self.model.state_dict(), "iter_step": self.iter_step},
                       os.path.join(self.checkpoints_path, self.model_params_subdir, name + ".pth"))
            torch.save({"epoch": epoch, "optimizer_state_dict": self.optimizer.state_dict()},
                       os.path.join(self.checkpoints_path, self.optimizer_params_subdir, name + ".pth"))
        return 0

    # ---- datasets (host plumbing, vsdf.py:147-178) -------------------------------------------------------------------------
    def gen_plot_dataset(self):
        data_conf = copy.deepcopy(dict(self.conf.get_config('dataset')))
        data_conf['img_res'] = [int(_ / 4.) for _ in data_conf['img_res']]
        data_conf.setdefault('data_dir_root', self.hparams.data_dir_root)
        self.plot_dataset = self._dataset_class(**data_conf)
        self.plot_dataloader = torch.utils.data.DataLoader(self.plot_dataset, batch_size=self.conf.get_int('plot.plot_nimgs'),
                                                           shuffle=False, collate_fn=self.plot_dataset.collate_fn)

    def gen_dataset(self, stg):
        self.train_dataset = self._dataset_class(**self.data_confs[stg])
        # the train loader draws the dataset's items through svs_hip.batches.CachedItems: the same items and the same use of
        # the random generators, assembled from a pixel grid built once instead of per item (checked against the dataset's own
        # method view by view; SVS_CACHED_ITEMS=0: the dataset itself)
        items = self.train_dataset
        if self._cached_items:
            from svs_hip.batches import CachedItems
            items = self.train_items = CachedItems(self.train_dataset)
        self.train_dataloader = torch.utils.data.DataLoader(items, batch_size=self.batch_size, shuffle=True,
                                                            collate_fn=items.collate_fn)
        self.eval_dataloader = torch.utils.data.DataLoader(self.train_dataset, batch_size=1, shuffle=False,
                                                           collate_fn=self.train_dataset.collate_fn)
        self.total_pixels = self.train_dataset.total_pixels
        self.img_res = self.train_dataset.img_res
        self.scale_factor = self.train_dataset.scale_factor
        self.n_batches = len(self.train_dataloader)
        self.device_batches = None          # (re)built by run() for the current dataset when the option is on

    # ---- one optimisation step (vsdf.py:196-235) ----------------------------------------------------------------------------
    def _mvs_views(self, ts):
        """The prior of the current MVS stage as `ops.cost_lookup` takes it.  The per-view tensors are formed ONCE per stage
        (get_mvs_input): a step must see the same device tensors as the step before -- a captured step is keyed on their
        addresses (a new `.contiguous()` copy of the depth range per step would be a new configuration per step)."""
        cache = self.__dict__.get("_mvs_view_cache")
        key = (id(self.costs), id(self.z_mvs), tuple(int(x) for x in self.trains_i))
        if cache is None or cache[0] != key:
            views = []
            for i, id_k in enumerate(self.trains_i):
                z = self.z_mvs[i]
                views.append(dict(K=self.train_dataset.intrinsics_all[id_k], c2w=self.train_dataset.pose_all[id_k],
                                  cost=self.costs[i], z_near=z[0, 0].contiguous(), z_far=z[0, -1].contiguous()))
            cache = self._mvs_view_cache = (key, views, (self.costs, self.z_mvs))      # (the dicts kept alive: id() stays theirs)
        same = -1
        for i, id_k in enumerate(self.trains_i):
            if int(ts[0]) == int(id_k):
                same = i
        return dict(views=cache[1], same_view=same, img_res=tuple(self.train_dataset.img_res),
                    inverse_depth=bool(self.hparams.inverse_depth) and self.stg == 0)

    def _to_device(self, key, t):
        """`t.cuda()` for the small per-step host tensors of a batch, without its host synchronisation: a pageable `.cuda()`
        blocks the host until the stream has drained (the whole previous step), so the loop around the step could not run
        ahead of the GPU.  The values go through a 4-slot ring of pinned buffers and leave with a non-blocking copy."""
        if not torch.is_tensor(t) or t.is_cuda or not torch.cuda.is_available():
            return t.cuda() if torch.is_tensor(t) and not t.is_cuda else t
        rings = self.__dict__.setdefault("_h2d_rings", {})
        ring = rings.setdefault((key, tuple(t.shape), t.dtype), dict(i=0, slots=[None] * 4))
        j = ring["i"] % 4
        ring["i"] += 1
        slot = ring["slots"][j]
        if slot is None:
            slot = ring["slots"][j] = dict(pin=torch.empty(t.shape, dtype=t.dtype).pin_memory(), ev=None)
        if slot["ev"] is not None:
            slot["ev"].synchronize()                   # the copy that last read this slot (4 steps ago) is done
        slot["pin"].copy_(t)
        d = slot["pin"].to("cuda", non_blocking=True)
        slot["ev"] = torch.cuda.Event()
        slot["ev"].record()
        return d

    def _epoch_overlapped(self):
        """One pass `for batch in self.train_dataloader: self.train_step(batch)` with the SAME batches and the same use of the
        random generators, but with the dataset's work for step i+1 (`change_sampling_idx`: torch.randperm over all pixels;
        `__getitem__`: the full pixel grid, the gathers; the collate) done by a persistent helper thread while the main thread
        enqueues step i.  The order in which the generators are consumed is the reference's: step i's own draws (sampler
        jitter, eikonal points; made first thing in the step) -> randperm for batch i+1 -> random.randint of __getitem__ ->
        step i+1's draws ...; the helper's job is submitted by the step right after its draws and awaited before the next
        step, so the generators are never used by two threads at once.  (`overlap_loader=False` / SVS_OVERLAP_LOADER=0:
        strictly sequential.  SVS_OVERLAP_NEXT=main keeps only the randperm in the helper and runs the DataLoader's `next()`
        in the main thread after the step is enqueued -- A/B'd on one box (tools/dev/loop_ab.py): 4.38 / 4.11 ms against
        3.91 / 4.10 for the default and 4.72 / 4.88 sequential: no better.)"""
        from concurrent.futures import ThreadPoolExecutor
        if self._loader_pool is None:
            self._loader_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="svs-next-batch")
        it = iter(self.train_dataloader)
        batch = next(it, None)
        fetch_in_helper = os.environ.get("SVS_OVERLAP_NEXT", "helper") == "helper"    # A/B switch

        def resample():
            change_sampling_idx(self.train_dataset, self.num_pixels)
            return next(it, None) if fetch_in_helper else None

        while batch is not None:
            fut = []
            self.step_fn.after_draws = lambda: fut.append(self._loader_pool.submit(resample))
            try:
                self.train_step(batch, self.hparams.use_mvs, _resample=False)
            finally:
                self.step_fn.after_draws = None
                if not fut:                              # (a step that made no draws: nothing ran beside it)
                    fut.append(self._loader_pool.submit(resample))
                got = fut[0].result()                    # re-raises what the helper raised
            batch = got if fetch_in_helper else next(it, None)

    def _shard_batch(self, model_input, ground_truth):
        """This rank's contiguous share of the batch's rays (every rank holds the same batch: sync_host_rng)."""
        if self.world == 1:
            return model_input, ground_truth
        mi = dict(model_input)
        mi["uv"] = shard_rays(model_input["uv"], self.rank, self.world)
        R = model_input["uv"].shape[1]
        gt = {k: (shard_rays(v, self.rank, self.world) if torch.is_tensor(v) and v.dim() >= 2 and v.shape[1] == R else v)
              for k, v in ground_truth.items()}
        return mi, gt

    def _launch_mode(self):
        """None: TrainStep's default (SVS_TRAIN_GRAPH, "auto": launch plans for short steps).  With the opt-in device-side
        batch source every step hands over NEW device tensors, which a planned step would copy into its static inputs one by
        one (measured slower than the eager step: 1.64 against 1.48 ms at 256 rays): eager launches there."""
        if getattr(self, "_device_batches", False):
            return False
        return None

    def _upload_batch(self, model_input, ground_truth):
        """The host tensors of a batch (pixels, camera, target colours: ~30 KB) to the device in ONE copy: packed into a slot
        of a 4-deep ring of pinned staging buffers and sent with a single non-blocking transfer into the slot's device
        buffer; the returned tensors are views of it (valid until the slot comes round again, four steps later).  Five
        separate transfers were five dependent DMA operations in front of every step's first kernel."""
        names = [("in", k) for k, v in model_input.items() if torch.is_tensor(v)] + [("gt", k) for k in ("rgb", "rgb_smooth")]
        src = [model_input[k] if w == "in" else ground_truth[k] for w, k in names]
        if (not torch.cuda.is_available() or any(t.is_cuda or t.dtype != torch.float32 for t in src)):
            mi = {k: self._to_device("in." + k, v) for k, v in model_input.items()}
            return mi, {k: self._to_device("gt." + k, ground_truth[k]) for k in ("rgb", "rgb_smooth")}
        key = tuple((w, k, tuple(t.shape)) for (w, k), t in zip(names, src))
        ring = self.__dict__.setdefault("_batch_ring", {})
        st = ring.get(key)
        if st is None:
            if len(ring) >= 8:
                ring.clear()
            total = sum(t.numel() for t in src)
            st = ring[key] = dict(i=0, slots=[dict(pin=torch.empty(total, dtype=torch.float32).pin_memory(),
                                                   dev=torch.empty(total, dtype=torch.float32, device="cuda"), ev=None)
                                              for _ in range(4)])
        slot = st["slots"][st["i"] % 4]
        st["i"] += 1
        if slot["ev"] is not None:
            slot["ev"].synchronize()                   # the transfer that last read this slot's staging buffer is done
        off, views = 0, []
        for t in src:
            n = t.numel()
            slot["pin"][off:off + n].copy_(t.reshape(-1))
            views.append(slot["dev"][off:off + n].view(t.shape))
            off += n
        ops.stage_in(slot["dev"], slot["pin"])
        slot["ev"] = torch.cuda.Event()
        slot["ev"].record()
        mi = {k: v for k, v in model_input.items() if not torch.is_tensor(v)}
        gt = {}
        for (w, k), v in zip(names, views):
            (mi if w == "in" else gt)[k] = v
        return mi, gt

    def train_step(self, batch, use_mvs=False, _resample=True):
        indices, model_input, ground_truth = batch
        model_input, ground_truth = self._shard_batch(model_input, ground_truth)
        if self.step_fn.takes_host_inputs(model_input["uv"].shape[1]):
            # (a planned step stages its inputs itself: the loader's host tensors go in as they are)
            model_input, gt = dict(model_input), {k: ground_truth[k] for k in ("rgb", "rgb_smooth")}
        else:
            model_input, gt = self._upload_batch(model_input, ground_truth)
        model_input['iter_step'] = self.iter_step
        if use_mvs and bool(self.hparams.inverse_depth) and self.stg >= 1:
            raise NotImplementedError                                      # vsdf.py:429-430
        loss_output, model_outputs = self.step_fn(model_input, gt, mvs=self._mvs_views(indices) if use_mvs else None, fast=1)
        if self.total_step % 50 == 0:
            mse = torch.mean((model_outputs['rgb_values'] - gt['rgb'].reshape(-1, 3).to(model_outputs['rgb_values'].device)) ** 2)
            if self.world > 1:
                # a rank's loss terms are its rays' share of the batch means (trainer.loss_norm): their sum over the ranks
                # is the batch's loss; the mse is a mean per rank
                import torch.distributed as dist
                keys = sorted(loss_output.keys())
                vec = torch.stack([torch.as_tensor(loss_output[k], dtype=torch.float32, device=mse.device).reshape(())
                                   for k in keys] + [mse / self.world])
                dist.all_reduce(vec)
                loss_output = dict(loss_output)
                loss_output.update({k: vec[i] for i, k in enumerate(keys)})
                mse = vec[-1]
            for k, v in loss_output.items():
                self.writer.add_scalar('t/' + k, v, self.total_step)
            beta = self.model.density.get_beta().item()
            self.writer.add_scalar('t/beta', beta, self.total_step)
            self.writer.add_scalar('t/alpha', 1. / beta, self.total_step)
            self.writer.add_scalar('t/psnr', (-10. * torch.log10(mse)).item(), self.total_step)
        if self.device_batches is None and _resample:
            change_sampling_idx(self.train_dataset, self.num_pixels)
        self.iter_step += 1
        self.total_step += 1
        return loss_output

    # ---- whole-image rendering (vsdf.py:237-287) ----------------------------------------------------------------------------------
    def render_step(self, batch, epoch, dataset, fast=-1):
        self.model.eval()
        indices, model_input, ground_truth = batch
        model_input = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in model_input.items()}
        model_input['iter_step'] = self.iter_step
        # data parallel: whole split_n_pixels chunks per rank, the image assembled with one all-gather per output
        model_outputs = renderer.render_image(self.model, model_input, dataset.total_pixels, split_n_pixels=self.split_n_pixels,
                                              fast=fast, rank=self.rank, world=self.world)
        # (1,H,W): the reference's lin2img(...)[0] keeps the channel axis, which the next MVS stage reads as the batch axis
        depth_cuda = renderer.depth_image(model_outputs, dataset.img_res, self.scale_factor)[None]
        mask_bin = ground_truth['mask'].reshape(-1, 3).cuda() == 1.
        mse = torch.mean((model_outputs['rgb_values'] - ground_truth['rgb'].reshape(-1, 3).cuda())[mask_bin] ** 2)
        self.last_val_psnr = -10. * torch.log10(mse)
        self.writer.add_scalar('val/psnr', self.last_val_psnr.item(), self.total_step)
        if self.rank == 0:
            self._plot(indices, model_input, model_outputs, ground_truth, epoch, dataset, mask_bin)
        self.total_step += 1
        return depth_cuda, None

    def get_plot_data(self, model_input, model_outputs, pose, rgb_gt):
        batch_size, num_samples, _ = rgb_gt.shape
        return {'rgb_gt': rgb_gt, 'pose': pose,
                'rgb_eval': model_outputs['rgb_values'].reshape(batch_size, num_samples, 3),
                'normal_map': (model_outputs['normal_map'].reshape(batch_size, num_samples, 3) + 1.) / 2.,
                'depth_map': model_outputs['depth_values'].reshape(batch_size, num_samples),
                'acc': model_outputs['weights'].sum(1).reshape(batch_size, num_samples)}

    def _plot(self, indices, model_input, model_outputs, ground_truth, epoch, dataset, mask_bin):
        """The reference's image dumps (volsdf/utils/plots.py, out of the path) if that module is importable."""
        try:
            import volsdf.utils.plots as plt
        except Exception:
            return
        cpu = {k: v.detach().cpu() for k, v in model_outputs.items()}
        plot_data = self.get_plot_data(model_input, cpu, model_input['pose'].cpu(), ground_truth['rgb'])
        stack = plt.stacked_plot(indices, plot_data, self.plots_dir, epoch, dataset.img_res, **self.plot_conf)
        stack[0][~mask_bin.cpu().reshape(list(dataset.img_res) + [3, ]).permute(2, 0, 1)] = 0
        self.writer.add_images('val/vis', torch.stack(stack, dim=0), self.total_step)

    def render_mvs(self, id_k, epoch):
        """Full-resolution depth of training view number `id_k` (position in the eval loader) -> ((1,H,W) on the device, None)."""
        ds = self.train_dataset
        ds.mode = 'test'
        try:
            ds.change_sampling_idx(-1)
            batch = next(itertools.islice(self.eval_dataloader, id_k, None))
            return self.render_step(batch, epoch, ds, fast=-1)
        finally:
            ds.mode = 'train'

    # ---- optimisation loop (vsdf.py:321-367) ----------------------------------------------------------------------------------------
    def _epochs(self, first, opt_stepN):
        """Epoch numbers first, first+1, ... ; stops after the epoch that carries the step count past `opt_stepN` steps
        beyond where this call started (the reference tests `>` after a whole pass over the views, vsdf.py:357)."""
        begin = self.iter_step
        for epoch in itertools.count(first):
            yield epoch, self.iter_step - begin
            if self.iter_step - begin > opt_stepN:
                return

    def _preview(self, epoch):
        """Low-resolution render of the first plot view + a `latest` checkpoint (vsdf.py:339-347)."""
        ds = self.plot_dataset
        ds.change_sampling_idx(-1)
        ds.mode = 'plot'
        batch = next(iter(self.plot_dataloader))
        ds.mode = 'train'
        self.render_step(batch, epoch, ds, fast=-1)
        self.save_checkpoints(epoch, latest_only=True)

    def run(self, opt_stepN=1e8):
        """Optimise for at least `opt_stepN` further steps in whole passes over the training views -> last epoch number.
        Same schedule as the reference: numbered checkpoint every `checkpoint_freq` epochs and at the end, preview
        every `render_freq` epochs and, during the first 6000 steps of the call, every 1000/len(views) epochs."""
        early_every = max(20 * 50 // self.ds_len, 1)
        epoch = self.start_epoch
        for epoch, done in self._epochs(self.start_epoch, opt_stepN):
            if epoch % self.checkpoint_freq == 0:
                self.save_checkpoints(epoch)
            if epoch % self.render_freq == 0 or (done <= 120 * 50 and epoch % early_every == 0):
                self._preview(epoch)
            if self._device_batches and self.world > 1:
                raise NotImplementedError("device_batches draws on the device per process; the data-parallel ranks need the "
                                          "same batch: use the DataLoader loop (default) or overlap_loader")
            if self._device_batches and self.device_batches is None and torch.cuda.is_available():
                from svs_hip.batches import DeviceBatches
                self.device_batches = DeviceBatches(self.train_dataset, self.num_pixels,
                                                    torch.device("cuda", torch.cuda.current_device()))
            if self.device_batches is not None:          # opt-in: batches drawn on the device (svs_hip/batches.py)
                for batch in self.device_batches:
                    self.train_step(batch, self.hparams.use_mvs)
                continue
            change_sampling_idx(self.train_dataset, self.num_pixels)
            if self._overlap_loader:
                self._epoch_overlapped()
                continue
            for batch in self.train_dataloader:
                self.train_step(batch, self.hparams.use_mvs)
        self.save_checkpoints(epoch)
        self.start_epoch = epoch
        return epoch

    # ---- MVS prior (vsdf.py:369-452) ---------------------------------------------------------------------------------------------------
    def get_mvs_input(self, outs_samples):
        # the previous stage's volumes go first: nothing may keep them on the GPU while this stage's are built (the per-view
        # dicts of _mvs_views and the look-up's constant cache hold references to them)
        self.__dict__.pop("_mvs_view_cache", None)
        ops.clear_lookup_caches()
        self.costs, self.z_mvs, self.bd_mvs = dict(), dict(), dict()
        sphere = self.conf.get_float('model.scene_bounding_sphere')
        for i in range(len(outs_samples)):
            prob_volume = outs_samples[i]['prob_volume']
            depth_values = outs_samples[i]['depth_values'] / self.scale_factor
            self.costs[i] = prob_volume.cuda()
            self.z_mvs[i] = depth_values.cuda()
            bd = depth_values[:, [0, -1], :, :]
            bd[:, 0, :, :] = torch.minimum(bd[:, 0, :, :], torch.ones_like(bd[:, 0, :, :]) * sphere)
            self.bd_mvs[i] = bd

    @torch.no_grad()
    def cost_mapping(self, z_vals, ts, xyz_raw):
        """-> (results_cost_j, results_cost_mvs, valid_mask), each (N_rays, N_samples)."""
        m = self._mvs_views(ts)
        return ops.cost_lookup(m["views"], m["same_view"], m["img_res"], xyz=xyz_raw, inverse_depth=m["inverse_depth"])

    def on_after_backward(self) -> None:
        """The NaN / Inf guard of vsdf.py:454-463 lives inside the fused clip + Adam launch (svs_clip_guard_adam): a
        gradient that is not finite is ZEROED and the Adam step still runs -- moments decay, parameters move by their
        momentum, the step count advances -- which is what the reference's `optimizer.zero_grad()` followed by
        `optimizer.step()` does under its pinned torch 1.9 (zero_grad leaves zero tensors, not None)."""
        return None
