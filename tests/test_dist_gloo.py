"""world_size-2 gloo test (CPU) of the data-parallel host logic: ray sharding, loss scaling by 1/world, flat parameter /
gradient buffers and the single all-reduce of the flat gradient (what bench.py --gpus N / TrainStep do over RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(2, 16), torch.nn.Softplus(beta=100), torch.nn.Linear(16, 3))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "s-volsdf_amd"))
    import torch.distributed as dist
    from svs_hip.trainer import FlatParams, allreduce_flat_grad, shard_rays
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _toy_model()
    fp = FlatParams(list(model.parameters()))
    uv = torch.arange(64, dtype=torch.float32).reshape(1, 32, 2) / 10.0           # the same global batch on every rank
    gt = torch.linspace(0, 1, 96).reshape(32, 3)
    k = 32 // world
    mine = shard_rays(uv, rank, world)
    assert mine.shape == (1, k, 2)
    fp.grad.zero_()
    out = model(mine[0])
    # each rank's mean is over its shard; dividing by world makes the all-reduced sum the global-batch mean
    loss = (out - gt[rank * k:(rank + 1) * k]).abs().mean() / world
    loss.backward()
    assert all(p.grad.data_ptr() >= fp.grad.data_ptr() for p in model.parameters()), "grads live in the flat buffer"
    allreduce_flat_grad(fp.grad, world)
    q.put((rank, fp.grad.clone().numpy(), fp.flat.clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_gradient_equals_full_batch():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the whole batch
    model = _toy_model()
    uv = torch.arange(64, dtype=torch.float32).reshape(1, 32, 2) / 10.0
    gt = torch.linspace(0, 1, 96).reshape(32, 3)
    (model(uv[0]) - gt).abs().mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).numpy()
    for rank, grad, flat in res:
        np.testing.assert_allclose(grad, ref, rtol=1e-5, atol=1e-7)
    np.testing.assert_array_equal(res[0][2], res[1][2])       # replicas stay identical


# --------------------------------------------------------------------------------------------------------------
# the step's own data-parallel arithmetic: loss normalisation by the GLOBAL ray / eikonal counts (trainer.loss_norm),
# ray shards of the pixel batch and of the host-drawn random numbers (shard_rays, slice_rng), one all-reduce (sum) of the
# flat gradient, identical optimiser step on every rank -- driven with the real model math (oracle/torch_ref.py
# restates VolSDFNetwork + VolSDFLoss with autograd; the HIP kernels cannot run here)
# --------------------------------------------------------------------------------------------------------------
R_GLOBAL, IT = 16, 50          # iteration 50 < anneal_rgb: every loss term is live (masked rgb, eikonal, MVS, sparse)


def _step_inputs():
    sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "s-volsdf_amd")]
    import synth
    K, pose = synth.make_camera()
    uv = synth.make_uv(R_GLOBAL, seed=3)
    rng = synth.make_train_rng(R_GLOBAL, seed=5)
    rs = np.random.default_rng(9)
    gt = rs.uniform(0, 1, (R_GLOBAL, 3)).astype(np.float32)
    gts = rs.uniform(0, 1, (R_GLOBAL, 3)).astype(np.float32)
    return synth.make_params(0), K, pose, uv, rng, gt, gts, synth.make_mvs_views(2)


def _shard_gradient(params, K, pose, uv, rng, gt, gts, views, lo, hi, norm):
    """Flat float32 gradient of the loss of rays [lo, hi) normalised by `norm` -> (FlatParams, z of the shard)."""
    import svs_oracle as orc
    import torch_ref as tref
    from svs_hip.trainer import FlatParams
    p = tref.to_torch(params, torch.float32)
    names = sorted(k for k in p if k != "density.beta") + ["density.beta"]
    fp = FlatParams([p[k] for k in names])
    dirs, cam, ds = orc.rays_from_uv(uv[lo:hi], pose, K)
    layers = orc.effective_weights(params, "implicit_network", 9)
    z, z_eik = orc.error_bound_sampler(lambda x: orc.sdf_vals(layers, x), dirs, cam, orc.get_beta(params["density.beta"]),
                                       fast=1, training=True, rng=rng)
    eik = np.concatenate([rng["eik_points"], (cam[None] + z_eik * dirs).astype(np.float32)], 0)
    out = tref.forward_differentiable(p, cam, dirs, z, eik, ds)
    xyz = (cam[None, None] + z[:, :, None] * dirs[:, None, :]).astype(np.float32)
    pj, pi, _ = orc.cost_mapping(xyz, 0, views, (576, 768))
    out["pj"], out["pi"] = torch.from_numpy(pj), torch.from_numpy(pi)
    total = tref.loss_fn(out, torch.from_numpy(gt[lo:hi]), torch.from_numpy(gts[lo:hi]), IT, norm=norm)
    fp.grad.zero_()
    total.backward()
    return fp, z, float(total.detach())


def _step_worker(rank, world, port, q):
    import torch.distributed as dist
    params, K, pose, uv, rng, gt, gts, views = _step_inputs()
    from svs_hip.trainer import allreduce_flat_grad, loss_norm, shard_rays
    from volsdf.model.network import VolSDFNetwork
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    k = R_GLOBAL // world
    mine = shard_rays(torch.from_numpy(uv)[None], rank, world)              # this rank's pixels ...
    assert np.array_equal(mine[0].numpy(), uv[rank * k:(rank + 1) * k])
    rng_t = {n: torch.from_numpy(np.ascontiguousarray(v)) for n, v in rng.items()}
    mine_rng = {n: v.numpy() for n, v in VolSDFNetwork.slice_rng(rng_t, rank * k, (rank + 1) * k).items()}   # ... and draws
    fp, z, total = _shard_gradient(params, K, pose, mine[0].numpy(), mine_rng, gt[rank * k:(rank + 1) * k],
                                   gts[rank * k:(rank + 1) * k], views, 0, k, loss_norm(k, world))
    local = fp.grad.clone()
    allreduce_flat_grad(fp.grad, world)                                       # THE collective of the step
    reduced = fp.grad.clone()
    opt = torch.optim.Adam(fp.params, lr=5e-4)
    torch.nn.utils.clip_grad_norm_(fp.params, 1.0)
    opt.step()
    q.put((rank, local.numpy(), reduced.numpy(), fp.flat.clone().numpy(), z, total))
    dist.barrier()
    dist.destroy_process_group()


def test_train_step_arithmetic_world2():
    """Two ranks x 8 rays == one process x 16 rays: summed shard losses, all-reduced gradient, parameters after the step."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_step_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    params, K, pose, uv, rng, gt, gts, views = _step_inputs()
    from svs_hip.trainer import loss_norm
    assert loss_norm(8, 2) == (16, 32) and loss_norm(1024, 1) == (1024, 2048)
    torch.set_num_threads(2)
    fp, z_full, total_full = _shard_gradient(params, K, pose, uv, rng, gt, gts, views, 0, R_GLOBAL, None)
    ref = fp.grad.clone().numpy()
    opt = torch.optim.Adam(fp.params, lr=5e-4)
    torch.nn.utils.clip_grad_norm_(fp.params, 1.0)
    opt.step()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    k = R_GLOBAL // world
    scale = np.abs(ref).max()
    for rank, local, grad, flat, z, total in res:
        assert np.array_equal(z, z_full[rank * k:(rank + 1) * k])            # fast = 1: sampling does not couple rays
        np.testing.assert_allclose(grad, ref, rtol=0, atol=2e-6 * scale)     # all-reduced == full-batch gradient
        assert np.abs(local - ref).max() > 0.05 * scale                      # (a single shard's is not)
    assert sum(t[5] for t in res) == pytest.approx(total_full, rel=1e-5)     # the shard losses add up to the batch loss
    np.testing.assert_array_equal(res[0][2], res[1][2])                      # same gradient bits on both ranks ...
    np.testing.assert_array_equal(res[0][3], res[1][3])                      # ... hence identical replicas after Adam
    d = np.abs(res[0][3] - fp.flat.numpy())
    assert d.max() <= 1.1e-3 and (d > 1e-5).mean() < 0.01                    # and the single-process step (Adam sign noise aside)


def test_shard_rays_rejects_ragged():
    sys.path.insert(0, os.path.join(ROOT, "s-volsdf_amd"))
    from svs_hip.trainer import shard_rays
    with pytest.raises(ValueError):
        shard_rays(torch.zeros(1, 10, 2), 0, 4)
    full = torch.arange(16.).reshape(1, 8, 2)
    parts = [shard_rays(full, r, 4) for r in range(4)]
    assert torch.equal(torch.cat(parts, 1), full)


def _render_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "s-volsdf_amd"))
    import importlib.util
    import torch.distributed as dist
    # renderer.py only needs torch: load it without importing the package's HIP bindings
    spec = importlib.util.spec_from_file_location("renderer", os.path.join(ROOT, "s-volsdf_amd", "svs_hip", "renderer.py"))
    renderer = importlib.util.module_from_spec(spec); spec.loader.exec_module(renderer)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Sampler:
        group_rays = None

    class Toy(torch.nn.Module):
        """stands in for VolSDFNetwork: every output is a function of the pixel alone"""
        def __init__(self):
            super().__init__()
            self.ray_sampler = Sampler()
            self.calls = []

        def forward(self, inp, fast=-1):
            uv = inp["uv"][0]
            self.calls.append((uv.shape[0], self.ray_sampler.group_rays))
            return {"rgb_values": torch.stack([uv[:, 0], uv[:, 1], uv.sum(1)], 1), "depth_values": uv[:, :1] * 2.0,
                    "weights": uv[:, :1].repeat(1, 5)}

    m = Toy().eval()
    N = 2300                               # 5 chunks of 500: 3 + 2 over two ranks, ragged last chunk
    uv = torch.stack([torch.arange(N, dtype=torch.float32), torch.arange(N, dtype=torch.float32) % 7], 1)[None]
    lo, hi = renderer.shard_pixels(N, 500, rank, world)
    out = renderer.render_image(m, {"uv": uv}, N, split_n_pixels=500, rays_per_launch=1000,
                                keys=("rgb_values", "depth_values", "weights"), rank=rank, world=world)
    q.put((rank, lo, hi, m.calls, {k: v.numpy() for k, v in out.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_image_render():
    """render_image over 2 ranks: whole 500-ray chunks per rank, per-chunk convergence groups set on the sampler, and
    the all-gathered image equals the single-process result."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_render_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, calls0, out0), (r1, lo1, hi1, calls1, out1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 1500, 1500, 2300)
    assert calls0 == [(1000, 500), (500, 500)] and calls1 == [(800, 500)]
    N = 2300
    uv = np.stack([np.arange(N, dtype=np.float32), np.arange(N, dtype=np.float32) % 7], 1)
    for out in (out0, out1):
        assert np.array_equal(out["rgb_values"], np.stack([uv[:, 0], uv[:, 1], uv.sum(1)], 1))
        assert np.array_equal(out["depth_values"], uv[:, :1] * 2.0)
        assert out["weights"].shape == (N, 5)


# --------------------------------------------------------------------------------------------------------------
# VolOpt-level data parallelism (volsdf/vsdf.py): joining the process group from the launcher's environment, host random
# generators continued from rank 0, the batch and the train-mode draws sharded the way TrainStep(shard_draws=True) does
# --------------------------------------------------------------------------------------------------------------
def _volopt_worker(rank, world, port, q):
    sys.path[:0] = [os.path.join(ROOT, "s-volsdf_amd")]
    import random
    from types import SimpleNamespace
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import volsdf.vsdf as vs
    from volsdf.model.network import VolSDFNetwork
    from volsdf.model.ray_sampler import ErrorBoundSampler
    assert vs.init_data_parallel() == (world, rank, rank) and dist.get_backend() == "gloo"
    # ranks arrive with DIFFERENT generator states ...
    torch.manual_seed(100 + rank); random.seed(200 + rank); np.random.seed(300 + rank)
    vs.sync_host_rng(world)
    # ... and continue from rank 0's: same view, same pixels
    view, perm, npr = random.randint(0, 48), torch.randperm(24 * 32)[:16], float(np.random.rand())
    # the batch every rank drew, and this rank's share of it (VolOpt._shard_batch)
    uv = torch.stack([perm % 32, perm // 32], 1).float()[None]
    gt = {"rgb": torch.rand(1, 16, 3), "rgb_smooth": torch.rand(1, 16, 3), "mask": torch.ones(1, 16, 3)}
    me = SimpleNamespace(world=world, rank=rank)
    mi, g = vs.VolOpt._shard_batch(me, {"uv": uv, "pose": torch.eye(4)[None]}, gt)
    # the step's draws: for the WHOLE batch (world x local rays), then this rank's rows (TrainStep._step, shard_draws)
    k = 16 // world
    sampler = SimpleNamespace(N_samples=64, N_samples_eval=128, N_samples_extra=32, inverse_sphere_bg=False)
    full = ErrorBoundSampler.draw_train_rng(sampler, k * world, torch.device("cpu"))
    mine = VolSDFNetwork.slice_rng(full, rank * k, (rank + 1) * k)
    # the logged loss terms: each rank's are its rays' share of the batch means; all-reduced like VolOpt.train_step does
    vec = torch.tensor([0.25 * (rank + 1), 1.5 * (rank + 1)])
    dist.all_reduce(vec)
    q.put((rank, view, perm.numpy(), npr, mi["uv"].numpy(), {n: v.numpy() for n, v in g.items()}, uv.numpy(),
           {n: v.numpy() for n, v in gt.items()}, {n: v.numpy() for n, v in mine.items()},
           {n: v.numpy() for n, v in full.items()}, vec.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_volopt_data_parallel_pieces_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_volopt_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = res
    assert a[1] == b[1] and np.array_equal(a[2], b[2]) and a[3] == b[3]          # same view / pixels / numpy draw on both ranks
    assert np.array_equal(a[6], b[6]) and all(np.array_equal(a[7][n], b[7][n]) for n in a[7])
    # the shards tile the batch: rays, colours ...
    assert np.array_equal(np.concatenate([a[4], b[4]], 1), a[6])
    for n in ("rgb", "rgb_smooth", "mask"):
        assert np.array_equal(np.concatenate([a[5][n], b[5][n]], 1), a[7][n])
    # ... and the draws (the extras' permutation is per batch, not per ray: identical on both ranks)
    for n, full in a[9].items():
        assert np.array_equal(full, b[9][n])
        if n == "perm":
            assert np.array_equal(a[8][n], full) and np.array_equal(b[8][n], full)
        else:
            assert np.array_equal(np.concatenate([a[8][n], b[8][n]], 0), full), n
    assert np.allclose(a[10], [0.75, 4.5]) and np.array_equal(a[10], b[10])


def test_init_data_parallel_without_launcher(monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "s-volsdf_amd"))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SVS_FORCE_DIST"):
        monkeypatch.delenv(k, raising=False)
    import torch.distributed as dist
    import volsdf.vsdf as vs
    assert vs.init_data_parallel() == (1, 0, 0) and not dist.is_initialized()


def test_no_ray_group_of_pure_padding():
    """A padded batch (TrainStep._pad_batch repeats the last ray up to the kernels' granularity) must not be split so that a
    ray group consists of padding only -- its loss would be a mean over zero rays; such a batch runs as one group."""
    sys.path.insert(0, os.path.join(ROOT, "s-volsdf_amd"))
    from types import SimpleNamespace
    from svs_hip.trainer import TrainStep
    me = SimpleNamespace(_force_groups=None, groups=[(0, 992), (992, 1008)], schedule={}, _n_valid=990)
    me._groups_raw = lambda R: TrainStep._groups_raw(me, R)
    assert TrainStep._groups_for(me, 1008) == [(0, 1008)]                  # rays 990..1007 are padding: the tail group is all padding
    me._n_valid = 1000
    assert TrainStep._groups_for(me, 1008) == [(0, 992), (992, 1008)]      # 8 real rays in the tail group: the split stays
    me._n_valid = 1008
    assert TrainStep._groups_for(me, 1008) == [(0, 992), (992, 1008)]


# --------------------------------------------------------------------------------------------------------------
# the flat gradient reduced in BUCKETS (trainer.grad_buckets / allreduce_range: the radiance bucket is reduced on a side stream
# beside the SDF backward, the SDF bucket at the end) against ONE all-reduce of the whole buffer
def _bucket_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "s-volsdf_amd"))
    import torch.distributed as dist
    from svs_hip.trainer import allreduce_flat_grad, allreduce_range, grad_buckets
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, n_sdf = 797883, 556545                                     # the DTU model's flat gradient and its SDF share
    g = torch.Generator().manual_seed(100 + rank)
    # gradient-like values over many orders of magnitude, a few exact zeros
    grad = torch.randn(n, generator=g) * torch.exp(torch.randn(n, generator=g) * 4.0)
    grad[torch.randint(0, n, (1000,), generator=g)] = 0.0
    whole = grad.clone()
    allreduce_flat_grad(whole, world)
    buckets = grad_buckets(n_sdf, n)
    assert buckets == [(n_sdf, n), (0, n_sdf)] and sum(hi - lo for lo, hi in buckets) == n
    pieces = grad.clone()
    (lo0, hi0), (lo1, hi1) = buckets
    work = allreduce_range(pieces, lo0, hi0, async_op=True)      # the early bucket, in flight ...
    pieces[lo1:hi1] *= 1.0                                         # ... while the step still writes the late one
    allreduce_range(pieces, lo1, hi1)
    work.wait()
    q.put((rank, torch.equal(whole, pieces), whole.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_equals_single_world2():
    """two buckets that tile the flat gradient == one all-reduce of the whole buffer, bit for bit, on both ranks; and the ranks
    hold identical sums (replicas stay identical)"""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(same for _, same, _ in res)
    np.testing.assert_array_equal(res[0][2], res[1][2])
