"""Benchmark of the S-VolSDF volume-rendering hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one pass of the per-iteration hot path over one 1024-ray batch per GPU = VolOpt.train_step
(volsdf/vsdf.py:196-219): rays -> 128 uniform samples -> SDF MLP -> error-bound / beta search -> 64 + 34 final
samples -> SDF MLP forward + d sdf/dx (100 352 + 2 048 points) -> radiance MLP -> alpha compositing -> MVS prior
lookup (3 views, 192x288x384 probability volumes) -> loss -> backward through compositing and both MLPs (incl. the
double backward through the normals) -> clip + NaN guard + Adam.  `--mode render` times the forward part only.
Inputs (weights, camera, pixel batch, prior volumes) are resident in HBM before the timed region.  Rays are
independent: with N GPUs each rank takes its own 1024-ray shard (`--scaling weak`, the default) or 1024/N rays of one
1024-ray batch (`--scaling strong`), and the step adds ONE RCCL all-reduce of the flat float32 gradient (3.19 MB).

Timed region: W warm-up steps (with --graph the step's launch sequence is captured into a hipGraph during them), then untimed
steps until `--settle` seconds of steady state have passed (clocks, allocator), then barrier + synchronize, EXACTLY K
steps, synchronize + barrier; the maximum over ranks is reported.

Rank 0 prints ONE compact JSON line LAST (<= 6 KB: `compact_line`, self-tested by `check_line`); the full record -- every
roofline row, the baselines' details, the secondary measurements -- goes to `bench_extras.json` next to this script (and to
gpurun_out/ when that directory exists).  `roofline` prices the kernel with the largest time per step; the per-launch
durations come from HIP events on each kernel's own launch stream (svs_hip/profiling.py) over steps run right AFTER the timed
region in the same process -- an event pair per launch would perturb the timed region, and a replayed graph's kernels cannot
be bracketed at all; the line carries that row and the five largest.  `cpu_baseline` times the CPU port on a bounded sample of
the same workload; `gpu_torch_baseline` the reference's step -- the sampler under no_grad, then plain PyTorch float32 autograd
+ clip + Adam -- on the same GPU.  Opt-in extras (never part of the default command): --other-precisions, --inline-ab,
--volopt-loop, --config4, --chamfer, --all-extras.

Precision.  The default (SVS_MLP_PRECISION=f16x2) evaluates every layer product, forward AND backward, as three fp16 MFMA
products of two-piece operands with float32 accumulation, and every activation block kept for the backward holds both pieces:
parameter gradients agree with float64 autograd like float32 autograd does (tests/test_gpu_train.py::
test_step_gradient_at_bench_geometry, 3e-5 of a tensor's largest entry).  `value` is measured on that mode.  With
--other-precisions the extras file also carries: `fast_grad_*` = SVS_MLP_PRECISION=f16x2_half (gradient-only blocks as ONE fp16
piece: a mixed-precision training step, 2e-4 ... 8e-4 gradient error) and `exact_f32_ms_per_step` = the float32-MFMA kernels.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "s-volsdf_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

F_SDF = 1_049_088          # FLOP per point per SDF-MLP forward
F_SDF_TRUNK = 918_016      # the same without the 256 feature rows of lin8 (the sampler's sdf-only evaluation)
F_RGB = 533_504            # FLOP per point per radiance forward
PEAK_F32_MFMA = 157.3e12   # MI355X dense float32 MFMA peak (MI355X_MICROARCH.md)
PEAK_F16_MFMA = 2.5e15     # dense fp16/bf16 MFMA peak; an fp16x2 product costs three fp16 MFMA products
PEAK_HBM = 8.0e12          # HBM3E bytes/s
BLOCK = 1024               # bytes per point of one 256-feature float32 activation block

# C-ABI entry points of the fused-MLP kernels: (kernel symbol fp16x2, symbol float32, bound, work per point, what)
# work: FLOP per point for "mfma" rows, algorithmic HBM bytes per point for "hbm" rows (every
# activation block a launch reads or writes once, BLOCK bytes per point each).
MLP_KERNELS = {
    "svs_sdf_outputs": ("svs::mlp::sdf_full_h2_kernel", "svs::mlp::sdf_full_kernel", "mfma", 2 * F_SDF,
                        "SDF MLP forward + input gradient + features"),
    "svs_sdf_vals": ("svs::mlp::sdf_only_h2_kernel", "svs::mlp::sdf_only_kernel", "mfma", F_SDF_TRUNK,
                     "SDF MLP forward of the sampler"),
    "svs_rgb_eval": ("svs::mlp::rgb_h2_kernel", "svs::mlp::rgb_kernel", "mfma", F_RGB, "radiance MLP forward"),
    "svs_rgb_bwd": ("svs::mlp::rgb_bwd_h2_kernel", "svs::mlp::rgb_bwd_kernel", "mfma", F_RGB, "radiance MLP backward"),
    "svs_sdf_bwd_a": ("svs::mlp::sdf_bwd_a_h2_kernel", "svs::mlp::sdf_bwd_a_kernel", "hbm", None,
                      "SDF MLP backward, second-order sweep"),
    "svs_sdf_bwd_b": ("svs::mlp::sdf_bwd_b_h2_kernel", "svs::mlp::sdf_bwd_b_kernel", "hbm", None,
                      "SDF MLP backward, backprop sweep"),
    "svs_wgrad_multi": ("svs::wgrad::h2::wgrad_h2_multi_kernel", "svs::wgrad::wgrad_kernel<8>", "hbm", None, "weight-gradient GEMMs"),
    "svs_lin8_row0_grad": ("svs::mlp::lin8_row0_h2_kernel", "svs::mlp::lin8_row0_kernel", "hbm", None,
                           "row 0 of lin8's weight gradient (h_8 and u_8 blocks)"),
}


def block_bytes():
    """Algorithmic HBM bytes per point of the HBM-bound launches, from the library's own block-format query."""
    from svs_hip import train
    return train.algorithmic_bytes_per_point()


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--volopt-child":
        rays, warm, steps, variants = sys.argv[2].split(":")[:4]
        model = (sys.argv[2].split(":") + ["dtu"])[4]
        print(json.dumps(_volopt_loop(int(rays), int(warm), int(steps), tuple(variants.split(",")), model=model)))
        return
    if len(sys.argv) == 3 and sys.argv[1] == "--allreduce-child":
        print(json.dumps(_allreduce_one_rank(int(sys.argv[2]))))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--mode", choices=["train", "render"], default="train")
    ap.add_argument("--model", choices=["dtu", "bmvs"], default="dtu",
                    help="dtu: VolSDFNetwork (configs[1], the headline metric); bmvs: VolSDFNetworkBG, fg + inverted-sphere "
                         "background (config 4), train mode only")
    ap.add_argument("--groups", choices=["auto", "none"], default="auto",
                    help="auto: the batch may run as two ray groups on concurrent streams, the first sized to whole rounds of "
                         "256 workgroups, so that the last partial round of every launch overlaps; TrainStep measures both "
                         "schedules during the first steps and keeps the faster (results do not depend on it)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --rays rays per GPU; strong: --rays rays in total, rays/N per GPU")
    ap.add_argument("--settle", type=float, default=1.0, help="seconds of untimed steady-state steps before the timed region")
    ap.add_argument("--graph", choices=["off", "on", "linear", "plan", "auto"], default="auto",
                    help="how the ~100 launches of a step are enqueued.  off: eager launches from Python; plan: the step's "
                         "captured sequence enqueued as plain launches by one library call (svs_plan_run); auto (default, also "
                         "VolOpt's): plan for batches of less than two rounds of workgroups (< 656 rays per GPU of the DTU model, "
                         "one or two ray groups: the host-bound sizes), eager otherwise; on / linear: hipGraphLaunch of the capture (with its stream topology / as one chain)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the per-kernel event timing (roofline = null)")
    ap.add_argument("--no-host-timing", action="store_true",
                    help="skip the 18 extra steps that time the host's enqueueing (`host_enqueue_ms_per_step`): counter passes "
                         "under rocprofv3 --pmc count on exactly --warmup + --steps steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-f32", action="store_true",
                    help="skip the extra runs in the other precisions (fast_grad_*, exact_f32_ms_per_step)")
    ap.add_argument("--no-gpu-torch", action="store_true", help="skip the PyTorch-eager comparator on the same GPU")
    ap.add_argument("--no-extras", action="store_true", help="skip the cost-volume / eval-render extras (`costvol`, `render_eval`)")
    ap.add_argument("--no-volopt-loop", action="store_true",
                    help="skip the end-to-end `VolOpt.run` measurement (`volopt_run` on the line)")
    ap.add_argument("--no-other-scaling", action="store_true",
                    help="N > 1: skip the second timed region in the other scaling mode (`other_scaling` on the line)")
    # opt-in extras: none of them is part of the default command (round 5's line lost its driver record to them); their results
    # go to bench_extras.json, never onto the final stdout line
    ap.add_argument("--other-precisions", action="store_true", help="extra: the same step at SVS_MLP_PRECISION=f16x2_half and f32")
    ap.add_argument("--inline-ab", action="store_true",
                    help="extra: the kernel table a second time with the radiance weight-gradient launch in line (SVS_RGB_WGRAD_SIDE=0)")
    ap.add_argument("--volopt-loop", action="store_true", help="extra: `VolOpt.run` end to end in a child process (1024 and 256 rays)")
    ap.add_argument("--config4", action="store_true",
                    help="extra: configs[3]'s projection from one GPU (child benches at 2048 / 256 rays, both models, all-reduce child)")
    ap.add_argument("--chamfer", action="store_true",
                    help="extra: short Chamfer optimisations per path on the analytic scene (tools/chamfer_parity.py; minutes)")
    ap.add_argument("--all-extras", action="store_true", help="all of the opt-in extras above")
    ap.add_argument("--extras-file", default=None,
                    help="where the full record goes (default: gpurun_out/bench_extras.json when that directory exists, and "
                         "bench_extras.json next to this script)")
    args = ap.parse_args()
    if args.all_extras:
        args.other_precisions = args.inline_ab = args.volopt_loop = args.config4 = args.chamfer = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain `python bench.py --gpus N`: this process has not touched a GPU; it starts the N ranks as
        # children (one process per GPU over RCCL, the launch line of the module docstring) and relays their output
        raise SystemExit(self_launch(args.gpus))

    import numpy as np
    import torch
    import synth
    from volsdf.utils.conf import dtu_model_conf
    from svs_hip import ops
    from svs_hip.trainer import TrainStep
    from volsdf.model.loss import VolSDFLoss
    from volsdf.model.network import VolSDFNetwork

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU fallback of the product path)"
    if os.environ.get("SVS_DIST_SHARE_GPU") == "1":      # validation aid: N ranks on one GPU, collectives over gloo
        local_rank %= torch.cuda.device_count()          # (SVS_DIST_BACKEND=gloo; exercises the N > 1 code path, measures nothing)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("SVS_FORCE_DIST") == "1":      # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        backend = os.environ.get("SVS_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    if args.scaling == "strong" and args.rays % world:
        raise SystemExit(f"--scaling strong: {args.rays} rays do not shard over {world} ranks")
    R = args.rays if args.scaling == "weak" else args.rays // world
    train = args.mode == "train"
    h2 = ops.is_h2(ops.default_precision())

    def make_model(seed_params=0):
        params = synth.make_params(seed_params)
        if args.model == "bmvs":
            from volsdf.utils.conf import bmvs_model_conf
            from volsdf.model.network_bg import VolSDFNetworkBG
            params = dict(params); params.update(synth.make_bg_params(0))
            model = VolSDFNetworkBG(bmvs_model_conf())
        else:
            model = VolSDFNetwork(dtu_model_conf())
        model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
        return params, model.to(dev).train()

    if args.model == "bmvs" and not train:
        raise SystemExit("--model bmvs is benchmarked in train mode")
    params, model = make_model()
    K, pose = synth.make_camera()

    def make_inputs(R, scaling):
        """each rank renders its own pixel shard of the view: R rays of a world x R ray batch (strong) / R rays of its own (weak)"""
        uv = synth.make_uv(R * world, seed=100)[rank * R:(rank + 1) * R] if scaling == "strong" else synth.make_uv(R, seed=100 + rank)
        inp = {"intrinsics": torch.from_numpy(K)[None].to(dev), "uv": torch.from_numpy(np.ascontiguousarray(uv))[None].to(dev),
               "pose": torch.from_numpy(pose)[None].to(dev)}
        rs = np.random.default_rng(11 + rank)
        gt = {"rgb": torch.from_numpy(rs.uniform(0, 1, (1, R, 3)).astype(np.float32)).to(dev),
              "rgb_smooth": torch.from_numpy(rs.uniform(0, 1, (1, R, 3)).astype(np.float32)).to(dev)}
        return inp, gt

    inp, gt_all = make_inputs(R, args.scaling)
    torch.manual_seed(1234 + rank)

    mvs = gt = None
    if train:
        # synthetic MVS prior at the real DTU stage-1 size: softmax(N(0,1)) over D = 192 at 288 x 384,
        # per-pixel hypotheses 1.5 .. 3.5, three views with x offsets 0, +-0.3
        gen = torch.Generator(device=dev); gen.manual_seed(7)
        views = []
        for j, dx in enumerate((0.0, 0.3, -0.3)):
            Kj, Pj = synth.make_camera(center=(dx, 0.0, -2.5), tilt=-0.12 * dx / 0.3)
            prob = torch.softmax(torch.randn(192, 288, 384, device=dev, generator=gen), 0)
            zm = torch.linspace(1.5, 3.5, 192, device=dev).view(-1, 1, 1) * (1 + 0.05 * (torch.rand(1, 288, 384, device=dev, generator=gen) * 2 - 1))
            views.append(dict(K=Kj, c2w=Pj, cost=prob, z_near=zm[0].contiguous(), z_far=zm[-1].contiguous()))
        mvs = dict(views=views, same_view=0, img_res=(576, 768), inverse_depth=False)
        gt = gt_all

    def make_step(mdl, inp=inp, gt=gt):
        if not train:
            def fwd():
                with torch.no_grad():
                    return mdl(inp, fast=1)
            return None, fwd
        loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=1.0,
                          sparse_weight=1.0, anneal_rgb=200, gce=0.5, confi=1e-3)       # config/ours.yaml:16-21
        t = TrainStep(mdl, loss, lr=5e-4, world=world, rank=rank, groups=None if args.groups == "none" else "auto",
                      graph={"off": False, "on": True, "linear": "linear", "plan": "plan", "auto": "auto"}[args.graph])
        return t, (lambda: t(inp, gt, mvs=mvs))

    ts, step = make_step(model)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # steady state before the clock starts: a fresh process runs its first steps slower (module loads, allocator growth,
    # graph capture) and the chip needs load to settle its clocks; a timed region of a few ms would measure that
    settle_steps, t_settle = 0, time.perf_counter()
    while time.perf_counter() - t_settle < args.settle and settle_steps < 5000:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        settle_steps += 10
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    del out

    # host side of a step: wall time the host thread needs to ENQUEUE a step (six steps into an empty queue, nothing
    # blocks on the device); a step whose kernels take less than this is host-bound
    host_ms = []
    for _ in range(0 if args.no_host_timing else 3):
        torch.cuda.synchronize()
        th = time.perf_counter()
        for _ in range(6):
            step()
        host_ms.append(1e3 * (time.perf_counter() - th) / 6)
    torch.cuda.synchronize()
    host_enqueue_ms = sorted(host_ms)[1] if host_ms else None

    other = None
    if world > 1 and train and not args.no_other_scaling and args.rays % world == 0:
        # the same job in the OTHER scaling mode, so that one driver run reports both: weak = every GPU its own --rays
        # batch; strong = ONE --rays batch sharded over the GPUs (config 4's partitioning: 2048 rays over 8 GPUs)
        o_scaling = "strong" if args.scaling == "weak" else "weak"
        R2 = args.rays // world if o_scaling == "strong" else args.rays
        inp2, gt2 = make_inputs(R2, o_scaling)
        _, model2 = make_model()
        ts2, step2 = make_step(model2, inp2, gt2)
        for _ in range(60):                      # past TrainStep's schedule tuning (steps 24..47)
            step2()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        n2 = max(50, args.steps // 2)
        for _ in range(n2):
            step2()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        d2 = torch.tensor([time.perf_counter() - t2], device=dev, dtype=torch.float64)
        dist.all_reduce(d2, op=dist.ReduceOp.MAX)
        d2 = float(d2.item())
        other = {"scaling": o_scaling, "rays_per_gpu": R2, "rays_total": R2 * world, "steps": n2,
                 "ms_per_step": 1e3 * d2 / n2, "value": world * R2 * n2 / d2, "unit": "rays/s",
                 "launch": ("launch plan" if any(c.plan is not None for c in ts2._captured.values()) else "eager launches")}
        del ts2, step2, model2

    S = model.ray_sampler.N_samples + model.ray_sampler.N_samples_extra + 2 - (1 if args.model == "bmvs" else 0)
    roofline = None
    if not args.no_kernel_timing:
        # every rank runs the extra steps (a train step contains the all-reduce); rank 0 reports its own kernel times
        sizes = [g[1] - g[0] for g in ts._groups_for(R)] if train else None
        roofline = kernel_roofline(ts, step, R, S, h2, train, ray_groups=sizes)
        if train and roofline is not None and args.inline_ab:
            # The dominant kernel (pass A of the SDF backward, HBM-bound) shares HBM with the radiance weight-gradient
            # launch that runs beside it on a side stream (the faster schedule: -1.2 % per step).  The same steps with that
            # launch IN LINE (SVS_RGB_WGRAD_SIDE=0) show what each kernel does with the memory system to itself.
            os.environ["SVS_RGB_WGRAD_SIDE"] = "0"
            try:
                alone = kernel_roofline(ts, step, R, S, h2, train, ray_groups=sizes)
            finally:
                os.environ.pop("SVS_RGB_WGRAD_SIDE", None)
            key = lambda r: (r["kernel"], r["what"], r["points_per_launch"])
            by = {key(r): r for r in alone["kernels"]}
            for r in roofline["kernels"]:
                o = by.get(key(r))
                if o is not None:
                    r["kernel_ms_in_line"], r["frac_in_line"] = o["kernel_ms"], o["frac"]
            top = by.get((roofline["kernel"], roofline["what"], roofline["points_per_launch"]))
            if top is not None:
                roofline["in_line"] = {"kernel_ms": top["kernel_ms"], "achieved": top["achieved"], "frac": top["frac"],
                                       "note": "the same kernel with the radiance weight-gradient launch in line instead of beside it "
                                               "(SVS_RGB_WGRAD_SIDE=0: +1.2 % per step, not the default); `achieved` / `frac` above "
                                               "are of the default schedule, where the two launches share HBM"}
        if dist:
            dist.barrier()
    if rank == 0:
        flop_per_ray = 128 * F_SDF + S * (2 * F_SDF + F_RGB) + 2 * (2 * F_SDF)
        if train:
            # backward: second-order sweep + backprop of the SDF MLP (2 x 8 layers), its two weight-gradient
            # contractions per layer, radiance backprop + weight gradients (approximate, SURVEY.md 8d: 0.92 GFLOP/ray)
            flop_per_ray += (S + 2) * (2 * F_SDF + 2 * F_SDF) + S * (2 * F_RGB)
        line = {
            "metric": "rendered rays/sec (1024-ray batch, 128 samples)",
            "value": world * R * args.steps / dt,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "host_enqueue_ms_per_step": host_enqueue_ms,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": dtype_string(ops.default_precision()),
            "data": "synthetic",
            "config": {"workload": ("configs[1]: VolOpt.train_step (forward + MVS prior lookup + loss + backward + clip/guard/Adam), "
                                    if train else "configs[1]: VolSDFNetwork.forward as called by VolOpt.train_step, forward only, ")
                                   + f"train mode, fast=1: {R} rays/GPU x 128 coarse + {S} composited samples + {2 * R} "
                                   "eikonal points, 8x256 SDF MLP + 4x256 radiance MLP"
                                   + (" + 32 inverted-sphere background samples per ray (8x256 bg implicit MLP + 128-wide bg "
                                      "radiance MLP): config 4, VolSDFNetworkBG" if args.model == "bmvs" else ""),
                       "mode": args.mode,
                       "mlp_precision": precision_note(ops.default_precision()),
                       "ray_groups": ([list(g) for g in ts._groups_for(R)] if train else [[0, R]]),
                       "ray_group_schedule": (ts.schedule.get(R) if (train and args.groups == "auto") else None),
                       "launch": ("eager launches" if not (train and ts.graph) else
                                  "launch plan (the captured step enqueued as plain launches by one library call: svs_plan_run) "
                                  "+ all-reduce / fused Adam" if (ts.graph in ("plan", "auto") and any(c.plan is not None for c in ts._captured.values())) else
                                  "eager launches" if ts.graph == "auto" else
                                  "hipGraph replay of the captured step + eager all-reduce / fused Adam"),
                       "launch_plan": (next((c.plan.info for c in ts._captured.values() if c.plan is not None), None)
                                       if train else None),
                       "rays_per_gpu": R, "rays_total": R * world, "settle_steps": settle_steps,
                       "flop_per_ray": flop_per_ray,
                       "model_flops_per_s": world * R * args.steps / dt * flop_per_ray},
            "roofline": roofline,
        }
        if other is not None:
            line["other_scaling"] = other
        if train and world == 1 and not args.no_gpu_torch and args.model == "dtu":
            line["gpu_torch_baseline"] = gpu_torch_baseline(ts, params, gt, R, dev, 1e3 * dt / args.steps, mvs)
        if train and h2 and world == 1 and args.model == "dtu" and args.other_precisions and not args.no_exact_f32:
            if ops.default_precision() == ops.F16X2:
                fg = other_precision_step_ms("f16x2_half", make_model, make_step, n=40, warm=60)
                line["fast_grad_ms_per_step"] = fg
                line["fast_grad_rays_per_s"] = R / (fg * 1e-3)
                line["fast_grad_note"] = ("SVS_MLP_PRECISION=f16x2_half: the same step with the backward's gradient-only activation "
                                          "blocks stored as one fp16 piece and one-product weight-gradient GEMMs -- a mixed-precision "
                                          "training step (parameter gradients 2e-4 ... 8e-4 of a tensor's largest entry off float64 "
                                          "autograd); NOT the figure `value` reports")
            line["exact_f32_ms_per_step"] = other_precision_step_ms("f32", make_model, make_step, n=10, warm=4)
        if train and world == 1 and args.model == "dtu" and (args.volopt_loop or args.config4) and not args.no_volopt_loop:
            if args.volopt_loop:
                line["volopt_run"] = volopt_loop(args.rays)
            if args.rays != 256:
                # config 4's per-GPU share when its 2048-ray batch is sharded over 8 GPUs: the loop at 256 rays (launch plans;
                # the host side decides here)
                line["volopt_run_256_rays"] = volopt_loop(256, variants=("default",))
            if args.config4:
                l256 = line.get("volopt_run_256_rays", line.get("volopt_run", {})).get("default", {}).get("ms_per_step")
                line["config4"] = config4_extra(dtu_loop_256_ms=l256)
        if world == 1 and train and args.model == "dtu" and not args.no_extras:
            # the other configurations of BASELINE.json, as extras measured after the timed region (same process, same box):
            # configs[2] = the CasMVSNet cost volume (tools/bench_costvol.py), and whole-image eval rendering, the
            # render_mvs path that hands depth maps to the MVS stages (tools/bench_render_eval.py)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            try:
                import bench_costvol
                line["costvol"] = bench_costvol.measure(dev)
            except Exception as e:                                           # an extra must not cost the headline line
                line["costvol"] = {"error": repr(e)}
            try:
                import bench_render_eval
                line["render_eval"] = bench_render_eval.measure(chunk_loop_too=False)
            except Exception as e:
                line["render_eval"] = {"error": repr(e)}
            # BASELINE.json's "Chamfer parity" on a scene with known geometry (tools/chamfer_parity.py; the 3000-step runs are
            # profiles/r05_chamfer_parity_prior.json): a short optimisation per path WITH a synthetic MVS prior (the reference's
            # stage-0 loss), rendered, fused and evaluated like a DTU scan
            try:
                if not args.chamfer:
                    raise _Skip()
                import chamfer_parity
                cp = chamfer_parity.measure(steps=600, seeds=(0, 1, 2), paths=("hip", "hip_f32", "torch_f32"), rays=512, timeout=600,
                                            prior=True, parallel=True,
                                            seeds_by_path={"hip": (0, 1, 2, 3, 4, 5, 6), "torch_f32": (0, 1, 2, 3, 4)})
                line["chamfer_parity"] = {"hip": cp.get("hip", {}).get("median_mm"), "hip_f32": cp.get("hip_f32", {}).get("median_mm"),
                                          "torch_f32": cp.get("torch_f32", {}).get("median_mm"), "spread": cp.get("spread_mm"),
                                          "unit": "mm", "steps": 600, "seeds": {"hip": 7, "hip_f32": 3, "torch_f32": 5},
                                          "statistic": "median over the seeds of a path (seven for the default HIP path, five for the torch "
                                                       "comparator, three for the float32 kernels)",
                                          "what": cp.get("what"),
                                          "note": "at 600 steps single runs of EVERY path scatter between 0.8 and 1.5 mm, the odd one up "
                                                  "to 2.4 (tools/dev/chamfer_600_distribution.py: six seeds per path), and the HIP paths are "
                                                  "not repeatable run to run (float atomics): medians, the runs (started side "
                                                  "by side on the one GPU) listed below; the statement with small error bars is the "
                                                  "3000-step one under long_runs",
                                          "mean": {k: v.get("overall_mm") for k, v in cp.items() if isinstance(v, dict) and "runs" in v},
                                          "range": {k: [v.get("min_mm"), v.get("max_mm")] for k, v in cp.items()
                                                    if isinstance(v, dict) and "runs" in v},
                                          "runs": {k: [{q: r.get(q) for q in ("seed", "accuracy_mm", "completeness_mm", "overall_mm",
                                                                                  "train_s", "error")} for r in v["runs"]]
                                                   for k, v in cp.items() if isinstance(v, dict) and "runs" in v},
                                          "long_runs": "profiles/r05_chamfer_parity_prior.json (3000 steps, three seeds per path, with the "
                                                       "synthetic MVS prior); profiles/r05_chamfer_parity*.json without a prior"}
            except _Skip:
                pass
            except Exception as e:
                line["chamfer_parity"] = {"error": repr(e)}
        if not args.no_cpu_baseline and args.model == "dtu" and world == 1:      # rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(params, K, pose, train=train)
        emit(line, args.extras_file)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


class _Skip(Exception):
    pass


MAX_LINE = 6144        # bytes of the final stdout line (the driver's record keeps a tail of ~8 KB)


def _r(x, sig=5):
    """floats to `sig` significant digits (the line is a record, not a checkpoint)"""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}")
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def compact_line(full):
    """The ONE line the driver parses: the contract's keys, `roofline` = the dominant kernel's row + the five largest rows,
    `cpu_baseline` = value / cores / kind / sample + one figure per thread count, and a few numbers of the secondary
    measurements.  Everything else is in bench_extras.json (`extras_file`)."""
    keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "host_enqueue_ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data")
    out = {k: full[k] for k in keys if k in full}
    c = full["config"]
    out["config"] = {"workload": c["workload"], "mode": c["mode"], "mlp_precision": c["mlp_precision"].split(":")[0],
                     "ray_groups": c["ray_groups"], "launch": c["launch"][:40], "rays_per_gpu": c["rays_per_gpu"],
                     "rays_total": c["rays_total"], "flop_per_ray": c["flop_per_ray"], "model_flops_per_s": c["model_flops_per_s"]}
    rf = full.get("roofline")
    if rf is not None:
        short = lambda k: k.split("::")[-1]
        top = {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "what",
                                  "launches_per_step", "kernel_ms", "ms_per_step", "points_per_launch", "work_per_point", "mfma_frac", "hbm_frac") if k in rf}
        top["timing"] = "HIP events on the kernel's launch stream, steps after the timed region"
        top["peak_basis"] = "mfma: 2500/3 TFLOP/s (fp16x2 = 3 fp16 products); hbm: 8 TB/s" if "fp16" in full["dtype"] else "f32 MFMA 157.3 TFLOP/s; hbm 8 TB/s"
        top["kernels"] = [{"kernel": short(r["kernel"]), "what": r["what"][:48], "bound": r["bound"], "launches_per_step": r["launches_per_step"],
                           "kernel_ms": r["kernel_ms"], "ms_per_step": r["ms_per_step"], "achieved": r["achieved"], "unit": r["unit"],
                           "frac": r["frac"], **({"hbm_frac": r["hbm_frac"]} if "hbm_frac" in r else {}),
                           **({"mfma_frac": r["mfma_frac"]} if "mfma_frac" in r else {})} for r in rf["kernels"][:6]]
        out["roofline"] = top
    else:
        out["roofline"] = None
    cb = full.get("cpu_baseline")
    if cb is not None:
        out["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"],
                               "by_threads": {str(r["threads"]): r["rays_per_s"] for r in cb.get("runs", []) if r["rays"] == cb.get("sample_rays")},
                               "host_cpu": cb.get("host_cpu"), "host_threads": cb.get("host_threads")}
    gt = full.get("gpu_torch_baseline")
    if gt is not None:
        out["gpu_torch_baseline"] = {k: gt[k] for k in ("value", "unit", "ms_per_step", "ratio_value_over_baseline") if k in gt}
        out["gpu_torch_baseline"]["what"] = "the reference's step in torch float32 eager on the same GPU (oracle/torch_ref.py)"
    if "other_scaling" in full:
        out["other_scaling"] = {k: full["other_scaling"][k] for k in ("scaling", "rays_per_gpu", "rays_total", "steps", "ms_per_step", "value")}
    cv = full.get("costvol")
    if isinstance(cv, dict):
        out["costvol"] = ({"error": cv["error"][:120]} if "error" in cv else
                          {k: cv[k] for k in cv if k != "roofline" and k != "workload"})
        if "roofline" in cv:
            name = lambda r: "warp" if r["bound"] == "hbm" else ("unet" if r["kernel"].startswith("CostRegNet") else "conv0")
            out["costvol"]["frac"] = {f"s{r['stage']}_{name(r)}": r.get("frac") for r in cv["roofline"] if "stage" in r}
    re_ = full.get("render_eval")
    if isinstance(re_, dict):
        out["render_eval"] = ({"error": re_["error"][:120]} if "error" in re_ else
                              {"image": re_.get("image"), "rays_per_s": re_.get("render_image_rays_per_s")})
    if rf is not None and any("sdf_full" in r["kernel"] for r in rf["kernels"]):
        # north_star asks >= 0.40 of the MFMA peak on the fused SDF MLP; the measured reason it stops below (ablation builds and
        # cycle stamps of rounds 2 and 5: NOTES/design_history_r01-r05.md section 4; `sdf_full_ablation` in the extras file)
        out["sdf_full_note"] = ("the training launch runs against TWO rooflines at once: beside its MFMAs it writes 17 and reads 8 "
                                "activation blocks per point (25.6 KB; PMC: 2.53 GB per launch) = hbm_frac of 8 TB/s, 2.7 TB/s of it "
                                "writes; one 512-register wave per SIMD issues in order: 166-173 cycles per k-step against 98.5 for "
                                "its three MFMAs (weight LDS-DMA, LDS fragment reads, barrier, epilogue, the block loads+stores share "
                                "the stream) at the 1.65-1.75 GHz the chip holds under fp16 MFMA load; sdf_only (no block traffic): "
                                "0.45 in isolation")
    out["extras_file"] = full.get("extras_file")
    return _r(out)


def check_line(text):
    """self-test of the line the driver will parse: strict JSON (no NaN / Infinity), bounded size, every roofline fraction in
    (0, 1] -- a fraction above 1 is a mis-priced row, not a fast kernel"""
    if len(text) >= MAX_LINE + 2048:
        raise AssertionError(f"bench line is {len(text)} bytes")
    d = json.loads(text, parse_constant=lambda c: (_ for _ in ()).throw(ValueError("non-finite number on the bench line: " + c)))
    json.dumps(d, allow_nan=False)
    rf = d.get("roofline")
    if rf is not None:
        for r in [rf] + rf.get("kernels", []):
            if not (0.0 < r["frac"] <= 1.0):
                raise AssertionError(f"roofline fraction outside (0, 1]: {r}")
    return d


SDF_FULL_ABLATION = {
    "source": "tools/ablate_fwd.py on one MI355X, round 5 (diagnostic builds -DSVS_ABL=<mask>; NOTES/design_history_r01-r05.md "
              "section 4): kernel time in ms for sdf_only / sdf_full / rgb with parts compiled out",
    "all": [0.356, 0.777, 0.188], "no_softplus": [0.333, 0.736, 0.182], "no_operand_split": [0.329, 0.674, 0.176],
    "no_chunk_wait_and_barrier": [0.341, 0.759, 0.179], "no_weight_fetch": [0.273, 0.624, 0.165],
    "cycles_per_k_step": {"three_mfma_only": 98.5, "skeleton_with_lds_reads_dma_barrier_softplus": 133, "sdf_only_trunk": 143,
                          "sdf_full_trunk": 166, "sdf_full_reverse": 173},
    "shader_clock_ghz_under_mfma_load": [1.65, 1.75],
    "reading": "no single part is the bound: every removed part shortens the kernel by its own issue time (the wave issues in "
               "order, one wave per SIMD holds all 512 registers), and removing vector work raises the MFMA duty and lowers the "
               "clock; without the weight fetch sdf_full would reach 0.43 of 833 TFLOP/s",
}


def emit(full, extras_file=None):
    """writes the full record to the extras file(s), prints the compact line LAST"""
    full.setdefault("sdf_full_ablation", SDF_FULL_ABLATION)
    paths = [extras_file] if extras_file else \
        ([os.path.join(ROOT, "gpurun_out", "bench_extras.json")] if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else []) + \
        [os.path.join(ROOT, "bench_extras.json")]
    written = []
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(full, f, indent=1, allow_nan=False)
            written.append(os.path.relpath(p, ROOT))
        except (OSError, ValueError) as e:
            print(f"bench.py: could not write {p}: {e!r}", file=sys.stderr)
    full["extras_file"] = written
    line = compact_line(full)
    if len(json.dumps(line, allow_nan=False, separators=(",", ":"))) > MAX_LINE:
        line.pop("sdf_full_note", None)
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) > MAX_LINE:                       # drop the least important parts, never the contract's keys
        for k in ("render_eval", "costvol", "other_scaling", "gpu_torch_baseline"):
            line.pop(k, None)
            text = json.dumps(line, allow_nan=False, separators=(",", ":"))
            if len(text) <= MAX_LINE:
                break
    try:
        check_line(text)
    except AssertionError as e:
        # the line still goes out (a record with a flagged row beats no record); the failure is loud and on the line
        line["self_test_failed"] = str(e)[:300]
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
        print("bench.py: SELF-TEST FAILED: " + str(e), file=sys.stderr, flush=True)
    sys.stdout.flush()
    print(text, flush=True)


def volopt_loop(rays, warm=60, steps=200, variants=("default", "sequential", "device_batches")):
    """`_volopt_loop` in a fresh child process (what a runner.py user's process looks like: no other TrainStep, no comparator
    runs, no extra streams before it -- inside this process, after the timed region and the other-precision runs, the
    256-ray loop measured 2.2 instead of 1.6 ms per step); falls back to this process if the child fails."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--volopt-child", f"{rays}:{warm}:{steps}:{','.join(variants)}"],
                           capture_output=True, text=True, timeout=600)
        res = json.loads(r.stdout.strip().splitlines()[-1])
        res["process"] = "child"
        return res
    except Exception as e:                                   # noqa: BLE001 -- a measurement aid must not cost the bench line
        res = _volopt_loop(rays, warm, steps, variants)
        res["process"] = f"in-process (child failed: {e!r:.100})"
        return res


def _allreduce_one_rank(n_floats, reps=200):
    """torch.distributed.all_reduce (RCCL) of the flat float32 gradient in a ONE-rank process group on this GPU: what the
    collective's call costs a step before any byte crosses xGMI (the step's only collective, trainer.TrainStep)."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        import socket
        s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s.getsockname()[1]); s.close()
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    g = torch.randn(n_floats, device=dev)
    for _ in range(20):
        dist.all_reduce(g)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dist.all_reduce(g)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    dist.destroy_process_group()
    return {"ms": ms, "bytes": 4 * n_floats, "ranks": 1}


def _child_json(argv, timeout=900, env=None):
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv, capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, **env) if env else None)
    # (the last JSON line: RCCL's banner is flushed to stdout when the process exits, behind it)
    return json.loads(next(l for l in reversed(r.stdout.strip().splitlines()) if l.startswith("{")))


def config4_extra(dtu_loop_256_ms=None):
    """BASELINE.json configs[3]: a 2048-ray batch sharded over the 8 GPUs of a node = 256 rays per GPU + one all-reduce of the
    flat gradient.  No 8-GPU node is the builder's to run on, so this is a PROJECTION from one GPU, stated as such: the bare
    step at 2048 and at 256 rays, `VolOpt.run` end to end at 256 rays (the loop a runner.py user gets: DataLoader, pixel draw,
    launch plan), the all-reduce call in a one-rank RCCL group, for the fg + background model (config 4's) and the DTU model;
    projected_strong_scaling_8 = t(2048 rays, 1 GPU) / (t(256-ray loop) + t(all-reduce)).  The all-reduce term is given twice:
    as measured with one rank (no xGMI traffic) and as a model of the 8-rank ring over xGMI (2 x 7/8 of the buffer per GPU at
    one link's ~50 GB/s effective, + 20 us per ring phase latency x 2 phases -- MI355X_MICROARCH.md's xGMI figures)."""
    quick = ["--no-cpu-baseline", "--no-exact-f32", "--no-gpu-torch", "--no-extras", "--no-volopt-loop", "--no-other-scaling",
             "--no-kernel-timing", "--steps", "100"]
    out = {}
    for model, n_grad in (("bmvs", 1361387), ("dtu", 797883)):
        row = {}
        try:
            for rays in (2048, 256):
                d = _child_json(["--model", model, "--rays", str(rays)] + quick)
                row[f"step_ms_{rays}_rays"] = d["ms_per_step"]
                row[f"launch_{rays}_rays"] = d["config"]["launch"][:40]
            if model == "bmvs":
                # the strict-float32 figure of config 4's model (float32 MFMA kernels for all four networks: csrc/svs_bg_f32.hip)
                row["exact_f32_step_ms_2048_rays"] = _child_json(["--model", model, "--rays", "2048"] + quick[:-1] + ["30"],
                                                                 env={"SVS_MLP_PRECISION": "f32"})["ms_per_step"]
            if model == "dtu" and dtu_loop_256_ms is not None:
                row["volopt_run_ms_256_rays"] = dtu_loop_256_ms
            else:
                row["volopt_run_ms_256_rays"] = _child_json(["--volopt-child", f"256:60:200:default:{model}"])["default"]["ms_per_step"]
            ar = _child_json(["--allreduce-child", str(n_grad)])
            row["allreduce_one_rank_ms"] = ar["ms"]
            row["gradient_bytes"] = ar["bytes"]
            ring = 1e3 * (2 * 7 / 8 * ar["bytes"] / 50e9) + 0.04
            row["allreduce_8_rank_ring_model_ms"] = ring
            t1 = row["step_ms_2048_rays"]
            row["projected_strong_scaling_8"] = t1 / (row["volopt_run_ms_256_rays"] + ar["ms"])
            row["projected_strong_scaling_8_ring_model"] = t1 / (row["volopt_run_ms_256_rays"] + ring)
            row["projected_strong_scaling_8_bare_step"] = t1 / (row["step_ms_256_rays"] + ar["ms"])
        except Exception as e:                                   # noqa: BLE001 -- an extra must not cost the bench line
            row["error"] = repr(e)[:200]
        out[model] = row
    out["note"] = ("PROJECTION from one GPU, no 8-GPU curve was measured here: configs[3] shards ONE 2048-ray batch over 8 GPUs "
                   "(256 rays each, SURVEY.md 8e); projected_strong_scaling_8 = step_ms_2048_rays / (volopt_run_ms_256_rays + "
                   "allreduce_one_rank_ms), `_ring_model` with the modelled 8-rank xGMI ring instead, `_bare_step` with the bare "
                   "256-ray step instead of the VolOpt.run loop; north_star asks >= 6")
    return out


def _volopt_loop(rays, warm=60, steps=200, variants=("default", "sequential", "device_batches"), model="dtu"):
    """What a runner.py user gets: `VolOpt.run` (the reference's optimisation loop, volsdf/vsdf.py:322-367) end to end on a
    synthetic in-memory scene with the SceneDataset interface at 576 x 768 (tests/synthetic_scene.py: full pixel grid per
    item, the reference's `change_sampling_idx`, one torch thread, as the reference's dataset does), `rays` pixels per
    step, no MVS prior, previews and checkpoints off -- ms per step including the DataLoader, for (a) the default loop
    (next batch prepared by a helper thread while the step is enqueued: same batches, same random streams), (b) the
    strictly sequential loop (`overlap_loader=False`), (c) the opt-in device-side batch source.  Outside the timed
    region; the step itself is the one `value` times."""
    import tempfile
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_volopt as tv
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp(prefix="svs_volopt_"))
    res = {}
    try:
        for name, kw in (("default", {}), ("sequential", dict(overlap_loader=False)), ("device_batches", dict(device_batches=True))):
            if name not in variants:
                continue
            a = tv.make_args()
            if model == "bmvs":                          # config 4: fg + inverted-sphere background model on a BlendedMVS-style scan
                import copy
                from volsdf.utils.conf import bmvs_model_conf
                a["vol"]["model"] = copy.deepcopy(dict(bmvs_model_conf()))
                a["vol"]["train"]["model_class"] = "volsdf.model.network_bg.VolSDFNetworkBG"
                a["vol"]["dataset"]["data_dir"] = "BlendedMVS"
            a["vol"]["dataset"]["img_res"] = [576, 768]
            a["vol"]["train"].update(num_pixels=rays, render_freq=10 ** 9, checkpoint_freq=10 ** 9)
            a["max_h"], a["max_w"] = 576, 768
            v = tv.build(a, **kw)
            v._preview = lambda *x, **k: None
            v.save_checkpoints = lambda *x, **k: None
            v.run(opt_stepN=warm)                   # kernel attribute set-up, TrainStep's schedule measurement (steps 24..47)
            torch.cuda.synchronize()
            n0, t0 = v.total_step, time.perf_counter()
            v.run(opt_stepN=steps)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            n = v.total_step - n0
            res[name] = {"ms_per_step": 1e3 * dt / n, "rays_per_s": rays * n / dt, "steps": n}
            del v
    finally:
        os.chdir(cwd)
    res["note"] = ("VolOpt.run end to end incl. the DataLoader over a SceneDataset-style dataset (host work per step: the "
                   "step's pixel indices -- torch.randperm(442 368)[:rays], drawn as its first `rays` shuffle iterations by "
                   "svs_randperm_prefix -- and the dataset's full pixel grid); `default` is what runner.py gets")
    return res


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a child and return its exit code (rank 0's
    JSON line goes to this process's stdout).  The parent never touches the HIP runtime: GPUs are counted from the KFD
    topology in sysfs (gpu_count(); where that is not readable the children report what they find)."""
    import socket
    import subprocess
    have = gpu_count()
    if 0 < have < n or (have == 0 and not os.path.isdir("/sys/class/kfd/kfd/topology/nodes")):
        print(f"bench.py --gpus {n}: this box shows {have} GPU(s) (KFD topology); one rank per GPU is needed -- "
              f"run with --gpus <= {have}" + (" (there is no CPU fallback of the product path)" if have == 0 else ""),
              file=sys.stderr, flush=True)
        return 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def kernel_roofline(ts, step, R, S, h2, train, n_steps=12, ray_groups=None):
    """Per-launch durations of every fused-MLP kernel over eager steps (HIP events on the launch streams), and the
    roofline of the one with the largest time per step."""
    import numpy as np
    import torch
    from svs_hip.profiling import LaunchTimer
    was = None
    if ts is not None:
        was, ts.graph = ts.graph, False
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with LaunchTimer(list(MLP_KERNELS)) as lt:
        for _ in range(n_steps):
            step()
        torch.cuda.synchronize()
    if ts is not None:
        ts.graph = was
    bpp = block_bytes() if train else {}
    rows = []
    for name, (k_h2, k_f32, bound, work, what) in MLP_KERNELS.items():
        ms = lt.times_ms(name)
        if not ms:
            continue
        metas = lt.meta[name]
        if name == "svs_wgrad_multi":
            # one launch = a list of jobs (svs_wgrad_job: one layer's products over n_points points); a launch may carry the
            # jobs of SEVERAL ray groups and of more than one network, so it is priced from its own job list: every operand
            # block a job reads, once, n_points x (bytes per point of that block) -- never from the number of jobs
            import ctypes
            from svs_hip import lib as _lib
            from svs_hip.train import LDW
            operand = 1024 if not h2 else bpp["wgrad_sdf"] // 34             # bytes per point of one operand block
            base = ts.accum.dWk.data_ptr() if ts is not None else 0          # accumulator slots: 0..8 SDF layers, 9..13 radiance
            kinds = {}
            for t, a in zip(ms, metas):
                jobs = ctypes.cast(a[0], ctypes.POINTER(_lib.WGradJob))
                n_jobs = int(a[1])
                nbytes, nflop, pts = 0, 0, {}
                for j in range(n_jobs):
                    jb = jobs[j]
                    nbytes += jb.n_points * (operand * (2 + (2 if jb.a1 else 0)) + (128 if jb.b_extra else 0))
                    # the launch's SECOND roofline: dW[256][256] += A B^T per operand pair (+ 16 columns for the extra rows)
                    nflop += jb.n_points * 2 * 256 * (256 * (2 if jb.a1 else 1) + (16 if jb.b_extra else 0))
                    slot, rem = divmod((jb.dW or 0) - base, 256 * LDW * 4)
                    net = ("sdf" if slot < 9 else "radiance") if (rem == 0 and 0 <= slot < 14) else "background"
                    pts.setdefault(net, {}).setdefault(slot, 0)
                    pts[net][slot] += jb.n_points
                # points of a network in this launch = those of its first layer's jobs (one job per ray group)
                n_pts = {net: v[min(v)] for net, v in pts.items()}
                kinds.setdefault((tuple(sorted(n_pts)), n_jobs, nbytes, sum(n_pts.values()), nflop), []).append(t)
            for (nets, n_jobs, nbytes, n_pts, nflop), tg in kinds.items():
                which = " + ".join(nets)
                rows.append(dict(entry=name, kernel=(k_h2 if h2 else k_f32), what=f"{what} ({which}: {n_jobs} layer jobs)",
                                 bound=bound, launches_per_step=len(tg) / n_steps, kernel_ms=float(np.mean(tg)),
                                 points_per_launch=float(n_pts), work_per_point=nbytes / max(n_pts, 1), flop_per_launch=float(nflop)))
            continue
        if name in ("svs_rgb_bwd", "svs_sdf_bwd_b"):
            pts = [int(a[0]) for a in metas]
        elif name == "svs_lin8_row0_grad":
            pts = [int(a[3]) for a in metas]
        else:
            pts = [int(a[1]) + int(a[6]) * int(a[7]) for a in metas]
        w = work if work is not None else bpp.get(name)
        # one row per launch SHAPE: the default step runs two ray groups of very different size (976 + 48 rays), and a
        # rate averaged over both would describe neither launch
        shapes = {}
        for t, n_pts in zip(ms, pts):
            shapes.setdefault(n_pts, []).append(t)
        for n_pts, tt in shapes.items():
            rows.append(dict(entry=name, kernel=(k_h2 if h2 else k_f32), what=what, bound=bound,
                             launches_per_step=len(tt) / n_steps, kernel_ms=float(np.mean(tt)),
                             points_per_launch=float(n_pts), work_per_point=w))
    for r in rows:
        r["ms_per_step"] = r["kernel_ms"] * r["launches_per_step"]
        per_launch = (r["work_per_point"] or 0) * r["points_per_launch"]
        rate = per_launch / (r["kernel_ms"] * 1e-3) if r["kernel_ms"] > 0 else 0.0
        if r["bound"] == "mfma":
            peak = PEAK_F16_MFMA / 3 if h2 else PEAK_F32_MFMA
            r.update(achieved=rate / 1e12, peak=peak / 1e12, unit="TFLOP/s", frac=rate / peak)
        else:
            r.update(achieved=rate / 1e9, peak=PEAK_HBM / 1e9, unit="GB/s", frac=rate / PEAK_HBM)
        if r.get("flop_per_launch") and r["kernel_ms"] > 0:
            # the weight-gradient GEMM is priced against HBM (every operand block read once), and it multiplies while it reads:
            # its algorithmic FLOP against the MFMA peak of the arithmetic, from the same launch time (NOTES/r06.md section 3f:
            # with the MFMAs compiled out the same copies run at 6.0 TB/s -- what bounds the launch is both at once)
            mf = r["flop_per_launch"] / (r["kernel_ms"] * 1e-3)
            r.update(mfma_TFLOPs=mf / 1e12, mfma_frac=mf / (PEAK_F16_MFMA / 3 if h2 else PEAK_F32_MFMA))
        if train and r["entry"] == "svs_sdf_outputs" and bpp.get("svs_sdf_outputs") and r["kernel_ms"] > 0:
            # the training launch of the fused SDF MLP is priced against the MFMA peak, but it also moves 25 activation blocks
            # per point (h, ghat, features out; h back in): its second roofline, from the same launch time
            hb = bpp["svs_sdf_outputs"] * r["points_per_launch"] / (r["kernel_ms"] * 1e-3)
            r.update(hbm_bytes_per_point=bpp["svs_sdf_outputs"], hbm_GBps=hb / 1e9, hbm_frac=hb / PEAK_HBM)
    rows.sort(key=lambda r: -r["ms_per_step"])
    # HBM traffic per launch: PMC counters of the committed profile of this same command (profiles/, see
    # tools/summarize_profiles.py); the bench itself cannot run the counters
    traffic, src_prof = {}, None
    try:
        # the latest round's summary: names are r<NN>_pmc_traffic.json (intermediate snapshots carry a suffix after r<NN>)
        prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles"))
                      if f.endswith("_pmc_traffic.json") and f[:1] == "r" and f[1:3].isdigit() and f[3] == "_")
        traffic = json.load(open(os.path.join(ROOT, "profiles", prof[-1])))["kernels"]
        src_prof = prof[-1]
    except Exception:
        pass
    top = dict(rows[0])
    # (the profile lists template instances: "svs::mlp::sdf_full_h2_kernel<true>"; its per-launch figure is the average
    # over the launches of the profiled step, i.e. over both ray groups)
    hit = [v for k, v in traffic.items() if k == top["kernel"] or k.startswith(top["kernel"] + "<")]
    top["traffic"] = max((v.get("hbm_bytes") or 0.0) for v in hit) if hit else None
    top["traffic_source"] = src_prof
    top["timing"] = (f"HIP events on each kernel's own launch stream over {n_steps} steps after the timed region (same "
                     "process, same launch sequence; kept out of the timed region because an event pair per launch "
                     "serialises the host side)")
    top["peak_note"] = ("mfma rows: algorithmic float32 FLOP; fp16x2 evaluates each product as three fp16 MFMA products: "
                        "peak = 2500 / 3 TFLOP/s.  hbm rows: algorithmic bytes = every activation block the launch "
                        "reads or writes, once" if h2 else "dense float32 MFMA peak / HBM3E peak")
    top["kernels"] = [{k: r[k] for k in ("kernel", "what", "bound", "launches_per_step", "kernel_ms", "ms_per_step",
                                         "points_per_launch", "work_per_point", "achieved", "unit", "frac",
                                         "hbm_bytes_per_point", "hbm_GBps", "hbm_frac", "flop_per_launch", "mfma_TFLOPs",
                                         "mfma_frac") if k in r} for r in rows]
    return top


def dtype_string(precision):
    from svs_hip import ops
    if precision == ops.F16X2:
        return ("f32 (every product, forward and backward, as three fp16 MFMA products of two-piece fp16 operands with f32 "
                "accumulate; activation blocks hold both pieces: float32 accuracy class)")
    if precision == ops.F16X2_HALF:
        return ("f32 forward (three fp16 MFMA products of two-piece operands), mixed-precision backward (gradient-only blocks as "
                "one fp16 piece, one-product weight gradients)")
    return "f32"


def precision_note(precision):
    from svs_hip import ops
    return {ops.F16X2: "fp16x2: two-piece fp16 operands on v_mfma_f32_32x32x16_f16, float32 accumulation, forward and backward "
                       "(float32-class accuracy: forward 2e-6, parameter gradients within 3e-5 of float64 autograd)",
            ops.F16X2_HALF: "fp16x2 forward; backward with one-piece gradient blocks (SVS_MLP_PRECISION=f16x2_half)",
            ops.F32: "float32 MFMA"}[precision]


def gpu_torch_baseline(ts, params, gt, R, dev, our_ms, mvs, reps=5, warm=2):
    """The reference's train step in plain PyTorch float32 on the SAME GPU (oracle/torch_ref.py), eager mode:
    the error-bounded sampler under no_grad (ray_sampler.py:67-219 at fast = 1: the SDF network on 128 points per ray, the
    beta search, inverse-CDF sampling, extras, sort; fresh draws every step), then the SDF and radiance MLPs with the double
    backward through the normals, compositing, loss, loss.backward(), clip_grad_norm_, the per-parameter NaN / Inf test
    (on_after_backward, vsdf.py:454-464), Adam.step, get_psnr (volsdf/vsdf.py:196-222 with
    network.py:206-279) and cost_mapping (the MVS prior look-up, vsdf.py:382-452, torch_ref.cost_mapping) at the rays and
    eikonal points of this process's last step.  The draws are made on the device here (the reference, and the timed step of
    this bench, draw from the CPU generator and upload); the dataset's randperm is outside both.  Checker-side code, outside
    the timed region."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import torch
    import torch_ref as tref
    keeps = [h[0] for h in ts._hold]
    outs = [r[1] for r in ts._results]
    cat = lambda xs: torch.cat([x.detach() for x in xs], 0).float()
    dirs, ds = (cat([k[n] for k in keeps]) for n in ("ray_dirs", "depth_scale"))
    cam = keeps[0]["cam_loc"].detach().float().reshape(3)
    eik = cat([k["src"].points for k in keeps])
    pj, pi = cat([o["pj"] for o in outs]), cat([o["pi"] for o in outs])
    p = {k: torch.tensor(np.asarray(v), dtype=torch.float32, device=dev, requires_grad=True) for k, v in params.items()}
    opt = torch.optim.Adam(list(p.values()), lr=5e-4)
    rgb, rgbs = gt["rgb"].reshape(-1, 3), gt["rgb_smooth"].reshape(-1, 3)
    n_rays = dirs.shape[0]

    def sdf_fn(x):
        sdf = tref.sdf_mlp(p, x)[:, 0]
        return torch.minimum(sdf, 20.0 * (3.0 - x.norm(2, 1)))          # get_sdf_vals, network.py:125-131

    def one(sampler=True, guard=True, lookup=True):
        opt.zero_grad(set_to_none=True)
        if sampler:
            with torch.no_grad():
                rng = dict(jitter=torch.rand(n_rays, 128, device=dev), u=torch.rand(n_rays, 64, device=dev),
                           perm=torch.randperm(128, device=dev), eik_idx=torch.randint(0, 98, (n_rays,), device=dev))
                beta0 = float(p["density.beta"].abs() + 1e-4)
                z, _ = tref.error_bound_sampler_train(sdf_fn, cam, dirs, beta0, rng, fast=1)
        else:
            z = z_ours
        out = tref.forward_differentiable(p, cam, dirs, z, eik, ds, device=dev)
        if lookup:
            xyz = cam.view(1, 1, 3) + z.unsqueeze(-1) * dirs.unsqueeze(1)
            out["pj"], out["pi"], _ = tref.cost_mapping(xyz, mvs["same_view"], mvs["views"], mvs["img_res"], mvs["inverse_depth"])
        else:
            out["pj"], out["pi"] = pj, pi
        tref.loss_fn(out, rgb, rgbs, 50).backward()
        torch.nn.utils.clip_grad_norm_(list(p.values()), 1.0)
        if guard:
            # on_after_backward (vsdf.py:454-464): a NaN / Inf test per parameter tensor, each a host synchronisation
            valid = True
            for q in p.values():
                if q.grad is not None:
                    valid = not (torch.isnan(q.grad).any() or torch.isinf(q.grad).any())
                    if not valid:
                        break
            if not valid:
                opt.zero_grad()
        opt.step()
        if guard:
            mse = torch.mean((out["rgb_values"] - rgb) ** 2)                   # get_psnr, vsdf.py:221 / rend_util.py:14-22
            return -10. * torch.log(mse) / torch.log(torch.tensor([10.], device=dev))

    z_ours = cat([k["z_vals"] for k in keeps])
    # the torch look-up at our samples gives our kernel's values (the comparator does the same work on the same data)
    with torch.no_grad():
        tj, ti, _ = tref.cost_mapping(cam.view(1, 1, 3) + z_ours.unsqueeze(-1) * dirs.unsqueeze(1), mvs["same_view"], mvs["views"],
                                      mvs["img_res"], mvs["inverse_depth"])
    lookup_err = max(float((tj - pj).abs().max()), float((ti - pi).abs().max()))
    if not lookup_err < 1e-4:
        raise AssertionError("comparator's prior look-up differs from the step's: %g" % lookup_err)

    def timed(**kw):
        for _ in range(warm):
            one(**kw)
        torch.cuda.synchronize()
        ts_ = []
        for _ in range(reps):
            t0 = time.perf_counter()
            one(**kw)
            torch.cuda.synchronize()
            ts_.append(time.perf_counter() - t0)
        return float(np.median(ts_))
    med = timed(sampler=True)
    med_noguard = timed(sampler=True, guard=False)
    med_nolookup = timed(sampler=True, guard=False, lookup=False)
    med_nosampler = timed(sampler=False, guard=False, lookup=False)
    return {"value": R / med, "unit": "rays/s", "ms_per_step": 1e3 * med, "rays": R, "kind": "port",
            "what": "oracle/torch_ref.py on cuda:0, torch float32 eager: error-bounded sampler (fast = 1, under no_grad) + "
                    "forward (SDF MLP + d sdf/dx via autograd, radiance MLP, compositing) + loss + backward (incl. the double "
                    "backward) + the MVS prior look-up (cost_mapping) + clip_grad_norm_ + the per-parameter NaN / Inf test of "
                    "on_after_backward + Adam + get_psnr",
            "ms_per_step_without_nan_test_and_psnr": 1e3 * med_noguard,
            "ms_per_step_without_those_and_prior_lookup": 1e3 * med_nolookup,
            "prior_lookup_max_abs_diff_vs_step": lookup_err,
            "ms_per_step_without_sampler": 1e3 * med_nosampler,
            "median_of": reps, "warmups": warm, "ratio_value_over_baseline": (1e3 * med) / our_ms,
            "torch": torch.__version__}


def other_precision_step_ms(precision, make_model, make_step, n=10, warm=4):
    """ms per step of the same workload with SVS_MLP_PRECISION=<precision> (f32: the exact float32-MFMA kernels; f16x2_half:
    one-piece gradient blocks), eager launches, `warm` untimed steps first (the ray-group schedule is measured during steps
    24-47: warm >= 48 times the schedule the step then keeps)."""
    import torch
    old = os.environ.get("SVS_MLP_PRECISION")
    os.environ["SVS_MLP_PRECISION"] = precision
    try:
        _, model = make_model()
        ts, step = make_step(model)
        ts.graph = False
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n
    finally:
        if old is None:
            os.environ.pop("SVS_MLP_PRECISION", None)
        else:
            os.environ["SVS_MLP_PRECISION"] = old


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """physical cores of the host: distinct (physical id, core id) pairs of /proc/cpuinfo (None if it cannot be read)"""
    try:
        cores, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if core is not None:
                    cores.add((phys, core))
                phys = core = None
        if core is not None:
            cores.add((phys, core))
        return len(cores) or None
    except OSError:
        return None


def gpu_count():
    """GPUs this process may use, WITHOUT touching the HIP runtime: the KFD topology lists every agent, GPUs are the nodes
    with a non-zero simd_count (CPUs have 0); HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES narrow it."""
    import glob
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(line.split()[:2] for line in open(f) if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def cpu_baseline(params, K, pose, train=True):
    """CPU port of the reference's PyTorch path on a bounded sample of the same workload: numpy oracle for the sampler and
    the MVS prior lookup, plain torch float32 autograd (oracle/torch_ref.py) for the differentiable part, clip + Adam.
    Sample: the first 256 rays of the benchmark's batch (the step is linear in the rays: the opt-in 1024-ray row shows it), median of
    3 / 5 / 3 steps after 1 warm-up at 1 torch thread (what the reference's trainer forces, volsdf/vsdf.py:21), at 32
    threads and at ALL PHYSICAL cores ("1 and all cores"; counted from /proc/cpuinfo); all hardware threads
    -- 256 on the round-4 box -- oversubscribe torch's intra-op pool (measured in round 4) and stay an opt-in; plus one
    row on the full 1024-ray batch at 32 threads.  For this leg the oracle's exp / expm1 / row sum are
    bound to numpy's (the bit-exact restatements of torch's routines emulate float32 fma in float64 and would make the
    baseline slower than a CPU path is)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import torch
    import svs_oracle as orc
    import synth
    import torch_ref as tref
    all_threads = os.cpu_count() or torch.get_num_threads()
    layers = orc.effective_weights(params, "implicit_network", 9)
    # a small prior (72 x 96 x 192, three views) for the lookup: its cost does not depend on the volume size
    rs = np.random.default_rng(3)
    views = []
    for dx in (0.0, 0.3, -0.3):
        Kj, Pj = synth.make_camera(center=(dx, 0.0, -2.5), tilt=-0.12 * dx / 0.3)
        logits = rs.normal(0, 1, (192, 72, 96)).astype(np.float32)
        prob = np.exp(logits - logits.max(0)); prob /= prob.sum(0)
        zm = np.broadcast_to(np.linspace(1.5, 3.5, 192, dtype=np.float32)[:, None, None], (192, 72, 96)).copy()
        views.append(dict(K=Kj, c2w=Pj, cost=prob.astype(np.float32), z_mvs=zm))

    def one(rays, rng, uv):
        if not train:
            orc.render_forward(params, uv, pose, K, beta_param=params["density.beta"], fast=1, training=True, rng=rng)
            return
        dirs, cam, ds = orc.rays_from_uv(uv, pose, K)
        z, z_eik = orc.error_bound_sampler(lambda x: orc.sdf_vals(layers, x), dirs, cam, orc.get_beta(0.1), fast=1,
                                           training=True, rng=rng)
        eik = np.concatenate([rng["eik_points"], (cam[None] + z_eik * dirs).astype(np.float32)], 0)
        p = tref.to_torch(params, torch.float32)
        out = tref.forward_differentiable(p, cam, dirs, z, eik, ds)
        xyz = (cam[None, None] + z[:, :, None] * dirs[:, None, :]).astype(np.float32)
        pj, pi, _ = orc.cost_mapping(xyz, 0, views, (576, 768))
        out["pj"], out["pi"] = torch.from_numpy(pj), torch.from_numpy(pi)
        tgt = torch.rand(rays, 3)
        total = tref.loss_fn(out, tgt, tgt, 250)
        total.backward()
        opt = torch.optim.Adam([v for v in p.values()], lr=5e-4)
        torch.nn.utils.clip_grad_norm_([v for v in p.values()], 1.0)
        opt.step()

    rows = []
    saved = torch.get_num_threads(), orc.ref_exp, orc.ref_expm1, orc.ref_sum
    with np.errstate(all="ignore"):
        orc.ref_exp = lambda x: np.exp(np.asarray(x, np.float32))
        orc.ref_expm1 = lambda x: np.expm1(np.asarray(x, np.float32))
        orc.ref_sum = lambda x: np.asarray(x, np.float32).sum(-1, keepdims=True, dtype=np.float32)
        try:
            # ALL host threads are not run by default: with torch's intra-op pool at 256 threads every small per-layer op
            # oversubscribes -- measured in round 4 on the GPU box: 197 s per 1024-ray step (5.2 rays/s) and 126 s per
            # 64-ray step (0.5 rays/s), against 5.8 s (177 rays/s) at 32 threads; SVS_CPU_BASELINE_ALL_THREADS=1 repeats it
            phys = physical_cores() or all_threads
            # ~35 s of CPU work in the default command; SVS_CPU_BASELINE_FULL=1 adds the whole 1024-ray batch at 32 threads
            plan = [(1, 256, 1, 3), (min(32, all_threads), 256, 1, 5), (phys, 256, 1, 3)]
            if os.environ.get("SVS_CPU_BASELINE_FULL") == "1":
                plan.append((min(32, all_threads), 1024, 1, 2))
            if all_threads > phys and os.environ.get("SVS_CPU_BASELINE_ALL_THREADS") == "1":
                plan.append((all_threads, 64, 0, 1))
            seen = set()
            for threads, rays, warm, reps in plan:
                if (threads, rays) in seen:
                    continue
                seen.add((threads, rays))
                torch.set_num_threads(threads)
                uv = synth.make_uv(rays, seed=5)
                rng = synth.make_train_rng(rays, seed=5)
                for _ in range(warm):
                    one(rays, rng, uv)
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    one(rays, rng, uv)
                    ts.append(time.perf_counter() - t0)
                rows.append(dict(threads=threads, rays=rays, rays_per_s=rays / float(np.median(ts)), median_s=float(np.median(ts)),
                                 reps=reps, warmups=warm))
        finally:
            torch.set_num_threads(saved[0])
            orc.ref_exp, orc.ref_expm1, orc.ref_sum = saved[1:]
    sample = [r for r in rows if r["rays"] == 256] or rows
    best = max(sample, key=lambda r: r["rays_per_s"])
    what = ("train step (numpy sampler + MVS prior lookup, torch float32 autograd, clip, Adam)" if train
            else "train-mode fast=1 forward, numpy oracle")
    full = next((r for r in rows if r["rays"] == 1024), None)
    return {"value": best["rays_per_s"], "unit": "rays/s", "cores": best["threads"], "kind": "port",
            "sample": f"{best['rays']} rays of the benchmark's 1024-ray batch through the same {what}; median of {best['reps']} "
                      f"steps after {best['warmups']} warm-up at the best of 1 / 32 / {phys} (all physical cores) torch threads",
            "sample_rays": best["rays"],
            "single_thread_rays_per_s": next(r["rays_per_s"] for r in rows if r["threads"] == 1),
            "physical_cores": phys,
            "all_physical_cores_rays_per_s": next((r["rays_per_s"] for r in rows if r["threads"] == phys and r["rays"] == 256), None),
            "all_threads_rays_per_s": next((r["rays_per_s"] for r in rows if r["threads"] == all_threads and all_threads != phys), None),
            "full_batch_rays_per_s": full["rays_per_s"] if full else None,
            "note": "the numpy parts are single-threaded; the step is linear in the rays (SVS_CPU_BASELINE_FULL=1: the whole batch "
                    f"at 32 threads); all {all_threads} hardware threads oversubscribe torch's intra-op pool (round 4: 5.2 rays/s; "
                    "SVS_CPU_BASELINE_ALL_THREADS=1 repeats it)",
            "runs": rows, "host_cpu": cpu_model_name(), "host_threads": all_threads}


if __name__ == "__main__":
    main()
