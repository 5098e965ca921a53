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
independent: with N GPUs each rank takes its own 1024-ray shard (weak scaling) and the step adds ONE RCCL
all-reduce of the flat float32 gradient (3.19 MB).

Rank 0 prints ONE JSON line; `roofline` prices the dominant kernel (fused SDF forward + input gradient) against the
matrix-core peak of the precision it runs in and, under "other", the weight-gradient GEMM launch against HBM;
`cpu_baseline` times the CPU port on a bounded sample of the same workload.  SVS_MLP_PRECISION=f32 selects the
float32-MFMA kernels instead of the default fp16x2 split-operand ones (same accuracy class).
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
F_RGB = 533_504            # FLOP per point per radiance forward
PEAK_F32_MFMA = 157.3e12   # MI355X dense float32 MFMA peak (MI355X_MICROARCH.md)
PEAK_F16_MFMA = 2.5e15     # dense fp16/bf16 MFMA peak; an fp16x2 product costs three fp16 MFMA products
PEAK_HBM = 8.0e12          # HBM3E bytes/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--mode", choices=["train", "render"], default="train")
    ap.add_argument("--model", choices=["dtu", "bmvs"], default="dtu",
                    help="dtu: VolSDFNetwork (configs[1], the headline metric); bmvs: VolSDFNetworkBG, fg + inverted-sphere "
                         "background (config 4), train mode only")
    ap.add_argument("--groups", choices=["auto", "none"], default="auto",
                    help="auto: the batch runs as two ray groups on concurrent streams, the first sized to whole rounds of "
                         "256 workgroups, so that the last partial round of every launch overlaps (results do not depend on it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("SVS_FORCE_DIST") == "1":      # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    R = args.rays
    params = synth.make_params(0)
    if args.model == "bmvs":
        from volsdf.utils.conf import bmvs_model_conf
        from volsdf.model.network_bg import VolSDFNetworkBG
        params = dict(params); params.update(synth.make_bg_params(0))
        model = VolSDFNetworkBG(bmvs_model_conf())
        if args.mode != "train":
            raise SystemExit("--model bmvs is benchmarked in train mode")
    else:
        model = VolSDFNetwork(dtu_model_conf())
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    model.to(dev).train()
    K, pose = synth.make_camera()
    uv = synth.make_uv(R, seed=100 + rank)          # each rank renders its own pixel shard of the view
    inp = {"intrinsics": torch.from_numpy(K)[None].to(dev), "uv": torch.from_numpy(uv)[None].to(dev),
           "pose": torch.from_numpy(pose)[None].to(dev)}
    torch.manual_seed(1234 + rank)

    # instrument the dominant kernel with events on the launch stream
    ev = []
    orig = ops.sdf_outputs

    def timed_sdf_outputs(pk, src, *a, **k):
        if src.S < 64 or not ev_on[0]:             # the sampler's forward-only launches are not this kernel
            return orig(pk, src, *a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(pk, src, *a, **k)
        e1.record()
        ev.append((e0, e1, src.n, src.S))
        return out

    ev_on = [False]
    ops.sdf_outputs = timed_sdf_outputs

    train = args.mode == "train"
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
        rs = np.random.default_rng(11 + rank)
        gt = {"rgb": torch.from_numpy(rs.uniform(0, 1, (1, R, 3)).astype(np.float32)).to(dev),
              "rgb_smooth": torch.from_numpy(rs.uniform(0, 1, (1, R, 3)).astype(np.float32)).to(dev)}
        loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=1.0,
                          sparse_weight=1.0, anneal_rgb=200, gce=0.5, confi=1e-3)       # config/ours.yaml:16-21
        ts = TrainStep(model, loss, lr=5e-4, world=world, rank=rank, groups=None if args.groups == "none" else "auto")

    wg_ev = []

    def step():
        if train:
            r = ts(inp, gt, mvs=mvs)
            if ev_on[0]:
                wg_ev.extend(b.timer_events for b in ts.bwd if b.timer_events)
            return r
        with torch.no_grad():
            return model(inp, fast=1)

    for _ in range(args.warmup):
        step()
    if train:
        for b in ts.bwd:
            b.time_wgrad = True
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev_on[0] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ev_on[0] = False
    if dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        # per-launch averages over ALL launches of the kernel in the timed region (what rocprofv3's kernel stats
        # average too): with ray groups there are two launches of different size per step, on concurrent streams
        launches = len(ev) / args.steps
        n_pts = float(np.mean([e[2] for e in ev]))
        pts_step = int(round(n_pts * launches))
        k_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        achieved = n_pts * 2 * F_SDF / (k_ms * 1e-3)
        S = ev[0][3]
        flop_per_ray = 128 * F_SDF + S * (2 * F_SDF + F_RGB) + 2 * (2 * F_SDF)
        if train:
            # backward: second-order sweep + backprop of the SDF MLP (2 x 8 layers), its two weight-gradient
            # contractions per layer, radiance backprop + weight gradients (approximate, SURVEY.md 8d: 0.92 GFLOP/ray)
            flop_per_ray += (S + 2) * (2 * F_SDF + 2 * F_SDF) + S * (2 * F_RGB)
        h2 = ops.default_precision() == ops.F16X2
        peak = PEAK_F16_MFMA / 3 if h2 else PEAK_F32_MFMA
        kname = "svs::mlp::sdf_full_h2_kernel" if h2 else "svs::mlp::sdf_full_kernel"
        roof_full = {"bound": "mfma", "kernel": kname + " (SDF MLP forward + input gradient + features)",
                     "achieved": achieved / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
                     "frac": achieved / peak, "traffic": None, "kernel_ms": k_ms, "launches_per_step": launches,
                     "points_per_launch": n_pts, "flop_per_point": 2 * F_SDF,
                     "peak_note": ("algorithmic float32 FLOP; fp16x2 evaluates each product as three fp16 MFMA products: "
                                   "peak = 2500 / 3 TFLOP/s" if h2 else "dense float32 MFMA peak")}
        # HBM traffic per launch: PMC counters of the committed profile of this same command (profiles/, see
        # tools/summarize_profiles.py); the bench itself cannot run the counters
        wname = "svs::wgrad::h2::wgrad_h2_multi_kernel" if h2 else "svs::wgrad::wgrad_kernel<8>"
        try:
            prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json"))
            pk = json.load(open(os.path.join(ROOT, "profiles", prof[-1])))["kernels"]
            traffic = {"sdf_full": pk.get(kname, {}).get("hbm_bytes"), "wgrad": pk.get(wname, {}).get("hbm_bytes")}
            src_prof = prof[-1]
        except Exception:
            traffic, src_prof = {"sdf_full": None, "wgrad": None}, None
        roof_full["traffic"] = traffic["sdf_full"]
        roof_full["traffic_source"] = src_prof
        roofline = roof_full
        if train and wg_ev:
            # the SDF weight gradients dW_l = abar_l h_l^T + ghat_l u_l^T (l = 0..7) and the feature head.  fp16x2: ONE
            # launch, HBM-bound: every operand block (32 KiB per 32 points) is read once; float32: 9 launches, MFMA-bound
            w_ms = float(np.mean([a.elapsed_time(b) for a, b in wg_ev]))
            w_launches = len(wg_ev) / args.steps
            n_tiles, n_main_tiles = (pts_step + 31) // 32, R * S // 32
            w_bytes = (8 * 4 * n_tiles + 2 * n_main_tiles) * 32768 / w_launches       # per launch
            w_flop = (2 * 2 * (F_SDF // 2 - 257 * 256) + 2 * 256 * 256) * pts_step / w_launches   # algorithmic: 2 pairs x 2 x rows x cols
            if h2:
                roof_w = {"bound": "hbm", "kernel": wname + " (all SDF weight gradients of a ray group in one launch)",
                          "achieved": w_bytes / (w_ms * 1e-3) / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                          "frac": w_bytes / (w_ms * 1e-3) / PEAK_HBM, "traffic": traffic["wgrad"], "kernel_ms": w_ms,
                          "launches_per_step": w_launches, "bytes_per_launch": w_bytes, "flop_per_launch": w_flop}
            else:
                roof_w = {"bound": "mfma", "kernel": wname + " (SDF weight gradients, 9 launches per step)",
                          "achieved": w_flop / (w_ms * 1e-3) / 1e12, "peak": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s",
                          "frac": w_flop / (w_ms * 1e-3) / PEAK_F32_MFMA, "traffic": traffic["wgrad"], "kernel_ms": w_ms / 9,
                          "launches_per_step": 9, "flop_per_step": w_flop}
            roof_w["traffic_source"] = src_prof
            # the dominant kernel is the one with the larger total time per step
            roofline = dict(roof_w, other=roof_full) if w_ms * w_launches > k_ms * launches else dict(roof_full, other=roof_w)
        line = {
            "metric": "rendered rays/sec (1024-ray batch, 128 samples)",
            "value": world * R * args.steps / dt,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("configs[1]: VolOpt.train_step (forward + MVS prior lookup + loss + backward + clip/guard/Adam), "
                                    if train else "configs[1]: VolSDFNetwork.forward as called by VolOpt.train_step, forward only, ")
                                   + f"train mode, fast=1: {R} rays/GPU x 128 coarse + {S} composited samples + {2 * R} "
                                   "eikonal points, 8x256 SDF MLP + 4x256 radiance MLP"
                                   + (" + 32 inverted-sphere background samples per ray (8x256 bg implicit MLP + 128-wide bg "
                                      "radiance MLP): config 4, VolSDFNetworkBG" if args.model == "bmvs" else ""),
                       "mode": args.mode,
                       "mlp_precision": ("fp16x2: two-piece fp16 operands on v_mfma_f32_32x32x16_f16, float32 accumulation "
                                         "(float32-class accuracy, same parity bounds)" if h2 else "float32 MFMA"),
                       "ray_groups": ([list(g) for g in ts.split_rays(R, S)] if (train and args.groups == "auto")
                                      else [[0, R]]),
                       "rays_per_gpu": R, "flop_per_ray": flop_per_ray,
                       "model_flops_per_s": world * R * args.steps / dt * flop_per_ray},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and args.model == "dtu":
            line["cpu_baseline"] = cpu_baseline(params, K, pose, train=train)
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(params, K, pose, train=True, rays=32, reps=3):
    """CPU port of the reference's PyTorch path on a bounded sample (`rays` rays of the same workload): numpy oracle for
    the sampler / forward, plain torch float32 autograd (oracle/torch_ref.py) for the differentiable part."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import torch
    import svs_oracle as orc
    import synth
    import torch_ref as tref
    cores = torch.get_num_threads()
    uv = synth.make_uv(rays, seed=5)
    rng = synth.make_train_rng(rays, seed=5)
    layers = orc.effective_weights(params, "implicit_network", 9)

    def one():
        if not train:
            orc.render_forward(params, uv, pose, K, beta_param=params["density.beta"], fast=1, training=True, rng=rng)
            return
        dirs, cam, ds = orc.rays_from_uv(uv, pose, K)
        z, z_eik = orc.error_bound_sampler(lambda x: orc.sdf_vals(layers, x), dirs, cam, orc.get_beta(0.1), fast=1,
                                           training=True, rng=rng)
        eik = np.concatenate([rng["eik_points"], (cam[None] + z_eik * dirs).astype(np.float32)], 0)
        p = tref.to_torch(params, torch.float32)
        out = tref.forward_differentiable(p, cam, dirs, z, eik, ds)
        tgt = torch.rand(rays, 3)
        total = tref.loss_fn(out, tgt, tgt, 250)
        total.backward()
        opt = torch.optim.Adam([v for v in p.values()], lr=5e-4)
        torch.nn.utils.clip_grad_norm_([v for v in p.values()], 1.0)
        opt.step()

    one()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        one()
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    what = "train step (numpy sampler + torch float32 autograd, clip, Adam)" if train else "train-mode fast=1 forward, numpy oracle"
    return {"value": rays / t, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{rays} rays of the same {what}, median of {reps}"}


if __name__ == "__main__":
    main()
