"""Benchmark of the S-VolSDF volume-rendering hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one pass of the per-iteration rendering path over one 1024-ray batch per GPU, exactly the model
call of VolOpt.train_step (volsdf/vsdf.py:205, train mode, fast=1): rays -> 128 uniform samples -> SDF MLP ->
error-bound / beta search -> 64 + 34 final samples -> SDF MLP forward + d sdf/dx (98 304+2 048 ... points) ->
radiance MLP -> alpha compositing -> eikonal points.  Inputs (weights, camera, pixel batch, random draws) are
resident in HBM before the timed region.  Rays are independent: with N GPUs each rank renders its own
1024-ray shard (weak scaling, no data-path collective).

Rank 0 prints ONE JSON line; `roofline` prices the dominant kernel (fused SDF forward+gradient) against the
float32 MFMA peak, `cpu_baseline` times the numpy oracle on a bounded sample of the same workload.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import synth
    from ref_shim import dtu_model_conf
    from svs_hip import ops
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
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    R = args.rays
    params = synth.make_params(0)
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
        if src.n < R * 90 or not ev_on[0]:
            return orig(pk, src, *a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(pk, src, *a, **k)
        e1.record()
        ev.append((e0, e1, src.n, src.S))
        return out

    ev_on = [False]
    ops.sdf_outputs = timed_sdf_outputs

    def step():
        return model(inp, fast=1)

    for _ in range(args.warmup):
        step()
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
        n_pts = ev[0][2]
        k_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        achieved = n_pts * 2 * F_SDF / (k_ms * 1e-3)
        S = ev[0][3]
        flop_per_ray = 128 * F_SDF + S * (2 * F_SDF + F_RGB) + 2 * (2 * F_SDF)
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
            "config": {"workload": "configs[1]: VolSDFNetwork.forward as called by VolOpt.train_step (train mode, fast=1): "
                                   f"{R} rays/GPU x 128 coarse + {S} composited samples + {2 * R} eikonal points, "
                                   "8x256 SDF MLP (fwd + d/dx) + 4x256 radiance MLP, forward only",
                       "rays_per_gpu": R, "flop_per_ray": flop_per_ray,
                       "model_flops_per_s": world * R * args.steps / dt * flop_per_ray},
            "roofline": {"bound": "mfma", "kernel": "sdf_full_kernel (SDF MLP forward + input gradient + features)",
                         "achieved": achieved / 1e12, "peak": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA, "traffic": None,
                         "kernel_ms": k_ms, "points_per_launch": n_pts, "flop_per_point": 2 * F_SDF},
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(params, K, pose)
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(params, K, pose, rays=48, reps=3):
    """numpy oracle (a port of the reference's PyTorch path) on a bounded sample: `rays` rays, same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import svs_oracle as orc
    import synth
    try:
        from threadpoolctl import threadpool_info
        cores = max([i.get("num_threads", 1) for i in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    uv = synth.make_uv(rays, seed=5)
    rng = synth.make_train_rng(rays, seed=5)
    orc.render_forward(params, uv[:8], pose, K, beta_param=params["density.beta"], fast=1, training=True,
                       rng={k: (v[:8] if v.shape[0] == rays else v) for k, v in rng.items()})
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        orc.render_forward(params, uv, pose, K, beta_param=params["density.beta"], fast=1, training=True, rng=rng)
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    return {"value": rays / t, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{rays} rays of the same train-mode fast=1 forward, numpy oracle, median of {reps}"}


if __name__ == "__main__":
    main()
