"""The data-parallel path of `VolOpt` END TO END with two ranks on a single-GPU box (validation aid; 8-GPU runs are the
driver's): both ranks share cuda:0, the collectives go over gloo (SVS_DIST_SHARE_GPU=1, SVS_DIST_BACKEND=gloo; RCCL
refuses two ranks on one device).  Everything else is the production path: init_data_parallel, sync_host_rng, the sharded
batch and draws, ONE all-reduce of the flat gradient per step, rank-0 checkpoints, the sharded render + all-gather.

    python tools/dev/dp_two_ranks.py single /tmp/dp_ref.pt          # the same job in one process
    SVS_DIST_SHARE_GPU=1 SVS_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
        --master-addr 127.0.0.1 --master-port 29541 tools/dev/dp_two_ranks.py dp /tmp/dp_ref.pt

Checks (dp mode): step 0 -- every rank's per-ray colours are the single-process rows of its shard bit for bit, the
all-reduced gradient equals the single-process gradient up to the float atomics' order; after 4 steps the replicas are
bit-identical across ranks and agree with the single process to Adam's sign noise; the MVS-stage depth map of a view is
the same on both ranks and equals the single-process render; only rank 0 wrote checkpoints."""
import os, random, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "s-volsdf_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import test_gpu_volopt as tv


def job():
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    args = tv.make_args()
    args["vol"]["train"]["num_pixels"] = 256
    if os.environ.get("DP_MODEL") == "bmvs":              # BASELINE config 4's model: fg + inverted-sphere background
        from volsdf.utils.conf import bmvs_model_conf
        args["vol"]["train"]["model_class"] = "volsdf.model.network_bg.VolSDFNetworkBG"
        args["vol"]["model"] = bmvs_model_conf()
        args["vol"]["dataset"]["data_dir"] = "BlendedMVS"
    v = tv.build(args, overlap_loader=False)
    depth0, _ = v.render_mvs(0, 0)                                  # before any step: identical parameters everywhere
    v.train_dataset.change_sampling_idx(v.num_pixels)
    rec = dict(depth0=depth0.cpu(), steps=[])
    it = iter(v.train_dataloader)
    for i in range(4):
        batch = next(it)
        lo = v.train_step(batch)
        torch.cuda.synchronize()
        res = v.step_fn._results
        rec["steps"].append(dict(rgb=torch.cat([o["rgb_values"] for _, o in res]).cpu(), loss={k: float(x) for k, x in lo.items()},
                                 grad=v.step_fn.fp.grad.cpu().clone() if i == 0 else None))
    rec["flat"] = v.step_fn.fp.flat.cpu().clone()
    v.save_checkpoints(0)
    rec["ckpt"] = os.path.exists(os.path.join(v.checkpoints_path, "ModelParameters", "latest.pth"))
    return v, rec


def main():
    mode, path = sys.argv[1], sys.argv[2]
    os.chdir(tempfile.mkdtemp(prefix="svs_dp_"))
    if mode == "single":
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            os.environ.pop(k, None)
        v, rec = job()
        assert v.world == 1
        torch.save(rec, path)
        print("single-process reference written:", path, {k: round(x, 6) for k, x in rec["steps"][0]["loss"].items()})
        return
    import torch.distributed as dist
    v, rec = job()
    world, rank = v.world, v.rank
    assert world == 2 and dist.get_backend() == "gloo"
    ref = torch.load(path)
    k = ref["steps"][0]["rgb"].shape[0] // world
    s0, r0 = rec["steps"][0], ref["steps"][0]
    assert torch.equal(s0["rgb"], r0["rgb"][rank * k:(rank + 1) * k]), "step 0: this rank's colours are not the single-process rows of its shard"
    gerr = float((s0["grad"] - r0["grad"]).abs().max() / r0["grad"].abs().max())
    assert gerr < 2e-5, gerr
    # the rank's loss terms are its share of the batch means: their sum over the ranks is the batch's loss
    mine = torch.tensor([s0["loss"][n] for n in sorted(s0["loss"])], dtype=torch.float64)
    dist.all_reduce(mine)
    want = torch.tensor([r0["loss"][n] for n in sorted(r0["loss"])], dtype=torch.float64)
    assert torch.allclose(mine, want, rtol=1e-5, atol=1e-7), (mine, want)
    # replicas after 4 steps: identical across ranks, Adam-sign-noise close to the single process
    flats = [torch.zeros_like(rec["flat"]) for _ in range(world)]
    dist.all_gather(flats, rec["flat"])
    assert torch.equal(flats[0], flats[1]), "replicas diverged"
    d = (rec["flat"] - ref["flat"]).abs()
    assert float(d.max()) <= 4.1e-3 and float((d > 2e-5).float().mean()) < 0.05, (float(d.max()), float((d > 2e-5).float().mean()))
    # the sharded render: same depth map on both ranks, equal to the single-process render
    depths = [torch.zeros_like(rec["depth0"]) for _ in range(world)]
    dist.all_gather(depths, rec["depth0"])
    assert torch.equal(depths[0], depths[1]) and torch.equal(rec["depth0"], ref["depth0"])
    cks = [None, None]
    dist.all_gather_object(cks, rec["ckpt"])
    assert cks == [True, False], cks                        # rank 0 writes, rank 1 does not (own temporary folder here)
    if rank == 0:
        print(f"two ranks on one GPU over gloo: step-0 shard colours bit-identical, all-reduced gradient within {gerr:.1e} of the "
              f"single-process gradient, loss terms add up, replicas identical after 4 steps (max |param - single| "
              f"{float(d.max()):.1e}), sharded render identical; rank-0-only checkpoints")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
