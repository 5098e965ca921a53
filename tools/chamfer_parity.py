"""Chamfer parity on a scene with KNOWN geometry (BASELINE.json: "rays/sec ...; Chamfer parity"; no dataset in the image).

    python tools/chamfer_parity.py [--steps 3000] [--seeds 0,1] [--paths hip,hip_f32,torch_f32] [--rays 512] [--out FILE]

The analytic scene of tests/golden/synth.py (sphere + box, Lambertian; rendered to five views by a numpy sphere tracer, three
of them training views: tests/synthetic_scene.py::AnalyticSceneDataset) goes through the reference's own pipeline:

    VolOpt.run (volsdf/vsdf.py:322-367)  ->  VolOpt.render_mvs of the training views (:237-287)  ->  filter_depth
    (runner.py:301-404: PFM depth maps + camera files in, fused PLY out)  ->  evals/eval_dtu.py (:92-196: accuracy, completeness
    and their mean in mm against the ground-truth cloud)

once per PATH and seed; the ground truth are the analytic surface points the training views see.  Paths:
    hip         the product: VolOpt.run on the HIP kernels at the default precision (fp16x2, float32 class)
    hip_f32     the same with SVS_MLP_PRECISION=f32 (float32 MFMA kernels)
    hip_det     the same as `hip` with SVS_DETERMINISTIC=1 (weight gradients summed in one fixed order, one stream: a seed gives
                one answer, run after run)
    torch_f32   the comparator: the same optimisation in plain PyTorch float32 autograd on the same GPU (oracle/torch_ref.py:
                the reference's sampler, networks, compositing, loss; Adam, clip) -- checker-side code; its trained weights
                are then rendered, fused and evaluated by the same renderer / fusion / evaluator as the other paths
Every (path, seed) runs in a child process (the precision is read at import).  Prints one JSON object: per path the
(accuracy, completeness, overall) of every seed and their mean; `spread` = the largest seed-to-seed difference of `overall`
within a path; `hip_minus_torch` = difference of the path means.  Parity = the HIP value inside the torch path's spread.
"""
import argparse
import json
import os
import random
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", os.path.join("tests", "golden"), "s-volsdf_amd"):
    q = os.path.join(ROOT, p)
    if q not in sys.path:
        sys.path.insert(0, q)

IMG_RES = (192, 256)
MM = 200.0                      # world millimetres per normalised scene unit (the dataset's scale_factor)


PRIOR_D, PRIOR_Z = 48, (1.8, 3.3)      # the synthetic MVS prior: 48 depth planes over the scene's depth range (normalised units)


def build_prior(ds):
    """An MVS prior for the training views from the ANALYTIC depth maps, in the form the MVS stage hands to
    `VolOpt.get_mvs_input` (runner.py:209-236: per view a probability volume (1, D, h, w) and its depth hypotheses in MVS
    units): D uniform planes over the scene's depth range at half the image resolution, probability = a Gaussian of one
    plane spacing around the true depth, normalised over the planes; pixels that miss the object carry no probability (the
    confidence test of the loss, loss.py:96-105, then applies the sparsity term to their rays).  What a converged CasMVSNet
    stage would give on this scene, without the network (the image holds no trained weights)."""
    import numpy as np
    import torch
    H, W = ds.img_res
    h, w = H // 2, W // 2
    planes = np.linspace(PRIOR_Z[0], PRIOR_Z[1], PRIOR_D).astype(np.float32)
    step = planes[1] - planes[0]
    outs = []
    for i in ds.trains_ids():
        r = ds.renders[i]
        depth = r["depth"].reshape(h, 2, w, 2)
        hit = r["mask"].reshape(h, 2, w, 2)
        n = hit.sum((1, 3))
        d = np.where(n > 0, (depth * hit).sum((1, 3)) / np.maximum(n, 1), 0.0).astype(np.float32)      # mean over the hit sub-pixels
        prob = np.exp(-0.5 * ((planes[:, None, None] - d[None]) / step) ** 2)
        prob = prob / np.maximum(prob.sum(0, keepdims=True), 1e-12) * (n[None] > 0)
        z = np.broadcast_to(planes[:, None, None], (PRIOR_D, h, w)) * MM
        outs.append(dict(prob_volume=torch.from_numpy(prob.astype(np.float32))[None],
                         depth_values=torch.from_numpy(np.ascontiguousarray(z, np.float32))[None]))
    return outs


def make_args(rays, prior=False):
    import test_gpu_volopt as tv
    a = tv.make_args(use_mvs=prior)
    a["vol"]["train"].update(dataset_class="synthetic_scene.AnalyticSceneDataset", num_pixels=rays, render_freq=10 ** 9,
                             checkpoint_freq=10 ** 9, plot_freq=10 ** 9)
    a["vol"]["dataset"].update(img_res=list(IMG_RES), scale_factor=MM)
    a["max_h"], a["max_w"] = IMG_RES
    return a


def train_hip(seed, steps, rays, prior=False):
    """VolOpt.run as runner.py drives it.  Without a prior the loss is the colour term + 0.1 eikonal (loss.py:80-114); with
    the synthetic MVS prior (build_prior) it is the reference's stage-0 loss: MVS term, annealed sparsity and rgb_smooth."""
    import numpy as np
    import torch
    import test_gpu_volopt as tv
    torch.manual_seed(seed); random.seed(seed); np.random.seed(seed)
    v = tv.build(make_args(rays, prior))
    v._preview = lambda *x, **k: None
    v.save_checkpoints = lambda *x, **k: None
    if prior:
        v.get_mvs_input(build_prior(v.train_dataset))
    t0 = time.perf_counter()
    v.run(opt_stepN=steps)
    torch.cuda.synchronize()
    return v, dict(steps=int(v.iter_step), train_s=time.perf_counter() - t0)


def train_torch(seed, steps, rays, prior=False):
    """The same optimisation in plain PyTorch float32 (oracle/torch_ref.py) on the GPU; the trained weights are loaded into a
    VolOpt (never stepped) for the common render -> fuse -> evaluate tail."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch_ref as tref
    import test_gpu_volopt as tv
    from svs_hip import ops
    torch.manual_seed(seed); random.seed(seed); np.random.seed(seed)
    v = tv.build(make_args(rays, prior))
    ds, dev = v.train_dataset, torch.device("cuda", 0)
    views = None
    if prior:
        views = []
        for i, o in zip(ds.trains_ids(), build_prior(ds)):
            z = (o["depth_values"][0] / MM).to(dev)
            views.append(dict(K=ds.intrinsics_all[i], c2w=ds.pose_all[i], cost=o["prob_volume"][0].to(dev),
                              z_near=z[0].contiguous(), z_far=z[-1].contiguous()))
        smooth = {i: ds.rgb_smooth[i].to(dev) for i in ds.trains_ids()}
    p = {k: t.detach().clone().float().to(dev).requires_grad_(True) for k, t in v.model.state_dict().items()}
    opt = torch.optim.Adam(list(p.values()), lr=5e-4)
    imgs = {i: ds.rgb_images[i].to(dev) for i in ds.trains_ids()}
    poses = {i: ds.pose_all[i].to(dev) for i in ds.trains_ids()}
    Ks = {i: ds.intrinsics_all[i].to(dev) for i in ds.trains_ids()}
    H, W = ds.img_res

    def sdf_fn(x):
        sdf = tref.sdf_mlp(p, x)[:, 0]
        return torch.minimum(sdf, 20.0 * (3.0 - x.norm(2, 1)))          # get_sdf_vals, network.py:125-131

    t0 = time.perf_counter()
    for it in range(steps):
        idx = ds.trains_ids()[random.randint(0, ds.num_views - 1)]        # scene_dataset.py:216-219
        sel = torch.randperm(H * W, device=dev)[:rays]                    # :275-279 (drawn on the device here)
        uv = torch.stack([(sel % W).float(), (sel // W).float()], -1)
        dirs, cam, dscale = ops.rays_from_uv(uv, poses[idx], Ks[idx])
        with torch.no_grad():
            rng = dict(jitter=torch.rand(rays, 128, device=dev), u=torch.rand(rays, 64, device=dev),
                       perm=torch.randperm(128, device=dev), eik_idx=torch.randint(0, 98, (rays,), device=dev))
            z, z_eik = tref.error_bound_sampler_train(sdf_fn, cam, dirs, float(p["density.beta"].abs() + 1e-4), rng, fast=1)
            eik = torch.cat([torch.empty(rays, 3, device=dev).uniform_(-3.0, 3.0), cam.view(1, 3) + z_eik.view(-1, 1) * dirs], 0)
        out = tref.forward_differentiable(p, cam, dirs, z, eik, dscale, device=dev)
        rgb = imgs[idx][sel]
        rgb_s = rgb
        if prior:
            xyz = cam.view(1, 1, 3) + z.unsqueeze(-1) * dirs.unsqueeze(1)
            out["pj"], out["pi"], _ = tref.cost_mapping(xyz, list(ds.trains_ids()).index(idx), views, (H, W), False)
            rgb_s = smooth[idx][sel]
        opt.zero_grad(set_to_none=True)
        tref.loss_fn(out, rgb, rgb_s, it).backward()
        torch.nn.utils.clip_grad_norm_(list(p.values()), 1.0)
        opt.step()
    torch.cuda.synchronize()
    info = dict(steps=steps, train_s=time.perf_counter() - t0)
    v.model.load_state_dict({k: t.detach() for k, t in p.items()}, strict=True)
    v.iter_step = steps
    return v, info


def ground_truth(ds):
    """the analytic surface points the training views see (sphere-traced at twice the image resolution), in millimetres"""
    import numpy as np
    import synth
    pts = []
    H, W = ds.img_res
    for i in ds.trains_ids():
        K = ds.intrinsics_all[i].numpy().astype(np.float64).copy()
        K[:2, :3] *= 2.0
        pts.append(synth.render_analytic_view(K, ds.pose_all[i].numpy(), (2 * H, 2 * W))["points"])
    return np.concatenate(pts, 0) * MM


def fuse_and_evaluate(v, workdir):
    """render_mvs -> scan folder in the reference's layout -> filter_depth_folder -> evaluate_scan"""
    import numpy as np
    import torch
    from PIL import Image
    from datasets.data_io import save_pfm
    from evals import eval_dtu
    from helpers.utils import write_cam
    from svs_hip import fusion
    ds = v.train_dataset
    H, W = ds.img_res
    ids = list(ds.trains_ids())
    scan, out = os.path.join(workdir, "scan"), os.path.join(workdir, "out")
    for d in (os.path.join(scan, "cams"), os.path.join(scan, "images"), os.path.join(out, "depth_est"), os.path.join(out, "confidence")):
        os.makedirs(d, exist_ok=True)
    t0 = time.perf_counter()
    for k, i in enumerate(ids):
        depth, _ = v.render_mvs(i, 0)                        # position in the eval loader = image index (shuffle=False)
        depth = depth[0].float().cpu().numpy()
        pose = ds.pose_all[i].numpy().astype(np.float64)
        E = np.eye(4)
        E[:3, :3] = pose[:3, :3].T
        E[:3, 3] = -pose[:3, :3].T @ (pose[:3, 3] * MM)      # world millimetres
        cam = np.zeros((2, 4, 4), np.float32)
        cam[0] = E
        cam[1, :3, :3] = ds.intrinsics_all[i].numpy()[:3, :3]
        cam[1, 3] = [depth.min(), 1.0, 192.0, depth.max()]
        write_cam(os.path.join(scan, "cams", "{:0>8}_cam.txt".format(i)), cam)
        img = (ds.rgb_images[i].numpy().reshape(H, W, 3) * 255.0 + 0.5).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(scan, "images", "{:0>8}.jpg".format(i)), quality=95)
        save_pfm(os.path.join(out, "depth_est", "{:0>8}.pfm".format(i)), depth.astype(np.float32))
        # confidence = the view's object mask (the DTU runs mask with the evaluation masks, runner.py:349-368)
        save_pfm(os.path.join(out, "confidence", "{:0>8}.pfm".format(i)), ds.masks[i][:, 0].numpy().reshape(H, W).astype(np.float32))
    torch.cuda.synchronize()
    render_s = time.perf_counter() - t0
    ply = os.path.join(workdir, "fused.ply")
    xyz, rgb, stats = fusion.filter_depth_folder(scan, out, ply, ids, conf=0.5, filter_dist=1, filter_diff=0.01, thres_view=1)
    stl = ground_truth(ds)
    lo, hi = stl.min(0) - 30.0, stl.max(0) + 30.0
    res = 4.0
    obs = np.ones(tuple(int(x) for x in np.ceil((hi - lo) / res) + 2), np.uint8)
    acc, comp, overall = eval_dtu.evaluate_scan(xyz.astype(np.float64), stl, obs, np.stack([lo, hi]).astype(np.float32), np.array([[res]]),
                                               np.array([0.0, 0.0, 0.0, 1.0]), shuffle_rng=np.random.default_rng(0))
    return dict(accuracy_mm=float(acc), completeness_mm=float(comp), overall_mm=float(overall), n_fused=int(len(xyz)),
                n_ground_truth=int(len(stl)), render_s=render_s)


def child(path, seed, steps, rays, prior=False):
    import torch
    assert torch.cuda.is_available(), "chamfer_parity needs the GPU"
    cwd = os.getcwd()
    work = tempfile.mkdtemp(prefix="svs_chamfer_")
    os.chdir(work)
    try:
        v, info = (train_torch if path == "torch_f32" else train_hip)(seed, steps, rays, prior)
        info.update(fuse_and_evaluate(v, work))
        from svs_hip import ops
        info["precision"] = "torch float32 autograd" if path == "torch_f32" else str(ops.default_precision())
        info["beta"] = float(v.model.density.get_beta())
    finally:
        os.chdir(cwd)
    print(json.dumps(info), flush=True)


def measure(steps=3000, seeds=(0, 1), paths=("hip", "hip_f32", "torch_f32"), rays=512, timeout=3000, prior=False, parallel=False,
            seeds_by_path=None):
    """seeds_by_path: {path: seeds} overriding `seeds` for a path (the HIP runs are cheap: more of them steady a median).  parallel: all (path, seed) runs as concurrent child processes on the one GPU (the plain-torch comparator is bound by its
    host-side launches: three of them side by side take as long as one); train_s of a run is then not a timing of anything."""
    def start(path, s):
        env = dict(os.environ)
        env.pop("SVS_MLP_PRECISION", None)
        env.pop("SVS_DETERMINISTIC", None)
        if path == "hip_f32":
            env["SVS_MLP_PRECISION"] = "f32"
        if path == "hip_det":
            env["SVS_DETERMINISTIC"] = "1"
        return subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", f"{path}:{s}:{steps}:{rays}:{int(prior)}"], env=env,
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)

    def finish(proc, s):
        try:
            out, err = proc.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            proc.kill()
            out, err = proc.communicate()
            return dict(seed=s, error="timeout: " + (err or out)[-400:])
        try:
            return dict(seed=s, **json.loads(next(l for l in reversed(out.strip().splitlines()) if l.startswith("{"))))
        except Exception:                                # noqa: BLE001
            return dict(seed=s, error=(err or out)[-600:])

    sb = lambda path: tuple((seeds_by_path or {}).get(path, seeds))
    # (keyed by position: a seed may be listed twice -- the repeatability check of the deterministic path)
    procs = {(path, i): start(path, s) for path in paths for i, s in enumerate(sb(path))} if parallel else {}
    res = {}
    for path in paths:
        runs = []
        for i, s in enumerate(sb(path)):
            runs.append(finish(procs[(path, i)] if parallel else start(path, s), s))
        ok = [x for x in runs if "overall_mm" in x]
        res[path] = dict(runs=runs)
        if ok:
            res[path].update(overall_mm=sum(x["overall_mm"] for x in ok) / len(ok),
                             accuracy_mm=sum(x["accuracy_mm"] for x in ok) / len(ok),
                             completeness_mm=sum(x["completeness_mm"] for x in ok) / len(ok),
                             spread_mm=max(x["overall_mm"] for x in ok) - min(x["overall_mm"] for x in ok),
                             median_mm=sorted(x["overall_mm"] for x in ok)[len(ok) // 2] if len(ok) % 2 else
                             0.5 * (sorted(x["overall_mm"] for x in ok)[len(ok) // 2 - 1] + sorted(x["overall_mm"] for x in ok)[len(ok) // 2]),
                             min_mm=min(x["overall_mm"] for x in ok), max_mm=max(x["overall_mm"] for x in ok))
    res["spread_mm"] = max([v.get("spread_mm", 0.0) for v in res.values() if isinstance(v, dict)] or [0.0])
    if "overall_mm" in res.get("hip", {}) and "overall_mm" in res.get("torch_f32", {}):
        res["hip_minus_torch_mm"] = res["hip"]["overall_mm"] - res["torch_f32"]["overall_mm"]
    res["paired"] = paired_differences(res)
    res["prior"] = bool(prior)
    res["what"] = (("WITH a synthetic MVS prior (48 planes, Gaussian around the analytic depth; MVS + annealed sparsity + rgb_smooth terms): "
                    if prior else "no MVS prior (colour + eikonal terms): ") + f"analytic sphere + box scene ({IMG_RES[0]} x {IMG_RES[1]} images, 3 training views, {MM:.0f} mm per unit): {steps} "
                   f"optimisation steps of {rays} rays per path and seed -> render_mvs -> filter_depth -> evaluate_scan against the "
                   "analytic surface points the training views see; overall = (accuracy + completeness) / 2 in mm")
    return res


def paired_differences(res, base="torch_f32"):
    """Per path: the differences to the comparator over the seeds BOTH ran (a seed fixes the initial weights of every path:
    the runs of a seed are paired by their starting point), their mean, the standard error of that mean and the ratio --
    |mean| < 2 SE = no difference between the paths that this many seeds can show."""
    out = {}
    ref = {r["seed"]: r["overall_mm"] for r in res.get(base, {}).get("runs", []) if "overall_mm" in r}
    for path, v in res.items():
        if path == base or not isinstance(v, dict) or "runs" not in v:
            continue
        d = [r["overall_mm"] - ref[r["seed"]] for r in v["runs"] if "overall_mm" in r and r["seed"] in ref]
        if len(d) < 2:
            continue
        n = len(d)
        mean = sum(d) / n
        sd = (sum((x - mean) ** 2 for x in d) / (n - 1)) ** 0.5
        se = sd / n ** 0.5
        out[f"{path}_minus_{base}"] = dict(n=n, mean_mm=mean, sd_mm=sd, se_mm=se, mean_over_se=(mean / se if se > 0 else None),
                                           within_2_se=bool(abs(mean) < 2 * se), differences_mm=d)
    return out


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        path, seed, steps, rays, prior = (sys.argv[2].split(":") + ["0"])[:5]
        child(path, int(seed), int(steps), int(rays), prior == "1")
    else:
        ap = argparse.ArgumentParser()
        ap.add_argument("--steps", type=int, default=3000)
        ap.add_argument("--seeds", default="0,1")
        ap.add_argument("--paths", default="hip,hip_f32,torch_f32")
        ap.add_argument("--rays", type=int, default=512)
        ap.add_argument("--out", default=None)
        ap.add_argument("--prior", action="store_true", help="optimise with the synthetic MVS prior (the reference's stage-0 loss)")
        ap.add_argument("--parallel", action="store_true", help="all (path, seed) runs side by side on the one GPU")
        a = ap.parse_args()
        out = measure(a.steps, tuple(int(s) for s in a.seeds.split(",")), tuple(a.paths.split(",")), a.rays, prior=a.prior,
                      parallel=a.parallel)
        text = json.dumps(out, indent=1)
        print(text)
        if a.out:
            with open(a.out, "w") as f:
                f.write(text + "\n")
