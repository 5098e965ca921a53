"""Sanity run: N optimisation steps (default 2000) of the DTU model on a smooth synthetic target with the default step
(the process's SVS_MLP_PRECISION, default fp16x2 with float32-class gradients; measured ray-group schedule) AND with the exact
float32-MFMA kernels (SVS_MLP_PRECISION=f32), same seeds, same pixel batches.  The colour and eikonal losses must fall,
beta must shrink, and the two precisions must follow the same trajectory: the end-to-end check that the gradient error of the
fp16x2 path (run with SVS_MLP_PRECISION=f16x2_half for the one-piece mode's 2e-4 ... 8e-4) does not
change what the optimiser does.  Prints a row every 250 steps and a
comparison at the end.      python tools/long_run.py [steps] [dtu|bmvs] [plain|full] [rays] [precision|launch]
  launch (5th argument): instead of two precisions, compare the two ways a step is enqueued -- eager launches from Python
        against launch plans (csrc/svs_plan.hip; what `auto` uses below 656 rays) -- at the given ray count.
  plain (default): colour + eikonal terms only, no annealing (round 3's run).
  full: the reference's whole loss (config/ours.yaml:16-21: anneal_rgb = 200, MVS prior term, sparse term) with synthetic prior
        volumes, so that the run crosses the colour annealing -- iterations 0 ... 199 train on the masked smooth target, 200+ on
        the image: the regime of the "iteration 250" gradient tests, where the radiance networks' tensors sit behind ReLU kinks.
  bmvs: the fg + inverted-sphere background model (config 4).  Its float32-MFMA variant does not exist,
        so the comparison there is default precision vs the one-piece mode, each against its own run-to-run spread.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "s-volsdf_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")]
import synth  # noqa: E402
from volsdf.utils.conf import dtu_model_conf  # noqa: E402
from volsdf.model.network import VolSDFNetwork  # noqa: E402
from volsdf.model.loss import VolSDFLoss  # noqa: E402
from svs_hip.trainer import TrainStep  # noqa: E402


MODEL, MODE, RAYS = "dtu", "plain", 1024


def run(precision, steps, graph=None):
    if precision in ("eager", "plan"):
        graph, precision = (False if precision == "eager" else "plan"), None
    if precision:
        os.environ["SVS_MLP_PRECISION"] = precision
    else:
        os.environ.pop("SVS_MLP_PRECISION", None)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    params = dict(synth.make_params(0))
    if MODEL == "bmvs":
        from volsdf.utils.conf import bmvs_model_conf
        from volsdf.model.network_bg import VolSDFNetworkBG
        params.update(synth.make_bg_params(0))
        m = VolSDFNetworkBG(bmvs_model_conf())
    else:
        m = VolSDFNetwork(dtu_model_conf())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    m.to(dev).train()
    full = MODE == "full"
    loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=1.0 if full else 0.0,
                      sparse_weight=1.0 if full else 0.0, anneal_rgb=200 if full else 0, gce=0.5, confi=1e-3)
    ts = TrainStep(m, loss, groups="auto", graph=graph)
    label = {False: "eager", "plan": "plan"}.get(graph, precision or "fp16x2")
    mvs = None
    if full:
        views = synth.make_mvs_views(3)
        mvs = dict(views=[dict(K=v["K"], c2w=v["c2w"], cost=torch.from_numpy(v["cost"]).to(dev), z_mvs=torch.from_numpy(v["z_mvs"]).to(dev))
                          for v in views], same_view=0, img_res=(576, 768), inverse_depth=False)
    K, pose = synth.make_camera()
    R, H, W = RAYS, 576, 768
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    img = np.stack([0.5 + 0.4 * np.sin(xx / 90.0), 0.5 + 0.4 * np.cos(yy / 70.0), 0.5 + 0.3 * np.sin((xx + yy) / 120.0)], -1).astype(np.float32)
    Kd, Pd = torch.from_numpy(K)[None].to(dev), torch.from_numpy(pose)[None].to(dev)
    hist, window = [], []
    for step in range(steps):
        uv = synth.make_uv(R, seed=step)
        # (the synthetic camera looks exactly at the origin: the principal point's ray goes through the sphere centre, where
        # the inverted-sphere background of the reference is 0 / 0 -- such a step is dropped by the NaN guard, as in the
        # reference; keep the runs comparable by keeping that one pixel out of the batches)
        uv = uv[~((uv[:, 0] == K[0, 2]) & (uv[:, 1] == K[1, 2]))]
        if len(uv) < R:
            uv = np.concatenate([uv, uv[:R - len(uv)] + 1], 0)
        gt_rgb = torch.from_numpy(img[uv[:, 1].astype(int), uv[:, 0].astype(int)])[None].to(dev)
        lo, _ = ts({"intrinsics": Kd, "uv": torch.from_numpy(uv)[None].to(dev), "pose": Pd}, {"rgb": gt_rgb, "rgb_smooth": gt_rgb},
                   mvs=mvs)
        if step >= steps - 100:
            window.append((float(lo["rgb_loss"]), float(lo["eikonal_loss"])))
        if step % 250 == 0 or step == steps - 1:
            info = ts.opt.info.cpu().numpy()
            hist.append((step, float(lo["rgb_loss"]), float(lo["eikonal_loss"]), float(info[0]), float(m.density.get_beta())))
            print(label, hist[-1], flush=True)
    p = ts.fp.flat
    w = np.asarray(window)
    res = dict(finite=bool(torch.isfinite(p).all()), max_abs_param=float(p.abs().max()), rgb_last100=float(w[:, 0].mean()),
               eik_last100=float(w[:, 1].mean()), beta=float(m.density.get_beta()), dropped=float(ts.opt.info[1]),
               schedule=ts.schedule.get(R))
    if graph:
        res["planned"] = any(c.plan is not None for c in ts._captured.values())
    print(label, res, flush=True)
    return hist, res


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    MODEL = sys.argv[2] if len(sys.argv) > 2 else "dtu"
    MODE = sys.argv[3] if len(sys.argv) > 3 else "plain"
    RAYS = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
    COMPARE = sys.argv[5] if len(sys.argv) > 5 else "precision"
    OTHER = "f32" if MODEL == "dtu" else "f16x2_half"
    BASE = None
    if COMPARE == "launch":
        BASE, OTHER = "eager", "plan"
    print(f"model {MODEL}, loss {MODE}, {n} steps of {RAYS} rays: {BASE or 'default precision'} vs {OTHER}")
    # two runs per precision: float atomics make any two runs differ in the last bits, and 2000 Adam steps amplify that --
    # the spread between two runs of the SAME precision is the yardstick for the difference between the precisions
    runs = {prec: [run(prec, n) for _ in range(2)] for prec in (BASE, OTHER)}
    runs[None] = runs[BASE]
    if COMPARE == "launch":
        assert all(r["planned"] for _, r in runs["plan"])
    (ha, ra), (ha2, ra2) = runs[None]
    (hb, rb), (hb2, rb2) = runs[OTHER]
    A = BASE or "fp16x2"
    print(f"\nstep   rgb_loss {A} #1 #2 / {OTHER} #1 #2              eikonal {A} #1 #2 / {OTHER} #1 #2")
    for a, a2, b, b2 in zip(ha, ha2, hb, hb2):
        print(f"{a[0]:5d}   {a[1]:.5f} {a2[1]:.5f} / {b[1]:.5f} {b2[1]:.5f}        {a[2]:.5f} {a2[2]:.5f} / {b[2]:.5f} {b2[2]:.5f}")
    for key in ("rgb_last100", "eik_last100", "beta"):
        print(f"{key:12s} {A} {ra[key]:.5f} {ra2[key]:.5f}   {OTHER} {rb[key]:.5f} {rb2[key]:.5f}")
    for _, r in runs[None] + runs[OTHER]:
        assert r["finite"] and r["dropped"] == 0.0
    first = 1 if MODE == "full" else 0        # (full: while the colour term is annealed -- steps < 200 -- its masked form is ~0)
    assert ha[-1][1] < (0.5 if MODE == "full" else 0.2) * ha[first][1], "the colour loss did not fall"
    mean = lambda k, rs: 0.5 * (rs[0][1][k] + rs[1][1][k])
    for key in ("rgb_last100", "eik_last100", "beta"):
        within = max(abs(ra[key] - ra2[key]), abs(rb[key] - rb2[key]))
        across = abs(mean(key, runs[None]) - mean(key, runs[OTHER]))
        print(f"{key}: between the two {'launch modes' if COMPARE == 'launch' else 'precisions'} {across:.5f}, between two runs of one {within:.5f}")
        assert across <= max(3.0 * within, 0.15 * mean(key, runs[OTHER])), f"default-precision and {OTHER} runs differ in {key}"
