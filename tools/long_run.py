"""Sanity run: N optimisation steps (default 2000) of the DTU model on a smooth synthetic target with the default step
(the process's SVS_MLP_PRECISION, default fp16x2 with float32-class gradients; measured ray-group schedule) AND with the exact
float32-MFMA kernels (SVS_MLP_PRECISION=f32), same seeds, same pixel batches.  The colour and eikonal losses must fall,
beta must shrink, and the two precisions must follow the same trajectory: the end-to-end check that the gradient error of the
fp16x2 path (run with SVS_MLP_PRECISION=f16x2_half for the one-piece mode's 2e-4 ... 8e-4) does not
change what the optimiser does.  Prints a row every 250 steps and a
comparison at the end.      python tools/long_run.py [steps]
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


def run(precision, steps):
    if precision:
        os.environ["SVS_MLP_PRECISION"] = precision
    else:
        os.environ.pop("SVS_MLP_PRECISION", None)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = VolSDFNetwork(dtu_model_conf())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(0).items()})
    m.to(dev).train()
    loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=0.0, sparse_weight=0.0,
                      anneal_rgb=0, gce=0.5, confi=1e-3)
    ts = TrainStep(m, loss, groups="auto")
    K, pose = synth.make_camera()
    R, H, W = 1024, 576, 768
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    img = np.stack([0.5 + 0.4 * np.sin(xx / 90.0), 0.5 + 0.4 * np.cos(yy / 70.0), 0.5 + 0.3 * np.sin((xx + yy) / 120.0)], -1).astype(np.float32)
    Kd, Pd = torch.from_numpy(K)[None].to(dev), torch.from_numpy(pose)[None].to(dev)
    hist, window = [], []
    for step in range(steps):
        uv = synth.make_uv(R, seed=step)
        gt_rgb = torch.from_numpy(img[uv[:, 1].astype(int), uv[:, 0].astype(int)])[None].to(dev)
        lo, _ = ts({"intrinsics": Kd, "uv": torch.from_numpy(uv)[None].to(dev), "pose": Pd}, {"rgb": gt_rgb, "rgb_smooth": gt_rgb})
        if step >= steps - 100:
            window.append((float(lo["rgb_loss"]), float(lo["eikonal_loss"])))
        if step % 250 == 0 or step == steps - 1:
            info = ts.opt.info.cpu().numpy()
            hist.append((step, float(lo["rgb_loss"]), float(lo["eikonal_loss"]), float(info[0]), float(m.density.get_beta())))
            print(precision or "fp16x2", hist[-1], flush=True)
    p = ts.fp.flat
    w = np.asarray(window)
    res = dict(finite=bool(torch.isfinite(p).all()), max_abs_param=float(p.abs().max()), rgb_last100=float(w[:, 0].mean()),
               eik_last100=float(w[:, 1].mean()), beta=float(m.density.get_beta()), dropped=float(ts.opt.info[1]),
               schedule=ts.schedule.get(R))
    print(precision or "fp16x2", res, flush=True)
    return hist, res


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    # two runs per precision: float atomics make any two runs differ in the last bits, and 2000 Adam steps amplify that --
    # the spread between two runs of the SAME precision is the yardstick for the difference between the precisions
    runs = {prec: [run(prec, n) for _ in range(2)] for prec in (None, "f32")}
    (ha, ra), (ha2, ra2) = runs[None]
    (hb, rb), (hb2, rb2) = runs["f32"]
    print("\nstep   rgb_loss fp16x2 #1 #2 / f32 #1 #2              eikonal fp16x2 #1 #2 / f32 #1 #2")
    for a, a2, b, b2 in zip(ha, ha2, hb, hb2):
        print(f"{a[0]:5d}   {a[1]:.5f} {a2[1]:.5f} / {b[1]:.5f} {b2[1]:.5f}        {a[2]:.5f} {a2[2]:.5f} / {b[2]:.5f} {b2[2]:.5f}")
    for key in ("rgb_last100", "eik_last100", "beta"):
        print(f"{key:12s} fp16x2 {ra[key]:.5f} {ra2[key]:.5f}   f32 {rb[key]:.5f} {rb2[key]:.5f}")
    for _, r in runs[None] + runs["f32"]:
        assert r["finite"] and r["dropped"] == 0.0
    assert ha[-1][1] < 0.2 * ha[0][1], "the colour loss did not fall"
    mean = lambda k, rs: 0.5 * (rs[0][1][k] + rs[1][1][k])
    for key in ("rgb_last100", "eik_last100", "beta"):
        within = max(abs(ra[key] - ra2[key]), abs(rb[key] - rb2[key]))
        across = abs(mean(key, runs[None]) - mean(key, runs["f32"]))
        print(f"{key}: between precisions {across:.5f}, between two runs of one precision {within:.5f}")
        assert across <= max(3.0 * within, 0.15 * mean(key, runs["f32"])), f"fp16x2 and float32 runs differ in {key}"
