"""The reference's per-scan loop (runner.py:164-300 + filter_depth :301-404 + evals/eval_dtu.py) end to end on the HIP
path, at toy sizes with random networks: three cascade stages of cost volumes (StageLoop / CascadeMVSNet / FeatureNet),
volume optimisation with the MVS priors and rendering of the training views (VolOpt), depth hand-off to the next
stage, depth / confidence / camera files (PFM, cam text), depth fusion to a PLY and its Chamfer evaluation.
A dataflow test: every stage consumes what the previous one produced, through the drop-in call surface."""
import os

import numpy as np
import pytest
import torch

import synth
from test_gpu_volopt import make_args

pytestmark = pytest.mark.gpu
F32 = np.float32


def test_scan_pipeline(tmp_path, monkeypatch):
    from datasets.data_io import read_pfm, save_pfm
    from evals import eval_dtu
    from helpers.utils import write_cam
    from models.CasMVSNet import CascadeMVSNet
    from svs_hip import fusion
    from svs_hip.stage_loop import StageLoop
    from volsdf.vsdf import VolOpt
    monkeypatch.chdir(tmp_path)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    G = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    # ---- MVS model (runner.py:128-154) and the scan's samples: 3 training views, each reference once ----
    model = CascadeMVSNet(refine=False, ndepths=[16, 8, 8], depth_interals_ratio=[4.0, 2.0, 1.0], share_cr=False,
                          cr_base_chs=[8, 8, 8], grad_method="detach")
    model.feature.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_featurenet_params(3).items()})
    for st, cin in enumerate((32, 16, 8)):
        model.cost_regularization[st].load_state_dict(
            {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_costreg_params(50 + st, cin).items()})
    model.to(dev).eval()
    H, W = 64, 96
    rng = np.random.default_rng(1)
    images = [G(rng.uniform(0, 1, (1, 3, H, W)).astype(F32)) for _ in range(3)]
    _, proj, depth_values = synth.make_mvs_sample(11, img_hw=(H, W), numdepth=16)
    samples = []
    for ref in range(3):
        order = [ref] + [v for v in range(3) if v != ref]
        samples.append(dict(imgs=torch.stack([images[v] for v in order], 1), depth_values=G(depth_values)[None],
                            proj_matrices={k: G(v[order])[None] for k, v in proj.items()},
                            filename=["scan24/{}/%08d{}" % ref]))

    # ---- volume optimiser (runner.py:164-170) ----
    args = make_args(use_mvs=True)
    args["vol"]["dataset"]["scale_factor"] = 90.0                           # VolSDF units (~5) -> MVS depth units (~450)
    vol_opt = VolOpt(args=args, batch_size=1, is_continue=False, timestamp="latest", checkpoint="latest", scan="scan24")
    vol_opt.trains_i = vol_opt.train_dataset.trains_ids()
    opt_stepNs, use_nerf_d = [6, 0, 0], [1, 0, 0]                           # config/ours.yaml: optimise at stage 0 only

    loop = StageLoop(model)
    outs_samples, view_extra = [None] * 3, [None] * 3
    depths = [None] * 3
    for stage_idx in range(3):
        outs, view_extras = loop.cost_volumes(stage_idx, samples, outs_samples, view_extra)                 # (a) cost volume
        if opt_stepNs[stage_idx] > 0 and use_nerf_d[stage_idx] > 0:                                          # (b) volume optimisation
            vol_opt.gen_dataset(stage_idx)
            vol_opt.stg = stage_idx
            vol_opt.loss.set_stg(stage_idx)
            vol_opt.get_mvs_input(outs)
            epoch = vol_opt.run(opt_stepNs[stage_idx]) if opt_stepNs[stage_idx] > 1 else 0
            for i, id_k in enumerate(vol_opt.trains_i):
                depths[i], _ = vol_opt.render_mvs(id_k, epoch)
                assert depths[i].is_cuda and bool(torch.isfinite(depths[i]).all())
            outs = StageLoop.hand_off_depth(outs, stage_idx, depths)
        outs_samples, view_extra = outs, view_extras
    assert loop.feature_calls == 3 and vol_opt.iter_step >= 6
    assert outs_samples[0]["depth"].shape == (1, H, W) and outs_samples[0]["stage1"]["depth"].shape == (1, 24, 32)

    # ---- depth / confidence / camera files (runner.py:252-296) ----
    outdir = tmp_path / "out" / "scan24"
    scan_folder = tmp_path / "scan24_in"
    for sub in ("depth_est", "confidence"):
        os.makedirs(outdir / sub)
    for sub in ("cams", "images"):
        os.makedirs(scan_folder / sub)
    from PIL import Image
    for v, o in enumerate(outs_samples):
        depth = o["depth"][0].cpu().numpy()
        up = lambda c: torch.nn.functional.interpolate(c[None], size=(H, W), mode="bilinear", align_corners=False)[0, 0].cpu().numpy()
        conf = up(o["stage1"]["photometric_confidence"]) * up(o["stage2"]["photometric_confidence"]) * o["photometric_confidence"][0].cpu().numpy()
        save_pfm(str(outdir / "depth_est" / ("%08d.pfm" % v)), depth)
        save_pfm(str(outdir / "confidence" / ("%08d.pfm" % v)), conf.astype(F32))
        cam = proj["stage3"][v].copy()
        cam[1, 3] = [float(depth_values[0]), float(depth_values[1] - depth_values[0]), 16, float(depth_values[-1])]
        write_cam(str(scan_folder / "cams" / ("%08d_cam.txt" % v)), cam)
        img = (images[v][0].permute(1, 2, 0).cpu().numpy() * 255).clip(0, 255).astype(np.uint8)
        Image.fromarray(img).save(str(scan_folder / "images" / ("%08d.jpg" % v)))
    back, _ = read_pfm(str(outdir / "depth_est" / "00000001.pfm"))
    assert np.array_equal(back, outs_samples[1]["depth"][0].cpu().numpy())

    # ---- fusion (runner.py:301-404): random networks are not multi-view consistent, so the thresholds are wide ----
    ply = str(tmp_path / "out" / "mvsnet024_l3.ply")
    xyz, rgb, stats = fusion.filter_depth_folder(str(scan_folder), str(outdir), ply, view_ids=[0, 1, 2], conf=0.0,
                                                 filter_dist=1e4, filter_diff=1e4, thres_view=1)
    assert len(stats) == 3 and xyz.shape[0] > 1000 and rgb.shape == xyz.shape and os.path.exists(ply)
    assert sorted(os.listdir(outdir / "mask"))[:3] == ["00000000_final.png", "00000000_geo.png", "00000000_photo.png"]

    # ---- Chamfer evaluation of that PLY against a stand-in ground truth (evals/eval_dtu.py) ----
    pts, col = fusion.read_ply_points(ply)
    assert pts.shape[0] == xyz.shape[0]
    lo, hi = pts.min(0), pts.max(0)
    res = float((hi - lo).max() / 40.0)
    dims = tuple(int(v) for v in np.ceil((hi - lo) / res).astype(int) + 2)
    stl = pts[::3] + np.random.default_rng(0).normal(0, 0.01 * res, pts[::3].shape)
    acc, comp, overall = eval_dtu.evaluate_scan(pts, stl, np.ones(dims, np.uint8), np.stack([lo, hi]).astype(F32), np.array([[res]]),
                                               np.array([0.0, 0.0, 1.0, 1e9]), downsample_density=0.02 * res, patch_size=60,
                                               max_dist=20 * res, shuffle_rng=np.random.default_rng(1))
    assert 0 <= acc < res and 0 <= comp < res and overall == pytest.approx((acc + comp) / 2)
