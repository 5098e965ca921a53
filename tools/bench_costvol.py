"""Secondary measurement (not the driver's bench line): CasMVSNet cost-volume build at BASELINE config 3
(640x512 image, D = 192/32/8, 3 views) on one MI355X.  Prints one JSON object with per-stage times and the
HBM roofline of the fused warp+variance kernel (algorithmic bytes: SURVEY.md section 8d)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "s-volsdf_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import synth  # noqa: E402
from models.CasMVSNet import CascadeMVSNet  # noqa: E402
from svs_hip import costvol  # noqa: E402


def measure(dev=None):
    """-> dict: per-stage ms of the three CasMVSNet stages of one reference view (config 3), and roofline-style rows for
    the fused warp + variance kernel (HBM: algorithmic bytes of SURVEY.md 8d) and the regularisation U-Net (MFMA: 2 x MACs
    per voxel x voxels, priced against 2500 / 3 TFLOP/s like the fp16x2 MLP kernels), conv0 on its own, FeatureNet."""
    dev = dev or torch.device("cuda:0")
    H, W = 512, 640
    feats, proj, depth_values = synth.make_mvs_sample(3, img_hw=(H, W))
    m = CascadeMVSNet(refine=False, ndepths=[192, 32, 8], depth_interals_ratio=[1.0, 0.5, 0.5], share_cr=False,
                      cr_base_chs=[8, 8, 8], grad_method="detach")
    for st, cin in enumerate((32, 16, 8)):
        m.cost_regularization[st].load_state_dict(
            {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_costreg_params(100 + st, cin).items()})
    m.to(dev).eval()
    G = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    sample = dict(imgs=torch.zeros(1, 3, 3, H, W, device=dev), depth_values=G(depth_values)[None],
                  proj_matrices={k: G(v)[None] for k, v in proj.items()})
    features = [{k: G(v)[None] for k, v in f.items()} for f in feats]
    res = {}
    macs = {0: 10152, 1: 6696, 2: 4968}
    # the three stages of a reference view back to back, several views in a row without a host synchronisation (the
    # stage loop of runner.py:178-243 does not synchronise either): stage boundaries are events on the stream, the
    # times are those of the last pass
    reps = 6
    marks = []
    for rep in range(reps):
        outputs = None
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        for st in range(3):
            outputs, _ = m(st, sample, features=features, extra=None, outputs=outputs, int_r=m.depth_interals_ratio[st])
            ev[st + 1].record()
        marks.append(ev)
    torch.cuda.synchronize()
    for st in range(3):
        res[f"stage{st + 1}_ms"] = min(mk[st].elapsed_time(mk[st + 1]) for mk in marks[2:])
    # the fused warp + variance alone, in the form the pipeline runs it (split-volume output feeding conv0)
    for st in range(3):
        key = f"stage{st + 1}"
        dv = outputs[key]["depth_values"]
        fs = [f[key] for f in features]
        ts = []
        for _ in range(5):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            e[0].record()
            costvol.warp_variance(fs, sample["proj_matrices"][key], dv, split=True)
            e[1].record()
            torch.cuda.synchronize()
            ts.append(e[0].elapsed_time(e[1]))
        C, D, h, w = fs[0].shape[1], dv.shape[1], dv.shape[2], dv.shape[3]
        bytes_ = 4 * (C * D * h * w + D * h * w + 3 * C * h * w)
        t = min(ts) * 1e-3
        res[key + "_warp_ms"] = t * 1e3
        res[key + "_warp_GBps"] = bytes_ / t / 1e9
        res[key + "_unet_TFLOPs"] = 2 * macs[st] * D * h * w / ((res[key + "_ms"] * 1e-3 - t)) / 1e12
        rows = res.setdefault("roofline", [])
        rows.append(dict(kernel="svs::costvol::warp_variance_reuse2_kernel", stage=st + 1, bound="hbm", kernel_ms=t * 1e3,
                         algorithmic_bytes=bytes_, achieved=bytes_ / t / 1e9, peak=8000.0, unit="GB/s", frac=bytes_ / t / 8.0e12))
        flop = 2 * macs[st] * D * h * w
        tu = res[key + "_ms"] * 1e-3 - t
        rows.append(dict(kernel="CostRegNet (11 launches)", stage=st + 1, bound="mfma", kernel_ms=tu * 1e3, algorithmic_flop=flop,
                         achieved=flop / tu / 1e12, peak=2500.0 / 3, unit="TFLOP/s", frac=flop / tu / (2.5e15 / 3)))
        if st == 0:
            # conv0 alone (32 -> 8 channels on the split volume the warp kernel wrote): 68 % of the stage-1 U-Net's MACs
            cr = m.cost_regularization[0]
            sv = costvol.warp_variance(fs, sample["proj_matrices"][key], dv, split=True)
            w0, b0 = cr.conv0.folded()
            if True:
                tc = []
                for _ in range(5):
                    sv = costvol.warp_variance(fs, sample["proj_matrices"][key], dv, split=True)
                    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                    e[0].record()
                    costvol.conv3d(sv, w0, b0, relu=True)
                    e[1].record()
                    torch.cuda.synchronize()
                    tc.append(e[0].elapsed_time(e[1]))
                f0 = 2 * 27 * 32 * 8 * D * h * w
                res["stage1_conv0_ms"] = min(tc)
                rows.append(dict(kernel="svs::conv::conv3d_pair_kernel<32,2>", stage=1, bound="mfma", kernel_ms=min(tc),
                                 algorithmic_flop=f0, achieved=f0 / (min(tc) * 1e-3) / 1e12, peak=2500.0 / 3, unit="TFLOP/s",
                                 frac=f0 / (min(tc) * 1e-3) / (2.5e15 / 3)))
    # FeatureNet (row f1) on one 512 x 640 image
    from models.CasMVSNet import FeatureNet
    net = FeatureNet(8, 3, 4, "fpn")
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_featurenet_params(1).items()})
    net.to(dev).eval()
    x = torch.rand(1, 3, H, W, device=dev)
    with torch.no_grad():
        net(x); torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        for _ in range(10):
            net(x)
        e[1].record(); torch.cuda.synchronize()
    res["featurenet_ms_per_image"] = e[0].elapsed_time(e[1]) / 10
    res["workload"] = "configs[2]: CasMVSNet 3-stage cost volume, 640x512 image, D = 192/32/8, 3 views, one reference view"
    return res


def main():
    print(json.dumps(measure()))


if __name__ == "__main__":
    main()
