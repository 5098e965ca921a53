"""Interleaved A/B (one process, alternating rounds) of TrainStep configurations in the bench.py train workload."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "s-volsdf_amd")):
    sys.path.insert(0, p)
import numpy as np, torch, synth
from volsdf.utils.conf import dtu_model_conf
from svs_hip.trainer import TrainStep
from volsdf.model.loss import VolSDFLoss
from volsdf.model.network import VolSDFNetwork

dev = torch.device("cuda:0")
R = 1024
K, pose = synth.make_camera()
inp = {"intrinsics": torch.from_numpy(K)[None].to(dev), "uv": torch.from_numpy(synth.make_uv(R, seed=1))[None].to(dev), "pose": torch.from_numpy(pose)[None].to(dev)}
gt = {"rgb": torch.rand(1, R, 3, device=dev), "rgb_smooth": torch.rand(1, R, 3, device=dev)}
views = []
for j, dx in enumerate((0.0, 0.3, -0.3)):
    Kj, Pj = synth.make_camera(center=(dx, 0.0, -2.5), tilt=-0.12 * dx / 0.3)
    prob = torch.softmax(torch.randn(192, 288, 384, device=dev), 0)
    zm = torch.linspace(1.5, 3.5, 192, device=dev).view(-1, 1, 1) * (1 + 0.05 * (torch.rand(1, 288, 384, device=dev) * 2 - 1))
    views.append(dict(K=Kj, c2w=Pj, cost=prob, z_near=zm[0].contiguous(), z_far=zm[-1].contiguous()))
mvs = dict(views=views, same_view=0, img_res=(576, 768), inverse_depth=False)
cfgs = {}
def spans(*sizes):
    out, lo = [], 0
    for n in sizes:
        out.append((lo, lo + n)); lo += n
    assert lo == R
    return out


SEL = os.environ.get("AB_GROUPS")
for name, groups in (("none", None), ("auto", "auto"), ("halves", [(0, 512), (512, 1024)]), ("q3+1", [(0, 768), (768, 1024)]),
                     ("656+368", spans(656, 368)), ("640+336+48", spans(640, 336, 48)),
                     ("336+336+352", spans(336, 336, 352)), ("320+320+384", spans(320, 320, 384)), ("384+384+256", spans(384, 384, 256)),
                     ("400+400+224", spans(400, 400, 224)), ("320+336+368", spans(320, 336, 368)), ("256x4", spans(256, 256, 256, 256)),
                     ("320x3+64", spans(320, 320, 320, 64)), ("496+480+48", spans(496, 480, 48)),
                     ("thirds", [(0, 352), (352, 704), (704, 1024)]), ("quarters", [(0, 256), (256, 512), (512, 768), (768, 1024)]),
                     ("3/8-3/8-1/4", [(0, 384), (384, 768), (768, 1024)]), ("eighths", [(128 * i, 128 * (i + 1)) for i in range(8)])):
    if SEL and name not in SEL.split(","):
        continue
    m = VolSDFNetwork(dtu_model_conf()); m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(0).items()}); m.to(dev).train()
    loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=1.0, sparse_weight=1.0, anneal_rgb=200, gce=0.5, confi=1e-3)
    if groups == "auto-rev":
        a = TrainStep.split_rays(R, 98)
        groups = [(0, R - a[0][1]), (R - a[0][1], R)] if len(a) == 2 else None
    cfgs[name] = TrainStep(m, loss, groups=groups)
for ts in cfgs.values():
    for _ in range(12):
        ts(inp, gt, mvs=mvs)
torch.cuda.synchronize()
res = {k: [] for k in cfgs}
for rnd in range(5):
    for name, ts in cfgs.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(25):
            ts(inp, gt, mvs=mvs)
        torch.cuda.synchronize()
        res[name].append(1e3 * (time.perf_counter() - t0) / 25)
for k, v in res.items():
    print(k, "ms/step per round:", [round(x, 2) for x in v], "median", round(float(np.median(v)), 3))
