"""How long does the host take to ENQUEUE one train step (no synchronisation inside the loop) vs the GPU to run it?"""
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
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = VolSDFNetwork(dtu_model_conf()); m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(0).items()}); m.to(dev).train()
K, pose = synth.make_camera()
inp = {"intrinsics": torch.from_numpy(K)[None].to(dev), "uv": torch.from_numpy(synth.make_uv(R, seed=1))[None].to(dev), "pose": torch.from_numpy(pose)[None].to(dev)}
gt = {"rgb": torch.rand(1, R, 3, device=dev), "rgb_smooth": torch.rand(1, R, 3, device=dev)}
loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1)
for groups in (None, "auto"):
    ts = TrainStep(m, loss, groups=groups)
    for _ in range(3):
        ts(inp, gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ts(inp, gt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"groups={groups}: host enqueue {1e3*(t1-t0)/10:.2f} ms/step, wall {1e3*(t2-t0)/10:.2f} ms/step")

import cProfile, pstats
ts = TrainStep(m, loss, groups=None)
for _ in range(3):
    ts(inp, gt)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    ts(inp, gt)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumtime").print_stats(45)
