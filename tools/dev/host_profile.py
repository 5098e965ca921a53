"""cProfile of the host side of one train step (two ray groups, MVS prior), sorted by own time; 10 steps that fit in the
GPU queue, so no call blocks on the device."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "s-volsdf_amd")):
    sys.path.insert(0, p)
import numpy as np, torch, synth
from volsdf.utils.conf import dtu_model_conf
from svs_hip.trainer import TrainStep
from volsdf.model.loss import VolSDFLoss
from volsdf.model.network import VolSDFNetwork
import cProfile, pstats

dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = VolSDFNetwork(dtu_model_conf()); m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(0).items()}); m.to(dev).train()
K, pose = synth.make_camera()
inp = {"intrinsics": torch.from_numpy(K)[None].to(dev), "uv": torch.from_numpy(synth.make_uv(R, seed=1))[None].to(dev), "pose": torch.from_numpy(pose)[None].to(dev)}
gt = {"rgb": torch.rand(1, R, 3, device=dev), "rgb_smooth": torch.rand(1, R, 3, device=dev)}
views = synth.make_mvs_views(3)
mvs = dict(views=[dict(K=v["K"], c2w=v["c2w"], cost=torch.from_numpy(v["cost"]).to(dev), z_mvs=torch.from_numpy(v["z_mvs"]).to(dev)) for v in views],
           same_view=0, img_res=(576, 768), inverse_depth=False)
loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=1.0, sparse_weight=1.0, anneal_rgb=200, gce=0.5, confi=1e-3)
ts = TrainStep(m, loss, groups="auto")
for _ in range(60):
    ts(inp, gt, mvs=mvs)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(6):
        ts(inp, gt, mvs=mvs)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"host enqueue {1e3*(t1-t0)/6:.2f} ms/step (6 steps, queue not full), groups {ts._groups_for(R)}")
pr = cProfile.Profile()
pr.enable()
for _ in range(6):
    ts(inp, gt, mvs=mvs)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
